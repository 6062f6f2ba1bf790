"""Minimal HDF5 writer / reader for Keras ``.weights.h5`` files (train.py:149-154, 237-238 of the reference).

The image has no h5py / libhdf5, so the subset of the HDF5 file format that ``h5py.File(path, 'w')`` produces for a
tree of groups and small dense arrays -- and that Keras' ``H5IOStore`` reads and writes -- is implemented here from the
format specification (HDF5 File Format Specification v3.0, sections III.A-III.F, IV.A):

  * superblock version 0, 8-byte offsets and lengths
  * "old style" groups: object header v1 with a Symbol Table message -> v1 B-tree (node type 0) -> symbol table
    nodes (SNOD) -> names in a local heap
  * datasets: object header v1 with Dataspace (v1), Datatype (v1: IEEE float32 / float64, two's-complement int32 /
    int64, little endian), Fill Value (v2) and Data Layout (v3, contiguous) messages; raw data stored contiguously

The reader additionally follows object-header continuation blocks and accepts dataspace v2, compact layout and
superblock v1, which libhdf5 may choose for the same content.  Anything else (chunked / compressed datasets, new-style
link messages, superblock v2+) raises ``NotImplementedError`` with the feature named -- never a silent misread.

Nothing here touches the GPU; ``wavenets_amd.io`` maps the tree to and from a model's variables.
"""
from __future__ import annotations

import struct
from typing import Dict, Union

import numpy as np

Tree = Dict[str, Union['Tree', np.ndarray]]

_SIG = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K, _INTERNAL_K = 16, 16            # symbol-table node holds 2 * leaf K entries; B-tree node 2 * internal K children
_HEAP_FREE_NULL = 1                      # H5HL_FREE_NULL: "no free block" in a local heap header


def _pad8(n: int) -> int:
  return (n + 7) & ~7


# ------------------------------------------------------------------------------------------ writer
class _Writer:
  def __init__(self):
    self.buf = bytearray(96)             # superblock (filled in last)

  def alloc(self, data: bytes) -> int:
    while len(self.buf) % 8:
      self.buf.append(0)
    addr = len(self.buf)
    self.buf += data
    return addr

  # ---- messages
  @staticmethod
  def _msg(mtype: int, data: bytes) -> bytes:
    data = data + b'\0' * (_pad8(len(data)) - len(data))
    return struct.pack('<HHB3x', mtype, len(data), 0) + data

  @staticmethod
  def _header(msgs) -> bytes:
    body = b''.join(msgs)
    # object header v1 prefix: version, reserved, #messages, reference count, header (message block) size, pad to 8
    return struct.pack('<BBHII4x', 1, 0, len(msgs), 1, len(body)) + body

  def dataset(self, arr: np.ndarray) -> int:
    arr = np.asarray(arr)
    if not arr.flags.c_contiguous:
      arr = arr.copy()                                    # (np.ascontiguousarray would turn a scalar into shape (1,))
    if arr.dtype == np.float32:
      dt = struct.pack('<B3BI', 0x11, 0x20, 31, 0, 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
    elif arr.dtype == np.float64:
      dt = struct.pack('<B3BI', 0x11, 0x20, 63, 0, 8) + struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
    elif arr.dtype in (np.int32, np.int64):
      dt = struct.pack('<B3BI', 0x10, 0x08, 0, 0, arr.dtype.itemsize) + struct.pack('<HH', 0, 8 * arr.dtype.itemsize)
    else:
      raise TypeError(f'unsupported dtype {arr.dtype}')
    raw = arr.astype(arr.dtype.newbyteorder('<'), copy=False).tobytes()
    data_addr = self.alloc(raw) if raw else _UNDEF
    space = struct.pack('<BBB5x', 1, arr.ndim, 0) + b''.join(struct.pack('<Q', d) for d in arr.shape)
    fill = struct.pack('<BBBB', 2, 2, 2, 0)               # v2: late allocation, fill if set, no user fill value
    layout = struct.pack('<BBQQ', 3, 1, data_addr, len(raw))
    return self.alloc(self._header([self._msg(0x0001, space), self._msg(0x0003, dt), self._msg(0x0005, fill),
                                    self._msg(0x0008, layout)]))

  def group(self, tree: Tree):
    """Writes a group and everything below it; returns (object header address, B-tree address, heap address)."""
    names = sorted(tree)                                  # symbol table entries are ordered by name (strcmp)
    if len(names) > 2 * _LEAF_K * 2 * _INTERNAL_K:
      raise NotImplementedError('group with more than 1024 members')
    child_addr = {}
    for n in names:
      v = tree[n]
      child_addr[n] = self.group(v)[0] if isinstance(v, dict) else self.dataset(np.asarray(v))
    # local heap: the empty string at offset 0, then the names, each NUL-terminated and padded to 8 bytes
    heap = bytearray(8)
    off = {}
    for n in names:
      b = n.encode('utf-8') + b'\0'
      off[n] = len(heap)
      heap += b + b'\0' * (_pad8(len(b)) - len(b))
    heap_data = self.alloc(bytes(heap))
    heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), _HEAP_FREE_NULL, heap_data))
    # symbol table nodes, each with room for 2 * leaf K entries
    chunks = [names[i:i + 2 * _LEAF_K] for i in range(0, len(names), 2 * _LEAF_K)]
    snods, keys = [], [0]
    for ch in chunks:
      ent = b''.join(struct.pack('<QQII16x', off[n], child_addr[n], 0, 0) for n in ch)
      ent += b'\0' * (40 * (2 * _LEAF_K - len(ch)))
      snods.append(self.alloc(b'SNOD' + struct.pack('<BBH', 1, 0, len(ch)) + ent))
      keys.append(off[ch[-1]])
    node = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), _UNDEF, _UNDEF)
    for i in range(2 * _INTERNAL_K):
      node += struct.pack('<QQ', keys[i] if i < len(keys) else 0, snods[i] if i < len(snods) else _UNDEF)
    node += struct.pack('<Q', keys[2 * _INTERNAL_K] if len(keys) > 2 * _INTERNAL_K else 0)
    btree_addr = self.alloc(node)
    hdr = self.alloc(self._header([self._msg(0x0011, struct.pack('<QQ', btree_addr, heap_addr))]))
    return hdr, btree_addr, heap_addr

  def finish(self, tree: Tree) -> bytes:
    hdr, btree, heap = self.group(tree)
    while len(self.buf) % 8:
      self.buf.append(0)
    sb = _SIG + struct.pack('<8B', 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack('<HHI', _LEAF_K, _INTERNAL_K, 0)
    sb += struct.pack('<QQQQ', 0, _UNDEF, len(self.buf), _UNDEF)
    sb += struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', btree, heap)          # root symbol table entry (cached)
    assert len(sb) == 96
    self.buf[:96] = sb
    return bytes(self.buf)


def write_h5(path: str, tree: Tree) -> None:
  """Write ``tree`` (nested dicts = groups, arrays = datasets) as an HDF5 file."""
  data = _Writer().finish(tree)
  with open(path, 'wb') as f:
    f.write(data)


# ------------------------------------------------------------------------------------------ reader
class _Reader:
  def __init__(self, data: bytes):
    self.d = data
    if data[:8] != _SIG:
      raise ValueError('not an HDF5 file (bad signature)')
    ver = data[8]
    if ver not in (0, 1):
      raise NotImplementedError(f'HDF5 superblock version {ver} (only 0 and 1: write the file with libver="earliest")')
    if data[13] != 8 or data[14] != 8:
      raise NotImplementedError('HDF5 offsets / lengths other than 8 bytes')
    self.leaf_k, self.internal_k = struct.unpack_from('<HH', data, 16)
    p = 24 + (4 if ver == 1 else 0)                        # v1 adds indexed-storage K + reserved
    self.base, _, self.eof, _ = struct.unpack_from('<QQQQ', data, p)
    self.root_hdr = struct.unpack_from('<Q', data, p + 32 + 8)[0]

  def _messages(self, addr: int):
    ver, _, nmsg, _, size = struct.unpack_from('<BBHII', self.d, addr)
    if ver != 1:
      raise NotImplementedError(f'object header version {ver} at {addr} (only version 1)')
    blocks = [(addr + 16, size)]
    out = []
    while blocks and len(out) < nmsg:
      p, left = blocks.pop(0)
      end = p + left
      while p + 8 <= end and len(out) < nmsg:
        mtype, msize, _flags = struct.unpack_from('<HHB', self.d, p)
        body = self.d[p + 8:p + 8 + msize]
        p += 8 + msize
        if mtype == 0x0010:                                # continuation: (offset, length) of another message block
          o, l = struct.unpack_from('<QQ', body, 0)
          blocks.append((self.base + o, l))
        out.append((mtype, body))
    return out

  def _heap_name(self, heap_addr: int, off: int) -> str:
    if self.d[heap_addr:heap_addr + 4] != b'HEAP':
      raise ValueError('bad local heap signature')
    seg = struct.unpack_from('<Q', self.d, heap_addr + 24)[0] + self.base
    end = self.d.index(b'\0', seg + off)
    return self.d[seg + off:end].decode('utf-8')

  def _btree_entries(self, node: int, heap: int, out: list):
    if self.d[node:node + 4] == b'SNOD':
      n = struct.unpack_from('<H', self.d, node + 6)[0]
      for i in range(n):
        name_off, hdr = struct.unpack_from('<QQ', self.d, node + 8 + 40 * i)
        out.append((self._heap_name(heap, name_off), self.base + hdr))
      return
    if self.d[node:node + 4] != b'TREE':
      raise ValueError('bad group B-tree signature')
    ntype, _level, used = struct.unpack_from('<BBH', self.d, node + 4)
    if ntype != 0:
      raise ValueError('group B-tree node of the wrong type')
    for i in range(used):
      child = struct.unpack_from('<Q', self.d, node + 24 + 16 * i + 8)[0]
      self._btree_entries(self.base + child, heap, out)

  def read(self, hdr: int):
    msgs = self._messages(hdr)
    types = {t for t, _ in msgs}
    if 0x0011 in types:                                    # old-style group
      btree, heap = struct.unpack_from('<QQ', next(b for t, b in msgs if t == 0x0011), 0)
      ents = []
      self._btree_entries(self.base + btree, self.base + heap, ents)
      return {name: self.read(addr) for name, addr in ents}
    if 0x0006 in types or 0x0002 in types:
      raise NotImplementedError('new-style group (link messages): write the file with libver="earliest"')
    return self._dataset(msgs)

  def _dataset(self, msgs):
    shape = dtype = None
    data = None
    for t, b in msgs:
      if t == 0x0001:
        ver, rank, flags = struct.unpack_from('<BBB', b, 0)
        p = 8 if ver == 1 else 4
        shape = struct.unpack_from('<%dQ' % rank, b, p) if rank else ()
      elif t == 0x0003:
        cls, b0, _b1, _b2, size = struct.unpack_from('<B3BI', b, 0)
        if b0 & 1:
          raise NotImplementedError('big-endian dataset')
        kind = cls & 0x0F
        if kind == 1 and size in (4, 8):
          dtype = np.dtype('<f%d' % size)
        elif kind == 0 and size in (1, 2, 4, 8):
          dtype = np.dtype(('<i%d' if b0 & 8 else '<u%d') % size)
        else:
          raise NotImplementedError(f'datatype class {kind} size {size}')
      elif t == 0x0008:
        ver, cls = struct.unpack_from('<BB', b, 0)
        if ver != 3:
          raise NotImplementedError(f'data layout message version {ver}')
        if cls == 1:
          addr, size = struct.unpack_from('<QQ', b, 2)
          data = b'' if addr == _UNDEF else self.d[self.base + addr:self.base + addr + size]
        elif cls == 0:
          size = struct.unpack_from('<H', b, 2)[0]
          data = b[4:4 + size]
        else:
          raise NotImplementedError('chunked dataset (Keras weight files store variables contiguously)')
      elif t == 0x000B:
        raise NotImplementedError('filtered (compressed) dataset')
    if shape is None or dtype is None or data is None:
      raise ValueError('object is neither a group nor a complete dataset')
    n = int(np.prod(shape)) if shape else 1
    if len(data) < n * dtype.itemsize:
      raise ValueError('dataset shorter than its dataspace')
    return np.frombuffer(data, dtype=dtype, count=n).reshape(shape).copy()


def read_h5(path: str) -> Tree:
  """Read an HDF5 file of the supported subset into nested dicts of numpy arrays."""
  with open(path, 'rb') as f:
    r = _Reader(f.read())
  return r.read(r.base + r.root_hdr)
