"""Data-parallel contract on CPU: 2 ranks over gloo, each computing the ORACLE's gradients on its
shard with the global-batch loss scaling, exchanged through wavenets_amd.dp -- must equal the
single-process step on the concatenated batch (src/model.py:328-336)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wavenet_oracle as O

KW = dict(blocks=3, channels=8, skip_channels=12, dilation_bound=4, final_layers_channels=[10],
          activation='leaky_relu', bits=5, l2_reg_factor=0.01)


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, out_dir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from wavenets_amd import dp
  torch.set_num_threads(1)
  cfg = O.OracleConfig(**KW)
  params = [p.double() for p in O.init_params(cfg, seed=0)]
  x = O.synthetic_waveform(4, 65, seed=3).double()
  rows = dp.shard_rows(4, world, rank)
  m = [torch.zeros_like(p) for p in params]
  v = [torch.zeros_like(p) for p in params]
  for step in (1, 2):
    loss, reg, grads, _ = O.loss_and_grads(x[rows], params, cfg, global_batch=4, n_replicas=world)
    flat = torch.cat([g.reshape(-1) for g in grads])
    lv = torch.stack([loss, reg])
    if step == 1:
      dp.allreduce_gradients(flat, lv)
    else:
      # the form WaveNet.train_step uses: gradients and {loss, reg_loss} in ONE bucket, one collective
      bucket = torch.cat([flat, lv])
      dp.allreduce_bucket(bucket)
      flat, lv = bucket[:flat.numel()], bucket[flat.numel():]
    out, off = [], 0
    for p in params:
      out.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
    grads = O.clip_by_norm_per_tensor(out, 1.0)
    params, m, v = O.keras_adam_step(params, grads, m, v, step, 5e-4)
  torch.save({'params': params, 'loss': lv}, os.path.join(out_dir, f'rank{rank}.pt'))
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_step_equals_single_process(tmp_path):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r0 = torch.load(tmp_path / 'rank0.pt')
  r1 = torch.load(tmp_path / 'rank1.pt')
  # replicas stay identical
  for a, b in zip(r0['params'], r1['params']):
    assert torch.equal(a, b)
  # and equal the single-process run on the whole batch
  cfg = O.OracleConfig(**KW)
  params = [p.double() for p in O.init_params(cfg, seed=0)]
  x = O.synthetic_waveform(4, 65, seed=3).double()
  m = [torch.zeros_like(p) for p in params]
  v = [torch.zeros_like(p) for p in params]
  for step in (1, 2):
    loss, reg, grads, _ = O.loss_and_grads(x, params, cfg)
    grads = O.clip_by_norm_per_tensor(grads, 1.0)
    params, m, v = O.keras_adam_step(params, grads, m, v, step, 5e-4)
  for a, b in zip(r0['params'], params):
    assert torch.allclose(a, b, atol=1e-12, rtol=0)
  assert abs(r0['loss'][0].item() - loss.item()) < 1e-9
  assert abs(r0['loss'][1].item() - reg.item()) < 1e-12


def test_shard_rows():
  from wavenets_amd import dp
  assert dp.shard_rows(64, 8, 3) == slice(24, 32)
  with pytest.raises(ValueError):
    dp.shard_rows(10, 4, 0)
  assert dp.world_size() == 1 and dp.rank() == 0
