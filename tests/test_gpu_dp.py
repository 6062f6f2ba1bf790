"""The product's data-parallel step under 2 ranks on real hardware (src/model.py:328-336,
train.py:203-205): WaveNet.train_step on row shards with the one-bucket SUM all-reduce must leave bit-equal
replicas and equal the single-process step on the concatenated batch.

 * test_two_ranks_one_gpu_gloo: both ranks share cuda:0 and exchange over gloo -- runs on a 1-GPU box, so the HIP
   path under world_size 2 (bucket layout, loss tail, post-reduce clipnorm, metric mean, per-rank dropout masks)
   is exercised wherever the GPU tests run;
 * test_two_ranks_nccl: one rank per GPU over RCCL; skipped when fewer than 2 devices are visible.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW = dict(blocks=6, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[48, 40],
          activation='leaky_relu', bits=8, l2_reg_factor=0.001)
# the shape family of the BASELINE networks (64 residual / 256 skip channels, first head conv 128 wide): training passes
# fold the skip path into the head and run the two-products-per-launch backward chain
KW_FOLDED = dict(blocks=5, channels=64, skip_channels=256, dilation_bound=16, final_layers_channels=[128, 64],
                 activation='leaky_relu', bits=8, l2_reg_factor=0.001)
# the exact BASELINE configs[1] / configs[2] network (30 blocks, dilations 1..512 x3), B_local = 2 x 3500 per rank
KW_CONFIGS1 = dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                   activation='leaky_relu', bits=8)
KWS = {'small': KW, 'folded': KW_FOLDED, 'configs1': KW_CONFIGS1}
GLOBAL_B, T, STEPS = 4, 300, 3
LENGTHS = {'configs1': 3500}                      # longer than the receptive field (3071)


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _data(net='small'):
  from wavenets_amd.data import synthetic_waveforms
  return synthetic_waveforms(GLOBAL_B, LENGTHS.get(net, T) + 1, seed=99, device='cpu')


def _run_steps(model, x):
  from wavenets_amd import Adam, MeanSquaredError
  model.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
  logs = None
  for _ in range(STEPS):
    logs = model.train_step(x)
  return logs


def _worker(rank, world, port, backend, out_dir, dropout, net='small'):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  dev_index = rank if backend == 'nccl' else 0
  torch.cuda.set_device(dev_index)
  dev = torch.device('cuda', dev_index)
  if backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
  else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
  from wavenets_amd import WaveNet, dp
  model = WaveNet(**KWS[net], dropout=dropout, device=dev, seed=7)
  x = _data(net)[dp.shard_rows(GLOBAL_B, world, rank)].to(dev)
  logs = _run_steps(model, x)
  torch.save({'params': model.flat_params.data.cpu(), 'logs': logs, 'drop_step': model._drop_step},
             os.path.join(out_dir, f'rank{rank}.pt'))
  dist.barrier()
  dist.destroy_process_group()


def _check(tmp_path, backend, dropout=0.0, net='small'):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), backend, str(tmp_path), dropout, net), nprocs=world, join=True)
  r0 = torch.load(tmp_path / 'rank0.pt')
  r1 = torch.load(tmp_path / 'rank1.pt')
  assert torch.equal(r0['params'], r1['params'])                 # replicas stay bit-identical
  assert r0['logs']['loss'] == r1['logs']['loss'] and r0['logs'].get('reg_loss') == r1['logs'].get('reg_loss')
  assert r0['logs']['mean_squared_error'] == r1['logs']['mean_squared_error']      # metric is reduced too
  return r0


@pytest.mark.parametrize('backend,net', [('gloo', 'small'), ('gloo', 'folded'), ('gloo', 'configs1'), ('nccl', 'folded'),
                                         ('nccl', 'configs1')])
def test_two_rank_train_step_equals_single_process(tmp_path, backend, net):
  if backend == 'nccl' and torch.cuda.device_count() < 2:
    pytest.skip('needs 2 GPUs')
  from wavenets_amd import WaveNet
  r0 = _check(tmp_path, backend, net=net)
  dev = torch.device('cuda', 0)
  single = WaveNet(**KWS[net], device=dev, seed=7)
  logs = _run_steps(single, _data(net).to(dev))
  err = (single.flat_params.data.cpu() - r0['params']).abs().max().item()
  # Two shards sum their gradients in a different order than one process over the whole batch (~1e-7 relative).  Keras
  # Adam turns that into parameter differences of up to a few percent of lr = 5e-4 per step on elements whose gradient
  # is itself ~1e-7 (update = lr * m / (sqrt(v) + 1e-7)): the 1.25 M-parameter BASELINE network has such elements.
  tol = 1e-4 if net == 'configs1' else 1e-5
  assert err < tol, err
  assert abs(logs['loss'] - r0['logs']['loss']) < 1e-5 * abs(logs['loss'])
  if 'reg_loss' in logs:
    assert abs(logs['reg_loss'] - r0['logs']['reg_loss']) < 1e-6 * max(1.0, abs(logs['reg_loss']))


def _nccl1_worker(rank, port, out_dir, early):
  """RCCL for real on a one-GPU box: a process group of ONE rank on backend "nccl" (train.py:203 with one visible device).
  dist.all_reduce then runs through librccl -- communicator creation bound to the device (device_id), the bucket + tail
  layout, the stream hand-over between RCCL's stream and the launch stream -- with a world of one."""
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
  from wavenets_amd import WaveNet, dp
  assert dp.initialized() and dp.world_size() == 1 and dist.get_backend() == 'nccl'
  # a bare collective first: the communicator is created here, on this device
  probe = torch.arange(8, dtype=torch.float32, device=dev)
  dp.allreduce_bucket(probe)
  assert torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32))
  model = WaveNet(**KW_CONFIGS1, device=dev, seed=7)
  model.early_logs = early                                       # None: automatic = the one-collective tail path
  x = _data('configs1').to(dev)
  logs = _run_steps(model, x)
  test_logs = model.test_step(x)
  torch.cuda.synchronize()
  torch.save({'params': model.flat_params.data.cpu(), 'logs': logs, 'test_logs': test_logs,
              'trips': (model.train_guard_trips, model.test_guard_trips)}, os.path.join(out_dir, 'nccl1.pt'))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('early', [None, True])
def test_nccl_world_size_one_drives_train_and_test_step(tmp_path, early):
  """BASELINE configs[2]'s exchange on the real backend as far as one GPU allows: WaveNet.train_step + test_step of the
  exact configs[1] network through dp.allreduce_bucket over RCCL (world size 1).  The data-parallel step path -- scalars
  from the all-reduced bucket tail through the pinned copy + event, or (early = True) a second tiny collective between
  forward and backward -- must reproduce the single-process step bit for bit: same kernels, same gradients, the reduce
  of one replica is the identity."""
  from wavenets_amd import WaveNet
  mp.spawn(_nccl1_worker, args=(_free_port(), str(tmp_path), early), nprocs=1, join=True)
  r = torch.load(tmp_path / 'nccl1.pt')
  dev = torch.device('cuda', 0)
  single = WaveNet(**KW_CONFIGS1, device=dev, seed=7)
  x = _data('configs1').to(dev)
  logs = _run_steps(single, x)
  test_logs = single.test_step(x)
  assert torch.equal(single.flat_params.data.cpu(), r['params'])
  assert logs == r['logs'] and test_logs == r['test_logs'], (logs, r['logs'], test_logs, r['test_logs'])
  assert r['trips'] == (0, 0)


def test_two_rank_dropout_masks_differ_per_replica(tmp_path):
  """MirroredStrategy draws an independent Dropout mask on every replica: the mask counter folds the rank in
  (call_index * world + rank + 1).  Replicas must still agree bit for bit after the reduce."""
  from wavenets_amd import _lib
  r0 = _check(tmp_path, 'gloo', dropout=0.2)
  assert r0['drop_step'] == STEPS
  L = _lib.lib()
  keys = {L.wn_dropout_key_for(7, 0, call * 2 + rank + 1) for call in range(STEPS) for rank in range(2)}
  assert len(keys) == 2 * STEPS


def test_bench_multi_rank_plumbing_rehearsal():
  """The SCALE command of the round (`python bench.py --gpus 8 --global-batch 64`, launched by the driver through
  torch.distributed.run) rehearsed at 2 ranks sharing this box's one GPU over gloo (WN_BENCH_BACKEND=gloo): the
  self-launch, the rank / device plumbing, the strong-scaling split of the global batch, the barrier + MAX-over-ranks
  timing and the one JSON line must work before a first real 8-GPU run.  (No throughput is asserted.)"""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, WN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--global-batch', '4', '--steps', '2', '--warmup', '1',
         '--length', '3500', '--no-cpu-baseline', '--no-generation', '--no-other-configs']
  res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert res.returncode == 0, res.stderr[-2000:]
  line = [ln for ln in res.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln][-1]
  out = json.loads(line)
  assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['steps'] == 2
  assert out['config']['global_batch'] == 4 and out['config']['parallelism'] == 'dp2'
  assert out['value'] > 0 and abs(out['value'] - 4 * 3500 * 2 / (out['ms_per_step'] * 2e-3)) < 1e-6 * out['value']
  assert out['roofline']['frac'] > 0 and 'cpu_baseline' not in out


def test_one_plan_shared_by_two_threads_and_streams():
  """a wn_plan is immutable and shareable across streams.  Two host threads drive ONE plan at the same
  time, each with its own execution state (wn_exec_create / wn_exec_bind), its own HIP stream, its own shape (so each
  caches different weight-gradient job tables), workspace and gradient buffer: losses and gradients are bit-identical to
  the same two calls made one after the other through the plan's own state."""
  import threading
  import torch
  from wavenets_amd import WaveNet, _lib
  from oracle import wavenet_oracle as O          # synthetic input generator only
  dev = torch.device('cuda', 0)
  kw = dict(blocks=6, channels=64, skip_channels=256, dilation_bound=32, final_layers_channels=[128, 256],
            activation='leaky_relu', bits=8)
  model = WaveNet(**kw, device=dev, seed=11)
  model.build((1, 8, 1))
  L = _lib.lib()
  shapes = [(3, 700), (2, 1333)]
  xs = [O.synthetic_waveform(B, T + 1, seed=5 + i).to(dev) for i, (B, T) in enumerate(shapes)]

  def call(i, stream_ptr):
    B, T = shapes[i]
    n = L.wn_plan_workspace_floats(model._plan, B, T, 1)
    ws = torch.empty(n, dtype=torch.float32, device=dev)
    grads = torch.zeros_like(model.flat_params)
    loss = torch.zeros(3, dtype=torch.float32, device=dev)
    for _ in range(3):                              # (repeated: the cached tables are reused, then nothing changes)
      _lib.check(L.wn_train_fwd_bwd(model._plan, _lib.ptr(model.flat_params), _lib.ptr(xs[i]), None, B, T, B, 1,
                                    _lib.ptr(grads), _lib.ptr(loss), None, _lib.ptr(ws), ws.numel(), stream_ptr))
    return loss, grads, ws

  ref = []
  for i in range(2):
    with torch.cuda.stream(torch.cuda.current_stream()):
      ref.append(call(i, _lib.stream_ptr())[:2])
  torch.cuda.synchronize()
  ref = [(l.clone(), g.clone()) for l, g in ref]

  out, errs = [None, None], []
  def worker(i):
    try:
      torch.cuda.set_device(0)
      ex = L.wn_exec_create(model._plan)
      assert ex
      L.wn_exec_bind(ex)
      st = torch.cuda.Stream()
      with torch.cuda.stream(st):
        out[i] = call(i, st.cuda_stream)
      st.synchronize()
      L.wn_exec_bind(None)
      L.wn_exec_destroy(ex)
    except Exception as e:                          # noqa: BLE001
      errs.append(e)
  ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
  for t in ths: t.start()
  for t in ths: t.join()
  assert not errs, errs
  torch.cuda.synchronize()
  for i in range(2):
    assert torch.equal(out[i][0], ref[i][0]) and torch.equal(out[i][1], ref[i][1]), i
