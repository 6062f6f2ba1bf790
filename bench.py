#!/usr/bin/env python3
"""Headline benchmark: WaveNet training-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--global-batch G]

A "step" is one full data-parallel training step as train.py:225-228 compiles it (src/model.py:309-348):
shift-by-one, forward, loss, backward, SUM all-reduce of the flat gradient over RCCL, per-tensor clipnorm +
Keras Adam, and -- because the reference compiles a MeanSquaredError metric -- the per-step
sample_waveform(pred) draw and the metric update, on one batch of synthetic 16 kHz mu-law waveforms already
resident in HBM.  Workload = BASELINE.json configs[1] (30-layer 3x10 mu-law-256 WaveNet, 64 residual / 256
skip channels, batch 8 x 16000 per GPU); N GPUs run configs[2]'s data-parallel form (weak scaling: per-GPU
batch 8, global batch 8 N -- global batch 64 at N = 8, as configs[2] names it).

  --gpus N > 1 without an enclosing torch.distributed.run: this script starts the N ranks itself (child
  processes of `python -m torch.distributed.run`, before anything here touches the GPU) and relays rank 0's line.
  --global-batch G: STRONG scaling -- the global batch is fixed at G, every rank takes G / N utterances
  (north_star: global batch 64); without it the line is weak scaling and a short strong-scaling measurement at
  global batch 64 is reported beside it ("strong_scaling").

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- SURVEY.md 8(d): the residual-block STACK forward.  achieved = N_blocks * 4 B T (2R + S) /
                  t_stack_fwd, t_stack_fwd = HIP events on the launch stream from the first block launch to the
                  end of the folded contraction over the blocks' gated activations (the skip tensors of the reference's
                  signature are produced and consumed there; training passes fold the head's first conv into it),
                  against the 8 TB/s HBM3E peak.  "fused_block_kernel" beside it: the dominant kernel on
                  its own counter bytes.
  cpu_baseline -- the CPU oracle (restated reference, PyTorch CPU) timed on this host's cores on a bounded
                  sample of the same workload
  generation   -- (N = 1) autoregressive generation speed on the same network, SURVEY.md 8(f)-1: queued sampler with
                  stochastic / deterministic draws and the sliding window, per utterance and aggregate at batch 8
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CFG2 = dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024,
            final_layers_channels=[128, 256], activation='leaky_relu', bits=8)
STRONG_GLOBAL_BATCH = 64   # north_star / configs[2]


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--batch', type=int, default=8, help='per-GPU batch (utterances), weak scaling')
  ap.add_argument('--global-batch', type=int, default=0,
                  help='strong scaling: fixed global batch, split over the ranks (0 = weak scaling)')
  ap.add_argument('--length', type=int, default=16000, help='predicted samples per utterance')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-strong-leg', action='store_true', help='skip the extra global-batch-64 measurement')
  ap.add_argument('--no-generation', action='store_true', help='skip the generation-speed leg (N = 1 only)')
  ap.add_argument('--no-dp-leg', action='store_true', help='skip the world-size-1 RCCL data-parallel leg (N = 1 only)')
  ap.add_argument('--no-other-configs', action='store_true', help='skip the other BASELINE configs (N = 1 only)')
  ap.add_argument('--cpu-budget', type=float, default=20.0)
  return ap.parse_args(argv)


def self_launch(args):
  """--gpus N outside torch.distributed.run: start the ranks as children and relay rank 0's JSON line.
  Runs before torch is imported, so this process never touches the GPU."""
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
  line = None
  for ln in res.stdout.splitlines():
    if ln.startswith('{') and '"metric"' in ln:
      line = ln
  if res.returncode != 0 or line is None:
    sys.stderr.write(res.stdout[-4000:])
    raise SystemExit(res.returncode or 1)
  print(line)
  raise SystemExit(0)


def generation_leg(dev, B: int = 8, n: int = 1000):
  """Autoregressive generation on the configs[1] network (the reference prints 'Speed of generation was ... samples/s',
  train.py:253-261): the queued (ring-buffer) sampler, stochastic draws as sample_waveform makes them, beside the sliding
  window of src/model.py:296-305.  Random-init weights, a random receptive field as the seed window."""
  import torch
  from wavenets_amd import WaveNet
  m = WaveNet(**CFG2, device=dev, seed=0)
  w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
  res = {'workload': f'configs[1] weights, batch {B}, {n} samples per utterance after a {m.receptive_field}-sample window'}
  for name, queued, det, steps in (('queued_stochastic', True, False, n), ('queued_deterministic', True, True, n),
                                   ('sliding_window', False, True, 20)):
    ts = []
    for k in (steps // 4, steps):                      # two lengths: the difference leaves the priming pass out
      m.generate(k, sample=w, use_queues=queued, deterministic=det)     # (workspace of this length allocated)
      best = None
      for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.generate(k, sample=w, use_queues=queued, deterministic=det)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
      ts.append(best)
    per = (ts[1] - ts[0]) / (steps - steps // 4)
    res[name] = {'ms_per_sample_step': per * 1e3, 'samples_per_s_per_utterance': 1.0 / per, 'samples_per_s_aggregate': B / per}
  del m
  # configs[3] weights (128 residual channels, mixture-of-logistics draws): the queued sampler's other kernel family
  m3 = WaveNet(**OTHER_CONFIGS['configs[3]'][0], device=dev, seed=0)
  w3 = (torch.rand(B, m3.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
  from wavenets_amd import _lib
  # (the default: one workgroup per block, rows handed from CU to CU inside the launch -- wn_gen_relay128_kernel; knob 2:
  # every block inside one workgroup per utterance tile -- wn_gen_chain128_kernel, the form of the earlier rounds)
  for name, knob in (('configs3_queued_stochastic', 0), ('configs3_queued_stochastic_one_workgroup', 1)):
    _lib.lib().wn_debug_set(2, knob)
    try:
      ts = []
      for k in (100, 400):
        m3.generate(k, sample=w3, use_queues=True, deterministic=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m3.generate(k, sample=w3, use_queues=True, deterministic=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    finally:
      _lib.lib().wn_debug_set(2, 0)
    per = (ts[1] - ts[0]) / 300
    res[name] = {'ms_per_sample_step': per * 1e3, 'samples_per_s_per_utterance': 1.0 / per, 'samples_per_s_aggregate': B / per}
  return res


# the other BASELINE.json configurations at their full size on one GPU (SURVEY.md 8(d) "Config -> constructor args")
OTHER_CONFIGS = {
    'configs[0]': (dict(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[], bits=8), 1,
                   '10-layer mu-law-256, dilations 1..512, 32 residual ch, head [], batch 1x16000'),
    'configs[3]': (dict(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                        activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16), 8,
                   '30-layer mixture-of-logistics head (10 mixtures), 128 residual / 256 skip ch, batch 8x16000 per GPU'),
    'configs[4]': (dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                        activation='leaky_relu', bits=8, conditioning='global', mapping_layers=[8, 16, 32],
                        mapping_activation='leaky_relu'), 8,
                   'global-conditioned (110-way one-hot speaker id, mapping [8,16,32]), 30 layers, batch 8x16000 per GPU'),
    # not a BASELINE.json config: the reference's own defaults (train.py:22-50) -- 5 blocks x 5 stacked dilated convs,
    # 32 channels, gaussian-8 on 16 bits, global conditioning (one-hot(gender, 2), src/utils.py:46-49), dropout 0.1,
    # dilation_bound 256, batch 64 x 8000 (recording_length 8000)
    'reference_default': (dict(blocks=5, layers_per_block=5, channels=32, dilation_bound=256, num_mixtures=8,
                               sampling_function='gaussian', bits=16, conditioning='global', mapping_layers=[8, 16, 32],
                               mapping_activation='leaky_relu', activation='leaky_relu', final_layers_channels=[128, 256],
                               dropout=0.1), 64, "the reference's default network (train.py:22-50): 5 blocks x 5 stacked dilated convs, "
                              '32 ch, gaussian-8, global conditioning, dropout 0.1, batch 64x8000', 8000, 2),
}


def other_configs_leg(dev, stack_profile, T: int = 16000, steps: int = 10, warmup: int = 3):
  """Train-step time and SURVEY 8(d) stack fraction of BASELINE configs[0], [3], [4] on one GPU (same step as the
  headline: Adam + clipnorm + the MSE metric's sample draw).  Host-timed between two device synchronisations."""
  import torch
  from wavenets_amd import WaveNet, Adam, MeanSquaredError, _lib
  from wavenets_amd.data import synthetic_waveforms
  res = {}
  T_default = T
  for name, entry in OTHER_CONFIGS.items():
    kw, B, desc = entry[0], entry[1], entry[2]
    T = entry[3] if len(entry) > 3 else T_default           # samples per utterance
    n_cond = entry[4] if len(entry) > 4 else 110            # width of the one-hot condition
    m = WaveNet(**kw, device=dev, seed=0)
    m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
    x = synthetic_waveforms(B, T + 1, seed=99, device=dev)
    data = x
    if kw.get('conditioning'):
      spk = torch.randint(0, n_cond, (B,), generator=torch.Generator().manual_seed(1))
      data = (x, torch.nn.functional.one_hot(spk, n_cond).float().to(dev))
    for _ in range(warmup):
      logs = m.train_step(data)
    # three blocks of `steps` steps, the median block reported (one block once came out 40 % slow on a shared host)
    blocks_ms = []
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        logs = m.train_step(data)
      torch.cuda.synchronize()
      blocks_ms.append((time.perf_counter() - t0) / steps * 1e3)
    dt = sorted(blocks_ms)[1] * 1e-3
    nblk = kw['blocks']
    nconv = nblk * kw.get('layers_per_block', 1)
    _, avg_ms, n_s, stack_ms, prep_ms = stack_profile(m, data, 5, nblk)
    R = kw['channels']
    S_eff = kw.get('skip_channels') or R
    stack_bytes = nblk * 4.0 * B * T * (2 * R + S_eff)
    t_stack = stack_ms + (prep_ms if _lib.lib().wn_debug_value(9) == 1 else 0.0)   # (prep runs beside the chain unless knob 9)
    res[name] = {'workload': desc, 'kernel_families': m.kernel_report(), 'ms_per_step': dt * 1e3, 'samples_per_s': B * T / dt, 'steps': steps,
                 'ms_per_step_blocks': blocks_ms,
                 'final_loss': logs['loss'],
                 'stack_fwd': {'t_stack_fwd_ms': t_stack, 't_fold_prep_ms': prep_ms, 'passes_timed': n_s,
                               'algorithmic_bytes': stack_bytes,
                               'achieved_GBps': stack_bytes / (t_stack * 1e-3) / 1e9 if t_stack > 0 else None,
                               'frac': stack_bytes / (t_stack * 1e-3) / 1e9 / HBM_PEAK_GBS if t_stack > 0 else None,
                               'block_launch_avg_ms': avg_ms}}
    del m
    torch.cuda.empty_cache()
  return res


def cpu_baseline(budget_s: float = 20.0):
  """Restated-reference CPU baseline: the oracle's train step on the bench workload's own network
  (configs[1]) over a bounded sample of its batch -- one of the 8 utterances per step."""
  import torch
  from oracle import wavenet_oracle as O
  # the GPU box exposes every host core but one GPU's share is 16 (more threads only thrash)
  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  cores = max(1, min(avail, 16))
  torch.set_num_threads(cores)
  ocfg = O.OracleConfig(**CFG2)
  params = O.init_params(ocfg, seed=0, bias_range=0.0)
  m = [torch.zeros_like(p) for p in params]
  v = [torch.zeros_like(p) for p in params]
  x = O.synthetic_waveform(1, 16001, seed=1234)
  t0 = time.time()
  _, params, m, v = O.train_step(x, params, m, v, 1, ocfg)          # warm-up
  one = time.time() - t0
  n = max(2, min(100, int(budget_s / max(one, 1e-3)) - 1))
  t0 = time.time()
  for i in range(n):
    _, params, m, v = O.train_step(x, params, m, v, i + 2, ocfg)
  dt = (time.time() - t0) / n
  return {'value': 16000.0 / dt, 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
          'sample': f'{n} train steps of the configs[1] network (30-layer, 64/256 ch, head [128,256]) on 1 of the '
                    f'8 utterances (batch 1x16000) with the PyTorch-CPU oracle, {dt * 1e3:.0f} ms/step'}


def _profile_rows(pattern):
  """Newest committed rocprofv3 summary matching profiles/<pattern> as a list of csv rows."""
  import csv
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
  if not files:
    return []
  with open(files[-1]) as f:
    return list(csv.DictReader(f))


def main():
  args = parse_args()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and world == 1 and 'RANK' not in os.environ:
    self_launch(args)                                     # does not return

  import ctypes as C
  import torch
  import torch.distributed as dist
  from wavenets_amd import WaveNet, Adam, MeanSquaredError, _lib
  from wavenets_amd.data import synthetic_waveforms    # (oracle/ is imported by the cpu_baseline leg only)

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  # WN_BENCH_BACKEND=gloo lets several ranks share ONE GPU (rehearsal of the multi-rank path on a
  # single-GPU box); the measured configuration is always nccl (= RCCL), one rank per GPU
  backend = os.environ.get('WN_BENCH_BACKEND', 'nccl')
  ndev = torch.cuda.device_count()
  dev_index = local_rank % max(ndev, 1) if backend != 'nccl' else local_rank
  torch.cuda.set_device(dev_index)
  dev = torch.device('cuda', dev_index)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(backend)

  T = args.length
  strong = args.global_batch > 0
  if strong:
    if args.global_batch % world:
      raise SystemExit(f'--global-batch {args.global_batch} is not divisible by {world} ranks')
    B = args.global_batch // world
  else:
    B = args.batch
  L = _lib.lib()
  nblk = CFG2['blocks']

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  def make_model():
    model = WaveNet(**CFG2, device=dev, seed=0)          # glorot kernels, zero biases, same on all ranks
    # the reference's own compile call (train.py:225-228): Adam + clipnorm and the MeanSquaredError metric, so
    # every step also draws sample_waveform(pred) and updates the metric (src/model.py:338-346)
    model.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
    return model

  def timed_steps(model, x, steps, warmup):
    """(total seconds for `steps` steps bracketed as the contract says, per-step host times, last logs)."""
    for _ in range(warmup):
      logs = model.train_step(x)
    sync()
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
      ts = time.perf_counter()
      logs = model.train_step(x)                         # ends with the step's one device-to-host read
      per.append(time.perf_counter() - ts)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), per, logs

  def stack_profile(m, data, steps, nblocks):
    """(block launches timed, avg ms per launch, stack passes timed, avg stack ms, avg fold-prep ms) over `steps`
    extra training steps with the library's event hooks armed."""
    _lib.check(L.wn_prof_enable(m._plan, nblocks * steps))
    _lib.check(L.wn_stack_prof_enable(m._plan, steps))
    for _ in range(steps):
      m.train_step(data)
    torch.cuda.synchronize()
    nl, av = C.c_int32(), C.c_float()
    _lib.check(L.wn_prof_read(m._plan, C.byref(nl), C.byref(av)))
    _lib.check(L.wn_prof_enable(m._plan, 0))
    npp, pm = C.c_int32(), C.c_float()
    _lib.check(L.wn_stack_prof_read_foldprep(m._plan, C.byref(npp), C.byref(pm)))
    ns_, sm = C.c_int32(), C.c_float()
    _lib.check(L.wn_stack_prof_read(m._plan, C.byref(ns_), C.byref(sm)))
    _lib.check(L.wn_stack_prof_enable(m._plan, 0))
    return nl.value, av.value, ns_.value, sm.value, pm.value

  model = make_model()
  x = synthetic_waveforms(B, T + 1, seed=1234 + rank, device=dev)
  # the timed region carries no measurement hooks: W warm-up steps, then exactly K steps between two barriers
  dt, per_step, logs = timed_steps(model, x, args.steps, args.warmup)
  # roofline inputs from EXTRA, untimed steps: HIP events on the launch stream around the block chain (per-launch
  # average of the fused block kernel), around the whole stack (first block launch -> end of the folded contraction) and
  # around the fold's per-pass weight-space preparation (V = W_s W_f0 and its fp16 images), which runs before the stack
  kernel_families = model.kernel_report()
  n_prof = max(3, min(args.steps, 10))
  n_l, avg_ms, n_s, stack_ms, prep_ms = stack_profile(model, x, n_prof, nblk)
  if world > 1:
    sync()
  ms_per_step = dt / args.steps * 1e3
  value = world * B * T * args.steps / dt

  # where the step goes (untimed extra steps): phase marks inside wn_train_fwd_bwd (HIP events on the launch
  # stream) + torch events around the gradient all-reduce and the optimizer
  phases = None
  from wavenets_amd import dp as _dp
  if rank == 0:
    _lib.check(L.wn_phase_enable(model._plan, 1))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    acc = [0.0] * 7
    nph = 3
    mse_metric = MeanSquaredError()
    for _ in range(nph):
      ev[0].record()
      loss_p, samp_p, y_p = model.loss_and_grads(x, want_sample=True)   # the sample draw rides in the loss phase
      ev[1].record()
      _dp.allreduce_gradients(model.flat_grads, loss_p)      # no-op for a single replica
      ev[2].record()
      model.optimizer.apply_gradients(model)
      ev[3].record()
      mse_p = mse_metric.update_state_device(y_p, samp_p)      # the library's reduction, as train_step queues it
      ev[4].record()
      torch.cuda.synchronize()
      ms4 = (C.c_float * 4)()
      _lib.check(L.wn_phase_read(model._plan, ms4))
      for i in range(4):
        acc[i] += ms4[i]
      acc[4] += ev[1].elapsed_time(ev[2])
      acc[5] += ev[2].elapsed_time(ev[3])
      acc[6] += ev[3].elapsed_time(ev[4])
    _lib.check(L.wn_phase_enable(model._plan, 0))
    names = ['forward', 'loss', 'backward_data', 'weight_gradients', 'allreduce', 'optimizer', 'metric']
    phases = {n: round(a / nph, 4) for n, a in zip(names, acc)}
  elif world > 1:                                         # the other ranks take part in the collectives
    for _ in range(3):
      loss_p, _, _ = model.loss_and_grads(x, want_sample=True)
      _dp.allreduce_gradients(model.flat_grads, loss_p)
      model.optimizer.apply_gradients(model)
  sync()

  # N = 1: the DATA-PARALLEL form of the same step on the real backend -- a process group of one rank on "nccl" (= RCCL):
  # dp.allreduce_bucket then runs librccl's all-reduce on the gradient bucket, and train_step takes its data-parallel
  # path (scalars out of the all-reduced bucket tail through a pinned copy + event; ONE collective per step) instead of
  # the single-replica early read.  `dp_mode.ms_per_step` is what that path costs against the headline.
  dp_mode = None
  if world == 1 and not args.no_dp_leg:
    try:
      import socket
      sk = socket.socket()
      sk.bind(('127.0.0.1', 0))
      port = sk.getsockname()[1]
      sk.close()
      dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
      nd = max(3, min(args.steps, 20))
      dp_mode = {'backend': 'nccl (RCCL)', 'world_size': 1}
      for key, early in (('one_collective_tail_read', None), ('early_read_second_small_collective', True)):
        model.early_logs = early
        dtd, _, _ = timed_steps(model, x, nd, 3)
        dp_mode[key] = {'ms_per_step': dtd / nd * 1e3, 'steps': nd, 'vs_headline': (dtd / nd * 1e3) / ms_per_step}
      dp_mode['ms_per_step'] = dp_mode['one_collective_tail_read']['ms_per_step']     # the default data-parallel path
      dp_mode['guard_trips'] = model.train_guard_trips
    except Exception as e:                                 # report, do not lose the headline
      dp_mode = {'error': f'{type(e).__name__}: {e}'[:300]}
    finally:
      model.early_logs = None
      if dist.is_initialized():
        dist.destroy_process_group()

  # the same step with the exact-fp32 MFMA kernels (debug knob 1), reported beside the default
  guard_trips = model.train_guard_trips                   # steps of this run repeated in exact fp32 by the range guard
  L.wn_debug_set(1, 1)
  nf = max(2, min(args.steps // 3, 10))
  dt_fp32, _, _ = timed_steps(model, x, nf, 1)
  dt_fp32 /= nf
  L.wn_debug_set(1, 0)

  # strong-scaling leg (north_star: global batch 64 split over the ranks): a short run beside the weak line
  strong_leg = None
  if not strong and not args.no_strong_leg and STRONG_GLOBAL_BATCH % world == 0:
    Bs = STRONG_GLOBAL_BATCH // world
    if Bs == B:
      strong_leg = {'global_batch': STRONG_GLOBAL_BATCH, 'per_gpu_batch': Bs, 'ms_per_step': ms_per_step, 'value': value,
                    'steps': args.steps, 'note': 'same run as the headline (per-GPU batch coincides)'}
    else:
      del model
      torch.cuda.empty_cache()
      model = make_model()
      xs = synthetic_waveforms(Bs, T + 1, seed=4321 + rank, device=dev)
      ns = max(3, min(args.steps, 10))
      dts, _, _ = timed_steps(model, xs, ns, 2)
      strong_leg = {'global_batch': STRONG_GLOBAL_BATCH, 'per_gpu_batch': Bs, 'ms_per_step': dts / ns * 1e3,
                    'value': STRONG_GLOBAL_BATCH * T * ns / dts, 'steps': ns}

  if rank == 0:
    per_ms = sorted(p * 1e3 for p in per_step)

    def pct(q):
      return per_ms[min(len(per_ms) - 1, int(round(q * (len(per_ms) - 1))))]

    # HBM bytes per launch of the dominant kernel and MFMA busy share from the committed rocprofv3 PMC passes
    # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); bench.py cannot run the counter passes itself
    traffic, mfma_busy = None, None
    for row in _profile_rows('r0*_train_pmc_hbm_traffic.csv'):
      if 'wn_layer_fwd_f16_kernel' in row['kernel'] and (B, T) == (8, 16000):
        traffic = float(row['total_bytes_corrected'])
    for row in _profile_rows('r0*_train_pmc_sq.csv'):
      if 'wn_layer_fwd_f16_kernel' in row['kernel']:
        # busy share of the chip's 1024 matrix pipes over the kernel's duration (tools/summarise_profiles.py); the round-3
        # files carry MFMA busy cycles per wave quad-cycle instead: x 2 waves per SIMD for the pipe's share
        mfma_busy = float(row['mfma_pipe_util']) if 'mfma_pipe_util' in row else 2.0 * float(row['mfma_busy_per_wave_cycle'])
    R, S = CFG2['channels'], CFG2['skip_channels']
    bytes_layer = 4.0 * B * T * (2 * R + S)          # read x, write x_out, write skip
    stack_bytes = nblk * bytes_layer
    # t_stack_fwd INCLUDES the fold's weight-space preparation of the pass (it exists only because the skip path is folded):
    # since round 4 it is forked onto the side stream AT the stack's start event and joined before the folded contraction,
    # so the event pair around the stack contains whatever it costs the chain (t_fold_prep_ms = its own duration on the
    # side stream, overlapped); with knob 9 it runs on the launch stream in front of the stack and is added
    prep_overlapped = L.wn_debug_value(9) != 1
    t_stack = stack_ms + (0.0 if prep_overlapped else prep_ms)
    achieved = stack_bytes / (t_stack * 1e-3) / 1e9 if t_stack > 0 else 0.0
    kern_gbs = (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None
    out = {
        'metric': 'audio samples/sec (training step, 16 kHz mu-law)',
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
        'vs_baseline': None, 'dtype': 'f32 (fp16 hi|lo split operands, fp32 accumulate)', 'data': 'synthetic',
        'ms_per_step_exact_fp32_mfma': dt_fp32 * 1e3, 'value_exact_fp32_mfma': world * B * T / dt_fp32,
        'range_guard_trips': guard_trips,
        'ms_per_step_median': pct(0.5), 'ms_per_step_p10': pct(0.1), 'ms_per_step_p90': pct(0.9),
        'math': 'fp32 tensors; contractions as fp16 hi/lo split, 3 products on v_mfma_f32_32x32x16_f16 with fp32 '
                'accumulate (|err| <= 6e-7 on O(1) results, parity-tested at 1e-4); exact-fp32 MFMA selectable',
        'exact_fp32_mfma': {'ms_per_step': dt_fp32 * 1e3, 'value': world * B * T / dt_fp32,
                            'note': 'the same step with v_mfma_f32_32x32x2_f32 contractions (no operand split): what a step '
                                    'costs when the range guard repeats it; range_guard_trips counts such repeats in the timed run'},
        'dp_mode': dp_mode,
        'phases_ms': phases, 'kernel_families': kernel_families,
        'config': {'workload': 'configs[1]: 30-layer (3x10) mu-law-256 WaveNet, 64 residual / 256 skip ch, '
                               f'head [128,256], batch {B}x{T} per GPU, full train step '
                               '(fwd+loss+bwd+allreduce+clipnorm-Adam+sample_waveform draw+MSE metric, as train.py:225-228 compiles it)',
                   'global_batch': world * B, 'samples_per_utterance': T, 'parallelism': f'dp{world}'},
        'per_gpu_samples_per_s': value / world,
        'strong_scaling': strong_leg,
        'final_loss': logs['loss'],
        'roofline': {'bound': 'hbm',
                     'kernel': 'residual-block stack forward: 30 x wn_layer_fwd_f16_kernel + the folded contraction over all '
                               "blocks' gated activations (wn_gemm_planes16s_kernel: skip sum and, in training passes, the head's "
                               'first conv in one product), SURVEY.md 8(d)',
                     'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'algorithmic_bytes_per_stack_pass': stack_bytes, 'algorithmic_bytes_per_block': bytes_layer,
                     't_stack_fwd_ms': t_stack, 't_blocks_and_folded_contraction_ms': stack_ms, 't_fold_prep_ms': prep_ms,
                     'passes_timed': n_s,
                     'fused_block_kernel': {'kernel': 'wn_layer_fwd_f16_kernel', 'avg_launch_ms': avg_ms,
                                            'launches_timed': n_l, 'counter_bytes_per_launch': traffic,
                                            'counter_GBps': kern_gbs,
                                            'counter_frac_of_peak': (kern_gbs / HBM_PEAK_GBS) if kern_gbs else None,
                                            'mfma_pipe_util': mfma_busy,
                                            'mfma_pipe_util_note': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 matrix pipes x kernel '
                                                                   'cycles); the kernel runs 2 waves per SIMD (8-wave workgroup per CU)'},
                     'note': 'measured in extra untimed steps after the timed region; traffic = HBM bytes per launch of the '
                             'fused block kernel (profiles/, FETCH x2 + WRITE); it writes x_out, z and the saved sigmoid -- '
                             'the skip tensors of the algorithmic signature are consumed inside the folded contraction, which '
                             "is part of the timed stack, as is the fold's per-pass weight preparation (forked at the stack's "
                             "start event onto a side stream, joined before the contraction)",
                     'fold_prep_overlapped': prep_overlapped},
    }
    if world == 1 and not args.no_other_configs:
      del model
      torch.cuda.empty_cache()
      out['other_configs'] = other_configs_leg(dev, stack_profile)
    if not args.no_cpu_baseline and world == 1:
      out['cpu_baseline'] = cpu_baseline(args.cpu_budget)
    if world == 1 and not args.no_generation:
      out['generation'] = generation_leg(dev)
    print(json.dumps(out))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
