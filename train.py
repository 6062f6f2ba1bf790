#!/usr/bin/env python3
"""Thin training driver with the reference's config vocabulary (train.py:22-60 of jirsat/wavenets).

  python train.py --configfile configs/cfg2.yaml                   (1 GPU)
  python -m torch.distributed.run --nproc-per-node N train.py ...   (data parallel, RCCL)

Keys are the reference's (including the misspelt ``use_resiudal``); ``dataset`` is either
``synthetic`` (default: SURVEY.md 8d two-tone generator) or a ``.npy`` file of shape (n, samples)
with waveforms in [-1, 1] (int16 arrays are divided by 2**15).  With ``conditioning: global`` every utterance
carries a class id (``labels``: a ``.npy`` of n integers; synthetic data: utterance index modulo ``condition_classes``)
that becomes a one-hot condition on each of its frames, as the reference does with the speaker's gender
(src/utils.py:46-49, train.py:98-124); batches are then ``(frames, condition)`` tuples (src/model.py:315-317).
Checkpoints follow the reference's ``weights-e{epoch:04d}-lr{lr}`` naming and resume-from-filename convention."""
import argparse
import os
import sys
import time
import wave

import numpy as np
import torch
import yaml

config = {                       # defaults of the reference, train.py:22-50
    'epochs': 500, 'lr': 0.0005, 'recording_length': 8000, 'batch_size': 64, 'apply_mulaw': False,
    'jit_compile': False, 'dataset': 'synthetic',
    'kernel_size': 2, 'channels': 32, 'blocks': 5, 'layers_per_block': 5, 'activation': 'leaky_relu',
    'conditioning': None, 'mapping_layers': [8, 16, 32], 'mapping_activation': 'leaky_relu', 'dropout': 0.1,
    'dilation_bound': 256, 'num_mixtures': 8, 'sampling_function': 'gaussian', 'bits': 16,
    'skip_channels': None, 'dilation_channels': None, 'use_resiudal': True, 'use_skip': True,
    'final_layers_channels': [128, 256], 'l2_reg_factor': 0,
    # additions of this driver
    'steps_per_epoch': 0, 'synthetic_utterances': 256, 'sample_rate': 16000, 'results_dir': './results',
    'preview_length': 0, 'condition_classes': 2, 'labels': None,
    'checkpoint_format': 'npz',    # 'h5': Keras .weights.h5 exchange files (weights only, as the reference writes them)
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--configfile', type=str)
  ap.add_argument('--epochs', type=int, default=None)
  args = ap.parse_args()
  if args.configfile is None:
    print('No config file provided, using default config')
    run_name = 'default'
  else:
    with open(args.configfile) as f:
      config.update(yaml.safe_load(f) or {})
    run_name = os.path.splitext(os.path.basename(args.configfile))[0]
  if args.epochs is not None:
    config['epochs'] = args.epochs

  import torch.distributed as dist
  from wavenets_amd import WaveNet, Adam, MeanSquaredError, callbacks, data, io, ops

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', device_id=dev)
  conditioned = config['conditioning'] is not None
  if conditioned and config['conditioning'] != 'global':
    raise SystemExit("only conditioning: global is supported (local conditioning is broken in the reference, src/model.py:136-137)")

  # ---- data: waveforms -> frames of recording_length + 1 (src/utils.py:22-85) ----
  L = int(config['recording_length'])
  if config['dataset'] == 'synthetic':
    raw = data.synthetic_waveforms(config['synthetic_utterances'], 4 * L + 1, seed=1234)[:, :, 0]
    raw = ops.inverse_mu_law(raw.to(dev)) if config['apply_mulaw'] else raw.to(dev)
  else:
    arr = np.load(config['dataset'])
    raw = torch.from_numpy(arr)
    raw = (data.normalise_int16(raw) if arr.dtype == np.int16 else raw.float()).to(dev)
  cond = None
  if conditioned:
    ncls = int(config['condition_classes'])
    labels = np.load(config['labels']).astype(np.int64) if config['labels'] else np.arange(raw.shape[0]) % ncls
    if len(labels) != raw.shape[0]:
      raise SystemExit('labels must hold one class id per utterance')
    parts = [data.preprocess_with_condition(w, int(l), ncls, L, config['apply_mulaw']) for w, l in zip(raw, labels)]
    frames = torch.cat([p[0] for p in parts], dim=0)
    cond = torch.cat([p[1] for p in parts], dim=0)
  else:
    frames = torch.cat([data.preprocess_waveform(w, L, config['apply_mulaw']) for w in raw], dim=0)
  per_rank = config['batch_size'] // world
  if per_rank < 1 or frames.shape[0] < config['batch_size']:
    raise SystemExit('not enough data for one global batch')
  if rank == 0:
    print(f'{frames.shape[0]} frames of {L + 1} samples; global batch {config["batch_size"]} on {world} GPU(s)')

  model = WaveNet(kernel_size=config['kernel_size'], channels=config['channels'], blocks=config['blocks'],
                  layers_per_block=config['layers_per_block'], activation=config['activation'],
                  conditioning=config['conditioning'], mapping_layers=config['mapping_layers'],
                  mapping_activation=config['mapping_activation'], dropout=config['dropout'],
                  dilation_bound=config['dilation_bound'], num_mixtures=config['num_mixtures'],
                  sampling_function=config['sampling_function'], bits=config['bits'],
                  skip_channels=config['skip_channels'], dilation_channels=config['dilation_channels'],
                  use_residual=config['use_resiudal'], use_skip=config['use_skip'],
                  final_layers_channels=config['final_layers_channels'], l2_reg_factor=config['l2_reg_factor'],
                  device=dev)
  if conditioned:                                                  # Keras builds on the first call (train.py:232-235):
    model.build([(per_rank, L, 1), (per_rank, cond.shape[1])])    # the condition width fixes the mapping net's shapes
  opt = Adam(learning_rate=config['lr'], clipnorm=1.0)           # train.py:225-226
  model.compile(optimizer=opt, metrics=[MeanSquaredError()])          # train.py:225-228
  print('Receptive field') if rank == 0 else None
  if rank == 0:
    print(model.receptive_field, ' samples')
    print(model.compute_receptive_field(config['sample_rate']), ' seconds')

  run_dir = os.path.join(config['results_dir'], run_name)
  initial_epoch = 0
  resume = io.find_resume(run_dir)
  if resume is not None:                                            # train.py:68-86
    ckpt, initial_epoch, lr = resume
    io.load_weights(model, ckpt, opt)
    opt.learning_rate = lr
    if rank == 0:
      print(f'resuming from {ckpt} (epoch {initial_epoch}, lr {lr})')

  plateau = callbacks.ReduceLROnPlateau()
  stopper = callbacks.EarlyStopping()
  nan_guard = callbacks.TerminateOnNaN()
  best = float('inf')
  n_batches = frames.shape[0] // config['batch_size']
  if config['steps_per_epoch']:
    n_batches = min(n_batches, int(config['steps_per_epoch']))
  g = torch.Generator(device='cpu').manual_seed(0)
  for epoch in range(initial_epoch, config['epochs']):
    perm = torch.randperm(frames.shape[0], generator=g)          # same permutation on every rank
    for metric in model.metrics:                                    # Keras resets every metric per epoch
      metric.reset_state()
    t0 = time.time()
    stop = False
    trips0 = model.train_guard_trips
    for i in range(n_batches):
      idx = perm[i * config['batch_size']:(i + 1) * config['batch_size']][rank * per_rank:(rank + 1) * per_rank]
      idx = idx.to(dev)
      logs = model.train_step((frames[idx], cond[idx]) if conditioned else frames[idx])
      if nan_guard.on_batch_end(logs['loss']):
        print('loss is not finite: terminating') if rank == 0 else None
        stop = True
        break
    loss = model.loss_tracker.result()
    if rank == 0:
      sps = n_batches * config['batch_size'] * L / max(time.time() - t0, 1e-9)
      extra = ''.join(f' - {k}: {v:.4f}' for k, v in logs.items() if k != 'loss')   # the compiled metrics, as Keras logs them
      # steps that left the fp16 range of the split-precision kernels and were repeated in exact fp32 (each costs a second,
      # ~2.7x slower step): shown only when there were any
      trips = model.train_guard_trips - trips0
      extra += f' - exact-fp32 repeats: {trips}' if trips else ''
      print(f'Epoch {epoch + 1}/{config["epochs"]} - loss: {loss:.4f}{extra} - lr: {opt.learning_rate:g} - {sps:,.0f} samples/s')
      if loss < best:                                               # ModelCheckpoint(save_best_only, monitor='loss')
        best = loss
        os.makedirs(run_dir, exist_ok=True)
        io.save_weights(model, os.path.join(run_dir, io.checkpoint_name(epoch + 1, opt.learning_rate,
                                                                         config['checkpoint_format'])), opt)
    plateau.on_epoch_end(loss, opt)
    if stop or stopper.on_epoch_end(loss, model):
      break

  # ---- timed generation + dumps (train.py:253-270) ----
  preview = int(config['preview_length']) or 4 * L
  if rank == 0:
    tic = time.time()
    nprev = min(config['batch_size'], 8)
    samples = model.generate(preview, batch_size=nprev, condition=cond[:nprev] if conditioned else None,
                             use_queues=config['layers_per_block'] == 1)
    torch.cuda.synchronize()
    tictoc = time.time() - tic
    print(f'Generation took {tictoc}s')
    print(f'Speed of generation was {preview / tictoc} samples/s')
    if config['apply_mulaw']:
      samples = ops.inverse_mu_law(samples)
    out_dir = os.path.join(run_dir, 'samples')
    os.makedirs(out_dir, exist_ok=True)
    arr = samples.cpu().numpy()
    np.save(os.path.join(out_dir, 'samples.npy'), arr)
    for i, s in enumerate(arr):
      with wave.open(os.path.join(out_dir, f'sample_{i}.wav'), 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(config['sample_rate']))
        w.writeframes((np.clip(s[:, 0], -1, 1) * 32767).astype('<i2').tobytes())
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
