#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ with the CPU oracle (fp64 arithmetic, stored as
fp32/int32).  The reference itself cannot be run here (no TensorFlow), so
these vectors pin the ORACLE against drift and give the HIP path fixed data to be checked
against; they are data only (inputs + expected outputs).

  python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import wavenet_oracle as O  # noqa: E402

CASES = {
    # tiny categorical model on fused-kernel shapes (32 channels), T = 2048 as SURVEY 8c suggests
    'cat32': dict(cfg=dict(blocks=6, channels=32, skip_channels=64, dilation_bound=32,
                           final_layers_channels=[48], activation='leaky_relu', bits=8), B=2, T=2048),
    # mixture of logistics head
    'mol32': dict(cfg=dict(blocks=4, channels=32, skip_channels=32, dilation_bound=8, final_layers_channels=[32],
                           activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16),
                  B=2, T=512),
    # global conditioning
    'cond32': dict(cfg=dict(blocks=4, channels=32, skip_channels=32, dilation_bound=8, final_layers_channels=[32],
                            activation='leaky_relu', conditioning='global', mapping_layers=[8, 16],
                            mapping_activation='leaky_relu', bits=8, cond_inputs=6), B=2, T=512),
}


def main():
  # quantiser / companding table
  bits_list = [8, 16]
  g = torch.Generator().manual_seed(123)
  x = torch.cat([torch.rand(4096, generator=g) * 2.2 - 1.1,
                 torch.tensor([-1.0, 1.0, 0.0, -1e-9, 1e-9, 0.5, -0.5, 0.9999999, -0.9999999])])
  out = {'x': x.numpy()}
  for b in bits_list:
    out[f'idx{b}'] = O.quantize(x, b).numpy().astype(np.int32)
  xm = torch.linspace(-1, 1, 2001, dtype=torch.float64)
  out['mu_in'] = xm.float().numpy()
  out['mu_out'] = O.mu_law(xm.float().double()).float().numpy()
  np.savez_compressed(os.path.join(HERE, 'quantiser.npz'), **out)

  for name, case in CASES.items():
    cfg = O.OracleConfig(**case['cfg'])
    params = O.init_params(cfg, seed=11, bias_range=0.1)
    B, T = case['B'], case['T']
    x = O.synthetic_waveform(B, T + 1, seed=21)
    cond = None
    if cfg.conditioning is not None:
      cond = torch.rand(B, cfg.cond_inputs, generator=torch.Generator().manual_seed(5))
    pd = [p.double() for p in params]
    loss, reg, grads, pred = O.loss_and_grads(x.double(), pd, cfg, cond.double() if cond is not None else None)
    logits = O.model_forward(x[:, :-1].double(), pd, cfg, cond.double() if cond is not None else None,
                             return_logits=True)
    d = {'x': x.numpy(), 'loss': np.float64(loss.item()), 'pred': pred.float().numpy()[:, -64:, :],
         'logits_tail': logits.float().numpy()[:, -64:, :]}
    if cond is not None:
      d['cond'] = cond.numpy()
    for i, (p, gr) in enumerate(zip(params, grads)):
      d[f'p{i}'] = p.numpy()
      d[f'g{i}'] = gr.float().numpy()
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **d)
    print(name, 'loss', loss.item(), 'params', sum(p.numel() for p in params))


if __name__ == '__main__':
  main()
