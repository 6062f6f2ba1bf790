"""Pure-Python structure of the network: argument validation, dilation schedule, receptive
field and the parameter list in Keras creation order.  No GPU, no library needed.

Follows src/model.py:52-70 (validation), :79-81 (schedule), :84-119 (layers), :122 (receptive
field), :124-149 (mapping net) and src/layers.py:49-120 (per-block layers).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

SUPPORTED_ACTIVATIONS = (None, 'linear', 'relu', 'leaky_relu', 'tanh', 'sigmoid', 'elu')


@dataclass
class ModelSpec:
  kernel_size: int
  channels: int
  blocks: int
  layers_per_block: int
  activation: Optional[str]
  conditioning: Optional[str]
  mapping_layers: List[int]
  mapping_activation: Optional[str]
  dropout: float
  dilation_bound: int
  num_mixtures: Optional[int]
  sampling_function: str
  bits: int
  skip_channels: Optional[int]
  dilation_channels: Optional[int]
  use_residual: bool
  use_skip: bool
  final_layers_channels: List[int]
  l2_reg_factor: float

  @property
  def D(self) -> int:
    return self.channels if self.dilation_channels is None else self.dilation_channels

  @property
  def out_channels(self) -> int:
    return 3 * self.num_mixtures if self.num_mixtures is not None else 2 ** self.bits

  @property
  def dilations(self) -> List[int]:
    max_power = int(math.log(self.dilation_bound, self.kernel_size))
    return [self.kernel_size ** (i % max_power)
            for i in range(self.layers_per_block * self.blocks)]

  @property
  def receptive_field(self) -> int:
    return 1 + sum(self.dilations) * (self.kernel_size - 1) + 1

  def cond_channels(self, cond_inputs: int) -> int:
    return self.mapping_layers[-1] if self.mapping_layers else cond_inputs

  def param_shapes(self, cond_inputs: int = 0) -> List[Tuple[str, Tuple[int, ...]]]:
    k, R, D, S = self.kernel_size, self.channels, self.D, self.skip_channels
    out = [('causal/kernel', (k, 1, R)), ('causal/bias', (R,))]
    cc = self.cond_channels(cond_inputs)
    for b in range(self.blocks):
      cin = R
      for i in range(self.layers_per_block):
        cout = 2 * D if i == self.layers_per_block - 1 else D
        out.append((f'block{b}/dil{i}/kernel', (k, cin, cout)))
        out.append((f'block{b}/dil{i}/bias', (cout,)))
        cin = cout
      out.append((f'block{b}/conv1/kernel', (1, D, R)))
      out.append((f'block{b}/conv1/bias', (R,)))
      if S is not None:
        out.append((f'block{b}/conv_skip/kernel', (1, D, S)))
        out.append((f'block{b}/conv_skip/bias', (S,)))
      if self.conditioning is not None:
        out.append((f'block{b}/conv_cond/kernel', (1, cc, 2 * D)))
        out.append((f'block{b}/conv_cond/bias', (2 * D,)))
    cprev = (S if S is not None else R) if self.use_skip else R
    for i, ch in enumerate(list(self.final_layers_channels) + [self.out_channels]):
      out.append((f'final{i}/kernel', (1, cprev, ch)))
      out.append((f'final{i}/bias', (ch,)))
      cprev = ch
    if self.conditioning == 'global':
      cin = cond_inputs
      for j, w in enumerate(self.mapping_layers):
        out.append((f'mapping{j}/kernel', (cin, w)))
        out.append((f'mapping{j}/bias', (w,)))
        cin = w
    return out


def validate(kernel_size, channels, blocks, layers_per_block, activation, conditioning,
             mapping_layers, mapping_activation, dropout, dilation_bound, num_mixtures,
             sampling_function, bits, skip_channels, dilation_channels, use_residual, use_skip,
             final_layers_channels, l2_reg_factor) -> ModelSpec:
  """Constructor checks of the reference, same messages (src/model.py:52-70,125-130)."""
  if conditioning not in ['global', 'local', None]:
    raise ValueError("Conditioning must be 'global', 'local' or None.")
  if kernel_size < 2:
    raise ValueError('Kernel size must be at least 2.')
  if math.log(dilation_bound, kernel_size) % 1 != 0:
    raise ValueError('dilation bound must be power of kernel_size.')
  if layers_per_block < 1:
    raise ValueError('Layers per block must be at least 1.')
  if blocks < 1:
    raise ValueError('Blocks must be at least 1.')
  if num_mixtures is not None and num_mixtures < 1:
    raise ValueError('Number of mixtures must be at least 1 or None.')
  if dropout < 0 or dropout > 1:
    raise ValueError('Dropout must be between 0 and 1.')
  if sampling_function not in ['categorical', 'logistic', 'gaussian']:
    raise ValueError('Sampling function must be categorical, ' + 'logistic or gaussian.')
  if sampling_function == 'categorical' and num_mixtures is not None:
    raise ValueError('Categorical sampling cannot be used with mixtures.')
  if mapping_layers is None:
    mapping_layers = []
  elif isinstance(mapping_layers, int):
    mapping_layers = [mapping_layers]
  elif not isinstance(mapping_layers, list):
    raise ValueError('Mapping layers must be a list of integers.')
  # the reference crashes on final_layers_channels=None (src/model.py:111); callers always pass a
  # list (train.py:223).  None is accepted here as the empty list.
  if final_layers_channels is None:
    final_layers_channels = []
  if sampling_function != 'categorical' and num_mixtures is None:
    raise ValueError('Mixture sampling functions need num_mixtures.')
  for act in (activation, mapping_activation):
    if act not in SUPPORTED_ACTIVATIONS:
      raise NotImplementedError(f'activation {act!r} is not supported by the gfx950 kernels '
                                f'(supported: {SUPPORTED_ACTIVATIONS})')
  if conditioning == 'local':
    # broken in the reference itself (Conv1D(kernel=1, ...) src/model.py:136-137; README "not tested")
    raise NotImplementedError('local conditioning is not implemented')
  return ModelSpec(kernel_size, channels, blocks, layers_per_block, activation, conditioning,
                   list(mapping_layers), mapping_activation, float(dropout), dilation_bound,
                   num_mixtures, sampling_function, bits, skip_channels, dilation_channels,
                   bool(use_residual), bool(use_skip), list(final_layers_channels),
                   float(l2_reg_factor or 0.0))
