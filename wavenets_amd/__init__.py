"""wavenets_amd: MI355X (gfx950) WaveNet training / generation hot path behind the class
surface of jirsat/wavenets (src/layers.py::WaveNetLayer, src/model.py::WaveNet)."""
from .layers import WaveNetLayer
from .model import WaveNet, MeanSquaredError
from .optim import Adam
from . import ops, dp, io, data, callbacks

__all__ = ['WaveNet', 'WaveNetLayer', 'Adam', 'MeanSquaredError', 'ops', 'dp', 'io', 'data', 'callbacks']
