import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
  # gpu tests are skipped (not failed) when somebody runs the whole suite on a CPU box
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU in this container')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _debug_knobs_from_env():
  """WN_KNOBS="4=1,16=3": run the suite under non-default kernel variants (tools/ A/B runs check parity this way)."""
  spec = os.environ.get('WN_KNOBS', '')
  if spec:
    from wavenets_amd import _lib
    for item in spec.split(','):
      k, v = item.split('=')
      _lib.lib().wn_debug_set(int(k), int(v))
  yield
