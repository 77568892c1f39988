import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line(
      'markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
  """The HIP library is git-ignored: build it in-tree if this checkout has not
  been built yet (hipcc cross-compiles gfx950 without a GPU).  The tests never
  fall back to anything else when it is missing."""
  lib = os.path.join(ROOT, 'swirl_fem_amd', 'libsfem_hip.so')
  if not os.path.exists(lib):
    import subprocess
    subprocess.run(['make', '-C', os.path.join(ROOT, 'swirl_fem_amd', 'csrc'),
                    '-j', str(min(8, os.cpu_count() or 1))], check=False,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
