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
  """The HIP library is git-ignored: (re)build it in-tree before the tests --
  `make` is an incremental no-op when the library is current, and a stale
  library from older sources must never be what gets tested (hipcc
  cross-compiles gfx950 without a GPU).  The tests never fall back to anything
  else when it is missing."""
  import subprocess
  res = subprocess.run(['make', '-C', os.path.join(ROOT, 'swirl_fem_amd', 'csrc'),
                        '-j', str(min(8, os.cpu_count() or 1))], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                       text=True)
  if res.returncode != 0:
    raise pytest.UsageError('building libsfem_hip.so failed:\n' +
                            res.stderr[-4000:])


def pytest_collection_modifyitems(config, items):
  """`gpu` tests are skipped (not failed) on a host without a GPU."""
  if not any('gpu' in item.keywords for item in items):
    return
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='needs a real MI355X (no GPU visible)')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
