"""Which lines of the stepper copy device tensors: wraps Tensor.clone / .contiguous (when it copies) / .copy_ and
counts bytes per calling line over one Taylor-Green step.  N=32 python scripts/count_copies.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv

N = int(os.environ.get('N', '32'))
counts = collections.Counter(); sizes = collections.Counter()
active = [False]


def where():
  for fr in reversed(traceback.extract_stack()[:-2]):
    if 'swirl_fem_amd' in fr.filename:
      return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:60]}'
  return 'other'


def wrap(name, copies):
  orig = getattr(torch.Tensor, name)
  def f(self, *a, **k):
    if active[0] and self.is_cuda and copies(self, *a, **k):
      w = name + ' @ ' + where(); counts[w] += 1; sizes[w] += self.numel() * self.element_size()
    return orig(self, *a, **k)
  setattr(torch.Tensor, name, f)

wrap('clone', lambda s, *a, **k: True)
wrap('contiguous', lambda s, *a, **k: not s.is_contiguous())
wrap('copy_', lambda s, *a, **k: True)
orig_step = drv.navier_stokes_step
calls = [0]
def step(*a, **k):
  calls[0] += 1
  active[0] = calls[0] == 3
  out = orig_step(*a, **k)
  active[0] = False
  return out
drv.navier_stokes_step = step
drv.taylor_green(n=N, order=7, reynolds=1600.0, dt=1e-3, steps=3, time_order=3, device='cuda:0', tol=1e-6)
tot = sum(sizes.values())
print(f'{sum(counts.values())} copies, {tot / 1e9:.2f} GB in one step')
for w, b in sizes.most_common(18):
  print(f'{b / 1e9:8.2f} GB {counts[w]:5d} x  {w}')
