"""Where a Navier-Stokes step spends its wall time (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
from swirl_fem_amd.linalg import cg as cg_mod
from swirl_fem_amd.navier_stokes import navier_stokes as ns
n = int(os.environ.get('N', '16')); p = int(os.environ.get('P', '7'))
orig_capture = cg_mod.CGRunner.capture
def capture(self):
  ok = orig_capture(self); print('capture ->', ok); return ok
cg_mod.CGRunner.capture = capture
T = {}
def timed(name, fn):
  def w(*a, **k):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(*a, **k); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.time() - t0; return r
  return w
ns.cg = timed('cg', ns.cg)
for name in ('C', 'filter'):
  setattr(ns.StokesSEM, name, timed(name, getattr(ns.StokesSEM, name)))
drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=1, device='cuda:0', tol=1e-6)
T.clear(); torch.cuda.synchronize(); t0 = time.time()
drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=3, device='cuda:0', tol=1e-6)
torch.cuda.synchronize(); print('total', time.time() - t0)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f'{k:10s} {v*1e3:8.1f} ms')
