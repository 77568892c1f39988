import json, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
from swirl_fem_amd.niles.datagen import datagen
from swirl_fem_amd.linalg import cg as cgmod
cfg = datagen.DatagenConfig()
sem = datagen.create_sem(cfg, torch.device('cuda', 0))
x = sem.velocity.mesh.node_coords
u0 = datagen.u_init_fn(x)
p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=u0.dtype, device=u0.device)
acc = {'capture': 0.0, 'init': 0.0, 'n': 0}
orig_cap = cgmod.CGRunner.capture
orig_init = cgmod.CGRunner.__init__
def cap(self):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  r = orig_cap(self)
  torch.cuda.synchronize(); acc['capture'] += time.perf_counter() - t0; acc['n'] += 1
  return r
def init(self, *a, **k):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  orig_init(self, *a, **k)
  torch.cuda.synchronize(); acc['init'] += time.perf_counter() - t0
cgmod.CGRunner.capture = cap
cgmod.CGRunner.__init__ = init
us, ps = (u0,) * 3, (p0,) * 3
Cus = tuple(sem.C(u) for u in us)
times = []
for it in range(30):
  if it == 10:
    acc.update(capture=0.0, init=0.0, n=0)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  f = datagen.forcing(x, us[-1], cfg.drag_coeff)
  u, p, Cu, aux = navier_stokes_step(sem, us, ps, Cus, reynolds=cfg.reynolds_number, dt=cfg.dt,
                                     time_order=3, forcing=f, tol=cfg.tol, atol=cfg.atol)
  us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
  torch.cuda.synchronize(); times.append(time.perf_counter() - t1)
print(json.dumps({'ms_per_step': 1e3 * float(np.mean(times[10:])), 'capture_ms_per_step': 1e3 * acc['capture'] / 20,
                  'init_ms_per_step': 1e3 * acc['init'] / 20, 'captures_per_step': acc['n'] / 20}))
