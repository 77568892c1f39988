"""Where a Kolmogorov-generator step spends its time (iterations, graphs on/off)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
from swirl_fem_amd.niles.datagen import datagen
cfg = datagen.DatagenConfig()
sem = datagen.create_sem(cfg, torch.device('cuda', 0))
x = sem.velocity.mesh.node_coords
u0 = datagen.u_init_fn(x)
p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=u0.dtype, device=u0.device)
for graphs in ('1', '0'):
  os.environ['SFEM_GRAPHS'] = graphs
  us, ps = (u0,) * 3, (p0,) * 3
  Cus = tuple(sem.C(u) for u in us)
  times, iters = [], []
  for _ in range(40):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f = datagen.forcing(x, us[-1], cfg.drag_coeff)
    u, p, Cu, aux = navier_stokes_step(sem, us, ps, Cus, reynolds=cfg.reynolds_number, dt=cfg.dt,
                                       time_order=3, forcing=f, tol=cfg.tol, atol=cfg.atol)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t1)
    iters.append((aux['u_star_info']['num_iterations'], aux['dp_info']['num_iterations']))
  print(json.dumps({'graphs': graphs, 'ms_per_step': 1e3 * float(np.mean(times[10:])), 'iters': iters[10:20]}), flush=True)
# pieces
def t(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
u = us[-1]; p = ps[-1]
print(json.dumps({'C': t(lambda: sem.C(u)), 'B': t(lambda: sem.B(u)), 'D': t(lambda: sem.D(u)),
                  'Dt': t(lambda: sem.Dt(p)), 'E': t(lambda: sem.E(p, cfg.dt, 3)), 'A': t(lambda: sem.A(u)),
                  'filter': t(lambda: sem.filter(u, 0.05)) if hasattr(sem, 'filter') else None}))
