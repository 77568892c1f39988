"""Times the pieces of the Schwarz pressure preconditioner on an n^3 p = 7
Taylor-Green stepper (HIP events, medians): env N (32)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
from swirl_fem_amd.navier_stokes import navier_stokes as ns
from swirl_fem_amd.navier_stokes import pressure_preconditioner as pc
n = int(os.environ.get('N', '32'))
sem, u, p, d = drv.taylor_green(n=n, order=7, reynolds=1600.0, dt=1e-3, steps=1,
                                time_order=3, device='cuda:0', tol=1e-6)
M = pc.make_pressure_preconditioner(sem, 'schwarz', 1e-3, 3)
E = ns._PressureOperator(sem, 1e-3, 3)
r = torch.randn_like(p)
r -= r.mean()


def timed(fn, reps=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(reps)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return round(ts[len(ts) // 2], 4)


Em, nn = M.pel.shape
rc = r.view(Em, nn).sum(dim=1)
print(json.dumps({
    'n': n, 'pressure_dofs': p.numel(), 'coarse_steps': M.coarse_iterations,
    'coarse_bounds': M.coarse_bounds,
    'E_apply_ms': timed(lambda: E(r)),
    'local_solve_ms': timed(lambda: M.local_solve(r)),
    'restrict_ms': timed(lambda: r.view(Em, nn).sum(dim=1)),
    'coarse_solve_ms': timed(lambda: M._coarse_solve(rc)),
    'project_ms': timed(lambda: M.project(r)),
    'whole_ms': timed(lambda: M(r))}))
