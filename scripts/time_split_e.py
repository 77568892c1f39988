"""Times the halves of the split pressure operator against grad_t / div."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd import _ops
from swirl_fem_amd.core import layout
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
n = int(os.environ.get('N', '48'))
dev = torch.device('cuda', 0)
sem = StokesSEM.create(unit_cube_mesh(n, ndim=3, a=0.0, b=2 * np.pi, periodic_dims=(0, 1, 2)), {}, order=7, device=dev)
op = sem._divgrad()
mesh = sem.velocity.mesh
Nv, Np = mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
p = torch.randn(Np, dtype=torch.float64, device=dev)
sem.E(p, dt=1e-3, time_order=3)
scale = sem._cache[('q_scale', 1e-3, 3)]
enc, zr, so = op._split_encoding()
w = layout.empty_component_major((Nv, 3), torch.float64, dev)
out = torch.empty(Np, dtype=torch.float64, device=dev)
args = (enc, op.penc, op.parts, op.host, 3, 8)
def t(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
shared = int(((enc.to(torch.int64) & (1 << 30)) != 0).sum())
print('n=%d shared slots %.1f%%  zero range %.1f%% of nodes' % (n, 100 * shared / enc.numel(), 100 * (zr[1] - zr[0]) / Nv))
print('grad_t(cm) %.3f  div(scale) %.3f | e_first %.3f  e_second %.3f  exchange %.3f ms' % (
    t(lambda: op.grad_t(p, out=w)), t(lambda: op.div(w, scale=scale, out=out)),
    t(lambda: _ops.stokes_e_first(p, w, out, *args, zr, scale, so)),
    t(lambda: _ops.stokes_e_second(w, out, *args, scale)),
    t(lambda: sem.velocity.exchange(w, inplace=True))))
