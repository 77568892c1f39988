"""Times the fused Stokes kernels (D, D^T, E, C) and the 3-component stiffness
on an n^3, p = 7 triply periodic box: one line of milliseconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
n = int(os.environ.get('N', '48'))
dev = torch.device('cuda', 0)
sem = StokesSEM.create(unit_cube_mesh(n, ndim=3, a=0.0, b=2 * np.pi, periodic_dims=(0, 1, 2)), {}, order=7, device=dev)
Nv, Np = sem.velocity.mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
u = torch.randn(Nv, 3, dtype=torch.float64, device=dev)
p = torch.randn(Np, dtype=torch.float64, device=dev)
def t(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
print('n=%d  D %.3f  Dt %.3f  E %.3f  A %.3f  C %.3f ms' % (
    n, t(lambda: sem.D(u)), t(lambda: sem.Dt(p)), t(lambda: sem.E(p, dt=1e-3, time_order=3)),
    t(lambda: sem.A(u)), t(lambda: sem.C(u), 5)))
