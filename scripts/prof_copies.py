import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
from swirl_fem_amd.navier_stokes import navier_stokes as ns
from swirl_fem_amd.linalg.cg import CGRunner
from swirl_fem_amd import _lib
sem, u, p, d = drv.taylor_green(n=16, order=7, reynolds=1600.0, dt=1e-3, steps=0,
                                time_order=3, device='cuda:0', tol=1e-6)
E = ns._PressureOperator(sem, 1e-3, 3)
b = torch.randn_like(p)
which = os.environ.get('WHICH', 'E')
parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device='cuda:0')
if which == 'E':
  for _ in range(50):
    E.apply_with_dot(b, parts)
elif which == 'grad_t':
  op = sem._divgrad()
  for _ in range(50):
    op.grad_t(b, component_major=True)
elif which == 'cg':
  run = CGRunner(E, E(b), tol=0.0, maxiter=10 ** 6, M=ns._NullspaceProjection(sem))
  for _ in range(50):
    run.step()
torch.cuda.synchronize()
