"""30 iterations of the Schwarz-preconditioned pressure CG on an n^3 p = 7
Taylor-Green stepper, for a kernel trace (env N, default 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
from swirl_fem_amd.navier_stokes import navier_stokes as ns
from swirl_fem_amd.navier_stokes import pressure_preconditioner as pc
from swirl_fem_amd.linalg.cg import CGRunner
n = int(os.environ.get('N', '32'))
sem, u, p, d = drv.taylor_green(n=n, order=7, reynolds=1600.0, dt=1e-3, steps=1,
                                time_order=3, device='cuda:0', tol=1e-6)
M = pc.make_pressure_preconditioner(sem, 'schwarz', 1e-3, 3)
E = ns._PressureOperator(sem, 1e-3, 3)
b = E(torch.randn_like(p))
run = CGRunner(E, b, tol=0.0, maxiter=10 ** 6, M=M)
torch.cuda.synchronize()
print('MARK start', flush=True)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(30):
  run.step()
t1.record(); torch.cuda.synchronize()
print('ms per iteration', t0.elapsed_time(t1) / 30)
