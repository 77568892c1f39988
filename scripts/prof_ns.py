"""NS step timing (run on the GPU box): 3D Taylor-Green, n^3 elements, order p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
n = int(os.environ.get('N', '16')); p = int(os.environ.get('P', '7'))
steps = int(os.environ.get('STEPS', '3'))
t0 = time.time()
sem, u, pr, diag = drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=1, device='cuda:0', tol=1e-6)
torch.cuda.synchronize(); print('setup + first step', time.time() - t0)
t0 = time.time()
sem, u, pr, diag = drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=steps, device='cuda:0', tol=1e-6)
torch.cuda.synchronize(); el = time.time() - t0
print('nodes', sem.velocity.mesh.num_nodes, 'iters', diag['cg_iterations'], 'div', diag['max_divergence'])
print('total', el)
