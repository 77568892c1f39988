"""Does the step time drift over a longer run / after other cases?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
def run(n, steps):
  prof = {}
  sem, u, p, diag = drv.taylor_green(n=n, order=7, reynolds=1600.0, dt=1e-3, steps=steps, tol=1e-6, device='cuda:0', profile=prof)
  print(n, [round(1e3 * t) for t in prof['step_s']], 'mem GB alloc/reserved', round(torch.cuda.memory_allocated() / 1e9, 1), round(torch.cuda.memory_reserved() / 1e9, 1), flush=True)
  del sem, u, p
  gc.collect(); torch.cuda.empty_cache()
  print('   after cleanup: alloc/reserved', round(torch.cuda.memory_allocated() / 1e9, 1), round(torch.cuda.memory_reserved() / 1e9, 1), flush=True)
run(32, 10)
run(48, 4)
run(32, 4)
run(64, 3)
