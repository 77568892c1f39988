"""Wall time per step of the 2D lid-driven cavity (config 3), eager vs graph."""
import os, subprocess, sys
if os.environ.get('CHILD'):
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import time, torch
  from swirl_fem_amd.examples import navier_stokes_driver as drv
  import inspect
  kw = dict(n=32, order=5, reynolds=100.0, dt=1e-3, device='cuda:0')
  drv.lid_driven_cavity(steps=2, **kw)
  torch.cuda.synchronize(); t0 = time.time()
  out = drv.lid_driven_cavity(steps=2, **kw)
  torch.cuda.synchronize(); t2 = time.time() - t0
  t0 = time.time()
  out = drv.lid_driven_cavity(steps=12, **kw)
  torch.cuda.synchronize(); t12 = time.time() - t0
  print('RESULT graphs=%s ms/step=%.2f (setup excluded by differencing) iters=%s' % (
      os.environ.get('SFEM_GRAPHS', '1'), (t12 - t2) / 10 * 1e3, out[-1]['cg_iterations'][-3:]))
else:
  for g in ('0', '1'):
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CHILD='1', SFEM_GRAPHS=g), capture_output=True, text=True, timeout=900)
    print([l for l in r.stdout.splitlines() if l.startswith('RESULT')] or r.stderr[-2500:], flush=True)
