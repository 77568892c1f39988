"""Wall time per Navier-Stokes step, eager vs HIP-graph CG (GPU box)."""
import os, subprocess, sys
if os.environ.get('CHILD'):
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import time, torch
  from swirl_fem_amd.examples import navier_stokes_driver as drv
  n = int(os.environ.get('N', '16')); p = int(os.environ.get('P', '7'))
  drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=1, device='cuda:0', tol=1e-6)
  torch.cuda.synchronize(); t0 = time.time()
  sem, u, pr, diag = drv.taylor_green(n=n, order=p, reynolds=1600.0, dt=1e-3, steps=3, device='cuda:0', tol=1e-6)
  torch.cuda.synchronize()
  print('RESULT graphs=%s n=%d p=%d ms/step=%.1f iters=%s' % (os.environ.get('SFEM_GRAPHS', '1'), n, p, (time.time() - t0) / 3 * 1e3, diag['cg_iterations']))
else:
  for n in ('16', '32'):
    for g in ('0', '1'):
      r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CHILD='1', SFEM_GRAPHS=g, N=n), capture_output=True, text=True, timeout=900)
      print([l for l in r.stdout.splitlines() if l.startswith('RESULT')] or r.stderr[-1500:], flush=True)
