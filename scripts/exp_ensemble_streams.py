"""Feasibility: an ensemble of Kolmogorov-generator members on concurrent HIP
streams (one host thread + one stream per member) against the members one
after the other.  Every member has its own StokesSEM here, so nothing is
shared but the device.
  B=8 STEPS=40 python scripts/exp_ensemble_streams.py
"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.niles.datagen import datagen

B = int(os.environ.get('B', '8')); STEPS = int(os.environ.get('STEPS', '40'))
WARM = int(os.environ.get('WARM', '10'))
dev = torch.device('cuda', 0)
cfg = datagen.DatagenConfig()


def make_member(seed):
  sem = datagen.create_sem(cfg, dev)
  x = sem.velocity.mesh.node_coords
  u0 = datagen.u_init_fn(x) * (1.0 + 0.05 * seed)
  p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=u0.dtype, device=dev)
  us, ps = (u0,) * 3, (p0,) * 3
  return dict(sem=sem, us=us, ps=ps, Cus=tuple(sem.C(u) for u in us))


def advance(m, steps):
  for _ in range(steps):
    u, p, Cu = datagen._solve_one_step(m['sem'], m['us'], m['ps'], m['Cus'], cfg)
    m['us'], m['ps'], m['Cus'] = m['us'][1:] + (u,), m['ps'][1:] + (p,), m['Cus'][1:] + (Cu,)


members = [make_member(b) for b in range(B)]
for m in members:
  advance(m, WARM)
torch.cuda.synchronize()

t0 = time.perf_counter(); advance(members[0], STEPS); torch.cuda.synchronize()
single = (time.perf_counter() - t0) / STEPS

t0 = time.perf_counter()
for m in members:
  advance(m, STEPS)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / STEPS

streams = [torch.cuda.Stream(dev) for _ in members]


def worker(m, s):
  with torch.cuda.stream(s):
    advance(m, STEPS)
    s.synchronize()


for rep in range(2):                       # first pass captures graphs per stream
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ts = [threading.Thread(target=worker, args=(m, s)) for m, s in zip(members, streams)]
  for t in ts: t.start()
  for t in ts: t.join()
  torch.cuda.synchronize()
  conc = (time.perf_counter() - t0) / STEPS
print(json.dumps({'members': B, 'single_ms': 1e3 * single, 'serial_ms': 1e3 * serial,
                  'concurrent_ms': 1e3 * conc, 'pc': os.environ.get('SFEM_PRESSURE_PC', '')}))
