"""grad_t (with the per-node Q of E) and div (with the fused dot): default
launches (facet table + chains where they pay) against index rows
(SFEM_STOKES_FACET=0) over orders, precisions and geometries."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
dev = torch.device('cuda', 0)
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
CASES = [(7, 'f64', 40, 'box'), (7, 'f64', 40, 'shear'), (7, 'f64', 40, 'jitter'),
         (7, 'f32', 40, 'box'), (7, 'f32', 40, 'shear'), (7, 'f32', 40, 'jitter'),
         (6, 'f64', 44, 'box'), (6, 'f64', 44, 'jitter'), (5, 'f64', 48, 'box'), (5, 'f64', 48, 'jitter')]
for order, dts, n, kind in CASES:
  dt = torch.float64 if dts == 'f64' else torch.float32
  pm = unit_cube_mesh(n, ndim=3)
  xyz = pm.node_coords.copy()
  if kind == 'jitter':
    xyz = xyz + 0.2 / n * np.random.default_rng(0).uniform(-1, 1, xyz.shape)
  elif kind == 'shear':
    xyz[:, 0] += 0.3 * xyz[:, 1] + 0.1 * xyz[:, 2]
  pm = pm.replace(node_coords=xyz)
  row = {'order': order, 'dtype': dts, 'n': n, 'kind': kind}
  for name, env in (('default', '1'), ('rows', '0')):
    os.environ['SFEM_STOKES_FACET'] = env
    sem = StokesSEM.create(pm, {'boundary': (1, 0.0)}, order=order, device=dev, dtype=dt)
    op = sem._divgrad()
    Nv, Np = sem.velocity.mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
    p = torch.randn(Np, dtype=dt, device=dev)
    scale = torch.rand(Nv, dtype=dt, device=dev) + 0.5
    w = layout.empty_component_major((Nv, 3), dt, dev)
    dots = torch.zeros(1024, dtype=torch.float64, device=dev)
    row[name] = [round(t(lambda: op.grad_t(p, out=w, scale=scale)), 3),
                 round(t(lambda: op.div(w, dot_with=p, dot_out=dots)), 3)]
    del sem, op, w
    torch.cuda.empty_cache()
  print(json.dumps(row), flush=True)
