import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
dev = torch.device('cuda', 0)
n = 48
pm = unit_cube_mesh(n, ndim=3)
xyz = pm.node_coords + 0.2 / n * np.random.default_rng(0).uniform(-1, 1, pm.node_coords.shape)
pm = pm.replace(node_coords=xyz)
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
sem = StokesSEM.create(pm, {'boundary': (1, 0.0)}, order=7, device=dev)
op = sem._divgrad()
Nv, Np = sem.velocity.mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
w = layout.empty_component_major((Nv, 3), torch.float64, dev); w.copy_(torch.randn(Nv, 3, dtype=torch.float64, device=dev))
wi = torch.randn(Nv, 3, dtype=torch.float64, device=dev)
print('div parts', [{k: (tuple(v.shape) if hasattr(v, 'shape') else v) for k, v in q.items()} for q in op._parts_for(w, div=True)])
print('row parts', [{k: (tuple(v.shape) if hasattr(v, 'shape') else v) for k, v in q.items()} for q in op.parts])
for rep in range(2):
  print('default route (component-major)', round(t(lambda: op.div(w)), 3))
  os.environ['SFEM_STOKES_FACET_DIV'] = 'all'
  print('chain div', round(t(lambda: op.div(w)), 3))
  os.environ['SFEM_STOKES_FACET_DIV'] = 'box'
  fp = op.facet_parts; op.facet_parts = None
  print('rows parts, component-major', round(t(lambda: op.div(w)), 3))
  print('rows parts, interleaved', round(t(lambda: op.div(wi)), 3))
  op.facet_parts = fp
