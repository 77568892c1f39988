"""div / grad_t of E on a deformed (multilinear) or sheared (affine) n^3 mesh: facet-table chain kernels vs index rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
n = int(os.environ.get('N', '48'))
kind = os.environ.get('KIND', 'jitter')
dev = torch.device('cuda', 0)
pm = unit_cube_mesh(n, ndim=3)
rng = np.random.default_rng(0)
xyz = pm.node_coords.copy()
if kind == 'jitter':
  xyz = xyz + 0.2 / n * rng.uniform(-1, 1, xyz.shape)
elif kind == 'shear':
  xyz[:, 0] += 0.3 * xyz[:, 1] + 0.1 * xyz[:, 2]
pm = pm.replace(node_coords=xyz)
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
for facet in ('1', '0'):
  os.environ['SFEM_STOKES_FACET'] = facet
  sem = StokesSEM.create(pm, {'boundary': (1, 0.0)}, order=7, device=dev)
  op = sem._divgrad()
  Nv, Np = sem.velocity.mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
  p = torch.randn(Np, dtype=torch.float64, device=dev)
  scale = torch.rand(Nv, dtype=torch.float64, device=dev) + 0.5
  w = layout.empty_component_major((Nv, 3), p.dtype, dev)
  parts = op._parts_for(w)
  dots = torch.zeros(1024, dtype=torch.float64, device=dev)
  print(kind, 'n=%d' % n, 'facet' if op.facet_parts is not None else 'rows',
        [(q['geo_mode'], 'facet' if 'facet_table' in q else 'rows') for q in parts],
        'grad_t(scale) %.3f  div %.3f  div(dot) %.3f ms' % (
            t(lambda: op.grad_t(p, out=w, scale=scale)), t(lambda: op.div(w)),
            t(lambda: op.div(w, dot_with=p, dot_out=dots))), flush=True)
  del sem, op, w; torch.cuda.empty_cache()
