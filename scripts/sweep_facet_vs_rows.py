"""Default (facet table / chains) against index rows (SFEM_FACET=0) over the
orders, precisions, geometries and field shapes the facet kernels cover:
one line per case, apply time in ms (HIP events, median of REPS)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core import layout, operators
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
dev = torch.device('cuda:0')
reps = int(os.environ.get('REPS', '15'))
CASES = [  # P, dtype, n, jitter, geometry, mass, ncomp
    (8, 'f64', 48, 0.0, 'auto', 0.5, 1), (8, 'f64', 48, 0.2, 'auto', 0.0, 1),
    (8, 'f64', 48, 0.2, 'auto', 0.5, 1), (8, 'f64', 48, 0.0, 'stored', 0.5, 1),
    (8, 'f64', 48, 0.0, 'auto', 0.5, 3), (8, 'f64', 48, 0.2, 'auto', 0.5, 3),
    (8, 'f64', 48, 0.0, 'stored', 0.5, 3),
    (8, 'f32', 48, 0.0, 'auto', 0.0, 1), (8, 'f32', 48, 0.2, 'auto', 0.0, 1),
    (8, 'f32', 48, 0.0, 'stored', 0.0, 1),
    (7, 'f64', 48, 0.0, 'auto', 0.0, 1), (7, 'f64', 48, 0.2, 'auto', 0.0, 1),
    (6, 'f64', 56, 0.0, 'auto', 0.0, 1), (6, 'f64', 56, 0.2, 'auto', 0.0, 1),
    (10, 'f64', 32, 0.0, 'auto', 0.0, 1), (10, 'f64', 32, 0.2, 'auto', 0.0, 1),
    (12, 'f64', 28, 0.0, 'auto', 0.0, 1), (12, 'f32', 32, 0.2, 'auto', 0.5, 1),
    (12, 'f32', 32, 0.0, 'stored', 0.5, 1), (9, 'f32', 40, 0.0, 'auto', 0.0, 1),
]
def time(op, u, mass):
  out = torch.empty_like(u)
  for _ in range(3): op.apply(u, mass, 1.0, out=out)
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in ev:
    a.record(); op.apply(u, mass, 1.0, out=out); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
for P, dts, n, jitter, geometry, mass, nc in CASES:
  dt = torch.float64 if dts == 'f64' else torch.float32
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt, jitter=jitter)
  mesh = part.mesh
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)))
  u = torch.randn(mesh.num_nodes, dtype=dt, device=dev) if nc == 1 else \
      layout.component_major(torch.randn(mesh.num_nodes, nc, dtype=dt, device=dev))
  row = {'P': P, 'dtype': dts, 'n': n, 'jitter': jitter, 'geometry': geometry, 'mass': mass, 'ncomp': nc}
  for name, env in (('default', '1'), ('rows', '0')):
    os.environ['SFEM_FACET'] = env
    op = operators.HelmholtzOperator.create(fes, mesh.physical_masks.get('boundary'), geometry)
    row[name] = round(time(op, u, mass), 4)
    if name == 'default':
      row['kernel'] = op.kernel_name(mass, 1.0)[:70]
    del op
  row['ratio'] = round(row['default'] / row['rows'], 3)
  print(json.dumps(row), flush=True)
  del part, mesh, fes, u
  torch.cuda.empty_cache()
