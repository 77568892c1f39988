import json, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core import layout, operators
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
dev = torch.device('cuda:0')
def time(op, u, mass, reps=15):
  out = torch.empty_like(u)
  for _ in range(3): op.apply(u, mass, 1.0, out=out)
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in ev:
    a.record(); op.apply(u, mass, 1.0, out=out); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
for P, dts, n, jitter, geometry, mass, nc in [(8, 'f64', 48, 0.0, 'stored', 0.5, 3), (8, 'f64', 48, 0.2, 'auto', 0.5, 3), (8, 'f64', 48, 0.0, 'auto', 0.5, 3), (8, 'f32', 48, 0.0, 'stored', 0.5, 3)]:
  dt = torch.float64 if dts == 'f64' else torch.float32
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt, jitter=jitter)
  mesh = part.mesh
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)))
  u = layout.component_major(torch.randn(mesh.num_nodes, nc, dtype=dt, device=dev))
  row = {'case': (P, dts, jitter, geometry)}
  op = operators.HelmholtzOperator.create(fes, mesh.physical_masks.get('boundary'), geometry)
  for name, env in (('vector_chains', '1'), ('one_element_vector', '0')):
    os.environ['SFEM_CHAIN_VECTOR'] = env
    row[name] = round(time(op, u, mass), 4)
  os.environ['SFEM_FACET'] = '0'
  op = operators.HelmholtzOperator.create(fes, mesh.physical_masks.get('boundary'), geometry)
  row['rows'] = round(time(op, u, mass), 4)
  del os.environ['SFEM_FACET']
  print(json.dumps(row), flush=True)
