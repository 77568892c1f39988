"""Timing experiment: order in which the fused apply visits the elements
(through the kernel's element list; mesh and node numbering unchanged)."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
n, P = 64, 8
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev)
mesh = part.mesh
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)))
E = mesh.num_elements
u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev)
out = torch.empty_like(u)
ident = np.arange(E)
def bitrev(k, bits):
  r = np.zeros_like(k)
  for b in range(bits):
    r |= ((k >> b) & 1) << (bits - 1 - b)
  return r
ijk = np.stack(np.unravel_index(ident, (n, n, n)), -1)
def morton(ijk):
  key = np.zeros(len(ijk), dtype=np.int64)
  for b in range(6):
    for d in range(3):
      key |= ((ijk[:, d] >> b) & 1) << (3 * b + d)
  return np.argsort(key, kind='stable')
orders = {
    'lexicographic (no list)': None,
    'lexicographic (list)': ident,
    'stride 4097': (ident * 4097) % E,
    'stride 65': (ident * 65) % E,
    'stride 257': (ident * 257) % E,
    'bit reversal': bitrev(ident, 18),
    'morton': morton(ijk),
    'random': np.random.default_rng(0).permutation(E),
    'x fastest': np.argsort(ijk[:, 0] + n * ijk[:, 1] + n * n * ijk[:, 2], kind='stable'),
    'checkerboard 2 colours': np.argsort((ijk.sum(1) % 2) * E + ident, kind='stable'),
    '8 colours': np.argsort(((ijk % 2) @ np.array([4, 2, 1])) * E + ident, kind='stable'),
}
for geometry in ('auto', 'stored'):
  op0 = fes.helmholtz_operator(mesh.physical_masks.get('boundary'), geometry=geometry)
  for name, order in orders.items():
    if order is None:
      op = op0
    else:
      lst = torch.as_tensor(order.astype(np.int32), device=dev).contiguous()
      assert len(op0.parts) == 1
      op = dataclasses.replace(op0, parts=[dict(op0.parts[0], elem_list=lst)])
    for _ in range(3):
      op.apply(u, 0.0, 1.0, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    lo, hi = op.zero_range
    for a, b in ev:
      out[lo:hi].zero_()
      a.record(); op.apply(u, 0.0, 1.0, out=out, zero=False); b.record()
    torch.cuda.synchronize()
    print(f'{geometry:7s} {name:28s} {np.mean([a.elapsed_time(b) for a, b in ev]):.4f} ms', flush=True)
  ref = op0.apply(u, 0.0, 1.0)
  chk = op.apply(u, 0.0, 1.0)
  print('   max rel diff last order vs default', float((ref - chk).abs().max() / ref.abs().max()))
  del op0, op
