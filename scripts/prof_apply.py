"""Small driver for rocprofv3: a few fused applies on the config-2 mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
reps = int(os.environ.get('REPS', '5'))
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev)
mesh = part.mesh
grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
op = fes.helmholtz_operator(mesh.physical_masks.get('boundary'))
u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev)
out = torch.empty_like(u)
if os.environ.get('LAYERED', '0') == '1':     # the apply CG issues (round 4)
  ext = op.new_extended()
  for _ in range(reps):
    op.apply_layered(u, ext, 0.0, 1.0)
  out = ext
else:
  for _ in range(reps):
    op.apply(u, 0.0, 1.0, out=out)
torch.cuda.synchronize()
print('done', float(out.abs().max()))
