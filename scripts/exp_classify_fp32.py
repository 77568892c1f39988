import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core import operators
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
dev = torch.device('cuda:0')
for P, dt, n in ((12, torch.float32, 32), (12, torch.float64, 24), (10, torch.float32, 32)):
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt, jitter=0.2)
  mesh = part.mesh
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)))
  op = operators.HelmholtzOperator.create(fes, mesh.physical_masks.get('boundary'), 'auto')
  print(P, dt, 'affine', op.num_affine, 'multi', op.num_multilinear, 'curved', op.num_curved,
        [(q['geo_mode'], 'facet' if 'facet_table' in q else 'rows', q['elem_list'].numel() if 'elem_list' in q else mesh.num_elements) for q in (op.facet_parts or [])])
