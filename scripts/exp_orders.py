"""Fused apply timing across polynomial orders (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
cases = [(3, 64), (4, 64), (5, 48), (6, 48), (7, 40), (8, 32), (9, 32), (10, 24)]
for P, n in cases:
  for dt in (torch.float64, torch.float32):
    dev = torch.device('cuda:0')
    part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt)
    mesh = part.mesh
    grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
    fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
    bm = mesh.physical_masks.get('boundary')
    u = torch.randn(mesh.num_nodes, dtype=dt, device=dev); out = torch.empty_like(u)
    res = []
    for geo in ('auto', 'stored'):
      op = fes.helmholtz_operator(bm, geo)
      for _ in range(3): op.apply(u, 0.0, 1.0, out=out)
      torch.cuda.synchronize()
      s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s0.record()
      for _ in range(20): op.apply(u, 0.0, 1.0, out=out, zero=False)
      s1.record(); torch.cuda.synchronize()
      ms = s0.elapsed_time(s1) / 20
      res.append(f'{geo} {ms:.4f} ms {mesh.num_nodes/ms/1e6:6.1f} GDOF/s')
      del op
    print(f'RESULT P={P} n={n} {str(dt)[6:]:8s} N={mesh.num_nodes:9d} ' + ' | '.join(res), flush=True)
    del fes, mesh, part, u, out; torch.cuda.empty_cache()
