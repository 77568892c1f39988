"""p=11 fp32 Helmholtz apply timing (GPU box): python scripts/exp_p11.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
for P, n, dt in ((12, 32, torch.float32), (10, 32, torch.float32), (9, 32, torch.float64), (12, 24, torch.float64)):
  dev = torch.device('cuda:0')
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt)
  mesh = part.mesh
  grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
  bm = mesh.physical_masks.get('boundary')
  u = torch.randn(mesh.num_nodes, dtype=dt, device=dev); out = torch.empty_like(u)
  for geo in ('auto', 'stored'):
    op = fes.helmholtz_operator(bm, geo)
    for _ in range(3): op.apply(u, 0.5, 1.0, out=out)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(20): op.apply(u, 0.5, 1.0, out=out, zero=False)
    s1.record(); torch.cuda.synchronize()
    ms = s0.elapsed_time(s1) / 20
    print(f'RESULT P={P} n={n} {dt} geo={geo} kernel_ms={ms:.4f} GDOF/s={mesh.num_nodes/ms/1e6:.1f} ns/kslot={ms*1e6/(mesh.elements.numel()/1e3):.2f}', flush=True)
    del op
  del fes, mesh, part, u, out; torch.cuda.empty_cache()
