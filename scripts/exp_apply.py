"""Timing experiments on the fused apply (run on the GPU box)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd import _ops

n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev)
mesh = part.mesh
grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
op = fes.helmholtz_operator(mesh.physical_masks.get('boundary'))
u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev)
out = torch.empty_like(u)
E, nn = mesh.elements.shape
alg = 4 * E * nn + 8 * mesh.num_nodes * 2 + 48 * E * nn
print('shared nodes', mesh.assembly_plan().num_shared, 'zero range', op.zero_range, 'N', mesh.num_nodes)

def timeit(label, fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s0.record()
  for _ in range(reps): fn()
  s1.record(); torch.cuda.synchronize()
  ms = s0.elapsed_time(s1) / reps
  print(f'{label:40s} {ms:8.3f} ms  {alg/ms/1e6:8.1f} GB/s alg')
  return ms

op_pp = fes.helmholtz_operator(mesh.physical_masks.get('boundary'), exploit_affine=False)
print('affine elements', op.num_affine, 'of', E)
timeit('kernel affine path', lambda: op.apply(u, 0.0, 1.0, out=out, zero=False))
timeit('kernel per-point path', lambda: op_pp.apply(u, 0.0, 1.0, out=out, zero=False))
timeit('kernel affine path', lambda: op.apply(u, 0.0, 1.0, out=out, zero=False))
timeit('apply affine (memset+kernel)', lambda: op.apply(u, 0.0, 1.0, out=out))
timeit('helmholtz affine', lambda: op.apply(u, 0.7, 1.0, out=out, zero=False))
timeit('mass only affine', lambda: op.apply(u, 1.0, 0.0, out=out, zero=False))
timeit('mass only per-point', lambda: op_pp.apply(u, 1.0, 0.0, out=out, zero=False))
a = torch.empty(mesh.num_nodes * 10, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
ms = timeit('torch copy 7.2GB+7.2GB', lambda: b.copy_(a))
print('copy GB/s', 2 * a.numel() * 8 / ms / 1e6)
