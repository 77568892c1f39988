"""Timing of the fused Stokes D / D^T kernels at scale (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout, operators
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.core.mesh_refiner import refine_premesh
n = int(os.environ.get('N', '48')); P = int(os.environ.get('P', '8'))
dev = 'cuda:0'
pm = unit_cube_mesh(n, ndim=3)
quad = Quadrature1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
vsp = FiniteElementSpace.create(refine_premesh(pm, Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize(device=dev), quad)
psp = FiniteElementSpace.create(refine_premesh(pm, Nodes1D.create(P - 2, NodeType.GAUSS_LEGENDRE)).finalize(device=dev), quad)
N, E = vsp.mesh.num_nodes, vsp.mesh.num_elements
nn, npp = P ** 3, (P - 2) ** 3
print('N', N, 'E', E, 'Np', psp.mesh.num_nodes)
def timeit(label, fn, bytes_, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s0.record()
  for _ in range(reps): fn()
  s1.record(); torch.cuda.synchronize()
  ms = s0.elapsed_time(s1) / reps
  print(f'{label:46s} {ms:8.3f} ms  {bytes_/ms/1e6:8.1f} GB/s (model)  {3*N/ms/1e6:7.2f} GDOF/s')
bm = vsp.mesh.physical_masks['boundary']
for geo in ('auto', 'multilinear', 'stored'):
  op = operators.StokesDivGrad.create(vsp, psp, bm, geo)
  ngeo = 9 * 8 * E * nn     # stored-factor model: 9 weighted cofactors per point
  base = 4 * E * nn + 3 * 8 * N + 8 * E * npp
  for cm in (False, True):
    u = torch.randn(N, 3, dtype=torch.float64, device=dev)
    if cm: u = layout.component_major(u)
    p = torch.randn(psp.mesh.num_nodes, dtype=torch.float64, device=dev)
    sc = torch.rand_like(u) + 0.5
    pout = torch.empty_like(p); out = torch.empty_like(u)
    tag = f'{geo} {"component-major" if cm else "row-major"}'
    timeit(f'div     {tag}', lambda: op.div(u, out=pout), base + ngeo)
    timeit(f'div*s   {tag}', lambda: op.div(u, scale=sc, out=pout), base + ngeo + 3 * 8 * N)
    timeit(f'grad_t  {tag}', lambda: op.grad_t(p, out=out), base + ngeo)
