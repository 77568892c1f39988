"""Generic form evaluation vs two-grid fused operator for a non-collocated
stiffness apply (the Poisson example's operator); GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core.fespace import FiniteElementSpace, grad
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.core.mesh_refiner import refine_premesh
for ndim, n, P in ((3, 32, 4), (3, 16, 8), (2, 512, 4)):
  q = (P - 1) + (ndim + 1) // 2
  mesh = refine_premesh(unit_cube_mesh(n, ndim=ndim), Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize(device='cuda:0')
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create(num_points=q, quadrature_type=NodeType.GAUSS_LEGENDRE))
  u = torch.randn(mesh.num_nodes, dtype=torch.float64, device='cuda:0')
  a = lambda f, v: lambda x: torch.vdot(grad(f)(x), grad(v)(x))
  generic = lambda: mesh.scatter(fes.local_covector(a, (fes.scalar_function(mesh.gather(u)), fes.scalar_function(None))))
  op = fes.helmholtz_operator(None)
  fused = lambda: op.apply(u, 0.0, 1.0)
  err = float((generic() - fused()).abs().max() / generic().abs().max())
  res = []
  for fn in (generic, fused):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10): fn()
    s1.record(); torch.cuda.synchronize()
    res.append(s0.elapsed_time(s1) / 10)
  print(f'RESULT {ndim}D n={n} P={P} q={q} N={mesh.num_nodes}: generic {res[0]:.3f} ms, two-grid {res[1]:.3f} ms ({res[0]/res[1]:.1f}x), rel diff {err:.1e}', flush=True)
