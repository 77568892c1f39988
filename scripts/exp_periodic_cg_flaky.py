"""Diagnostic: repeat the 2x2x2 periodic emulated CG and print per-solve data."""
import sys, types
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_emulated_8ranks as T

class MP:
  def setattr(self, obj, name, val): setattr(obj, name, val)

from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.distributed import blocks, solver
from swirl_fem_amd.linalg.cg import cg
DEV = T.DEV
grid = (2, 2, 2); n, P = 2, 5
quad = Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE))
kw = dict(device=DEV, lo=0.0, hi=2*np.pi, periodic_dims=(0, 1, 2))
def rhs(x):
  return torch.sin(x[:, 0]) * torch.cos(2 * x[:, 1]) + torch.cos(x[:, 2]) * torch.sin(x[:, 0]) + 0.3
whole = blocks.build_block_partition(tuple(n * g for g in grid), P, (1, 1, 1), 0, **kw)
gm = whole.mesh
gop = FiniteElementSpace.create(gm, quad).helmholtz_operator(None)
for tol in (1e-12, 1e-10):
  xg, info_g = cg(gop.linear_operator(1.0, 1.0), gop.apply(rhs(gm.node_coords), 1.0, 0.0), M=gm.exchange, tol=tol, maxiter=2000)
  print('global', tol, info_g['num_iterations'], float(info_g['residual']))
  lookup = dict(zip(whole.global_keys.tolist(), range(gm.num_nodes)))
  for rep in range(4):
    mail = T.Mailbox(8)
    T._install_transport(MP(), mail)
    def rank_main(rank):
      part = blocks.build_block_partition(n, P, grid, rank, **kw)
      mesh = part.mesh
      fes = FiniteElementSpace.create(mesh, quad)
      op = fes.helmholtz_operator(None)
      ids = torch.as_tensor([lookup[k] for k in part.global_keys.tolist()], device=DEV)
      b_loc = op.apply(rhs(mesh.node_coords), 1.0, 0.0)
      out = []
      x, info = solver.cg(op.linear_operator(1.0, 1.0), b_loc, part.plan, tol=tol, maxiter=2000)
      out.append((float((x - xg[ids]).abs().max() / xg.abs().max()), info['num_iterations'], float(info['residual'])))
      x, info = cg(op.linear_operator(1.0, 1.0), b_loc, M=mesh.exchange, tol=tol, maxiter=2000, reduce_fn=lambda t: mail.all_reduce(t))
      out.append((float((x - xg[ids]).abs().max() / xg.abs().max()), info['num_iterations'], float(info['residual'])))
      return out
    res = T._run_ranks(mail, rank_main)
    print(rep, [res[r] for r in (0, 7)])
