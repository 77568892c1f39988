"""Fused element operators (build-side fast paths behind the reference API).

`HelmholtzOperator` is the collocated  H = lambda0 * B + lambda1 * A  operator
of the reference's callers -- examples/poisson.py:141-154 (A, B),
navier_stokes/navier_stokes.py:220-236, :295-307, :431 -- as ONE kernel:
gather, sum-factorised apply with 6 (+1) stored geometric factors per point,
Dirichlet mask and direct-stiffness summation (`sfem_helmholtz_apply`).
"""

from __future__ import annotations

import dataclasses

import numpy as np
import torch

from swirl_fem_amd import _ops

MAX_FUSED_P = 12


def supports_fused(fespace) -> str | None:
  """None if the fused kernel applies, else the reason it does not."""
  P = fespace.mesh.gridpoints_1d.num_points
  if not fespace.is_collocated:
    return 'quadrature points differ from the nodes'
  if fespace.mesh.ndim not in (2, 3):
    return f'ndim={fespace.mesh.ndim}'
  if not 2 <= P <= MAX_FUSED_P:
    return f'P={P} outside 2..{MAX_FUSED_P}'
  d = fespace.interpolator._differentiation_matrix_1d()
  if not np.allclose(d[::-1, ::-1], -d, rtol=0,
                     atol=1e-12 * max(1.0, np.abs(d).max())):
    return 'node set is not symmetric about 0'
  return None


@dataclasses.dataclass(eq=False)
class HelmholtzOperator:
  fespace: object
  geo: torch.Tensor              # (E, ng + 1, Q)
  enc: torch.Tensor              # (E, n) encoded indices
  dmat: np.ndarray               # (P, P) host
  zero_range: tuple

  @classmethod
  def create(cls, fespace, dirichlet_mask=None) -> 'HelmholtzOperator':
    why = supports_fused(fespace)
    if why is not None:
      raise NotImplementedError(f'fused Helmholtz kernel unavailable: {why}')
    mesh = fespace.mesh
    w = torch.as_tensor(fespace.quadrature.weights_nd(mesh.ndim),
                        dtype=fespace.dtype, device=fespace.device)
    geo = _ops.helmholtz_setup(fespace.invjacs, fespace.jacdets, w)
    plan = mesh.assembly_plan()
    mask = None
    if dirichlet_mask is not None:
      mask = torch.as_tensor(dirichlet_mask, device=fespace.device)
      mask = (mask != 0).to(torch.uint8).contiguous()
    enc = _ops.encode_elements(mesh.elements, mask, plan.multiplicity)
    return cls(fespace=fespace, geo=geo, enc=enc,
               dmat=fespace.interpolator._differentiation_matrix_1d(),
               zero_range=plan.zero_range)

  def apply(self, u, lambda0=0.0, lambda1=1.0, out=None):
    """u (N,) or (N, nc) -> mask * scatter((l0 B + l1 A)_local(gather(u)))."""
    mesh = self.fespace.mesh
    if u.shape[0] != mesh.num_nodes:
      raise ValueError(f'expected {mesh.num_nodes} nodal values, got '
                       f'{tuple(u.shape)}')
    u = u.to(self.fespace.dtype).contiguous()
    if out is None:
      out = torch.empty_like(u)
    return _ops.helmholtz_apply(
        u, out, self.enc, self.geo, self.dmat, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1, self.zero_range)

  def apply_local(self, u_local, lambda0=0.0, lambda1=1.0):
    """Element-local action (E, n[, nc]) -> (E, n[, nc]); no gather/scatter."""
    mesh = self.fespace.mesh
    return _ops.helmholtz_local(
        u_local.to(self.fespace.dtype), self.geo, self.dmat, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1)
