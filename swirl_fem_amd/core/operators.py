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


AFFINE_RTOL = 1e-11


def _affine_mask(fespace):
  """Elements whose Jacobian is constant over the quadrature points."""
  ij, jd = fespace.invjacs, fespace.jacdets
  spread = (ij.amax(dim=1) - ij.amin(dim=1)).abs().amax(dim=(1, 2))
  scale = ij.abs().amax(dim=(1, 2, 3))
  dspread = jd.amax(dim=1) - jd.amin(dim=1)
  tol = AFFINE_RTOL if ij.dtype == torch.float64 else 1e-5
  return (spread <= tol * scale) & (dspread.abs() <= tol * jd.abs().amax(dim=1))


@dataclasses.dataclass(eq=False)
class HelmholtzOperator:
  fespace: object
  geo: torch.Tensor | None       # per-point factors of non-affine elements
  enc: torch.Tensor              # (E, n) encoded indices
  dmat: np.ndarray               # (P, P) host
  zero_range: tuple
  geo_elem: torch.Tensor | None = None    # (E, 8) affine constants
  geo_index: torch.Tensor | None = None   # (E,) slot in geo or -1
  weights: np.ndarray | None = None       # (P,) host
  num_affine: int = 0

  @classmethod
  def create(cls, fespace, dirichlet_mask=None,
             exploit_affine=True) -> 'HelmholtzOperator':
    why = supports_fused(fespace)
    if why is not None:
      raise NotImplementedError(f'fused Helmholtz kernel unavailable: {why}')
    mesh = fespace.mesh
    w = torch.as_tensor(fespace.quadrature.weights_nd(mesh.ndim),
                        dtype=fespace.dtype, device=fespace.device)
    geo_elem = geo_index = None
    num_affine = 0
    affine = _affine_mask(fespace) if exploit_affine else None
    if affine is not None and bool(affine.any()):
      num_affine = int(affine.sum())
      general = ~affine
      geo_index = torch.where(
          affine, torch.full_like(affine, -1, dtype=torch.int64),
          torch.cumsum(general, 0) - 1).to(torch.int32).contiguous()
      geo_elem = _ops.helmholtz_setup_affine(fespace.invjacs, fespace.jacdets)
      if num_affine < mesh.num_elements:
        geo = _ops.helmholtz_setup(fespace.invjacs[general].contiguous(),
                                   fespace.jacdets[general].contiguous(), w)
      else:
        geo = geo_index = None         # every element affine
    else:
      geo = _ops.helmholtz_setup(fespace.invjacs, fespace.jacdets, w)
    plan = mesh.assembly_plan()
    mask = None
    if dirichlet_mask is not None:
      mask = torch.as_tensor(dirichlet_mask, device=fespace.device)
      mask = (mask != 0).to(torch.uint8).contiguous()
    enc = _ops.encode_elements(mesh.elements, mask, plan.multiplicity)
    return cls(fespace=fespace, geo=geo, enc=enc,
               dmat=fespace.interpolator._differentiation_matrix_1d(),
               zero_range=plan.zero_range, geo_elem=geo_elem,
               geo_index=geo_index,
               weights=np.asarray(fespace.quadrature.weights),
               num_affine=num_affine)

  def apply(self, u, lambda0=0.0, lambda1=1.0, out=None, *, zero=True):
    """u (N,) or (N, nc) -> mask * scatter((l0 B + l1 A)_local(gather(u))).

    `zero=False` skips clearing the shared-node range of `out` (the caller has
    cleared it; used by bench.py to time the kernel alone).
    """
    mesh = self.fespace.mesh
    if u.shape[0] != mesh.num_nodes:
      raise ValueError(f'expected {mesh.num_nodes} nodal values, got '
                       f'{tuple(u.shape)}')
    u = u.to(self.fespace.dtype).contiguous()
    if out is None:
      out = torch.empty_like(u)
    return _ops.helmholtz_apply(
        u, out, self.enc, self.geo, self.dmat, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1,
        self.zero_range if zero else (0, 0), self.geo_elem, self.geo_index,
        self.weights)

  def apply_local(self, u_local, lambda0=0.0, lambda1=1.0):
    """Element-local action (E, n[, nc]) -> (E, n[, nc]); no gather/scatter."""
    mesh = self.fespace.mesh
    return _ops.helmholtz_local(
        u_local.to(self.fespace.dtype), self.geo, self.dmat, mesh.ndim,
        mesh.gridpoints_1d.num_points, lambda0, lambda1, self.geo_elem,
        self.geo_index, self.weights)
