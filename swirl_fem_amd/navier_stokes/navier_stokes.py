"""Spectral-element fractional-step Stokes / Navier-Stokes operators.

Drop-in for `swirl_fem/navier_stokes/navier_stokes.py`: `extk_coeffs` :49-58,
`bdfk_coeffs` :61-70, `BCType` :81-85, `dirichlet_bc` :88-94, `StokesPressure`
:98-140, `StokesVelocity` :144-245, `StokesSEM` :249-494 (`create`, `B`, `Bi`,
`A`, `C`, `D_local`, `Dt_local`, `D`, `Dt`, `Q`, `E`, `stokes_one_step`,
`filter`, `vorticity`).  P_N - P_{N-2} spaces: velocity on GLL(order+1) nodes,
pressure on GL(order-1) nodes, shared GLL(order+1) quadrature, convection
over-integrated on GLL(order+1+2) points.

Kernels: the velocity stiffness / mass / Helmholtz operators run the fused
gather-apply-scatter kernel (`sfem_helmholtz_apply`, all components in one
launch), divergence and pressure gradient `sfem_stokes_div` /
`sfem_stokes_grad_t`, convection `sfem_stokes_convect_local` between two
interpolations; the pressure mass matrix and the filter go through the
two-grid / generic sum-factorised basis kernels
(`sfem_basis_eval`, `sfem_basis_eval_t`); both PCG solves of a time step use
the device-resident CG (`linalg/cg.py`).  `lax.custom_linear_solve` of the
reference (:436-452) only matters for differentiation through the solve and is
a plain `cg` call here.
"""

from __future__ import annotations

import dataclasses
import enum
import os
from collections.abc import Sequence
from functools import partial
from typing import Any

import numpy as np
import torch

from swirl_fem_amd import switches
from swirl_fem_amd.core import autodiff
from swirl_fem_amd.core import basis
from swirl_fem_amd.core import layout
from swirl_fem_amd.core import operators
from swirl_fem_amd.core.fespace import div
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.fespace import grad
from swirl_fem_amd.core.interpolation import BarycentricInterpolator
from swirl_fem_amd.core.interpolation import Nodes1D
from swirl_fem_amd.core.interpolation import NodeType
from swirl_fem_amd.core.interpolation import Quadrature1D
from swirl_fem_amd.core.mesh import Mesh
from swirl_fem_amd.core.mesh_refiner import refine_premesh
from swirl_fem_amd.core.premesh import Premesh
from swirl_fem_amd.linalg.cg import cg
from swirl_fem_amd.linalg.cg import symmetric_solve
from swirl_fem_amd import _lib
from swirl_fem_amd import _ops

# pylint: disable=invalid-name


def _uniform_lagrange(k: int, at, derivative: bool) -> np.ndarray:
  """Values (or first derivatives) at `at` of the k + 1 Lagrange polynomials
  on the unit-spaced time levels 0, 1, ..., k (oldest first), in exact
  rational arithmetic."""
  from fractions import Fraction
  out = []
  for j in range(k + 1):
    others = [m for m in range(k + 1) if m != j]
    denom = Fraction(1)
    for m in others:
      denom *= j - m
    if not derivative:
      num = Fraction(1)
      for m in others:
        num *= Fraction(at) - m
    else:               # sum over the factor that is differentiated
      num = Fraction(0)
      for skip in others:
        term = Fraction(1)
        for m in others:
          if m != skip:
            term *= Fraction(at) - m
        num += term
    out.append(float(num / denom))
  return np.array(out, dtype=np.float64)


def extk_coeffs(k: int) -> np.ndarray:
  """Order-k extrapolation to the new time level from the last k + 1 levels,
  oldest first (reference navier_stokes.py:49-58: Newton-Cotes nodes on
  [-1, 1] evaluated one step beyond the end)."""
  return _uniform_lagrange(k, k + 1, derivative=False)


def bdfk_coeffs(k: int) -> np.ndarray:
  """Backward differentiation formula of order k: the derivative at the newest
  of k + 1 unit-spaced levels, oldest first (reference navier_stokes.py:61-70,
  whose factor h undoes the [-1, 1] scaling)."""
  return _uniform_lagrange(k, k, derivative=True)


def _solve(differentiable, A, b, **kwargs):
  """`cg(A, b, **kwargs)`; with `differentiable`, autograd sees the solve as
  `x = A^-1 b` (adjoint by a second solve, `linalg.cg.symmetric_solve`)."""
  members = kwargs.pop('members', 1)
  if members > 1:
    # an ensemble on the replicated mesh: one recurrence per member
    # (linalg/cg_ensemble.py)
    if kwargs.pop('reduce_fn', None) is not None:
      raise NotImplementedError('ensembles of a partitioned mesh')
    from swirl_fem_amd.linalg import cg_ensemble as ens
    if not differentiable:
      return ens.cg_ensemble(A, b, members, **kwargs)
    info = {}
    x = ens.symmetric_solve_ensemble(A, b, members, info_out=info, **kwargs)
    return x, info
  if not differentiable:
    return cg(A, b, **kwargs)
  info = {}
  x = symmetric_solve(A, b, info_out=info, **kwargs)
  return x, info


def _FUSED_DOTS():
  # SFEM_FUSED_DOTS=0: the pressure CG computes its inner products itself
  return switches.get('SFEM_FUSED_DOTS') != '0'


class _PressureOperator:
  """`p -> E p` for `cg`, handing it p . E p from inside the `D` kernel."""

  def __init__(self, sem, dt, time_order):
    self.sem, self.dt, self.time_order = sem, dt, time_order
    if (sem._divgrad() is not None and _FUSED_DOTS() and
        switches.get('SFEM_SPLIT_E') != '1'):
      self.apply_with_dot = self._apply_with_dot

  def __call__(self, p):
    return self.sem.E(p, dt=self.dt, time_order=self.time_order)

  def _apply_with_dot(self, p, partials):
    return self.sem.E(p, dt=self.dt, time_order=self.time_order,
                      dot_out=partials)


class _NullspaceProjection:
  """The default pressure preconditioner as an object `cg` can ask for
  r . M r together with M r (one pass less per iteration)."""

  def __init__(self, sem):
    self.sem = sem
    if not sem.is_partitioned and _FUSED_DOTS() and sem.members == 1:
      self.apply_with_dot = self._apply_with_dot

  def __call__(self, p):
    return _pressure_project_out_nullspace(self.sem, p)

  def _apply_with_dot(self, r, scalars, slot):
    return _pressure_project_out_nullspace(self.sem, r, (scalars, slot))

  def ensemble_mean_projection(self):
    """`mean_projection` for an ensemble: (w of ONE member, its total); every
    member has its own mean (`linalg/cg_ensemble.py`)."""
    sem = self.sem
    pmesh = sem.pressure.pspace.mesh
    gi = pmesh.exchange_gather_indices
    if sem.members == 1 or (gi is not None and gi.numel() != 0):
      return None
    b1, total = _pressure_mass_ones(sem, pmesh.dtype, pmesh.device)
    return b1, float(total)

  def mean_projection(self):
    """(w, total) when this preconditioner IS r -> r - (w . r / total) 1 (one
    partition, pressure nodes inside their elements: QQ^T is the identity), so
    that `cg` can fold it into its vector updates; else None."""
    sem = self.sem
    pmesh = sem.pressure.pspace.mesh
    gi = pmesh.exchange_gather_indices
    if (sem.members > 1 or sem.is_partitioned or pmesh.axis_name is not None or
        pmesh.neighbor_plan is not None or
        (gi is not None and gi.numel() != 0)):
      return None
    _pressure_mass_ones(sem, pmesh.dtype, pmesh.device)
    b1, total = sem._cache[('pressure_mass_ones', pmesh.dtype)]
    return b1, float(total)


class _MassPreconditioner:
  """`z = (d_max / d) . QQ^T r` with d the assembled lumped velocity mass: an
  opt-in preconditioner for the Helmholtz solve of the stepper (beyond the
  reference, whose solve uses M = QQ^T alone, navier_stokes.py:436-438).

  H = (beta / dt) B + mu A, and B is diagonal on GLL nodes: B^-1 H = beta / dt
  (I + (dt mu / beta) B^-1 A) is close to a multiple of the identity whenever
  the step is advection-dominated (dt mu / beta times the largest eigenvalue of
  B^-1 A small: 3e-7 x O(1e5) for the Taylor-Green case of config 4), while H
  itself inherits the spread of the GLL weights.  The factor d_max makes the
  stopping rule at least as strict as the reference's: d_max / d >= 1, so
  r . M r <= tol^2 b . b implies r . QQ^T r <= tol^2 b . b.  Commutes with
  QQ^T (d is equal on the copies of a node), hence symmetric."""

  capturable = True

  def __init__(self, sem):
    d = sem.velocity.exchange(sem.velocity_mass_diag)
    top = sem._global_sum_max(d.max().reshape(1))
    self.sem = sem
    self.factor = (top / d) * sem.velocity.interior_mask
    self._laid_out = {}

  def __call__(self, r):
    z = self.sem.velocity.exchange(r)
    f = self._laid_out.get(z.stride())
    if f is None:                  # (the solve's vectors: one layout per caller)
      f = self._laid_out[z.stride()] = layout.like(self.factor, z).clone()
    return f * z


class _SolutionProjection:
  """Successive right-hand sides (Fischer, "Projection techniques for
  iterative solution of A x = b with successive right-hand sides", Comput.
  Methods Appl. Mech. Engrg. 163, 1998): the pressure increments of the last
  steps, E-orthonormalised, span a space that holds most of the next one.
  `guess(b)` is the E-orthogonal projection of the solution onto that space
  (= the best start CG can be given from it), `update` adds the part of the
  new solution that was not in the space.  The stopping rule of the solve is
  unchanged (`cg` tests r . M r against tol^2 b . b, whatever x0 is), so the
  result agrees with the unprojected solve to the solver tolerance; what
  changes is the iteration count.  Beyond the reference (whose stepper starts
  every pressure solve from zero, navier_stokes.py:446-452): opt-in.

  Storage: 2 L pressure vectors (x_i and E x_i); cost per step: one more
  application of E and four passes over the basis -- against hundreds of
  iterations."""

  def __init__(self, size: int, reduce=None, members: int = 1):
    """`reduce`: in-place sum over the partitions (or None); `members`: an
    ensemble keeps one basis per member (the vectors are `(B Np,)`, member m
    in [m Np, (m + 1) Np): every coefficient below has a member axis)."""
    self.size, self.reduce, self.members = int(size), reduce, int(members)
    self.X = self.W = None        # (l, B, Np) each
    self.count = 0

  def _dots(self, X, v):
    """(l, B, Np) x (B, Np) -> (l, B), over all ranks."""
    out = torch.einsum('lbn,bn->lb', X, v)
    return out if self.reduce is None else self.reduce(out)

  def guess(self, b):
    if self.count == 0:
      return None
    X = self.X[:self.count]
    alpha = self._dots(X, b.reshape(self.members, -1))          # (l, B)
    return torch.einsum('lb,lbn->bn', alpha, X).reshape(b.shape)

  def update(self, x, x0, apply_e):
    B = self.members
    d = x if x0 is None else x - x0
    if self.X is None:
      self.X = torch.zeros((self.size, B, x.numel() // B), dtype=x.dtype,
                           device=x.device)
      self.W = torch.zeros_like(self.X)
    if self.count == self.size:
      # full: start again from the newest solution (it carries what the old
      # basis knew about the current time level)
      self.count, d = 0, x
    w = apply_e(d).reshape(B, -1)
    d = d.reshape(B, -1)
    if self.count:
      X, W = self.X[:self.count], self.W[:self.count]
      beta = self._dots(X, w)
      d = d - torch.einsum('lb,lbn->bn', beta, X)
      w = w - torch.einsum('lb,lbn->bn', beta, W)
    nrm2 = self._dots(d[None], w)[0]                            # (B,)
    ok = nrm2 > 0
    scale = torch.where(ok, torch.rsqrt(torch.where(ok, nrm2,
                                                    torch.ones_like(nrm2))),
                        torch.zeros_like(nrm2))
    self.X[self.count] = d * scale[:, None]
    self.W[self.count] = w * scale[:, None]
    self.count += 1


def _pressure_mass_ones(sem, dtype, device):
  """(B 1, 1 . B 1) of the pressure space, built once."""
  key = ('pressure_mass_ones', dtype)
  if key not in sem._cache:
    ones = torch.ones(sem.pressure.pspace.mesh.num_nodes, dtype=dtype,
                      device=device)
    b1 = sem.pressure.B(ones)
    if sem.members > 1:            # the members are copies of one mesh
      b1 = b1[:b1.numel() // sem.members].contiguous()
    sem._cache[key] = (b1, sem._global_sum(torch.sum(b1).reshape(1)))
  return sem._cache[key]


def _pressure_project_out_nullspace(sem, p, dot_result=None):
  """Remove the nullspace (all 1s vector) from p."""
  w = sem.pressure.exchange(p)
  # The reference applies the pressure mass matrix twice per call,
  # 1.B(w) / 1.B(1) (:73-78).  B is symmetric, so 1.B(w) = (B 1).w: the
  # vector B 1 is built once and every later call is a dot product.
  b1, total = _pressure_mass_ones(sem, p.dtype, p.device)
  if sem.members > 1:
    # every member has its own mean (b1: the weights of one member)
    if 'ens_project_partials' not in sem._cache:
      sem._cache['ens_project_partials'] = (
          torch.empty((sem.members, _lib.SFEM_ENS_GROUPS), dtype=torch.float64,
                      device=p.device), float(total))
    partials, total_host = sem._cache['ens_project_partials']
    return _ops.ens_subtract_weighted_mean(w, b1, total_host, sem.members,
                                           partials)
  if not sem.is_partitioned:
    # one partition: dot and subtraction as two launches, nothing on the host
    if 'project_partials' not in sem._cache:
      sem._cache['project_partials'] = (
          torch.empty(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                      device=p.device), float(total))
    partials, total_host = sem._cache['project_partials']
    return _ops.subtract_weighted_mean(w, b1, total_host, partials,
                                       dot_result=dot_result)
  return w - sem._global_sum(torch.vdot(b1, w).reshape(1)) / total


@enum.unique
class BCType(enum.Enum):
  """Types of boundary conditions."""
  DIRICHLET = 'dirichlet'
  NEUMANN = 'neumann'


def dirichlet_bc(mesh: Mesh, boundary_conditions) -> torch.Tensor:
  """Interior mask (1 inside, 0 on Dirichlet groups), shape (N,)."""
  interior_mask = torch.ones(mesh.num_nodes, dtype=mesh.dtype,
                             device=mesh.device)
  for physical_group, (bctype, unused_bcvalue) in boundary_conditions.items():
    if physical_group not in mesh.physical_masks and (
        mesh.axis_name is not None or mesh.neighbor_plan is not None):
      continue           # this rank's block does not touch that boundary
    if bctype == BCType.DIRICHLET:
      interior_mask = interior_mask * (
          1 - mesh.physical_masks[physical_group].to(mesh.dtype))
  return interior_mask


def _replace(obj, **kw):
  return dataclasses.replace(obj, **kw)


@dataclasses.dataclass(frozen=True, eq=False)
class StokesPressure:
  """Pressure space for the Stokes problem (Gauss-Legendre nodes)."""
  pspace: FiniteElementSpace

  @classmethod
  def create(cls, premesh: Premesh, quadrature: Quadrature1D, order: int,
             device=None, dtype=None, axis_name=None,
             rank=None) -> 'StokesPressure':
    gridpoints_1d = Nodes1D.create(num_points=order - 1,
                                   node_type=NodeType.GAUSS_LEGENDRE)
    pmesh = refine_premesh(premesh, gridpoints_1d=gridpoints_1d).finalize(
        axis_name, rank=rank, device=device, dtype=dtype)
    return cls(pspace=FiniteElementSpace.create(mesh=pmesh,
                                                quadrature=quadrature))

  replace = _replace

  def gather(self, p):
    return self.pspace.mesh.gather(p)

  def scatter(self, p):
    return self.pspace.mesh.scatter(p)

  def B(self, p):
    """Apply the pressure mass matrix."""
    def l(u, v):
      return lambda x: u(x) * v(x)

    u = self.pspace.scalar_function(self.gather(p))
    v = self.pspace.scalar_function(None)
    return self.scatter(self.pspace.local_covector(l, (u, v)))

  def exchange(self, p):
    """Apply QQ^T."""
    return self.pspace.mesh.exchange(p)


@dataclasses.dataclass(frozen=True, eq=False)
class StokesVelocity:
  """Velocity space for the Stokes system (Gauss-Lobatto-Legendre nodes)."""
  vspace: FiniteElementSpace
  overint_space: FiniteElementSpace
  interior_mask: torch.Tensor          # (N, 1)
  diag_qqt: torch.Tensor               # (N,)
  num_convection_overint_nodes: int = 2

  @classmethod
  def create(cls, premesh: Premesh, order: int, boundary_conditions,
             num_convection_overint_nodes: int = 2, device=None,
             dtype=None, axis_name=None, rank=None,
             neighbor_plan=None) -> 'StokesVelocity':
    gridpoints_1d = Nodes1D.create(
        num_points=order + 1, node_type=NodeType.GAUSS_LOBATTO_LEGENDRE)
    vmesh = refine_premesh(premesh, gridpoints_1d=gridpoints_1d).finalize(
        axis_name, rank=rank, device=device, dtype=dtype)
    if neighbor_plan is not None:
      # `premesh` is this rank's own block (distributed/blocks.py): the
      # refined block mesh numbers its nodes exactly like the block builder's
      # mesh, so the builder's neighbour plan applies as it is
      if neighbor_plan.has_local_images:
        # periodic along a direction with a single block: the mesh keeps the
        # gather / unique indices of its own images, which the plan sums
        # before and copies back after the neighbour exchange
        vmesh = vmesh.replace(axis_name='blocks', neighbor_plan=neighbor_plan)
      else:
        cat = (np.concatenate(neighbor_plan.indices).astype(np.int32)
               if neighbor_plan.indices else np.zeros(0, np.int32))
        vmesh = vmesh.replace(
            axis_name='blocks', neighbor_plan=neighbor_plan,
            exchange_gather_indices=torch.as_tensor(cat, device=vmesh.device),
            exchange_unique_indices=None)
    vspace = FiniteElementSpace.create(
        mesh=vmesh, quadrature=Quadrature1D.create_from_nodes_1d(gridpoints_1d))
    interior_mask = dirichlet_bc(vmesh, boundary_conditions)
    if vmesh.axis_name is not None and vmesh.node_indices is not None:
      # padding nodes of an uneven partition carry no equation
      interior_mask = interior_mask * (vmesh.node_indices >= 0).to(
          interior_mask.dtype)
    interior_mask = interior_mask[:, None]
    overint_gridpoints_1d = Nodes1D.create(
        num_points=gridpoints_1d.num_points + num_convection_overint_nodes,
        node_type=NodeType.GAUSS_LOBATTO_LEGENDRE)
    overint_space = FiniteElementSpace.create(
        mesh=vmesh,
        quadrature=Quadrature1D.create_from_nodes_1d(overint_gridpoints_1d))
    diag_qqt = vmesh.scatter(torch.ones(
        tuple(vmesh.elements.shape), dtype=vmesh.dtype, device=vmesh.device))
    if vmesh.axis_name is not None:
      # multiplicity over all partitions; padding nodes (count 0) -> 1
      diag_qqt = vmesh.exchange(diag_qqt).clamp(min=1.0)
    return cls(vspace=vspace, overint_space=overint_space, diag_qqt=diag_qqt,
               interior_mask=interior_mask,
               num_convection_overint_nodes=num_convection_overint_nodes)

  replace = _replace

  @property
  def local_shape(self):
    m = self.vspace.mesh
    return (m.num_elements, m.num_nodes_per_element, m.ndim)

  @property
  def mesh(self) -> Mesh:
    return self.vspace.mesh

  def C(self, u):
    """Apply the convection operator with overintegration."""
    return self.interior_mask * self.scatter(self.C_local(self.gather(u)))

  def gather(self, u):
    """(N, d) -> (E, n, d)."""
    if autodiff.needs_grad(u):
      return autodiff.gather_rows(u, self.mesh.elements)
    return _ops.gather_rows(u, self.mesh.elements)

  def scatter(self, u):
    """(E, n, d) -> (N, d)."""
    if autodiff.needs_grad(u):
      return autodiff.scatter_add(u, self.mesh.elements, self.mesh.num_nodes,
                                  ncomp=u.shape[-1])
    return _ops.scatter_add(u, self.mesh.elements, self.mesh.num_nodes,
                            ncomp=u.shape[-1])

  def exchange(self, u, inplace=False):
    """Apply QQ^T to every component (`inplace`: `u` may be overwritten)."""
    mesh = self.mesh
    gi = mesh.exchange_gather_indices
    if gi is None or gi.numel() == 0:
      return u
    if mesh.axis_name is None:
      if mesh.exchange_unique_indices is None:
        return u if inplace else u.clone()
      if autodiff.needs_grad(u):
        return autodiff.exchange_local(u, gi, mesh.exchange_unique_indices)
      return _ops.exchange_local(u, gi, mesh.exchange_unique_indices,
                                 inplace=inplace)
    if autodiff.needs_grad(u):
      raise NotImplementedError('autograd through the partitioned exchange')
    from swirl_fem_amd.distributed import comm
    if inplace:
      return comm.neighbor_exchange_(u, mesh.neighbor_plan)
    return comm.neighbor_exchange(u, mesh.neighbor_plan)

  def _fused(self):
    """Fused operator without mask, or None if the space is not eligible."""
    if operators.supports_fused(self.vspace) is None:
      return self.vspace.helmholtz_operator(None)
    return None

  def A_local(self, u_local):
    """Apply the velocity stiffness operator locally."""
    op = self._fused()
    if op is not None:
      if autodiff.needs_grad(u_local):
        return autodiff.helmholtz_local(op, u_local, 0.0, 1.0)
      return op.apply_local(u_local, 0.0, 1.0)

    def a(u, v):
      return lambda x: torch.einsum('ij,ij->', grad(u)(x), grad(v)(x))

    u = self.vspace.vector_function(u_local)
    v = self.vspace.vector_function(None)
    return self.vspace.local_covector(a, (u, v))

  def B_local(self, u_local):
    """Apply the velocity mass operator locally."""
    op = self._fused()
    if op is not None:
      if autodiff.needs_grad(u_local):
        return autodiff.helmholtz_local(op, u_local, 1.0, 0.0)
      return op.apply_local(u_local, 1.0, 0.0)

    def l(u, v):
      return lambda x: torch.vdot(u(x), v(x))

    u = self.vspace.vector_function(u_local)
    v = self.vspace.vector_function(None)
    return self.vspace.local_covector(l, (u, v))

  def C_local(self, u_local):
    """Apply the local convection operator."""
    cache = self.overint_space._cache
    if 'convection' not in cache:
      try:
        cache['convection'] = operators.ConvectionOperator.create(
            self.overint_space)
      except NotImplementedError:      # fall back to the generic form below
        cache['convection'] = None
    # (the generic form below is what autograd can differentiate)
    if cache['convection'] is not None and not autodiff.needs_grad(u_local):
      return cache['convection'].apply_local(u_local)

    def c(u, w, v):
      return lambda x: torch.einsum('i,ij,j->', u(x), grad(w)(x), v(x))

    u = self.overint_space.vector_function(u_local)
    v = self.overint_space.vector_function(None)
    return self.overint_space.local_covector(c, (u, u, v))


@dataclasses.dataclass(frozen=True, eq=False)
class StokesSEM:
  """Linear operators for a Stokes solver using spectral elements."""
  velocity: StokesVelocity
  pressure: StokesPressure
  velocity_mass_diag: torch.Tensor      # (N, d)
  # > 1: this is `ensemble(members)` of a single-mesh StokesSEM
  members: int = 1
  _cache: dict = dataclasses.field(default_factory=dict, repr=False,
                                   compare=False)

  @classmethod
  def create(cls, premesh: Premesh, boundary_conditions, order: int,
             num_convection_overint_nodes: int = 2, *, device=None,
             dtype=None, axis_name=None, rank=None,
             neighbor_plan=None) -> 'StokesSEM':
    """`axis_name` / `rank`: build this rank's partition of a partitioned
    premesh (one process per GPU; the reference has no partitioned
    Navier-Stokes path).  Fields are then consistent across partitions, right
    hand sides unassembled, `M = exchange` assembles inside the solves and the
    inner products are all-reduced.

    `neighbor_plan`: alternatively `premesh` is already this rank's own block
    and the plan says which of its nodes other ranks hold too
    (`BlockPartition.premesh` / `.plan` of `distributed/blocks.py`: nothing of
    the global mesh is ever built)."""
    if premesh.order != 1:
      raise ValueError(f'Expected mesh order 1; got {premesh.order}.')
    quadrature = Quadrature1D.create(
        num_points=order + 1, quadrature_type=NodeType.GAUSS_LOBATTO_LEGENDRE)
    pressure = StokesPressure.create(premesh, quadrature, order, device=device,
                                     dtype=dtype, axis_name=axis_name,
                                     rank=rank)
    velocity = StokesVelocity.create(premesh, order, boundary_conditions,
                                     num_convection_overint_nodes,
                                     device=device, dtype=dtype,
                                     axis_name=axis_name, rank=rank,
                                     neighbor_plan=neighbor_plan)
    ones = torch.ones(velocity.local_shape, dtype=velocity.mesh.dtype,
                      device=velocity.mesh.device)
    velocity_mass_diag = velocity.scatter(velocity.B_local(ones))
    return cls(velocity=velocity, pressure=pressure,
               velocity_mass_diag=velocity_mass_diag)

  def replace(self, **kw):
    kw.setdefault('_cache', {})
    return dataclasses.replace(self, **kw)

  # --------------------------------------------------------------- ensembles
  def ensemble(self, members: int) -> 'StokesSEM':
    """The same operators for an ensemble of `members` flows at once.

    The reference `jax.vmap`s its solver step over an ensemble
    (niles/train.py:232, :262-264).  Here the ensemble is ONE StokesSEM on
    `members` disjoint copies of the mesh (`Mesh.replicate`): a batch of
    fields `(B, N, ...)` is the field `(B N, ...)` of that mesh
    (`flatten` / `unflatten` are views), every operator -- B, C, D, Dt, E, H,
    filter -- is the ordinary kernel launched once for all members, and the
    two solves of `stokes_one_step` run one CG recurrence per member
    (`linalg/cg_ensemble.py`: own step lengths, stop test and iteration
    count, results equal to the member-by-member solves).  Everything that
    takes this object (`navier_stokes_step`, the generator's step) works on
    the ensemble unchanged, autograd included (the cotangent of a solve is
    one more ensemble solve).  One partition.
    """
    members = int(members)
    if self.members != 1:
      raise ValueError('this StokesSEM is an ensemble already')
    if members == 1:
      return self
    if self.is_partitioned:
      raise NotImplementedError('ensembles of a partitioned mesh')
    vel, prs = self.velocity, self.pressure
    vmesh = vel.mesh.replicate(members)
    pmesh = prs.pspace.mesh.replicate(members)
    velocity = vel.replace(
        vspace=FiniteElementSpace.create(mesh=vmesh,
                                         quadrature=vel.vspace.quadrature),
        overint_space=FiniteElementSpace.create(
            mesh=vmesh, quadrature=vel.overint_space.quadrature),
        interior_mask=vel.interior_mask.repeat(members, 1),
        diag_qqt=vel.diag_qqt.repeat(members))
    pressure = prs.replace(pspace=FiniteElementSpace.create(
        mesh=pmesh, quadrature=prs.pspace.quadrature))
    return StokesSEM(velocity=velocity, pressure=pressure,
                     velocity_mass_diag=self.velocity_mass_diag.repeat(
                         members, 1), members=members)

  def flatten(self, batched: torch.Tensor) -> torch.Tensor:
    """(B, N, ...) -> (B N, ...): the ensemble's field (a view)."""
    if batched.shape[0] != self.members:
      raise ValueError(f'expected {self.members} members; got '
                       f'{batched.shape[0]}')
    return batched.reshape((-1,) + tuple(batched.shape[2:]))

  def unflatten(self, field: torch.Tensor) -> torch.Tensor:
    """(B N, ...) -> (B, N, ...)."""
    return field.reshape((self.members, -1) + tuple(field.shape[1:]))

  # ------------------------------------------------------------- partitions
  @property
  def is_partitioned(self) -> bool:
    return self.velocity.mesh.axis_name is not None

  def _global_sum(self, t):
    """Sum of a device tensor over the partitions (identity on one rank)."""
    if self.is_partitioned:
      from swirl_fem_amd.distributed import comm
      comm.all_reduce_sum_(t)
    return t

  def _reduce_fn(self):
    return self._global_sum if self.is_partitioned else None

  def _global_sum_max(self, t):
    """Maximum of a device tensor over the partitions."""
    if self.is_partitioned:
      from swirl_fem_amd.distributed import comm
      comm.all_reduce_max_(t)
    return t

  # ----------------------------------------------------------------- operators
  def B(self, u):
    """Apply the (diagonal) mass operator to a velocity field."""
    return self.velocity.interior_mask * self.velocity_mass_diag * u

  def Bi(self, u):
    """Apply the inverse mass operator to a velocity field."""
    if 'diag_qqti' not in self._cache:
      self._cache['diag_qqti'] = 1 / self.velocity.exchange(
          self.velocity_mass_diag)
    return self._cache['diag_qqti'] * self.velocity.exchange(u)

  def _masked_operator(self):
    """Fused velocity operator with the Dirichlet rows zeroed, or None."""
    if 'masked_op' not in self._cache:
      op = None
      if operators.supports_fused(self.velocity.vspace) is None:
        dirichlet = (self.velocity.interior_mask[:, 0] == 0)
        op = operators.HelmholtzOperator.create(self.velocity.vspace,
                                                dirichlet)
      self._cache['masked_op'] = op
    return self._cache['masked_op']

  def A(self, u):
    """Apply the stiffness operator to a velocity field."""
    op = self._masked_operator()
    if op is not None:
      if autodiff.needs_grad(u):
        return self._fused_apply_diff(op, u, 0.0, 1.0)
      return op.apply(u, 0.0, 1.0)
    return self.velocity.interior_mask * self.velocity.scatter(
        self.velocity.A_local(self.velocity.gather(u)))

  def H(self, u, mass_coeff: float, mu: float):
    """Helmholtz operator `mass_coeff * B + mu * A` (reference :431) in one
    fused kernel; equals `mass_coeff * self.B(u) + mu * self.A(u)`."""
    op = self._masked_operator()
    if op is not None:
      if autodiff.needs_grad(u):
        return self._fused_apply_diff(op, u, mass_coeff, mu)
      return op.apply(u, mass_coeff, mu)
    return mass_coeff * self.B(u) + mu * self.A(u)

  def _fused_apply_diff(self, op, u, l0, l1):
    """Fused apply that autograd can differentiate (`autodiff._HelmholtzApply`:
    the cotangent goes through the unmasked twin of the operator)."""
    return autodiff.helmholtz_apply(op, self.velocity._fused(),
                                    self.velocity.interior_mask[:, 0], u, l0,
                                    l1)

  def C(self, u):
    """Apply the convection operator to a velocity field."""
    return self.velocity.C(u)

  def D_local(self, u_local):
    """Apply the local operator D."""
    def b(v, q):
      return lambda x: div(v)(x) * q(x)

    v = self.velocity.vspace.vector_function(u_local)
    p = self.pressure.pspace.scalar_function(None)
    return self.pressure.pspace.local_covector(b, (v, p))

  def Dt_local(self, p_local):
    """Apply the local operator D^T."""
    def b(v, q):
      return lambda x: div(v)(x) * q(x)

    v = self.velocity.vspace.vector_function(None)
    p = self.pressure.pspace.scalar_function(p_local)
    return self.velocity.vspace.local_covector(b, (v, p))

  def _divgrad(self):
    """Fused D / D^T kernels (`operators.StokesDivGrad`), or None when the
    spaces are not eligible (then the generic q-function path runs)."""
    if 'divgrad' not in self._cache:
      op = None
      if operators.supports_fused_stokes(self.velocity.vspace,
                                         self.pressure.pspace) is None:
        dirichlet = (self.velocity.interior_mask[:, 0] == 0)
        op = operators.StokesDivGrad.create(self.velocity.vspace,
                                            self.pressure.pspace, dirichlet)
      self._cache['divgrad'] = op
    return self._cache['divgrad']

  def D(self, u):
    """Velocity divergence matrix."""
    op = self._divgrad()
    if op is not None:
      if autodiff.needs_grad(u):
        return autodiff.stokes_div(op, self._divgrad_free(), u)
      return op.div(u)
    return self.pressure.scatter(self.D_local(self.velocity.gather(u)))

  def _divgrad_free(self):
    """`_divgrad()` without the Dirichlet mask: its `grad_t` is the exact
    transpose of `div` (the cotangent rule of the fused divergence)."""
    if 'divgrad_free' not in self._cache:
      self._cache['divgrad_free'] = operators.StokesDivGrad.create(
          self.velocity.vspace, self.pressure.pspace, None)
    return self._cache['divgrad_free']

  def Dt(self, p):
    """Apply the pressure gradient operator."""
    op = self._divgrad()
    if op is not None:
      if autodiff.needs_grad(p):
        return autodiff.stokes_grad_t(op, self.velocity.interior_mask[:, 0], p)
      return op.grad_t(p)
    return self.velocity.interior_mask * self.velocity.scatter(
        self.Dt_local(self.pressure.gather(p)))

  def Q(self, u, dt: float, time_order: int):
    """Apply the operator Q = (dt / beta_k) B^-1."""
    beta_k = bdfk_coeffs(time_order)[-1]
    return (dt / beta_k) * self.Bi(u)

  def E(self, p, dt: float, time_order: int, dot_out=None):
    """Apply the operator E = D Q D^T.  `dot_out`: SFEM_DOT_SLOTS device
    doubles that accumulate partial sums of p . E p (for `cg`)."""
    op = None if autodiff.needs_grad(p) else self._divgrad()
    if op is not None:
      # two kernels: D^T, then D with Q = (dt / beta_k) diag(QQ^T B)^-1 folded
      # into its gather (plus the exchange on periodic / partitioned meshes)
      key = ('q_scale', float(dt), int(time_order))
      if key not in self._cache:
        if 'diag_qqti' not in self._cache:
          self._cache['diag_qqti'] = 1 / self.velocity.exchange(
              self.velocity_mass_diag)
        beta_k = float(bdfk_coeffs(time_order)[-1])
        q = (dt / beta_k) * self._cache['diag_qqti']
        # the lumped mass is the same for every component: one factor per node
        same = bool((q == q[:, :1]).all())
        self._cache[key] = (q[:, 0].contiguous() if same
                            else layout.component_major(q))
      # component-major intermediate: the shared-node atomics of one component
      # then hit whole lines (D^T 1.3 ms instead of 2.2 ms at 48^3, p = 7)
      if op.penc is None and switches.get('SFEM_SPLIT_E') == '1':
        # opt-in: nodes held by one element stay in registers between D^T and
        # D.  4 % faster than the two kernels below while their atomics ran in
        # slot order; with the sorted shared scatter D^T alone dropped by a
        # third and the plain pair wins (3.95 vs 4.71 ms at 64^3, p = 7)
        if dot_out is not None:
          raise NotImplementedError('fused p . E p with the split E')
        return op.e_apply(
            p, scale=self._cache[key],
            exchange=partial(self.velocity.exchange, inplace=True))
      # index-row kernels in 2D: D^T with a position per writer, the sums
      # formed by the class kernel that serves the periodic images anyway --
      # no atomics, nothing cleared, one launch less (`scripts/bench_ns.py`,
      # ms per step atomic -> layered: Kolmogorov generator 10.1 -> 8.9,
      # lid-driven cavity 35.6 -> 31.6, an ensemble of 8 flows 28.2 -> 22.5;
      # 3D index rows gain nothing, `scripts/time_layered_e.py`: order 4 at
      # 32^3 0.171 -> 0.169 ms per E, order 9 at 24^3 0.67 -> 0.73: opt-in)
      lay = switches.get('SFEM_STOKES_LAYERED')
      if (lay != '0' and not self.is_partitioned and
          (lay == '1' or op.vspace.mesh.ndim == 2) and
          op.supports_layered_e()):
        return op.e_layered(p, scale=self._cache[key], dot_with=p,
                            dot_out=dot_out)
      # Q is the same on every copy of a node, so it commutes with the
      # assembly and the exchange: applied inside D^T (whose scatter waits on
      # atomics anyway) it spares D one gather per node
      w = self.velocity.exchange(
          op.grad_t(p, component_major=True, scale=self._cache[key]),
          inplace=True)
      return op.div(w, dot_with=p, dot_out=dot_out)
    if dot_out is not None:
      raise NotImplementedError('fused p . E p needs the fused Stokes kernels')
    return self.D(self.Q(self.Dt(p), dt=dt, time_order=time_order))

  # ------------------------------------------------------------- time stepping
  def stokes_one_step(self, us: Sequence[torch.Tensor],
                      ps: Sequence[torch.Tensor], f, mu: float, dt: float,
                      time_order: int, alpha: float = 0.05, u_boundary=None,
                      pressure_preconditioner=None,
                      project_out_nullspace=True, tol: float = 1e-8,
                      atol: float = 0, pressure_projection: int | None = None,
                      velocity_preconditioner=None
                      ) -> tuple[torch.Tensor, torch.Tensor, Any]:
    """Evolves the Stokes system by one fractional step (reference :350-458).

    `velocity_preconditioner` (beyond the reference; None = the switch
    SFEM_VELOCITY_PC, default 'exchange' = the reference's M = QQ^T): 'mass'
    (`_MassPreconditioner`) or a callable r -> z for the Helmholtz solve.

    `pressure_projection` (beyond the reference; None = the switch
    SFEM_PRESSURE_PROJECTION, default 0 = off): number of earlier pressure
    increments kept to start the pressure solve from their span
    (`_SolutionProjection`); same result to the solver tolerance, fewer
    iterations -- the state lives in this object, so use it for ONE time
    series with fixed dt and time_order."""
    default_projection = pressure_preconditioner is None and \
        project_out_nullspace
    if default_projection:
      pressure_preconditioner = _NullspaceProjection(self)

    ext_coeffs = extk_coeffs(k=1)
    p_ext = sum(float(ext_coeffs[-i]) * ps[-i]
                for i in range(1, len(ext_coeffs) + 1))
    f = f + self.Dt(p_ext)

    bdf = bdfk_coeffs(time_order)
    beta_hist, beta_k = bdf[:-1], float(bdf[-1])
    H_ = lambda u: self.H(u, beta_k / dt, mu)
    fused_h = self._masked_operator()
    if fused_h is not None and not autodiff.needs_grad(f, u_boundary, *us,
                                                       *ps):
      # the same apply as an object `cg` can ask for p . H p (accumulated in
      # the kernel's scatter stage: one dot pass and one scalar launch less
      # per iteration)
      H_ = fused_h.linear_operator(beta_k / dt, float(mu))
    f = f - self.B((1 / dt) * sum(float(c) * u for c, u in zip(beta_hist, us)))
    if u_boundary is not None:
      f = f - H_(u_boundary)

    # component-major storage for the Helmholtz solve: every kernel of the CG
    # then works on contiguous component strips (same (N, d) shape for callers)
    # (an ensemble keeps the dense layout: member m is the contiguous block
    # [m N d, (m + 1) N d) there)
    ens = {'members': self.members} if self.members > 1 else {}
    if not ens:
      f = layout.component_major(f)
    # single-partition solves replay each CG iteration as one HIP graph launch
    # (iterations on small meshes are launch-bound); SFEM_GRAPHS=0 disables it
    graph = (self.velocity.mesh.axis_name is None and
             switches.get('SFEM_GRAPHS') != '0')
    # differentiable step (reference: lax.custom_linear_solve(symmetric=True),
    # :436-452): the cotangent of a solve is one more solve
    diff = autodiff.needs_grad(f, u_boundary, *us, *ps)
    if diff:
      # f and u_star vanish on the Dirichlet rows; saying so to autograd keeps
      # the cotangent solve on the same (masked) system
      f = self.velocity.interior_mask * f
    # the two solves of a step use the same operators step after step: their
    # recorded iterations are kept (SFEM_GRAPH_REUSE=0: record every solve)
    if (graph and not diff and self._reduce_fn() is None and
        switches.get('SFEM_GRAPH_REUSE') != '0'):
      ws = self._cache.setdefault('cg_workspaces', {})
      keep = lambda *key: dict(workspace=ws,
                               key=key + (tol, atol, self.members))
    else:
      keep = lambda *key: {}
    # (replaying an iteration pays while it is launch-bound; on vectors of
    # tens of millions of values there is nothing to save -- 64^3 elements,
    # p = 7: 2.14 against 2.20 s per step eager on one box, equal within the
    # noise on another -- and the eager loop does not hold a second set of
    # solver vectors)
    limit = int(switches.get('SFEM_GRAPH_MAX_NUMEL'))
    small = lambda b: graph and b.numel() <= limit
    if velocity_preconditioner is None:
      velocity_preconditioner = switches.get('SFEM_VELOCITY_PC')
    vpc = velocity_preconditioner
    if vpc == 'exchange':
      M_v = self.velocity.exchange
    elif vpc == 'mass':
      if 'velocity_mass_pc' not in self._cache:
        self._cache['velocity_mass_pc'] = _MassPreconditioner(self)
      M_v = self._cache['velocity_mass_pc']
    elif callable(vpc):
      M_v = vpc
    else:
      raise ValueError(f'unknown velocity preconditioner {vpc!r}')
    u_star, info = _solve(diff, H_, f, M=M_v, tol=tol,
                          atol=atol, graph=small(f),
                          reduce_fn=self._reduce_fn(), **ens,
                          **(keep('H', beta_k / dt, float(mu),
                                  vpc if isinstance(vpc, str) else id(vpc))
                             if small(f) else {}))
    if diff:
      u_star = self.velocity.interior_mask * u_star
    if u_boundary is not None:
      u_star = u_star + u_boundary
    aux = {'u_star_info': info}

    u_star = self.filter(u_star, alpha=alpha)

    rhs = -self.D(u_star)
    E_ = _PressureOperator(self, dt, time_order)
    if pressure_projection is None:
      pressure_projection = int(switches.get('SFEM_PRESSURE_PROJECTION'))
    hist, dp0 = None, None
    if pressure_projection > 0 and not diff:
      key = ('pressure_projection', float(dt), int(time_order),
             int(pressure_projection))
      hist = self._cache.get(key)
      if hist is None:
        hist = self._cache[key] = _SolutionProjection(
            pressure_projection, self._reduce_fn(), self.members)
      dp0 = hist.guess(rhs)
    dp, info = _solve(diff, E_, rhs, x0=dp0,
                      M=pressure_preconditioner, tol=tol, atol=atol,
                      # (a caller's preconditioner may do things a recorded
                      # iteration cannot hold: sparse products, host logic)
                      graph=small(rhs) and (default_projection or getattr(
                          pressure_preconditioner, 'capturable', False)),
                      reduce_fn=self._reduce_fn(), **ens,
                      **(keep('E', float(dt), int(time_order),
                              None if default_projection
                              else id(pressure_preconditioner))
                         if small(rhs) and (default_projection or getattr(
                             pressure_preconditioner, 'capturable', False))
                         else {}))
    if hist is not None:
      hist.update(dp, dp0, E_)
    aux['dp_info'] = info

    u = u_star + self.Q(self.Dt(dp), dt=dt, time_order=time_order)
    p = p_ext + dp
    return u, p, aux

  def filter(self, u, alpha=0.05):
    """Filter-based stabilisation: blend with the P-1 interpolant (:460-482)."""
    vmesh = self.velocity.mesh
    grid = vmesh.gridpoints_1d
    low = Nodes1D.create(num_points=grid.num_points - 1,
                         node_type=grid.node_type)
    low_interp = BarycentricInterpolator(ndim=vmesh.ndim, gridpoints_1d=grid,
                                         evalpoints_1d=low)
    high_interp = BarycentricInterpolator(ndim=vmesh.ndim, gridpoints_1d=low,
                                          evalpoints_1d=grid)
    u_local = self.velocity.gather(u)
    filtered_local = basis.interp(high_interp, basis.interp(low_interp,
                                                            u_local))
    filtered = (1 / self.velocity.diag_qqt[:, None]) * (
        self.velocity.exchange(self.velocity.scatter(filtered_local))
        if self.is_partitioned else self.velocity.scatter(filtered_local))
    return (1 - alpha) * u + alpha * filtered

  def vorticity(self, u):
    """Vorticity (2D) of a velocity field, averaged at shared nodes."""
    uf = self.velocity.vspace.vector_function(self.velocity.gather(u))

    def _vorticity(x):
      g = grad(uf)(x)
      return g[1, 0] - g[0, 1]

    vort_local = self.velocity.vspace._evaluate(_vorticity)
    vmesh = self.velocity.vspace.mesh
    return (1. / self.velocity.diag_qqt) * vmesh.scatter(vort_local)
