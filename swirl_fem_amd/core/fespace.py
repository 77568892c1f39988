"""Finite element space on a `Mesh`: q-functions, integration, operator action.

API of the reference `swirl_fem/core/fespace.py`: `NodalQFunction` family
:76-225, `grad` :233-241, `div` :244-248, `FiniteElementSpace.create` :306-348,
`scalar_function` :364-370, `vector_function` :372-379, `integrate` :381-403,
`local_covector` :405-471.

How it runs here
  * `create` gathers element coordinates and computes `invjacs (E,Q,d,d)`,
    `jacdets (E,Q)` (signed) and `quad_coords (E,Q,d)` with
    `sfem_geom_factors` (sum-factorised, closed-form d x d inverse).
  * Nodal q-functions evaluate with `sfem_basis_eval` (values / physical
    gradients at all quadrature points of all elements).
  * `local_covector` does not trace: the form is evaluated once on `QExpr`
    values (core/qexpr.py); the placeholder's slot yields the coefficient
    fields `(c0, c1)` and `sfem_basis_eval_t` applies the exact transpose
    `sum_q w detJ (I^T c0 + G^T J^-T c1)` -- what `jax.linear_transpose`
    produces in the reference (:466-471).
  * The collocated mass / stiffness / Helmholtz operators have a fused
    gather->apply->scatter kernel, reached through `helmholtz_operator`.
"""

from __future__ import annotations

import dataclasses
from typing import Protocol

import numpy as np
import torch

from swirl_fem_amd import _ops
from swirl_fem_amd.core import autodiff
from swirl_fem_amd.core import interpolation
from swirl_fem_amd.core.interpolation import BarycentricInterpolator
from swirl_fem_amd.core.interpolation import Quadrature1D
from swirl_fem_amd.core.mesh import Mesh
from swirl_fem_amd.core.qexpr import QExpr
from swirl_fem_amd.core import qexpr


# ---------------------------------------------------------------- q-functions
class QFunction(Protocol):
  """A function from the mesh to R^k, called on the point variable `x`
  (reference core/fespace.py:35-56).  Here `x` is a `QExpr` carrying all
  quadrature points of all elements at once, and so is the result."""

  def __call__(self, x):
    ...


class Form(Protocol):
  """Maps q-functions to the scalar q-function that gets integrated
  (reference core/fespace.py:59-72), e.g.
  `lambda u, v: lambda x: torch.vdot(grad(u)(x), grad(v)(x))`."""

  def __call__(self, *args: QFunction) -> QFunction:
    ...


@dataclasses.dataclass(eq=False)
class NodalQFunction:
  """A nodal function of a `FiniteElementSpace` (u_local None = placeholder)."""
  fespace: 'FiniteElementSpace'
  value_shape: tuple
  u_local: torch.Tensor | None = None

  def __post_init__(self):
    expected = (self.fespace.num_elements,
                self.fespace.mesh.num_nodes_per_element) + self.value_shape
    if self.u_local is not None and tuple(self.u_local.shape) != expected:
      raise ValueError('shape:', tuple(self.u_local.shape))

  def _evaluate(self) -> torch.Tensor:
    raise NotImplementedError

  def _placeholder(self) -> QExpr:
    raise NotImplementedError

  def __call__(self, x=None) -> QExpr:
    del x   # nodal values suffice (reference fespace.py:162-165)
    if self.u_local is None:
      return self._placeholder()
    return QExpr(self._evaluate())

  def _u3(self):
    E, n = self.u_local.shape[:2]
    return self.u_local.reshape(E, n, -1)


class ScalarNodalQFunction(NodalQFunction):
  """Scalar function interpolated from nodal values (fespace.py:171-179)."""

  def __init__(self, fespace, u_local=None):
    super().__init__(fespace, (), u_local)

  def _evaluate(self):
    val, _ = self.fespace._basis(self._u3(), want_val=True, want_grad=False)
    return val[..., 0]

  def _placeholder(self):
    return QExpr(shape=(), pullback=lambda ct: (ct, None))


class ScalarNodalQFunctionGrad(NodalQFunction):
  """Gradient of a scalar nodal function (fespace.py:183-195)."""

  def __init__(self, fespace, u_local=None):
    super().__init__(fespace, (), u_local)

  def _evaluate(self):
    _, g = self.fespace._basis(self._u3(), want_val=False, want_grad=True)
    return g[..., 0]                                    # (E, Q, d)

  def _placeholder(self):
    d = self.fespace.mesh.ndim
    return QExpr(shape=(d,), pullback=lambda ct: (None, ct))


class VectorNodalQFunction(NodalQFunction):
  """Vector function interpolated from nodal values (fespace.py:199-209)."""

  def __init__(self, fespace, u_local=None):
    super().__init__(fespace, (fespace.mesh.ndim,), u_local)

  def _evaluate(self):
    val, _ = self.fespace._basis(self._u3(), want_val=True, want_grad=False)
    return val

  def _placeholder(self):
    d = self.fespace.mesh.ndim
    return QExpr(shape=(d,), pullback=lambda ct: (ct, None))


class VectorNodalQFunctionGrad(NodalQFunction):
  """Gradient [j, k] = d u_k / d x_j of a vector function (:213-225)."""

  def __init__(self, fespace, u_local=None):
    super().__init__(fespace, (fespace.mesh.ndim,), u_local)

  def _evaluate(self):
    _, g = self.fespace._basis(self._u3(), want_val=False, want_grad=True)
    return g                                            # (E, Q, d, d)

  def _placeholder(self):
    d = self.fespace.mesh.ndim
    return QExpr(shape=(d, d), pullback=lambda ct: (None, ct))


def grad(f):
  """Gradient of a q-function (fespace.py:233-241)."""
  if isinstance(f, ScalarNodalQFunction):
    return ScalarNodalQFunctionGrad(fespace=f.fespace, u_local=f.u_local)
  if isinstance(f, VectorNodalQFunction):
    return VectorNodalQFunctionGrad(fespace=f.fespace, u_local=f.u_local)

  # A plain function of the coordinate x: differentiate pointwise with
  # autograd (the reference falls back to jax.grad here).
  def _grad_f(x: QExpr) -> QExpr:
    with torch.enable_grad():
      xv = x.val.detach().clone().requires_grad_(True)
      out = f(QExpr(xv))
      out = out.val if isinstance(out, QExpr) else out
      (g,) = torch.autograd.grad(out.sum(), xv)
    return QExpr(g.detach())

  return _grad_f


def div(f):
  """Divergence of a vector-valued q-function (fespace.py:244-248)."""
  def _divf(x):
    return qexpr.trace(grad(f)(x))
  return _divf


# ------------------------------------------------------------ the FE space
@dataclasses.dataclass(frozen=True, eq=False)
class FiniteElementSpace:
  """Nodal finite element space on a mesh with a tensor quadrature."""
  mesh: Mesh
  quadrature: Quadrature1D
  interpolator: BarycentricInterpolator
  _cache: dict = dataclasses.field(default_factory=dict, repr=False,
                                   compare=False)

  @classmethod
  def create(cls, mesh: Mesh, quadrature: Quadrature1D) -> 'FiniteElementSpace':
    interpolator = BarycentricInterpolator(
        ndim=mesh.ndim, gridpoints_1d=mesh.gridpoints_1d,
        evalpoints_1d=quadrature.nodes)
    return cls(mesh=mesh, quadrature=quadrature, interpolator=interpolator,
               _cache={})

  # The reference computes and stores `invjacs (E,Q,d,d)`, `jacdets (E,Q)` and
  # `quad_coords (E,Q,d)` in `create` (core/fespace.py:330-348): 13 reals per
  # quadrature point, 14 GB at 64^3 elements / p = 7.  Here they are built on
  # first use (one `sfem_geom_factors` launch): the fused operators evaluate
  # the geometry of multilinear elements in registers and never ask for them.
  def _geometry(self):
    if 'geometry' not in self._cache:
      i1, g1 = self._matrices()
      self._cache['geometry'] = _ops.geom_factors(
          self.mesh.element_coords(), i1, g1, self.mesh.ndim,
          self.mesh.gridpoints_1d.num_points, self.quadrature.num_points,
          want_quad_coords=True)
    return self._cache['geometry']

  @property
  def invjacs(self) -> torch.Tensor:
    return self._geometry()[0]

  @property
  def jacdets(self) -> torch.Tensor:
    return self._geometry()[1]

  @property
  def quad_coords(self) -> torch.Tensor:
    return self._geometry()[2]

  def replace(self, **kw):
    kw.setdefault('_cache', {})
    return dataclasses.replace(self, **kw)

  # ------------------------------------------------------------- properties
  @property
  def num_elements(self) -> int:
    return self.mesh.num_elements

  @property
  def num_quadrature_points_per_element(self) -> int:
    return int(self.quadrature.num_points ** self.mesh.ndim)

  @property
  def dtype(self):
    return self.mesh.dtype

  @property
  def device(self):
    return self.mesh.device

  @property
  def is_collocated(self) -> bool:
    return self.interpolator.is_collocated

  def _matrices(self):
    return _device_matrices(self.interpolator, self.dtype, self.device,
                            self._cache)

  def wdet(self) -> torch.Tensor:
    """`jacdets * quadrature weights`, shape (E, Q)."""
    if 'wdet' not in self._cache:
      w = torch.as_tensor(self.quadrature.weights_nd(self.mesh.ndim),
                          dtype=self.dtype, device=self.device)
      self._cache['wdet'] = (self.jacdets * w[None, :]).contiguous()
    return self._cache['wdet']

  def _basis(self, u3, want_val, want_grad):
    """u3 (E, n, nc) -> values (E,Q,nc), physical gradients (E,Q,d,nc)."""
    i1, g1 = self._matrices()
    u3 = u3.to(self.dtype)
    ev = autodiff.basis_eval if autodiff.needs_grad(u3) else _ops.basis_eval
    return ev(
        u3, i1, g1, self.invjacs if want_grad else None, self.mesh.ndim,
        self.mesh.gridpoints_1d.num_points, self.quadrature.num_points,
        self.is_collocated, want_val, want_grad)

  # ------------------------------------------------------------ q-functions
  def _evaluate(self, f) -> torch.Tensor:
    """Evaluates a q-function at every element's quadrature points."""
    out = f(QExpr(self.quad_coords))
    if isinstance(out, QExpr):
      if out.is_linear:
        raise ValueError('cannot evaluate a placeholder q-function')
      return out.val
    out = torch.as_tensor(out, dtype=self.dtype, device=self.device)
    if out.dim() == 0:
      return out.expand(self.num_elements,
                        self.num_quadrature_points_per_element)
    return out

  def scalar_function(self, u_local):
    expected = (self.num_elements, self.mesh.num_nodes_per_element)
    if u_local is not None and tuple(u_local.shape) != expected:
      raise ValueError(
          f'Expecting shape {expected} but got {tuple(u_local.shape)=}')
    return ScalarNodalQFunction(fespace=self, u_local=u_local)

  def vector_function(self, u_local):
    expected = (self.num_elements, self.mesh.num_nodes_per_element,
                self.mesh.ndim)
    if u_local is not None and tuple(u_local.shape) != expected:
      raise ValueError(
          f'Expecting shape {expected} but got {tuple(u_local.shape)=}')
    return VectorNodalQFunction(fespace=self, u_local=u_local)

  def integrate(self, f) -> torch.Tensor:
    """Quadrature of a scalar q-function over the mesh (0-dim tensor)."""
    w = self._evaluate(f)
    expected = (self.num_elements, self.num_quadrature_points_per_element)
    if tuple(w.shape) != expected:
      raise ValueError(
          'Expecting an array of shape (num elements, num quadrature points), '
          f'that is ({expected}) but got: {tuple(w.shape)}')
    res = torch.zeros(1, dtype=torch.float64, device=self.device)
    _ops.dot(w.to(self.dtype).contiguous().reshape(-1),
             self.wdet().reshape(-1), res, 0)
    return res[0].to(self.dtype)

  def local_covector(self, form, funs) -> torch.Tensor:
    """Local covector `(E, n) + value_shape` of the functional obtained by
    fixing every argument of the multilinear `form` except the placeholder."""
    def _is_input(f):
      return isinstance(f, NodalQFunction) and f.u_local is None

    if sum(_is_input(f) for f in funs) != 1:
      raise ValueError('Exactly one `QFunction` must be a nodal function and '
                       'have `None` as nodal values')
    placeholder = [f for f in funs if _is_input(f)][0]
    if placeholder.fespace is not self:
      # mixed forms (e.g. div(v) q): the covector lives in the space that
      # `local_covector` is called on, like the reference (:465-470).
      pass
    expr = form(*funs)(QExpr(self.quad_coords))
    if not isinstance(expr, QExpr) or not expr.is_linear:
      raise ValueError('the form does not depend on the placeholder function')
    if expr.shape != ():
      raise ValueError(f'the form must be scalar valued, got {expr.shape}')
    E, Q = self.num_elements, self.num_quadrature_points_per_element
    ones = torch.ones((E, Q), dtype=self.dtype, device=self.device)
    c0, c1 = expr.pullback(ones)
    value_shape = placeholder.value_shape
    nc = int(np.prod(value_shape)) if value_shape else 1
    d = self.mesh.ndim
    if c0 is not None:
      c0 = c0.to(self.dtype).reshape(E, Q, nc)
    if c1 is not None:
      c1 = c1.to(self.dtype).reshape(E, Q, d, nc)
    i1, g1 = self._matrices()
    ev_t = (autodiff.basis_eval_t if autodiff.needs_grad(c0, c1)
            else _ops.basis_eval_t)
    out = ev_t(
        c0, c1, i1, g1, self.invjacs, self.wdet(), d,
        self.mesh.gridpoints_1d.num_points, self.quadrature.num_points, nc,
        self.is_collocated)
    return out.reshape((E, self.mesh.num_nodes_per_element) + value_shape)

  # -------------------------------------------------------- fused operators
  def helmholtz_operator(self, dirichlet_mask=None, geometry='auto',
                         assembly='auto'):
    """Fused `out = mask * scatter((l0 B + l1 A)_local(gather(u)))`.

    `geometry`: 'auto' evaluates the geometric factors of affine / multilinear
    elements in registers and stores 6 factors per point only for curved
    elements; 'stored' stores them for every element (same results to
    rounding).  `assembly`: how shared nodes are summed, see
    `operators.HelmholtzOperator.create`.
    """
    from swirl_fem_amd.core import operators
    # cached per (mask object, options); the entry keeps the mask alive so that
    # its id cannot be recycled by another tensor
    key = ('helmholtz', None if dirichlet_mask is None else id(dirichlet_mask),
           geometry, assembly)
    hit = self._cache.get(key)
    if hit is not None and hit[0] is dirichlet_mask:
      return hit[1]
    if not self.is_collocated and assembly in ('auto', 'atomic') and (
        operators.supports_two_grid(self) is None):
      # quadrature != nodes: interpolate, fused element kernel on the
      # quadrature grid, transposed interpolation
      op = operators.TwoGridHelmholtzOperator.create(
          self, dirichlet_mask, 'stored' if geometry == 'stored' else 'auto')
    else:
      op = operators.HelmholtzOperator.create(self, dirichlet_mask, geometry,
                                              assembly)
    self._cache[key] = (dirichlet_mask, op)
    return op


def _device_matrices(interpolator, dtype, device, cache):
  key = ('mats', dtype, str(device))
  if key not in cache:
    i1, _ = interpolation.matrices_1d(interpolator.gridpoints_1d,
                                      interpolator.evalpoints_1d)
    g1 = interpolator._interp_grad_matrix_1d()
    cache[key] = tuple(
        torch.as_tensor(np.array(m), dtype=dtype, device=device)
        for m in (i1, g1))
  return cache[key]
