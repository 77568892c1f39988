"""1D node families, quadrature rules and barycentric interpolation matrices.

Host-side (NumPy, fp64) setup data for the MI355X operator kernels.  Mirrors the
public surface of the reference module `swirl_fem/core/interpolation.py`
(`NodeType` :29, `Nodes1D` :38-91, `Quadrature1D` :95-140,
`BarycentricInterpolator` :143-292) so that solver code written against the
reference keeps working.

Differences in *how* (not *what*):
  * the 1D matrices are built with vectorised NumPy instead of Python loops and
    are cached per (grid, eval) pair -- the reference rebuilds them every trace;
  * the d-fold Kronecker products (`interpolation_matrix`,
    `interpolation_matrix_grad`) are still available for API parity but the
    device kernels never form them: they contract the 1D factors axis by axis
    (sum factorisation, see `csrc/sfem_basis.hip`);
  * `interpolate` / `interpolate_grad` act on a single element like the
    reference (`:254-263`, `:288-292`) but run the batched HIP basis kernels.
"""

from __future__ import annotations

import dataclasses
import enum
import functools

import numpy as np
import scipy.special


@enum.unique
class NodeType(enum.Enum):
  """Distributions of collocation / quadrature nodes on [-1, 1]."""
  NEWTON_COTES = 'newton_cotes'
  GAUSS_LEGENDRE = 'gauss_legendre'
  GAUSS_LOBATTO_LEGENDRE = 'gauss_lobatto_legendre'
  SINGLE = 'single_point'


@dataclasses.dataclass(frozen=True, eq=False)
class Nodes1D:
  """A sequence of 1D nodes on the reference element [-1, 1]."""

  num_points: int
  node_type: NodeType
  node_values: np.ndarray

  @classmethod
  def create_single_point(cls, node_value) -> 'Nodes1D':
    return cls(num_points=1, node_type=NodeType.SINGLE,
               node_values=np.array([node_value], dtype=np.float64))

  @classmethod
  def create(cls, num_points: int, node_type: NodeType) -> 'Nodes1D':
    if node_type == NodeType.NEWTON_COTES:
      values = np.linspace(-1, 1, num=num_points, dtype=np.float64)
    elif node_type == NodeType.GAUSS_LEGENDRE:
      values, _ = np.polynomial.legendre.leggauss(deg=num_points)
    elif node_type == NodeType.GAUSS_LOBATTO_LEGENDRE:
      # interior GLL nodes = roots of P'_{n-1} = Gauss-Jacobi(1,1) nodes
      if num_points == 2:
        inner = np.array([], dtype=np.float64)
      else:
        inner, _ = scipy.special.roots_jacobi(num_points - 2, alpha=1, beta=1)
      values = np.concatenate([[-1.], inner, [1.]])
    else:
      raise ValueError(f'Node type not recognized: {node_type}')
    return cls(num_points=num_points, node_type=node_type, node_values=values)

  def is_continuous(self) -> bool:
    """Whether the end points are nodes (C0 continuity across elements)."""
    return bool(self.node_values[0] == -1.0 and self.node_values[-1] == 1.0)

  def replace(self, **kw) -> 'Nodes1D':
    return dataclasses.replace(self, **kw)

  def __eq__(self, other):
    if not isinstance(other, Nodes1D) or self.node_type != other.node_type:
      return False
    if self.node_type == NodeType.SINGLE:
      return bool(np.allclose(self.node_values, other.node_values, rtol=0,
                              atol=np.finfo(self.node_values.dtype).eps))
    return self.num_points == other.num_points

  def __hash__(self):
    if self.node_type == NodeType.SINGLE:
      return hash((self.node_type, float(self.node_values[0])))
    return hash((self.node_type, self.num_points))

  def _cache_key(self):
    if self.node_type == NodeType.SINGLE:
      return (self.node_type.value, float(self.node_values[0]))
    return (self.node_type.value, self.num_points)


@dataclasses.dataclass(frozen=True, eq=False)
class Quadrature1D:
  """A 1D quadrature rule on [-1, 1]."""

  num_points: int
  quadrature_type: NodeType
  nodes: Nodes1D
  weights: np.ndarray

  @classmethod
  def create_from_nodes_1d(cls, nodes: Nodes1D) -> 'Quadrature1D':
    n = nodes.num_points
    if nodes.node_type == NodeType.GAUSS_LEGENDRE:
      _, weights = np.polynomial.legendre.leggauss(deg=n)
    elif nodes.node_type == NodeType.GAUSS_LOBATTO_LEGENDRE:
      weights = (2 / (n * (n - 1))) / np.square(
          scipy.special.eval_legendre(n - 1, nodes.node_values))
    elif nodes.node_type == NodeType.NEWTON_COTES:
      weights = (1 / (n - 1)) * np.array([1.] + (n - 2) * [2.] + [1.])
    else:
      raise ValueError(f'Quadrature type not recognized: {nodes.node_type}')
    return cls(num_points=n, quadrature_type=nodes.node_type, nodes=nodes,
               weights=weights)

  @classmethod
  def create(cls, num_points: int, quadrature_type: NodeType) -> 'Quadrature1D':
    return cls.create_from_nodes_1d(
        Nodes1D.create(num_points=num_points, node_type=quadrature_type))

  def weights_nd(self, ndim: int) -> np.ndarray:
    """Tensor-product weights, flattened with axis 0 slowest."""
    return functools.reduce(np.outer, [self.weights] * ndim).reshape(-1)

  def replace(self, **kw) -> 'Quadrature1D':
    return dataclasses.replace(self, **kw)


def barycentric_weights(gridpoints_1d: Nodes1D) -> np.ndarray:
  """Closed-form barycentric weights per node family (reference :180-208)."""
  n = gridpoints_1d.num_points
  signs = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
  if gridpoints_1d.node_type == NodeType.NEWTON_COTES:
    return signs * scipy.special.binom(n - 1, np.arange(n))
  if gridpoints_1d.node_type == NodeType.GAUSS_LEGENDRE:
    quad = Quadrature1D.create_from_nodes_1d(gridpoints_1d)
    return signs * np.sqrt((1 - np.square(quad.nodes.node_values)) *
                           quad.weights)
  if gridpoints_1d.node_type == NodeType.GAUSS_LOBATTO_LEGENDRE:
    quad = Quadrature1D.create_from_nodes_1d(gridpoints_1d)
    return signs * np.sqrt(quad.weights)
  raise ValueError(f'Gridpoint type not supported: {gridpoints_1d.node_type}')


@functools.lru_cache(maxsize=None)
def _matrices_1d_cached(grid_key, eval_key, grid_vals, eval_vals, grid_type):
  """(interp (q,P), diff (P,P)) for a (grid, eval) pair; cached."""
  grid = Nodes1D(num_points=len(grid_vals), node_type=NodeType(grid_type),
                 node_values=np.array(grid_vals, dtype=np.float64))
  w = barycentric_weights(grid)
  x = np.array(grid_vals, dtype=np.float64)
  xe = np.array(eval_vals, dtype=np.float64)

  # second (true) barycentric form, vectorised; rows that hit a node exactly
  # become unit rows (exact floating-point comparison is intentional).
  diff = xe[:, None] - x[None, :]
  hit = diff == 0.0
  with np.errstate(divide='ignore', invalid='ignore'):
    terms = w[None, :] / diff
  # summation order along the node axis follows a left-to-right Python sum
  denom = np.zeros(len(xe))
  for j in range(len(x)):
    denom = denom + terms[:, j]
  with np.errstate(invalid='ignore'):
    interp = terms / denom[:, None]
  row_hit = hit.any(axis=1)
  interp[row_hit] = hit[row_hit].astype(np.float64)

  dx = x[:, None] - x[None, :]
  np.fill_diagonal(dx, 1.0)
  dmat = (w[None, :] / w[:, None]) / dx
  np.fill_diagonal(dmat, 0.0)
  np.fill_diagonal(dmat, -dmat.sum(axis=1))
  interp.setflags(write=False)
  dmat.setflags(write=False)
  return interp, dmat


def matrices_1d(gridpoints_1d: Nodes1D, evalpoints_1d: Nodes1D):
  """Returns `(I, D)`: I[a,i]=l_i(xe_a) of shape (q,P); D[i,j]=l_j'(x_i)."""
  return _matrices_1d_cached(
      gridpoints_1d._cache_key(), evalpoints_1d._cache_key(),
      tuple(float(v) for v in gridpoints_1d.node_values),
      tuple(float(v) for v in evalpoints_1d.node_values),
      gridpoints_1d.node_type.value)


class BarycentricInterpolator:
  """Barycentric interpolation from tensor grid points to evaluation points."""

  def __init__(self, ndim: int, gridpoints_1d: Nodes1D, evalpoints_1d: Nodes1D):
    self.ndim = ndim
    self.gridpoints_1d = gridpoints_1d
    self.evalpoints_1d = evalpoints_1d

  # -- 1D building blocks -------------------------------------------------
  def _barycentric_weights(self) -> np.ndarray:
    return barycentric_weights(self.gridpoints_1d)

  def _interpolation_matrix_1d(self) -> np.ndarray:
    return np.array(matrices_1d(self.gridpoints_1d, self.evalpoints_1d)[0])

  def _differentiation_matrix_1d(self) -> np.ndarray:
    return np.array(matrices_1d(self.gridpoints_1d, self.evalpoints_1d)[1])

  def _interp_grad_matrix_1d(self) -> np.ndarray:
    """(I @ D): derivative of the interpolant, sampled at the eval points."""
    i1, d1 = matrices_1d(self.gridpoints_1d, self.evalpoints_1d)
    return i1 @ d1

  @property
  def is_collocated(self) -> bool:
    """Grid and evaluation points coincide (interpolation is the identity)."""
    return self.gridpoints_1d == self.evalpoints_1d

  # -- dense Kronecker forms (API parity; never used by the kernels) -------
  def interpolation_matrix(self) -> np.ndarray:
    return functools.reduce(np.kron,
                            [self._interpolation_matrix_1d()] * self.ndim)

  def interpolation_matrix_grad(self) -> np.ndarray:
    i1 = self._interpolation_matrix_1d()
    g1 = self._interp_grad_matrix_1d()
    mats = []
    for k in range(self.ndim):
      row = [g1 if j == k else i1 for j in range(self.ndim)]
      mats.append(functools.reduce(np.kron, row))
    return np.stack(mats, axis=-1)

  # -- single-element application (device tensors) -------------------------
  def interpolate(self, x):
    """Interpolates one element's nodal values `(P**ndim,)` to eval points."""
    from swirl_fem_amd.core import basis
    n = self.gridpoints_1d.num_points ** self.ndim
    if tuple(x.shape) != (n,):
      raise AssertionError(tuple(x.shape))
    if self.is_collocated:
      return x
    return basis.interp(self, x.reshape(1, n, 1)).reshape(-1)

  def interpolate_grad(self, x):
    """Reference-space gradient `(q**ndim, ndim)` of one element's values."""
    from swirl_fem_amd.core import basis
    n = self.gridpoints_1d.num_points ** self.ndim
    if tuple(x.shape) != (n,):
      raise AssertionError(tuple(x.shape))
    out = basis.ref_grad(self, x.reshape(1, n, 1))  # (1, Q, d, 1)
    return out.reshape(out.shape[1], self.ndim)
