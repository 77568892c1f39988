"""Multi-threaded CPU restatement of the reference's CG iteration -- TEST /
BENCH INFRASTRUCTURE ONLY (imported by tests/ and by bench.py's cpu_baseline
leg; never by swirl_fem_amd/).

Same algorithm as `oracle/sfem_oracle.py` (= the reference): dense Kronecker
gradient matrix G (Q, n, d) applied to every element (core/interpolation.py:
288-292), 9 + 1 stored geometric arrays (core/fespace.py:338-346), the
transposed application (core/fespace.py:458-471), scatter-add, un-fused PCG
(linalg/cg.py:75-86) -- but with the element-batch contractions handed to
torch-CPU GEMMs so that all host cores work, which is how XLA-CPU would run
the reference's einsums.  Checked against the NumPy oracle in
tests/test_oracle_pins.py; parity is pinned there, not here.
"""

from __future__ import annotations

import time

import numpy as np
import torch

from oracle import sfem_oracle as O


class StiffnessCG:
  """Dirichlet Laplacian A = mask * scatter(A_loc(gather(.))) + plain CG."""

  def __init__(self, node_coords, elements, P, dirichlet):
    # Same arrays as `O.FESpace` builds (fespace.py:338-346: jacs[m,q,i,j] =
    # sum_n x[m,n,j] G[q,n,i], inverse, signed determinant), with the batched
    # contraction and the 3x3 inverses handed to torch so that all host cores
    # work (the NumPy einsum takes 37 s for 16^3 elements at p = 7).
    node_coords = np.asarray(node_coords, dtype=np.float64)
    elements = np.asarray(elements)
    d = node_coords.shape[1]
    interp = O.Interpolator(d, P, 'gll', P, 'gll')
    weights = O.weights_nd(O.quadrature_weights(P, 'gll'), d)
    G = interp.interpolation_matrix_grad()                     # (Q, n, d)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    self.num_nodes = node_coords.shape[0]
    self.elements = t(elements.astype(np.int64))
    Q, n, _ = G.shape
    self.Q, self.n, self.d = Q, n, d
    self.G2 = t(G.transpose(0, 2, 1).reshape(Q * d, n))        # (Q d, n)
    E = elements.shape[0]
    xe = t(node_coords)[self.elements]                         # (E, n, d)
    # (E, d_j, n) @ (n, Q d_i) -> (E, j, q, i) -> jacs (E, q, i, j)
    jacs = (xe.transpose(1, 2) @ self.G2.T).reshape(E, d, Q, d).permute(
        0, 2, 3, 1).contiguous()
    self.invjacs = torch.linalg.inv(jacs)                      # (E, Q, d, d)
    self.wdet = torch.linalg.det(jacs) * t(weights)[None, :]   # (E, Q)
    self.interior = t(1.0 - np.asarray(dirichlet, dtype=np.float64))
    self.P1 = P
    self.D1 = t(O.differentiation_matrix_1d(O.nodes_1d(P, 'gll'), 'gll'))

  def apply(self, u):
    E = self.elements.shape[0]
    ul = u[self.elements]                                      # gather (E, n)
    ref = (ul @ self.G2.T).reshape(E, self.Q, self.d)          # GEMM
    grad = torch.einsum('mqi,mqji->mqj', ref, self.invjacs)
    flux = torch.einsum('mqj,mqji->mqi', grad * self.wdet[..., None],
                        self.invjacs)
    loc = flux.reshape(E, self.Q * self.d) @ self.G2           # GEMM (E, n)
    out = torch.zeros(self.num_nodes, dtype=u.dtype)
    out.index_add_(0, self.elements.reshape(-1), loc.reshape(-1))
    return self.interior * out

  def apply_sum_factorised(self, u):
    """The same operator with the element gradient contracted axis by axis
    with the 1D matrix (O(P^4) per element instead of O(P^6)) -- the variant
    SURVEY 8(d) asks to report beside the dense one.  3D only."""
    E, P, d = self.elements.shape[0], self.P1, self.d
    ul = u[self.elements].reshape(E, P, P, P)
    D = self.D1
    ref = torch.stack([torch.einsum('am,emij->eaij', D, ul),
                       torch.einsum('im,eamj->eaij', D, ul),
                       torch.einsum('jm,eaim->eaij', D, ul)],
                      dim=-1).reshape(E, self.Q, d)
    grad = torch.einsum('mqi,mqji->mqj', ref, self.invjacs)
    flux = torch.einsum('mqj,mqji->mqi', grad * self.wdet[..., None],
                        self.invjacs).reshape(E, P, P, P, d)
    loc = (torch.einsum('am,eaij->emij', D, flux[..., 0]) +
           torch.einsum('im,eaij->eamj', D, flux[..., 1]) +
           torch.einsum('jm,eaij->eaim', D, flux[..., 2]))
    out = torch.zeros(self.num_nodes, dtype=u.dtype)
    out.index_add_(0, self.elements.reshape(-1), loc.reshape(-1))
    return self.interior * out

  def cg_iterations(self, b, iters=None, budget_s=None, sum_factorised=False):
    """Un-fused CG body; returns (x, iterations, seconds)."""
    apply = self.apply_sum_factorised if sum_factorised else self.apply
    return self._cg(apply, b, iters, budget_s)

  def _cg(self, apply, b, iters, budget_s):
    self_apply = apply
    x = torch.zeros_like(b)
    r = b - self_apply(x)
    p = r.clone()
    gamma = torch.dot(r, r)
    self_apply(p)                                  # warm-up
    k, t0 = 0, time.perf_counter()
    while True:
      Ap = self_apply(p)
      alpha = gamma / torch.dot(p, Ap)
      x = x + alpha * p
      r = r - alpha * Ap
      g2 = torch.dot(r, r)
      p = r + (g2 / gamma) * p
      gamma = g2
      k += 1
      el = time.perf_counter() - t0
      if (iters is not None and k >= iters) or (
          budget_s is not None and k >= 3 and el > budget_s):
        return x, k, el
