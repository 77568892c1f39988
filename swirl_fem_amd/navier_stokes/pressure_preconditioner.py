"""Two-level Schwarz preconditioner for the pressure operator E = D Q D^T.

The reference's stepper takes a `pressure_preconditioner` callable
(navier_stokes/navier_stokes.py:354, :419-420, :449-452) and ships none beyond
the projection that removes the constant mode (:73-78); with it the pressure
solve of a P_N - P_{N-2} discretisation needs hundreds of iterations per step
(380 - 680 on one 64^3 block of BASELINE config 4), and their number grows
with the polynomial order and with the number of elements across the domain.
This module is an opt-in for that hook -- results agree to the solver
tolerance, the default stays the reference's:

    M^-1 r = P ( sum_e R_e^T E~_e^+ R_e r  +  R_0^T E_0^+ R_0 r )

* local part: the pressure nodes of an element belong to it alone, so the
  diagonal block E_ee = D_e Q D_e^T is a natural subdomain problem.  With the
  element taken as a box in its own coordinates and Q (= (dt / beta) x inverse
  assembled velocity mass x Dirichlet mask) as a tensor product along the
  element's axes -- both exact on Cartesian meshes, an approximation elsewhere
  that only costs iterations -- E_ee = sum_d k_d^2 (A_d in direction d) (x) (B
  in the others) with 1D matrices A_d = G diag(f_d) G^T, B_d = H diag(f_d) H^T,
  G = I^T W D and H = I^T W (I: pressure -> velocity nodes, W: quadrature
  weights, D: 1D derivative).  The generalised eigenvectors of (A_d, B_d)
  diagonalise it: two small tensor contractions per direction and a division
  (fast diagonalisation, Lynch, Rice & Thomas 1964; for E: Fischer, J. Comput.
  Phys. 133, 1997).  The element's constant mode is left to the coarse level
  (pseudo-inverse);
* coarse part: R_0 sums a vector over each element (piecewise constants),
  E_0 = R_0 E R_0^T is assembled exactly from the elements' boundary fluxes
  int_e d(phi_n)/dx_c (sparse, 27 entries per row on a structured mesh) and
  applied through a fixed Chebyshev polynomial of the Jacobi-scaled E_0
  (`sfem_ell_chebyshev`: linear and symmetric, which a truncated inner CG is
  not; spectrum bounds by Lanczos at setup); as a dense pseudo-inverse up to
  4096 elements; and on a uniform, fully periodic box -- where E_0 is a
  circulant stencil, verified against E_0 itself at setup -- exactly, by two
  FFTs (`_circulant_coarse`);
* P: the reference's nullspace projection.
"""

from __future__ import annotations

import numpy as np
import torch

from swirl_fem_amd.navier_stokes import navier_stokes as ns


DENSE_COARSE_MAX = 4096     # elements up to which E_0^+ is kept as a matrix


def _centre_cofactors(xe, P, d):
  """(E, d) squared row norms k_a^2 = sum_c (det J dxi_a / dx_c)^2 of the
  Jacobian of the element's multilinear map at its centre."""
  E = xe.shape[0]
  x = xe.reshape((E,) + (P,) * d + (d,))
  rows = []
  for a in range(d):
    hi = x.select(1 + a, P - 1)
    lo = x.select(1 + a, 0)
    diff = hi - lo                                  # (E, P.., d) over the face
    # corners of the face: mean over the 2^(d-1) edges along axis a
    corner = diff
    for ax in range(d - 1):
      corner = corner.index_select(1 + ax, torch.tensor(
          [0, P - 1], device=x.device))
    rows.append(corner.reshape(E, -1, d).mean(dim=1) / 2)
  J = torch.stack(rows, dim=1)                      # J[a, c] = dx_c / dxi_a
  det = torch.linalg.det(J)
  K = det[:, None, None] * torch.linalg.inv(J).transpose(1, 2)  # K[a][c]
  return (K ** 2).sum(dim=2)


class SchwarzPressurePreconditioner:
  """`z = M^-1 r` for `cg(E, b, M=...)`; see the module docstring."""

  def __init__(self, sem, dt, time_order, coarse_iterations=None,
               coarse_solver='chebyshev'):
    self.sem = sem
    # an ensemble (`StokesSEM.ensemble`): the local part is element-wise
    # anyway, the coarse problem is one block per member
    self.members = sem.members
    vmesh = sem.velocity.mesh
    pmesh = sem.pressure.pspace.mesh
    d = self.d = vmesh.ndim
    P = self.P = vmesh.gridpoints_1d.num_points
    Pp = self.Pp = pmesh.gridpoints_1d.num_points
    # partitions: the local part is element-wise (no communication); the
    # coarse level is solved redundantly on every rank after one all-gather of
    # the element sums -- for uniform periodic boxes cut into blocks
    # (BASELINE config 4), where it is a circulant stencil (`_circulant_blocks`)
    self.partitioned = sem.is_partitioned
    dev, dtype = vmesh.device, vmesh.node_coords.dtype
    E = vmesh.num_elements
    self.pel = pmesh.elements.to(torch.int64)       # (E, Pp^d)
    # --- Q on the velocity nodes (what `StokesSEM.E` applies between D^T
    # and D): (dt / beta) / QQ^T(mass), zero on Dirichlet rows
    beta_k = float(ns.bdfk_coeffs(time_order)[-1])
    mass = sem.velocity.exchange(sem.velocity_mass_diag)
    q = (dt / beta_k) / mass * sem.velocity.interior_mask
    q = q[:, 0].contiguous()                        # same for every component
    vel = vmesh.elements.to(torch.int64)
    qloc = q[vel].reshape((E,) + (P,) * d)          # (E, P, P[, P])
    ref = qloc[(slice(None),) + (1,) * d]           # an element-interior node
    t = ref ** (1.0 / d)
    f = []
    for a in range(d):
      idx = [slice(None)] + [1] * d
      idx[1 + a] = slice(None)
      f.append(qloc[tuple(idx)] / t[:, None] ** (d - 1))    # (E, P)
    k2 = _centre_cofactors(vmesh.element_coords(), P, d)     # (E, d)
    # --- 1D matrices
    interp = np.asarray(sem.pressure.pspace.interpolator.
                        _interpolation_matrix_1d())            # (P, Pp)
    w = np.asarray(sem.velocity.vspace.quadrature.weights)
    dmat = np.asarray(sem.velocity.vspace.interpolator.
                      _differentiation_matrix_1d())
    H = interp.T * w[None, :]                                   # (Pp, P)
    G = H @ dmat
    # --- generalised eigenproblems, one per distinct factor row
    S_all, lam_all, case = [], [], []
    offset = 0
    for a in range(d):
      fa = f[a].cpu().numpy()
      scale = np.abs(fa).max(axis=1, keepdims=True)
      key = np.round(fa / scale, 10) * scale        # exact on uniform meshes
      uniq, inv = np.unique(key, axis=0, return_inverse=True)
      A = np.einsum('pi,ci,qi->cpq', G, uniq, G)
      B = np.einsum('pi,ci,qi->cpq', H, uniq, H)
      L = np.linalg.cholesky(B)
      Li = np.linalg.inv(L)
      lam, V = np.linalg.eigh(Li @ A @ Li.transpose(0, 2, 1))
      S = Li.transpose(0, 2, 1) @ V                 # S^T B S = I, S^T A S = lam
      S_all.append(S)
      lam_all.append(np.maximum(lam, 0.0))
      case.append(inv.reshape(-1) + offset)
      offset += len(uniq)
    self.S = torch.as_tensor(np.concatenate(S_all), dtype=dtype, device=dev)
    lam = torch.as_tensor(np.concatenate(lam_all), dtype=dtype, device=dev)
    self.case = [torch.as_tensor(c, device=dev) for c in case]
    self.case32 = torch.stack(self.case).to(torch.int32).contiguous()
    # eigenvalues of the element blocks, pseudo-inverted
    shape = lambda a: [E] + [Pp if b == a else 1 for b in range(d)]
    ev = sum((k2[:, a, None] * lam[self.case[a]]).reshape(shape(a))
             for a in range(d))
    top = ev.reshape(E, -1).max(dim=1).values.reshape([E] + [1] * d)
    self.inv_ev = torch.where(ev > 1e-10 * top, 1.0 / ev,
                              torch.zeros_like(ev)).contiguous()
    # element e owns the pressure nodes [e n, (e + 1) n): no index array
    ident = torch.arange(self.pel.numel(), device=dev).reshape(self.pel.shape)
    self.pel_arg = None if torch.equal(self.pel, ident) else self.pel.contiguous()
    # --- coarse level
    self.coarse_solver = coarse_solver
    self._build_coarse(q, dtype, dev)
    # steps for a residual reduction of ~ 1e-2 on [lmin, lmax]:
    # sqrt(kappa) ln(2 / eps) / 2
    lmin, lmax = self.coarse_bounds
    self.coarse_iterations = (int(coarse_iterations) if coarse_iterations
                              else max(8, int(np.ceil(
                                  np.sqrt(lmax / lmin) * np.log(200.0) / 2))))
    self.project = ns._NullspaceProjection(sem)
    # with the dense coarse level every step of an apply is plain device work
    # on fixed buffers: a CG iteration that uses it can be replayed as a graph
    self.capturable = self.E0_pinv is not None

  # ------------------------------------------------------------- coarse level
  def _build_coarse(self, q, dtype, dev):
    """E_0 = R_0 E R_0^T from the elements' flux vectors
    g_e[n, c] = int_e d(phi_n)/dx_c (= D^T_local of the constant 1)."""
    import scipy.sparse as sp
    sem = self.sem
    vmesh = sem.velocity.mesh
    E, n = vmesh.elements.shape
    d = self.d
    ones = torch.ones((E, self.Pp ** d), dtype=dtype, device=dev)
    g = sem.Dt_local(ones)                                       # (E, n, d)
    # node classes: periodic images are one node for the operator
    ids = vmesh.elements.to(torch.int64)
    if vmesh.node_indices is not None:
      ids = torch.as_tensor(vmesh.node_indices, device=dev).to(
          torch.int64)[ids]
    gmax = float(g.abs().max())
    rows, cols, vals = [], [], []
    for c in range(d):
      nz = g[..., c].abs() > 1e-13 * gmax
      e_idx, loc = torch.nonzero(nz, as_tuple=True)
      rows.append(e_idx)
      cols.append(ids[e_idx, loc] * d + c)
      vals.append(g[e_idx, loc, c])
    rows, cols, vals = (torch.cat(v).cpu().numpy() for v in (rows, cols, vals))
    N = int(vmesh.num_nodes)
    Gs = sp.csr_matrix((vals, (rows, cols)), shape=(E, N * d))
    Gs.sum_duplicates()
    # the representative of a class carries the operator's q
    qn = np.repeat(q.cpu().numpy(), d)
    E0 = (Gs.multiply(qn[None, :]) @ Gs.T).tocsr()
    E0.sum_duplicates()
    self.coarse_diag = torch.as_tensor(E0.diagonal(), dtype=dtype, device=dev)
    Es = self.coarse_size = E // self.members
    # rows of equal length (ELL): y = sum_k vals[:, k] x[cols[:, k]] is plain
    # gather / multiply / add, which a HIP graph can hold (`_coarse_solve`)
    width = int(np.diff(E0.indptr).max())
    cols = np.zeros((E, width), dtype=np.int64)
    vals = np.zeros((E, width), dtype=np.float64)
    row = np.repeat(np.arange(E), np.diff(E0.indptr))
    slot = np.arange(E0.nnz) - np.repeat(E0.indptr[:-1], np.diff(E0.indptr))
    cols[row, slot] = E0.indices
    vals[row, slot] = E0.data
    self.E0_cols = torch.as_tensor(cols, device=dev)
    self.E0_vals = torch.as_tensor(vals, dtype=dtype, device=dev)
    # column-major copies for the Chebyshev kernel (`sfem_ell_chebyshev`)
    self.E0_cols_t = self.E0_cols.t().to(torch.int32).contiguous()
    self.E0_vals_t = self.E0_vals.t().contiguous()
    self._coarse_graph = None
    # constants are in the kernel of E_0 when nothing pins the pressure
    resid = np.abs(E0 @ np.ones(E)).max() / max(np.abs(E0.diagonal()).max(),
                                                1e-300)
    self.coarse_singular = bool(resid < 1e-8)
    if self.members > 1:
      # the members' blocks are equal: pseudo-inverse / spectrum of the first
      E0 = E0[:Es, :Es].tocsr()
      E = Es
    # small coarse problems (a few thousand elements: launch-bound steps): the
    # pseudo-inverse as a dense matrix, one matrix-vector product per apply
    self.E0_pinv = None
    self.E0_fft = None
    self.E0_blocks = None
    if E <= DENSE_COARSE_MAX and not self.partitioned:
      dense = torch.as_tensor(E0.toarray(), dtype=torch.float64, device=dev)
      w, V = torch.linalg.eigh(dense)
      keep = w > 1e-10 * w.abs().max()
      winv = torch.where(keep, 1.0 / torch.where(keep, w, torch.ones_like(w)),
                         torch.zeros_like(w))
      self.E0_pinv = ((V * winv[None, :]) @ V.t()).to(dtype).contiguous()
      self.coarse_bounds = (1.0, 2.0)          # (not used)
      return
    if self.partitioned:
      self.E0_fft = None
      self.E0_blocks = self._circulant_blocks(E0, dtype, dev)
      if self.E0_blocks is None:
        raise NotImplementedError(
            'Schwarz pressure preconditioner on a partitioned mesh needs a '
            'uniform, fully periodic box cut into blocks of at least 3 '
            'elements per direction (coarse solve by FFT)')
      self.coarse_bounds = (1.0, 2.0)          # (not used)
      return
    # a uniform, fully periodic box: E_0 is a (block-)circulant 27-point
    # stencil, which the discrete Fourier transform diagonalises -- the exact
    # pseudo-inverse for two small FFTs
    self.E0_fft = None
    if self.members == 1 and self.coarse_solver == 'chebyshev':
      self.E0_fft = self._circulant_coarse(E0, dtype, dev)
      if self.E0_fft is not None:
        self.coarse_bounds = (1.0, 2.0)        # (not used)
        return
    # spectrum of D^-1 E_0 on the complement of the constants: Lanczos (host,
    # SciPy, setup only) for the two ends; the Chebyshev polynomial is built
    # for [lmin, lmax] and stays positive definite as long as lmax is a bound
    import scipy.sparse.linalg as spla
    dm = 1.0 / np.sqrt(E0.diagonal())
    Ssym = sp.diags(dm) @ E0 @ sp.diags(dm)
    lmax = float(spla.eigsh(Ssym, k=1, which='LA', tol=1e-3,
                            return_eigenvectors=False)[0])
    # smallest non-zero eigenvalue: a few Lanczos steps on the deflated
    # operator (shift-invert would need a factorisation of the coarse matrix)
    k = 2 if self.coarse_singular else 1
    try:
      low = spla.eigsh(Ssym, k=k, which='SA', tol=1e-2, maxiter=20 * E,
                       return_eigenvectors=False)
      lmin = float(np.sort(low)[-1])
    except spla.ArpackNoConvergence as exc:
      got = np.sort(exc.eigenvalues)
      lmin = float(got[-1]) if len(got) >= k else lmax / (4.0 * E ** (2.0 / d))
    self.coarse_bounds = (0.8 * max(lmin, 1e-12 * lmax), 1.05 * lmax)

  def _circulant_coarse(self, E0, dtype, dev):
    """(perm, 1 / symbol) if the elements form a full tensor grid on which E_0
    is translation-invariant with periodic wrap-around (checked against E_0
    itself on a random vector), else None."""
    vmesh = self.sem.velocity.mesh
    d = self.d
    xc = vmesh.element_coords().mean(dim=1).cpu().numpy()        # (E, d)
    E = xc.shape[0]
    span = np.ptp(xc, axis=0).max() or 1.0
    idx, shape = [], []
    for a in range(d):
      vals = np.unique(np.round(xc[:, a] / span, 9))
      shape.append(len(vals))
      idx.append(np.searchsorted(vals, np.round(xc[:, a] / span, 9)))
    if int(np.prod(shape)) != E or min(shape) < 3:
      return None
    lin = np.ravel_multi_index(tuple(idx), tuple(shape))
    if len(np.unique(lin)) != E:
      return None
    perm = np.empty(E, dtype=np.int64)       # grid position -> element
    perm[lin] = np.arange(E)
    row = E0.getrow(int(perm[0]))
    stencil = np.zeros(shape)
    off = np.unravel_index(lin[row.indices], tuple(shape))
    np.add.at(stencil, off, row.data)
    symbol = np.fft.rfftn(stencil)
    if np.abs(symbol.imag).max() > 1e-9 * np.abs(symbol.real).max():
      return None
    symbol = symbol.real
    rng = np.random.default_rng(0)
    v = rng.standard_normal(E)
    want = E0 @ v
    got = np.empty(E)
    axes = tuple(range(d))
    got[perm] = np.fft.irfftn(symbol * np.fft.rfftn(v[perm].reshape(shape)),
                              s=shape, axes=axes).reshape(-1)
    if np.abs(got - want).max() > 1e-9 * np.abs(want).max():
      return None
    top = np.abs(symbol).max()
    inv = np.where(np.abs(symbol) > 1e-10 * top,
                   1.0 / np.where(symbol == 0, 1.0, symbol), 0.0)
    ident = bool((perm == np.arange(E)).all())
    return (None if ident else torch.as_tensor(perm, device=dev),
            torch.as_tensor(inv, dtype=dtype, device=dev), tuple(shape))

  def _circulant_blocks(self, E0, dtype, dev):
    """The circulant coarse operator of a periodic box held as blocks by the
    ranks: (own grid positions, everyone's grid positions, 1 / symbol, global
    shape), or None.  `E0` holds this rank's rows with the columns of its own
    elements; the rows of elements away from the block's faces are complete
    and give the stencil; the symbol is checked against them."""
    from swirl_fem_amd.distributed import comm
    vmesh = self.sem.velocity.mesh
    d = self.d
    xc = vmesh.element_coords().mean(dim=1)                     # (E, d)
    E = xc.shape[0]
    hi = comm.all_reduce_max_(xc.max(dim=0).values.clone())
    lo = -comm.all_reduce_max_((-xc).max(dim=0).values.clone())
    xc, lo, hi = xc.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy()
    shape, gidx = [], []
    for a in range(d):
      vals = np.unique(np.round(xc[:, a] / max(hi[a] - lo[a], 1e-300), 9))
      if len(vals) < 2:
        return None
      h = (vals[1] - vals[0]) * (hi[a] - lo[a])
      n = int(round((hi[a] - lo[a]) / h)) + 1
      k = np.round((xc[:, a] - lo[a]) / h).astype(np.int64)
      if np.abs((xc[:, a] - lo[a]) / h - k).max() > 1e-6 or n < 3:
        return None
      shape.append(n)
      gidx.append(k)
    glin = np.ravel_multi_index(tuple(gidx), tuple(shape))
    if len(np.unique(glin)) != E:
      return None
    nnz = np.diff(E0.indptr)
    full = 3 ** d
    if nnz.max() != full:
      return None
    row = E0.getrow(int(np.argmax(nnz)))
    i0 = int(np.argmax(nnz))
    stencil = np.zeros(shape)
    off = tuple((gidx[a][row.indices] - gidx[a][i0]) % shape[a]
                for a in range(d))
    np.add.at(stencil, off, row.data)
    axes = tuple(range(d))
    symbol = np.fft.rfftn(stencil, axes=axes)
    if np.abs(symbol.imag).max() > 1e-9 * np.abs(symbol.real).max():
      return None
    symbol = symbol.real
    # check on the complete rows: E_0 v against the FFT product for a v that
    # is the same on every rank
    rng = np.random.default_rng(0)
    V = rng.standard_normal(shape)
    Y = np.fft.irfftn(symbol * np.fft.rfftn(V, axes=axes), s=shape, axes=axes)
    rows = np.nonzero(nnz == full)[0]
    want = (E0[rows] @ V.reshape(-1)[glin])
    got = Y.reshape(-1)[glin[rows]]
    ok = np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    flag = torch.tensor([0.0 if ok else 1.0], device=dev)
    if float(comm.all_reduce_max_(flag)) != 0.0:
      return None
    top = np.abs(symbol).max()
    inv = np.where(np.abs(symbol) > 1e-10 * top,
                   1.0 / np.where(symbol == 0, 1.0, symbol), 0.0)
    mine = torch.as_tensor(glin, device=dev)
    everyone = torch.cat(comm.all_gather(mine))
    if everyone.numel() != int(np.prod(shape)):
      return None
    return (mine, everyone, torch.as_tensor(inv, dtype=dtype, device=dev),
            tuple(shape))

  def coarse_matvec(self, x):
    return (self.E0_vals * x[self.E0_cols]).sum(dim=1)

  def _coarse_solve(self, b):
    """E_0^+ b to the accuracy a preconditioner needs: a fixed Chebyshev
    polynomial of the Jacobi-scaled coarse matrix (`sfem_ell_chebyshev`: one
    small launch per step, no inner products, exactly linear and symmetric);
    `coarse_solver = 'cg'` keeps the truncated CG of the first version."""
    if self.E0_blocks is not None:
      from swirl_fem_amd.distributed import comm
      mine, everyone, inv, shape = self.E0_blocks
      G = torch.empty(everyone.numel(), dtype=b.dtype, device=b.device)
      G[everyone] = torch.cat(comm.all_gather(b.contiguous()))
      X = torch.fft.irfftn(torch.fft.rfftn(G.reshape(shape)) * inv, s=shape)
      return X.reshape(-1)[mine]
    if self.E0_fft is not None:
      perm, inv, shape = self.E0_fft
      v = (b if perm is None else b[perm]).reshape(shape)
      x = torch.fft.irfftn(torch.fft.rfftn(v) * inv, s=shape).reshape(-1)
      if perm is None:
        return x
      out = torch.empty_like(x)
      out[perm] = x
      return out
    if self.E0_pinv is not None and self.coarse_solver == 'chebyshev':
      if self.members > 1:        # (the pseudo-inverse is symmetric)
        return (b.view(self.members, -1) @ self.E0_pinv).reshape(-1)
      return torch.mv(self.E0_pinv, b)
    if self.coarse_solver == 'cg':
      if self.members > 1:
        raise NotImplementedError("coarse_solver='cg' for an ensemble")
      return self._coarse_solve_cg(b)
    from swirl_fem_amd import _ops
    centre = lambda v: (v.view(self.members, -1) - v.view(
        self.members, -1).mean(dim=1, keepdim=True)).reshape(-1)
    if self.coarse_singular:
      b = centre(b)
    lmin, lmax = self.coarse_bounds
    x = _ops.ell_chebyshev(self.E0_cols_t, self.E0_vals_t,
                           1.0 / self.coarse_diag, b.contiguous(),
                           self.coarse_iterations, lmin, lmax)
    return centre(x) if self.coarse_singular else x

  def _coarse_solve_cg(self, b):
    """The coarse solve as ONE graph launch: its few hundred tiny kernels are
    recorded once (fixed iteration count, fixed buffers) and replayed."""
    if self._coarse_graph is None:
      self._coarse_in = torch.zeros_like(b)
      for _ in range(2):                         # warm-up outside the capture
        self._coarse_iterate(self._coarse_in)
      torch.cuda.synchronize()
      graph = torch.cuda.CUDAGraph()
      import gc                  # (no collection inside a capture: cg.capture)
      gc.collect()
      was_enabled = gc.isenabled()
      gc.disable()
      try:
        with torch.cuda.graph(graph):
          self._coarse_out = self._coarse_iterate(self._coarse_in)
        self._coarse_graph = graph
      except Exception:                  # pylint: disable=broad-except
        torch.cuda.synchronize()
        self._coarse_graph = False
      finally:
        if was_enabled:
          gc.enable()
    if self._coarse_graph is False:
      return self._coarse_iterate(b)
    self._coarse_in.copy_(b)
    self._coarse_graph.replay()
    return self._coarse_out

  def _coarse_iterate(self, b):
    """`coarse_iterations` Jacobi-preconditioned CG iterations on E_0 y = b
    (all on the device, no convergence test: a fixed linear-in-practice map)."""
    if self.coarse_singular:
      b = b - b.mean()
    dinv = 1.0 / self.coarse_diag
    x = torch.zeros_like(b)
    r = b.clone()
    z = dinv * r
    p = z.clone()
    rz = torch.dot(r, z)
    tiny = torch.finfo(b.dtype).tiny
    for _ in range(self.coarse_iterations):
      Ap = self.coarse_matvec(p)
      alpha = rz / torch.clamp(torch.dot(p, Ap), min=tiny)
      x = x + alpha * p
      r = r - alpha * Ap
      z = dinv * r
      rz_new = torch.dot(r, z)
      p = z + (rz_new / torch.clamp(rz, min=tiny)) * p
      rz = rz_new
    if self.coarse_singular:
      x = x - x.mean()
    return x

  # -------------------------------------------------------------------- apply
  def _mode(self, t, a, transpose):
    """Contracts axis a of t (E, Pp, ..) with the element's S (or S^T)."""
    S = self.S[self.case[a]]                          # (E, Pp, Pp)
    if transpose:
      S = S.transpose(1, 2)
    t = t.movedim(1 + a, 1)
    shape = t.shape
    out = torch.bmm(S, t.reshape(shape[0], shape[1], -1)).reshape(shape)
    return out.movedim(1, 1 + a)

  def local_solve(self, r):
    """One kernel (`sfem_fdm_solve`); `local_solve_torch` is the same with
    batched matrix products (the test's cross-check)."""
    from swirl_fem_amd import _ops
    if self.Pp > 10:
      return self.local_solve_torch(r)
    return _ops.fdm_solve(r.contiguous(), self.pel_arg, self.S, self.case32,
                          self.inv_ev, self.d, self.Pp)

  def local_solve_torch(self, r):
    E, d, Pp = self.pel.shape[0], self.d, self.Pp
    t = r[self.pel].reshape((E,) + (Pp,) * d)
    for a in range(d):
      t = self._mode(t, a, True)
    t = t * self.inv_ev
    for a in range(d):
      t = self._mode(t, a, False)
    z = torch.empty_like(r)
    z[self.pel.reshape(-1)] = t.reshape(-1)
    return z

  def _fused_setup(self):
    """(weights per node, their element sums, total) for the one-pass closing
    of `__call__`, or None where the pieces run separately (partitions, an
    exchange on the pressure space, pressure nodes not numbered by element)."""
    if not hasattr(self, '_fused'):
      sem = self.sem
      pmesh = sem.pressure.pspace.mesh
      gi = pmesh.exchange_gather_indices
      self._fused = None
      from swirl_fem_amd import switches
      if (self.pel_arg is None and self.Pp <= 10 and not self.partitioned and
          (gi is None or gi.numel() == 0) and
          switches.get('SFEM_PC_FUSED') != '0'):
        b1, total = ns._pressure_mass_ones(sem, pmesh.dtype, pmesh.device)
        full = b1.repeat(self.members).contiguous()
        self._fused = (full, full.view(self.pel.shape).sum(dim=1),
                       float(total))
    return self._fused

  def __call__(self, r):
    fused = self._fused_setup()
    if fused is not None:
      # the element sums of r and the elements' shares of the mean come out
      # of the local solve; coarse correction and mean removal are one pass
      from swirl_fem_amd import _ops
      w, w_elem, total = fused
      E, n = self.pel.shape
      z, rc, share = _ops.fdm_solve_sums(r.contiguous(), None, self.S,
                                         self.case32, self.inv_ev, w, self.d,
                                         self.Pp)
      yc = self._coarse_solve(rc)
      B = self.members
      shift = (share.view(B, -1).sum(dim=1) +
               (yc * w_elem).view(B, -1).sum(dim=1)) / total
      return _ops.add_element_constants_(z, yc, shift, n, E // B)
    z = self.local_solve(r)
    E, n = self.pel.shape
    if self.pel_arg is None:           # element e owns [e n, (e + 1) n)
      yc = self._coarse_solve(r.view(E, n).sum(dim=1))
      z.view(E, n).add_(yc[:, None])
    else:
      yc = self._coarse_solve(r[self.pel].sum(dim=1))
      z = z.index_add(0, self.pel.reshape(-1),
                      yc[:, None].expand(-1, n).reshape(-1))
    return self.project(z)


def make_pressure_preconditioner(sem, name, dt, time_order):
  """The preconditioner a driver asks for by name ('schwarz'), kept per
  (dt, time_order) in the stepper object; None / 'projection' = the
  reference's default."""
  if name in (None, 'projection'):
    return None
  if name != 'schwarz':
    raise ValueError(f'unknown pressure preconditioner {name!r}')
  key = ('pressure_pc', name, float(dt), int(time_order))
  if key not in sem._cache:
    from swirl_fem_amd import switches
    its = switches.get('SFEM_PC_COARSE_ITERS')
    sem._cache[key] = SchwarzPressurePreconditioner(
        sem, dt, time_order, coarse_iterations=int(its) if its else None)
  return sem._cache[key]
