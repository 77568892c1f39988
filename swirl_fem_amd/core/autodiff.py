"""`torch.autograd` rules for the HIP primitives of the generic operator path.

The reference is differentiable end to end because JAX transposes its linear
pieces (`jax.linear_transpose`, core/fespace.py:458-471) and differentiates
the rest; the training loop of `niles/train.py:227-293` relies on that to
push gradients through `stokes_one_step`.  The HIP kernels are opaque to
autograd, so each linear primitive gets its transpose here, expressed through
the primitive it is adjoint to:

    gather          <->  scatter_add          (core/gather_scatter.py:121-133)
    exchange (QQ^T)  =   its own transpose    (:189-261)
    basis_eval      <->  basis_eval_t         (core/fespace.py:178-225, :458-471)

Callers (`Mesh`, `FiniteElementSpace`, `StokesVelocity`, `basis.interp`)
switch to these when an operand requires grad; nothing changes for plain
tensors.  Solves are differentiated by `linalg.cg.symmetric_solve`.

The fused operator kernels are linear maps too and carry their transposes the
same way, so a differentiated Navier-Stokes step keeps the fused path for its
linear operators (only the quadratic convection term drops to the generic
q-function path):

    helmholtz apply  y = M S L G u   ->  u_bar = S L G (M y_bar)   (L symmetric)
    helmholtz local  y = L u         ->  u_bar = L y_bar
    stokes div       y = D u         ->  u_bar = D^T y_bar  (unmasked grad_t)
    stokes grad_t    y = M D^T p     ->  p_bar = D (M y_bar)
"""

from __future__ import annotations

import torch

from swirl_fem_amd import _ops


def needs_grad(*tensors) -> bool:
  """True when autograd is recording and one of `tensors` takes part in it."""
  return torch.is_grad_enabled() and any(
      isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _valid(indices, g, trailing=0):
  """Zeroes the cotangent at SENTINEL (-1) slots: they read a constant."""
  ok = (indices >= 0).to(g.dtype)
  return g * ok.reshape(ok.shape + (1,) * trailing)


class _Gather(torch.autograd.Function):
  """(N,) -> indices.shape, SENTINEL slots read `fill`."""

  @staticmethod
  def forward(ctx, u, indices, fill):
    ctx.indices, ctx.num_nodes = indices, u.shape[0]
    return _ops.gather(u, indices, fill)

  @staticmethod
  def backward(ctx, g):
    g = _valid(ctx.indices, g).contiguous()
    return _ops.scatter_add(g, ctx.indices, ctx.num_nodes), None, None


class _GatherRows(torch.autograd.Function):
  """(N, nc) -> indices.shape + (nc,), SENTINEL rows are zero."""

  @staticmethod
  def forward(ctx, x, indices):
    ctx.indices, ctx.num_nodes = indices, x.shape[0]
    return _ops.gather_rows(x, indices)

  @staticmethod
  def backward(ctx, g):
    g = _valid(ctx.indices, g, 1).contiguous()
    return _ops.scatter_add(g, ctx.indices, ctx.num_nodes,
                            ncomp=g.shape[-1]), None


class _ScatterAdd(torch.autograd.Function):
  """indices.shape [+ (nc,)] -> (N,) / (N, nc)."""

  @staticmethod
  def forward(ctx, u_local, indices, num_nodes, ncomp):
    ctx.indices, ctx.vector = indices, u_local.dim() > indices.dim()
    return _ops.scatter_add(u_local, indices, num_nodes, ncomp=ncomp)

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous()
    if ctx.vector:
      return _ops.gather_rows(g, ctx.indices), None, None, None
    return _ops.gather(g, ctx.indices, 0.0), None, None, None


class _ExchangeLocal(torch.autograd.Function):
  """Unpartitioned QQ^T: symmetric, so the cotangent takes the same route."""

  @staticmethod
  def forward(ctx, u, gather_indices, unique_indices):
    ctx.gi, ctx.ui = gather_indices, unique_indices
    return _ops.exchange_local(u, gather_indices, unique_indices)

  @staticmethod
  def backward(ctx, g):
    return _ops.exchange_local(g.contiguous(), ctx.gi, ctx.ui), None, None


class _BasisEval(torch.autograd.Function):
  """u (E, n, nc) -> the requested ones of val (E, Q, nc), grad (E, Q, d, nc)."""

  @staticmethod
  def forward(ctx, u_local, interp1, grad1, invjac, ndim, P, q, collocated,
              want_val, want_grad):
    if want_grad and invjac is None:
      raise NotImplementedError(
          'autograd through a reference-space gradient (no inverse Jacobian)')
    val, grad = _ops.basis_eval(u_local, interp1, grad1, invjac, ndim, P, q,
                                collocated, want_val, want_grad)
    ctx.args = (interp1, grad1, invjac, ndim, P, q, collocated, want_val,
                want_grad, u_local.shape[-1])
    return tuple(t for t in (val, grad) if t is not None)

  @staticmethod
  def backward(ctx, *gs):
    (interp1, grad1, invjac, ndim, P, q, collocated, want_val, want_grad,
     nc) = ctx.args
    gs = list(gs)
    gval = gs.pop(0) if want_val else None
    ggrad = gs.pop(0) if want_grad else None
    ref = gval if gval is not None else ggrad
    if ref is None:
      return (None,) * 10
    E = ref.shape[0]
    # basis_eval_t weights its input by wdet: ones give the bare transpose
    ones = torch.ones((E, q ** ndim), dtype=ref.dtype, device=ref.device)
    out = _ops.basis_eval_t(gval, ggrad, interp1, grad1, invjac, ones, ndim, P,
                            q, nc, collocated)
    return (out,) + (None,) * 9


class _BasisEvalT(torch.autograd.Function):
  """(c0 (E, Q, nc), c1 (E, Q, d, nc)) -> (E, n, nc), weighted by wdet."""

  @staticmethod
  def forward(ctx, c0, c1, interp1, grad1, invjac, wdet, ndim, P, q, nc,
              collocated):
    ctx.args = (interp1, grad1, invjac, wdet, ndim, P, q, collocated,
                c0 is not None, c1 is not None)
    return _ops.basis_eval_t(c0, c1, interp1, grad1, invjac, wdet, ndim, P, q,
                             nc, collocated)

  @staticmethod
  def backward(ctx, g):
    (interp1, grad1, invjac, wdet, ndim, P, q, collocated, has0,
     has1) = ctx.args
    val, grad = _ops.basis_eval(g.contiguous(), interp1, grad1, invjac, ndim,
                                P, q, collocated, has0, has1)
    g0 = val * wdet[:, :, None] if has0 else None
    g1 = grad * wdet[:, :, None, None] if has1 else None
    return (g0, g1) + (None,) * 9


class _HelmholtzApply(torch.autograd.Function):
  """`op.apply(u, l0, l1)` of a fused `HelmholtzOperator` with Dirichlet mask;
  `op_free` is the same operator without mask, `keep` (N,) is 1 off the mask."""

  @staticmethod
  def forward(ctx, u, op, op_free, keep, l0, l1):
    ctx.op_free, ctx.keep, ctx.l0, ctx.l1 = op_free, keep, l0, l1
    return op.apply(u, l0, l1)

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous()
    if ctx.keep is not None:
      g = g * (ctx.keep if g.dim() == 1 else ctx.keep[:, None])
    return ctx.op_free.apply(g, ctx.l0, ctx.l1), None, None, None, None, None


class _HelmholtzLocal(torch.autograd.Function):
  """`op.apply_local(u_local, l0, l1)`: symmetric per element."""

  @staticmethod
  def forward(ctx, u_local, op, l0, l1):
    ctx.op, ctx.l0, ctx.l1 = op, l0, l1
    return op.apply_local(u_local, l0, l1)

  @staticmethod
  def backward(ctx, g):
    return ctx.op.apply_local(g.contiguous(), ctx.l0, ctx.l1), None, None, None


class _StokesDiv(torch.autograd.Function):
  """`op.div(u)`; `op_free` = the same `StokesDivGrad` without Dirichlet mask
  (its `grad_t` is the exact transpose of `div`)."""

  @staticmethod
  def forward(ctx, u, op, op_free):
    ctx.op_free = op_free
    return op.div(u)

  @staticmethod
  def backward(ctx, g):
    return ctx.op_free.grad_t(g.contiguous()), None, None


class _StokesGradT(torch.autograd.Function):
  """`op.grad_t(p)` = M D^T p; the transpose is D (M .)."""

  @staticmethod
  def forward(ctx, p, op, keep):
    ctx.op, ctx.keep = op, keep
    return op.grad_t(p)

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous()
    if ctx.keep is not None:
      g = g * ctx.keep[:, None]
    return ctx.op.div(g), None, None


def helmholtz_apply(op, op_free, keep, u, l0, l1):
  return _HelmholtzApply.apply(u, op, op_free, keep, float(l0), float(l1))


def helmholtz_local(op, u_local, l0, l1):
  return _HelmholtzLocal.apply(u_local, op, float(l0), float(l1))


def stokes_div(op, op_free, u):
  return _StokesDiv.apply(u, op, op_free)


def stokes_grad_t(op, keep, p):
  return _StokesGradT.apply(p, op, keep)


# ------------------------------------------------------------- entry points
def gather(u, indices, fill):
  return _Gather.apply(u, indices, fill)


def gather_rows(x, indices):
  return _GatherRows.apply(x.contiguous(), indices)


def scatter_add(u_local, indices, num_nodes, ncomp=1):
  return _ScatterAdd.apply(u_local.contiguous(), indices, num_nodes, ncomp)


def exchange_local(u, gather_indices, unique_indices):
  return _ExchangeLocal.apply(u.contiguous(), gather_indices, unique_indices)


def basis_eval(u_local, interp1, grad1, invjac, ndim, P, q, collocated,
               want_val, want_grad):
  """Same contract as `_ops.basis_eval`: returns `(val, grad)`."""
  outs = list(_BasisEval.apply(u_local, interp1, grad1, invjac, ndim, P, q,
                               collocated, want_val, want_grad))
  val = outs.pop(0) if want_val else None
  grad = outs.pop(0) if want_grad else None
  return val, grad


def basis_eval_t(c0, c1, interp1, grad1, invjac, wdet, ndim, P, q, nc,
                 collocated):
  return _BasisEvalT.apply(c0, c1, interp1, grad1, invjac, wdet, ndim, P, q,
                           nc, collocated)
