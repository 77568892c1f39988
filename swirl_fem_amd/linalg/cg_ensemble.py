"""Conjugate gradients for an ensemble of right-hand sides on one operator.

The reference trains on ensembles by `jax.vmap`-ing its solver step
(niles/train.py:232, :262-264): B solves of linalg/cg.py:30-97 that share the
operator and nothing else -- every member has its own step lengths, its own
stopping test and its own iteration count.  On the small meshes that path
uses (64 x 64 quads, order 8) one member leaves the GPU waiting on launches,
so walking the ensemble on the host costs B times the single solve.

Here the B members are B disjoint copies of the mesh seen as one mesh
(`Mesh.replicate`, `StokesSEM.ensemble`): a vector is `(B N, ...)`, member m
owns the contiguous rows [m N, (m + 1) N), `A` and `M` are the ordinary
operators of that mesh (linear, member by member) and serve all members in
one launch each.  The recurrence runs per member inside the `sfem_ens_*`
kernels (csrc/sfem_cg_ensemble.hip): an iteration is the same seven launches
(four with the mean projection of the pressure solve folded in) whatever B is, members that have stopped become no-ops, and each member's
iterates are those of its own single solve (inner products: stored partial
sums added in a fixed order -- no atomics, nothing to clear).
"""

from __future__ import annotations

import gc

import torch

from swirl_fem_amd import _lib, _ops, switches

MAX_KEPT_RUNNERS = 8


class EnsembleCGRunner:
  """State of one ensemble solve; `step()` enqueues one iteration of every
  member that is still running."""

  def __init__(self, A, b, members, x0=None, *, tol=1e-5, atol=0.0,
               maxiter=None, M=None):
    if not isinstance(b, torch.Tensor) or not b.is_cuda:
      raise RuntimeError('cg_ensemble runs on MI355X device tensors (one '
                         'tensor per field; there is no CPU fallback)')
    members = int(members)
    if members < 1 or b.shape[0] % members:
      raise ValueError(f'{b.shape[0]} rows do not split into {members} '
                       'members')
    self.A, self.M, self.members = A, M, members
    self.tol, self.atol = tol, atol
    self.shape = tuple(b.shape)
    self.maxiter = (10 * (b.numel() // members) if maxiter is None
                    else maxiter)
    dev = b.device
    self.scalars = torch.zeros((members, _lib.SFEM_ENS_NSCALARS),
                               dtype=torch.float64, device=dev)
    self.partials = torch.zeros((members, 2, _lib.SFEM_ENS_GROUPS),
                                dtype=torch.float64, device=dev)
    self.x = torch.zeros(self.shape, dtype=b.dtype, device=dev)
    self.r = torch.empty_like(self.x)
    self.p = torch.empty_like(self.x)
    # M r = r - (w . r / total) 1 per member folded into the vector updates
    # (`sfem_ens_update_r_mean`): z = M r is never stored
    self.mean = None
    probe = getattr(M, 'ensemble_mean_projection', None)
    found = probe() if (probe is not None and b.dim() == 1 and
                        switches.get('SFEM_FUSED_MEAN') != '0') else None
    if found is not None:
      w, total = found
      self.mean = (w.to(b.dtype).contiguous(), float(total),
                   torch.zeros((members, 3, _lib.SFEM_ENS_GROUPS),
                               dtype=torch.float64, device=dev))
    self.issued = 0
    self._graph = None
    self._capture_failed = False
    self.restart(b, x0)

  def _dense(self, t):
    if tuple(t.shape) != self.shape:
      raise ValueError(f'operator returned shape {tuple(t.shape)} for a '
                       f'field of shape {self.shape}')
    return t.contiguous()

  def matches(self, b, members, tol, atol, maxiter) -> bool:
    return (tuple(b.shape) == self.shape and b.dtype == self.x.dtype and
            b.device == self.x.device and int(members) == self.members and
            tol == self.tol and atol == self.atol and
            (maxiter is None or maxiter == self.maxiter))

  def restart(self, b, x0=None):
    """A new right-hand side for the same operators: everything is rewritten
    in place, a recorded iteration stays valid."""
    B = self.members
    b = b.contiguous()
    if x0 is None:
      self.x.zero_()
      self.r.copy_(b)
    else:
      self.x.copy_(x0)
      self.r.copy_(b - self._dense(self.A(self.x)))
    z = self.r if self.M is None else self._dense(self.M(self.r))
    self.p.copy_(z)
    _ops.ens_dot(b, b, B, self.partials, 0)
    _ops.ens_dot(self.r, z, B, self.partials, 1)
    _ops.ens_init(self.scalars, self.partials, B, self.maxiter, self.tol,
                  self.atol)
    self.issued = 0

  def _iterate(self):
    B = self.members
    Ap = self._dense(self.A(self.p))
    _ops.ens_dot(self.p, Ap, B, self.partials, 0)
    if self.mean is not None:
      w, total, sums = self.mean
      _ops.ens_update_r_mean(self.r, Ap, w, B, self.scalars, self.partials,
                             sums)
      _ops.ens_close_mean(self.scalars, self.partials, sums, total, B,
                          self.maxiter)
      _ops.ens_update_xp_mean(self.x, self.p, self.r, B, self.scalars)
      self.issued += 1
      return
    _ops.ens_update_r(self.r, Ap, B, self.scalars, self.partials)
    z = self.r if self.M is None else self._dense(self.M(self.r))
    _ops.ens_dot(self.r, z, B, self.partials, 1)
    _ops.ens_close(self.scalars, self.partials, B, self.maxiter)
    _ops.ens_update_xp(self.x, self.p, z, B, self.scalars)
    self.issued += 1

  def step(self):
    if self._graph is not None:
      self._graph.replay()
      self.issued += 1
    else:
      self._iterate()

  def capture(self) -> bool:
    """Records one iteration into a HIP graph (see `CGRunner.capture`)."""
    if self._graph is not None:
      return True
    self._iterate()                # warm caches eagerly
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    issued = self.issued
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
      with torch.cuda.graph(graph):
        self._iterate()
    except Exception:              # pylint: disable=broad-except
      torch.cuda.synchronize()
      self.issued = issued
      self._capture_failed = True
      return False
    finally:
      if was_enabled:
        gc.enable()
    self.issued = issued
    self._graph = graph
    return True

  def done(self) -> bool:
    """Synchronising poll: have all members stopped?"""
    return bool((self.scalars[:, 7] != 0).all().item())

  def info(self):
    scal = self.scalars.cpu()
    status = [_lib.CG_STATUS.get(int(v), 'unknown') for v in scal[:, 10]]
    iters = [int(v) for v in scal[:, 8]]
    status = ['maxiter' if s == 'running' and self.issued >= self.maxiter
              else s for s in status]
    worst = next((s for s in status if s != 'converged'), 'converged')
    return {'residual': self.scalars[:, 0].clone(),
            'num_iterations': max(iters), 'member_iterations': iters,
            'status': worst, 'member_status': status}


def cg_ensemble(A, b, members, x0=None, *, tol=1e-5, atol=0.0, maxiter=None,
                M=None, check_every=16, graph=False, workspace=None, key=None):
  """Solves A x_m = b_m for the `members` members stacked in `b`.

  Args:
    A, M: operator and preconditioner of the replicated mesh (they act on
      `(B N, ...)` fields member by member; default M = identity).
    b: `(B N, ...)` device tensor, member m in rows [m N, (m + 1) N).
    tol, atol: every member stops on ITS r^T M r <= max(tol^2 b_m.b_m, atol^2).
    graph, workspace, key: as for `linalg.cg.cg`.
  Returns:
    (x, info): info['num_iterations'] is the largest count,
    info['member_iterations'] / ['member_status'] list the members,
    info['residual'] is the (B,) tensor of final r.M r.
  """
  reuse = graph and workspace is not None and key is not None
  run = workspace.get(key) if reuse else None
  if run is not None and run.matches(b, members, tol, atol, maxiter):
    workspace[key] = workspace.pop(key)
    run.restart(b, x0)
  else:
    run = EnsembleCGRunner(A, b, members, x0, tol=tol, atol=atol,
                           maxiter=maxiter, M=M)
    if reuse:
      workspace.pop(key, None)
      workspace[key] = run
      while len(workspace) > MAX_KEPT_RUNNERS:
        workspace.pop(next(iter(workspace)))
  if (graph and run.maxiter > 2 and run._graph is None and
      not run._capture_failed and not run.done()):
    run.capture()
  while run.issued < run.maxiter:
    for _ in range(min(check_every, run.maxiter - run.issued)):
      run.step()
    if run.done():
      break
  info = run.info()
  if info['status'].startswith('breakdown'):
    import warnings
    warnings.warn(f"cg_ensemble: a member stopped with status "
                  f"'{info['status']}': its x is the last iterate, not a "
                  'solution to the requested tolerance', RuntimeWarning,
                  stacklevel=2)
  return (run.x.clone() if reuse else run.x), info


class _EnsembleSymmetricSolve(torch.autograd.Function):
  """x_m = A^-1 b_m for every member, with the adjoint solved by the same
  routine (A symmetric): see `linalg.cg.symmetric_solve`."""

  @staticmethod
  def forward(ctx, b, A, members, kwargs, info_out):
    x, info = cg_ensemble(A, b.detach(), members, **kwargs)
    if info_out is not None:
      info_out.update(info)
    ctx.A, ctx.members, ctx.kwargs = A, members, kwargs
    return x

  @staticmethod
  def backward(ctx, grad_x):
    kwargs = dict(ctx.kwargs)
    kwargs.pop('x0', None)         # (a start for b says nothing about g)
    grad_b, _ = cg_ensemble(ctx.A, grad_x.detach().contiguous(), ctx.members,
                            **kwargs)
    return grad_b, None, None, None, None


def symmetric_solve_ensemble(A, b, members, info_out=None, **kwargs):
  """`cg_ensemble(A, b, members, **kwargs)[0]` that autograd differentiates
  with respect to `b` (one more ensemble solve for the cotangent: what
  `jax.vmap` of `lax.custom_linear_solve(symmetric=True)` does for the
  reference's solves, navier_stokes/navier_stokes.py:436-452)."""
  return _EnsembleSymmetricSolve.apply(b, A, members, kwargs, info_out)
