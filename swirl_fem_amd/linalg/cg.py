"""Preconditioned conjugate gradients on MI355X device tensors.

Signature and semantics of the reference `swirl_fem/linalg/cg.py:30-97`:
residual measured as r^T M r (:68-73), `maxiter = 10 * size` (:57-59),
tolerance `max(tol^2 b.b, atol^2)` (:65-66), pytrees of arrays, user `dot_fn`,
returns `(x, {'residual', 'num_iterations'})`.

The reference keeps the loop on the device with `lax.while_loop` (:94-95).
Here the five scalars of the recurrence live in a small device array and every
vector update is a fused HIP kernel that reads them there (`sfem_dot*`,
`sfem_cg_update_r`, `sfem_cg_update_xp`, `sfem_cg_scalars`), so an iteration
issues no host synchronisation.  The stopping test is evaluated on the device
each iteration; once it fires every later kernel is a no-op, and the host only
polls the flag every `check_every` iterations.  Iterates and iteration count
are therefore identical to a loop that tests every iteration.

A preconditioner that offers `mean_projection()` (M r = r - (w.r / total) 1,
the nullspace projection of the pressure solve) is folded into the two vector
updates: z = M r is never stored and r . z comes out of the sums update_r takes
anyway -- 9 vector passes per iteration instead of 12.

Per iteration (M = identity): A(p) with p.Ap fused into the operator's scatter
stage when it offers `apply_with_dot`; r -= a Ap fused with r.r; then
x += a p and p = r + b p in one kernel  ->  8 N-vector passes besides the
apply (SURVEY 8d's fused model: 11; the reference's un-fused loop: 13).
"""

from __future__ import annotations

import os

import torch

from swirl_fem_amd import switches
from swirl_fem_amd import _lib
from swirl_fem_amd import _ops
from swirl_fem_amd.core import layout


def _leaves(t):
  if isinstance(t, dict):
    return [l for k in sorted(t) for l in _leaves(t[k])]
  if isinstance(t, (list, tuple)):
    return [l for x in t for l in _leaves(x)]
  return [t]


def _map(f, *ts):
  t0 = ts[0]
  if isinstance(t0, dict):
    return {k: _map(f, *[t[k] for t in ts]) for k in t0}
  if isinstance(t0, (list, tuple)):
    return type(t0)(_map(f, *xs) for xs in zip(*ts))
  return f(*ts)


def _as_vec(t):
  if not isinstance(t, torch.Tensor):
    raise TypeError(f'cg operates on device tensors, got {type(t)}')
  return t


MAX_KEPT_RUNNERS = 6   # solver states a `workspace` of `cg` holds at most
# vectors below this size are served by the caches: the lazy x update would
# only cost memory there (same threshold as the non-temporal vector kernels;
# SFEM_LAZY_X_MIN_MB overrides it, the tests set 0)
LAZY_X_MIN_BYTES = 1 << 28
RR_PARTIALS = 1 << 16      # stored r.r sums: one per workgroup of the r update


def _lazy_min_bytes():
  mb = switches.get('SFEM_LAZY_X_MIN_MB')
  return LAZY_X_MIN_BYTES if mb is None else int(float(mb) * (1 << 20))


class _Scalars:
  """Device-resident CG scalars (see include/sfem.h, SFEM_CG_NSCALARS)."""
  GAMMA, PAP, GAMMA_NEW, ALPHA, BETA, BB, ATOL2, DONE, ITERS = range(9)
  STATUS = 10

  def __init__(self, device):
    self.t = torch.zeros(_lib.SFEM_CG_NSCALARS, dtype=torch.float64,
                         device=device)
    # strip of partial sums for operators that fuse p.Ap into their scatter
    self.partials = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                                device=device)

  def interface_correction(self, slot, a, b, interface):
    """scalars[slot] -= sum_interface w a b: turns the plain local dot of two
    *consistent* vectors into this rank's share of the global inner product
    (`NeighborPlan.interface_weights`); the interface is O(N^(2/3)) nodes."""
    idx, w = interface
    if idx.numel() == 0:
      return
    for x, y in zip(_leaves(a), _leaves(b)):
      _ops.dot_indexed(x, layout.like(y, x), idx, w, self.t, slot, -1.0)

  def dot_into(self, slot, a, b, dot_fn, reduce_fn, interface=None):
    """scalars[slot] = <a, b> summed over the leaves of the pytrees."""
    la, lb = _leaves(a), _leaves(b)
    if dot_fn is None:
      first = True
      for x, y in zip(la, lb):
        _ops.dot(layout.flat(x), layout.flat(layout.like(y, x)), self.t, slot,
                 accumulate=not first)
        first = False
    else:
      total = sum(dot_fn(x, y) for x, y in zip(la, lb))
      self.t[slot] = torch.as_tensor(total, dtype=torch.float64,
                                     device=self.t.device)
    if interface is not None:
      self.interface_correction(slot, a, b, interface)
    if reduce_fn is not None:
      reduce_fn(self.t[slot:slot + 1])


class CGRunner:
  """State of one CG solve; `step()` enqueues exactly one iteration.

  `cg()` drives it until the device flag reports convergence; `bench.py`
  steps it a fixed number of times.
  """

  def __init__(self, A, b, x0=None, *, tol=1e-5, atol=0.0, maxiter=None,
               M=None, dot_fn=None, reduce_fn=None, interface=None):
    if interface is not None and (M is not None or dot_fn is not None):
      raise ValueError('interface weights apply to the plain dot of consistent '
                       'vectors (M = None, dot_fn = None)')
    b_leaves = [_as_vec(l) for l in _leaves(b)]
    if not all(l.is_cuda for l in b_leaves):
      raise RuntimeError('swirl_fem_amd.linalg.cg runs on MI355X device '
                         'tensors (there is no CPU fallback)')
    self.A, self.M, self.dot_fn, self.reduce_fn = A, M, dot_fn, reduce_fn
    self.interface = interface
    self.tol, self.atol = tol, atol
    if maxiter is None:
      maxiter = 10 * sum(l.numel() for l in b_leaves)
    self.maxiter = maxiter
    device = b_leaves[0].device
    dense = lambda t: t if (t.is_contiguous() or
                            layout.is_component_major(t)) else t.contiguous()
    b = _map(dense, b)
    self._x = (_map(torch.zeros_like, b) if x0 is None
               else _map(lambda t, bb: layout.like(t, bb).clone(), x0, b))
    self.lazy = None
    self.s = s = _Scalars(device)
    S = _Scalars
    self.identity_m = M is None
    s.dot_into(S.BB, b, b, dot_fn, reduce_fn, interface)
    self.r = _map(lambda bb, ax: bb - layout.like(ax, bb), b, A(self._x))
    z = self.r if self.identity_m else M(self.r)
    self._p = _map(lambda t, rr: layout.like(t, rr).clone(), z, self.r)
    s.dot_into(S.GAMMA, self.r, z, dot_fn, reduce_fn, interface)
    # operators exposing `apply_with_dot` hand back p.Ap with the apply
    self.fused_dot = (dot_fn is None and hasattr(A, 'apply_with_dot') and
                      isinstance(self._p, torch.Tensor))
    self.parts = s.partials if self.fused_dot else None
    _ops.cg_scalars(s.t, 2, maxiter, tol, atol, self.parts)
    self.fuse_rr = self.identity_m and dot_fn is None
    # r.r spread over 64 slots: update_r then streams with 128 workgroups per
    # CU.  When something needs the complete sum in the named slot right after
    # the update (an all-reduce, the interface correction) one tiny scalar
    # launch folds the slots first.
    if self.fuse_rr:
      self.fuse_rr = 2
    self.fold_rr = self.fuse_rr == 2 and (reduce_fn is not None or
                                          interface is not None)
    # M r = r - (w . r / total) 1 folded into the two vector updates
    self.mean = None
    probe = getattr(M, 'mean_projection', None)
    if (probe is not None and dot_fn is None and reduce_fn is None and
        interface is None and isinstance(self.r, torch.Tensor) and
        switches.get('SFEM_FUSED_MEAN') != '0'):
      found = probe()
      if found is not None:
        w, total = found
        self.mean = (layout.flat(layout.like(w.to(self.r.dtype), self.r)),
                     float(total),
                     torch.zeros(_lib.SFEM_CG_MEAN_SUMS, dtype=torch.float64,
                                 device=device))
    # Layered assembly: the operator leaves the contributions of shared nodes
    # in layers of an extended Ap (plain stores: no atomics, no cleared range)
    # and `r -= alpha Ap` adds them up where it streams Ap anyway, in a fixed
    # order -- same sums, bitwise reproducible.  One partition, scalar field.
    # On partitions (`reduce_fn`, `interface`) the operator hands back the
    # layered result with its interface nodes already whole and exchanged.
    self.layered = None
    if (self.fused_dot and self.mean is None and self._p.dim() == 1 and
        hasattr(A, 'apply_layered_with_dot')):
      self.layered = A.layer_plan()
    # ... and with the two inner products of the iteration summed from STORED
    # partial sums in a fixed order (one double per wave of the apply, one per
    # workgroup of the r update) the whole iteration is bitwise reproducible
    # from run to run; costs one more scalar launch per iteration
    self.det = None
    if (self.layered is not None and reduce_fn is None and interface is None
        and switches.get('SFEM_DETERMINISTIC') != '0'):
      # (+ SFEM_FOLD_GROUPS: scratch of the two-stage sums behind the slots)
      pad = _lib.SFEM_FOLD_GROUPS
      self.det = (torch.zeros(A.layered_dot_slots() + pad, dtype=torch.float64,
                              device=device),
                  torch.zeros(RR_PARTIALS + pad, dtype=torch.float64,
                              device=device))
      self._reproducible_start(b, z)
    # Lazy solution update (`sfem_cg_update_xp_lazy`): x is touched every m-th
    # iteration only, the directions in between wait in a ring -- bitwise the
    # same x, 4.25 instead of 5 vector passes in the x / p update at m = 4.
    # Worth m - 1 more vectors only where the iteration streams from HBM.
    m = int(switches.get('SFEM_LAZY_X'))
    if (m >= 2 and self.mean is None and isinstance(self._p, torch.Tensor) and
        self._p.is_contiguous() and
        self._p.numel() * self._p.element_size() >= _lazy_min_bytes()):
      m = min(m, _lib.SFEM_CG_LAZY_MAX)
      n = self._p.numel()
      ring = torch.empty((m, (n + 3) // 4 * 4), dtype=self._p.dtype,
                         device=device)
      ring[0, :n] = self._p.reshape(-1)
      self._p = None
      self.lazy = (ring, n, torch.zeros(1 + _lib.SFEM_CG_LAZY_MAX,
                                        dtype=torch.float64, device=device))
      self._shape = tuple(self.r.shape)
    self.issued = 0
    self._graph = None
    self._capture_failed = False

  def _reproducible_start(self, b, z):
    """b.b and gamma_0 = r.z again, as tree reductions without atomics (the
    device dot accumulates its workgroups' sums in arrival order), and the
    stopping rule re-initialised from them: the start of a reproducible solve."""
    s, S = self.s, _Scalars
    tree = lambda u, v: (u.double() * v.double()).sum()
    s.t[S.BB] = tree(b, b)
    s.t[S.GAMMA] = tree(self.r, z)
    _ops.cg_scalars(s.t, 2, self.maxiter, self.tol, self.atol, self.parts)

  @property
  def vector_passes(self):
    """N-vector reads + writes of one iteration outside the operator (M = I):
    r -= alpha Ap reads r, Ap and writes r; the x / p update reads x, p, r and
    writes x, p -- or, lazily, (4 m + 1) / m of them on average."""
    if self.lazy is None:
      return 8
    m = self.lazy[0].shape[0]
    return 3 + (4 * m + 1) / m

  @property
  def p(self):
    """The current search direction p_k (k = iterations issued)."""
    if self.lazy is None:
      return self._p
    ring, n, _ = self.lazy
    return ring[self.issued % ring.shape[0], :n].view(self._shape)

  @property
  def x(self):
    """The iterate after the iterations issued so far (with the lazy update:
    after adding the terms that were still held back)."""
    if self.lazy is not None:
      self._flush()
      ring, n, state = self.lazy
      _ops.cg_flush_x(layout.flat(self._x), ring, self.s.t, state)
    return self._x

  def matches(self, b, tol, atol, maxiter) -> bool:
    """Whether `restart(b)` can take this right-hand side: same leaves
    (shape, dtype, device, memory layout) and the same stopping rule (a
    captured iteration carries tol / atol / maxiter as kernel arguments)."""
    mine, theirs = _leaves(self.r), _leaves(b)
    if len(mine) != len(theirs) or (tol, atol) != (self.tol, self.atol):
      return False
    if maxiter is None:
      maxiter = 10 * sum(l.numel() for l in theirs)
    if maxiter != self.maxiter:
      return False
    # (component-major and row-major (N, d) fields have the same shape: the
    # strides tell them apart; `restart` repacks a dense b of another layout)
    return all(isinstance(t, torch.Tensor) and t.shape == m.shape and
               t.dtype == m.dtype and t.device == m.device and
               (t.stride() == m.stride() or t.is_contiguous())
               for m, t in zip(mine, theirs))

  def restart(self, b, x0=None):
    """The same solve (A, M, stopping rule) for a new right-hand side: x, r,
    p and the scalars are rewritten IN PLACE, so an iteration captured into a
    HIP graph stays valid -- a time stepper that solves with the same
    operators every step then records each of them once (recording costs
    about 1 ms per solve, a third of a Kolmogorov-generator step)."""
    s, S = self.s, _Scalars
    if x0 is None:
      _map(lambda xx: xx.zero_(), self._x)
    else:
      _map(lambda xx, t: xx.copy_(layout.like(t, xx)), self._x, x0)
    self.issued = 0
    if self.lazy is not None:
      self.lazy[2].zero_()
    s.dot_into(S.BB, b, b, self.dot_fn, self.reduce_fn, self.interface)
    _map(lambda rr, bb, ax: rr.copy_(layout.like(bb, rr) - layout.like(ax, rr)),
         self.r, b, self.A(self._x))
    z = self.r if self.identity_m else self.M(self.r)
    if z is not self.r:
      _map(lambda pp, zz: pp.copy_(layout.like(zz, pp)), self.p, z)
    else:
      _map(lambda pp, rr: pp.copy_(rr), self.p, self.r)
    s.dot_into(S.GAMMA, self.r, z, self.dot_fn, self.reduce_fn, self.interface)
    _ops.cg_scalars(s.t, 2, self.maxiter, self.tol, self.atol, self.parts)
    if self.det is not None:
      self._reproducible_start(b, z)
    if self.mean is not None:
      self.mean[2].zero_()
    self.issued = 0

  def step(self):
    """One iteration of cg.py:75-86, all on the device."""
    if self._graph is not None:
      self._graph.replay()
      self.issued += 1
      return
    self._step_eager()

  def _step_eager(self):
    s, S = self.s, _Scalars
    A, M, dot_fn, reduce_fn = self.A, self.M, self.dot_fn, self.reduce_fn
    args = (self.maxiter, self.tol, self.atol, self.parts)
    merged = self.fused_dot and reduce_fn is None
    if self.fused_dot and self.det is not None:
      Ap = A.apply_layered_with_dot(self.p, self.det[0], per_wave=True)
      _ops.cg_scalars_n(s.t, 5, self.maxiter, self.tol, self.atol,
                        self.det[0],
                        self.det[0].numel() - _lib.SFEM_FOLD_GROUPS)
    elif self.fused_dot:
      Ap = (A.apply_layered_with_dot(self.p, s.partials)
            if self.layered is not None
            else A.apply_with_dot(self.p, s.partials))
      # Without an all-reduce between the sum of the partials and alpha the
      # iteration needs ONE scalar launch: phase 5 also closes the previous
      # iteration (beta, gamma, counter, convergence flag) -- see `_flush`.
      _ops.cg_scalars(s.t, 5 if merged else 3, *args)
      if reduce_fn is not None:
        reduce_fn(s.t[S.PAP:S.PAP + 1])
    else:
      Ap = A(self.p)
    if self.fused_dot:
      pass
    elif dot_fn is None:
      # slot PAP is zero here: cleared by phase 2 / phase 1
      for xx, yy in zip(_leaves(self.p), _leaves(Ap)):
        _ops.dot(layout.flat(xx), layout.flat(layout.like(yy, xx)), s.t,
                 S.PAP, accumulate=True)
      if self.interface is not None:
        s.interface_correction(S.PAP, self.p, Ap, self.interface)
      if reduce_fn is not None:
        reduce_fn(s.t[S.PAP:S.PAP + 1])
    else:
      s.dot_into(S.PAP, self.p, Ap, dot_fn, reduce_fn)
    if not merged:
      _ops.cg_scalars(s.t, 0, *args)
    if self.mean is not None:
      w, total, sums = self.mean
      _ops.cg_update_r_mean(layout.flat(self.r),
                            layout.flat(layout.like(Ap, self.r)), w, s.t, sums)
      _ops.cg_update_xp_mean(layout.flat(self._x), layout.flat(self.p),
                             layout.flat(self.r), s.t, sums, total)
      if not merged:
        _ops.cg_scalars(s.t, 1, *args)
      self.issued += 1
      return
    if self.layered is not None and self.det is not None and self.fuse_rr:
      n = _ops.cg_update_r_layered_det(
          self.r, Ap, self.layered.layers, s.t,
          self.det[1][:RR_PARTIALS], masks=self.layered.masks)
      _ops.cg_scalars_n(s.t, 8, self.maxiter, self.tol, self.atol,
                        self.det[1], n)
    elif self.layered is not None:
      _ops.cg_update_r_layered(self.r, Ap, self.layered.layers, s.t,
                               self.fuse_rr, masks=self.layered.masks)
    else:
      for rr, aa in zip(_leaves(self.r), _leaves(Ap)):
        _ops.cg_update_r(layout.flat(rr), layout.flat(layout.like(aa, rr)),
                         s.t, self.fuse_rr)
    if self.fuse_rr:
      z = self.r
      if self.fold_rr:
        _ops.cg_scalars(s.t, 7, *args)
      if self.interface is not None:
        s.interface_correction(S.GAMMA_NEW, self.r, self.r, self.interface)
      if reduce_fn is not None:
        reduce_fn(s.t[S.GAMMA_NEW:S.GAMMA_NEW + 1])
    else:
      if (dot_fn is None and hasattr(M, 'apply_with_dot') and
          isinstance(self.r, torch.Tensor)):
        # the preconditioner accumulates r . M r itself (slot is zero here)
        z = M.apply_with_dot(self.r, s.t, S.GAMMA_NEW)
        if reduce_fn is not None:
          reduce_fn(s.t[S.GAMMA_NEW:S.GAMMA_NEW + 1])
      elif dot_fn is None:
        z = self.r if self.identity_m else M(self.r)
        for xx, yy in zip(_leaves(self.r), _leaves(z)):
          _ops.dot(layout.flat(xx), layout.flat(layout.like(yy, xx)), s.t,
                   S.GAMMA_NEW, accumulate=True)
        if reduce_fn is not None:
          reduce_fn(s.t[S.GAMMA_NEW:S.GAMMA_NEW + 1])
      else:
        # after convergence the done flag guards every consumer of this slot
        z = self.r if self.identity_m else M(self.r)
        s.dot_into(S.GAMMA_NEW, self.r, z, dot_fn, reduce_fn)
    # x += alpha p rides with the p update (p is in registers there): 8 vector
    # passes per iteration instead of 9, same arithmetic
    if self.lazy is not None:
      ring, n, state = self.lazy
      _ops.cg_update_xp_lazy(layout.flat(self._x), ring,
                             layout.flat(layout.like(z, self.r)), s.t, state)
    else:
      for xx, pp, zz in zip(_leaves(self._x), _leaves(self.p), _leaves(z)):
        _ops.cg_update_xp(layout.flat(xx), layout.flat(pp),
                          layout.flat(layout.like(zz, pp)), s.t)
    if not merged:
      _ops.cg_scalars(s.t, 1, *args)
    self.issued += 1

  def capture(self) -> bool:
    """Records one iteration into a HIP graph; later `step()` calls replay it.

    A CG iteration on a small mesh is a few dozen short launches (kernels,
    memsets, scalar updates) and is bound by launch overhead, not by the GPU;
    one graph launch per iteration removes that.  Everything an iteration
    touches lives at fixed addresses (x, r, p, the scalars; temporaries come
    from the graph's private pool) and no kernel needs the host, so the replay
    is exact.  Returns False (and stays eager) if the operator or the
    preconditioner does something that cannot be captured.
    """
    if self._graph is not None:
      return True
    if self.reduce_fn is not None:
      return False                 # collectives stay on the eager path
    if self.lazy is not None:
      # a recorded iteration has fixed operands; the ring rotates them
      if self.issued:
        return False
      ring, n, _ = self.lazy
      self._p = ring[0, :n].view(self._shape).clone()
      self.lazy = None
    self.step()                    # warm caches / lazy setup eagerly
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    issued = self.issued
    # The cyclic collector must not run while the stream is capturing: an
    # older solver state that sits in a reference cycle (stepper -> cache ->
    # runner -> operator -> stepper) owns a graph and its memory pool, and
    # freeing those in the middle of a capture aborts the process.  Collect
    # first, then keep the collector off until the capture has ended.
    import gc
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
      with torch.cuda.graph(graph):
        self._step_eager()
    except Exception:              # pylint: disable=broad-except
      torch.cuda.synchronize()
      self.issued = issued
      self._capture_failed = True  # (a kept runner does not try again)
      return False
    finally:
      if was_enabled:
        gc.enable()
    self.issued = issued           # capture records, it does not execute
    self._graph = graph
    return True

  def _flush(self):
    """Closes the iteration that the single-scalar-launch scheme leaves open
    (no-op otherwise), so that the scalars describe the last `step()`."""
    if self.fused_dot and self.reduce_fn is None:
      _ops.cg_scalars(self.s.t, 6, self.maxiter, self.tol, self.atol, None)

  def done(self) -> bool:
    """Synchronising poll of the device convergence flag."""
    self._flush()
    return bool(self.s.t[_Scalars.DONE].item() != 0.0)

  def info(self):
    self._flush()
    scal = self.s.t.cpu()
    status = _lib.CG_STATUS.get(int(scal[_Scalars.STATUS].item()), 'unknown')
    if status == 'running' and self.issued >= self.maxiter:
      status = 'maxiter'
    # 'residual' and 'num_iterations' as the reference returns them
    # (cg.py:96-97); 'status' says why the loop ended: 'converged', 'maxiter',
    # or a breakdown the reference would have reported as convergence
    # ('breakdown_gamma': r.Mr negative / not finite, 'breakdown_pAp': p.Ap
    # zero / not finite) -- x is then the last iterate, NOT a solution to `tol`.
    return {'residual': self.s.t[_Scalars.GAMMA].clone(),
            'num_iterations': int(scal[_Scalars.ITERS].item()),
            'status': status}


def cg(A, b, x0=None, *, tol=1e-5, atol=0.0, maxiter=None, M=None,
       dot_fn=None, reduce_fn=None, interface=None, check_every=16,
       graph=False, workspace=None, key=None):
  """Solves A x = b with (preconditioned) conjugate gradients.

  Args:
    A: linear operator on pytrees of device tensors.
    b: right-hand side pytree.
    x0: initial guess (default zeros).
    tol, atol: stop when r^T M r <= max(tol^2 b.b, atol^2).
    maxiter: iteration cap (default 10 * size).
    M: preconditioner (default identity).
    dot_fn: optional custom inner product `(a, b) -> scalar tensor`; the default
      is the fused device dot.
    reduce_fn: optional in-place reduction applied to every inner product
      (partitioned meshes pass an RCCL all-reduce; reference callers pass a
      psum-ing `dot_fn`).
    interface: optional `(idx, w)` from `NeighborPlan.interface_weights`: all
      vectors are consistent across partitions (A returns the *assembled*
      result) and every plain dot is corrected on the interface nodes to count
      each global node once.  Mathematically the same iterates as the
      reference's partitioned convention (unassembled A, M = exchange,
      navier_stokes.py:436-438) without the extra vector z = M r.
    graph: replay the iteration as one HIP graph launch (`CGRunner.capture`);
      worth it when an iteration is launch-bound (small meshes, long solves).
      `A` and `M` must then be pure device work on fixed operands.
    check_every: the host polls the device convergence flag this often.
    workspace, key: with `graph=True`, a dict owned by the caller and a
      hashable key that stands for (A, M): the solver state and the recorded
      iteration are kept under `workspace[key]` and reused by later calls with
      the same key, stopping rule and vector shapes (`CGRunner.restart`).  The
      caller promises that A and M of those calls are the same operators.
  Returns:
    (x, info) with info = {'residual': gamma, 'num_iterations': k,
    'status': 'converged' | 'maxiter' | 'breakdown_gamma' | 'breakdown_pAp'}.
  """
  if not _leaves(b):
    return b, {'residual': 0.0, 'num_iterations': 0, 'status': 'converged'}
  reuse = (graph and workspace is not None and key is not None and
           dot_fn is None and reduce_fn is None and interface is None)
  run = workspace.get(key) if reuse else None
  if run is not None and run.matches(b, tol, atol, maxiter):
    workspace[key] = workspace.pop(key)      # most recently used goes last
    run.restart(b, x0)
  else:
    run = CGRunner(A, b, x0, tol=tol, atol=atol, maxiter=maxiter, M=M,
                   dot_fn=dot_fn, reduce_fn=reduce_fn, interface=interface)
    if reuse:
      workspace.pop(key, None)
      workspace[key] = run
      # (a stepper whose coefficients change every step must not collect
      # solver states without bound: the oldest ones go)
      while len(workspace) > MAX_KEPT_RUNNERS:
        workspace.pop(next(iter(workspace)))
  if (graph and dot_fn is None and run.maxiter > 2 and run._graph is None and
      not run._capture_failed and not run.done()):
    run.capture()
  while run.issued < run.maxiter:
    for _ in range(min(check_every, run.maxiter - run.issued)):
      run.step()
    if run.done():
      break
  info = run.info()
  if info['status'].startswith('breakdown'):
    # the reference would have returned this as a converged solve; callers
    # that ignore `info` (the Stokes solves) at least get told
    import warnings
    warnings.warn(f"cg stopped with status '{info['status']}' after "
                  f"{info['num_iterations']} iterations: x is the last "
                  'iterate, not a solution to the requested tolerance',
                  RuntimeWarning, stacklevel=2)
  # (a kept runner overwrites its x in the next solve)
  return (_map(lambda t: t.clone(), run.x) if reuse else run.x), info


class _SymmetricSolve(torch.autograd.Function):
  """x = A^-1 b with the adjoint solved by the same routine (A symmetric)."""

  @staticmethod
  def forward(ctx, b, A, kwargs, info_out):
    x, info = cg(A, b.detach(), **kwargs)
    if info_out is not None:
      info_out.update(info)
    ctx.A, ctx.kwargs = A, kwargs
    return x

  @staticmethod
  def backward(ctx, grad_x):
    # d<x, g>/db = A^-T g = A^-1 g
    grad_b, _ = cg(ctx.A, grad_x.detach().contiguous(), **ctx.kwargs)
    return grad_b, None, None, None


def symmetric_solve(A, b, info_out=None, **kwargs):
  """`cg(A, b, **kwargs)[0]` that autograd can differentiate with respect to
  `b`: the cotangent is obtained by a second solve with the same symmetric
  operator, which is what `lax.custom_linear_solve(symmetric=True)` does for
  the reference's solves (navier_stokes/navier_stokes.py:436-452).  Gradients
  with respect to tensors hidden inside `A` are not propagated.  `info_out`: a
  dict that receives the forward solve's `info`."""
  return _SymmetricSolve.apply(b, A, kwargs, info_out)
