"""Partitioned Krylov solves on consistent vectors.

The reference runs CG on a partitioned mesh with the *unassembled* local
operator and puts QQ^T into the preconditioner slot (`M = exchange`,
navier_stokes/navier_stokes.py:436-438), which costs an extra N-vector `z`
and four more vector passes per iteration than the single-partition loop.

Here every CG vector is kept *consistent* (equal on all partitions holding a
node).  The operator adds the neighbours' contributions in place on the
interface nodes only (`comm.neighbor_exchange_`), `p . A p` falls out of the
fused apply kernel from the unassembled local result, and `r . r` is the plain
local sum minus a correction over the O(N^(2/3)) interface nodes
(`NeighborPlan.interface_weights`).  One iteration therefore costs what the
single-GPU iteration costs plus a packed neighbour exchange and two scalar
all-reduces.  In exact arithmetic the iterates are those of the reference
convention (x identical, r_here = QQ^T r_reference).
"""

from __future__ import annotations

import torch

from swirl_fem_amd.distributed import comm
from swirl_fem_amd.linalg import cg as cg_lib


def _exchange_layered_(ext, num_nodes, layer_plan, plan, group):
  """QQ^T on an extended (layered) operator result: the interface nodes are
  made whole first (`sfem_fold_layers_at`: their layers added up and cleared),
  then exchanged in place like any nodal vector."""
  from swirl_fem_amd import _ops
  _ops.fold_layers_at(ext, plan.interface_nodes(ext.device), num_nodes,
                      layer_plan.layers)
  comm.neighbor_exchange_(ext[:num_nodes], plan, group)
  return ext


class PartitionedOperator:
  """Assembled action `u -> QQ^T A_local(u)` of a rank-local operator."""

  def __init__(self, local_op, plan: comm.NeighborPlan, group=None):
    self.local_op, self.plan, self.group = local_op, plan, group
    if hasattr(local_op, 'apply_with_dot'):
      self.apply_with_dot = self._apply_with_dot
    if hasattr(local_op, 'apply_layered_with_dot'):
      self.apply_layered_with_dot = self._apply_layered_with_dot
      self.layer_plan = local_op.layer_plan

  def _apply_layered_with_dot(self, u, partials, per_wave=False):
    # the rank-local result in layers (no atomics, nothing cleared); only the
    # interface nodes are assembled here, the rest inside `r -= alpha Ap`
    ext = self.local_op.apply_layered_with_dot(u, partials, per_wave)
    return _exchange_layered_(ext, u.shape[0], self.local_op.layer_plan(),
                              self.plan, self.group)

  def __call__(self, u):
    return comm.neighbor_exchange_(self.local_op(u), self.plan, self.group)

  def _apply_with_dot(self, u, partials):
    # u . (unassembled local result), summed over ranks, is the global u.Au
    w = self.local_op.apply_with_dot(u, partials)
    return comm.neighbor_exchange_(w, self.plan, self.group)


class OverlappedHelmholtz:
  """`u -> QQ^T (l0 B + l1 A)_local u` with the interface exchange hidden
  behind the interior elements:

      boundary elements  ->  pack + post send/recv  ->  interior elements
                                       (RCCL stream)        (compute stream)
                         ->  wait + unpack-add

  An element is a boundary element when it holds an interface node; those
  are ~9 % of a 64^3 block, so the exchange (about 1.2 M values per GPU at
  p = 7) overlaps with ~90 % of the apply.  Also hands CG its fused p.Ap.
  """

  def __init__(self, op, plan: comm.NeighborPlan, lambda0=0.0, lambda1=1.0,
               group=None):
    mesh = op.fespace.mesh
    on_iface = torch.zeros(mesh.num_nodes + 1, dtype=torch.bool,
                           device=op.enc.device)
    on_iface[plan.interface_nodes(op.enc.device)] = True
    boundary = on_iface[mesh.elements.to(torch.int64)].any(dim=1)  # -1 -> pad
    self.boundary_op, self.interior_op = op.split(boundary)
    self.plan, self.group = plan, group
    self.lambda0, self.lambda1 = lambda0, lambda1
    self.num_boundary_elements = int(boundary.sum())
    self.num_nodes = mesh.num_nodes
    # layered assembly over BOTH halves (one plan: every writer of a facet,
    # whichever half launches it, has its own layer); None = atomics
    self._layers = self._ext = None
    from swirl_fem_amd.core import operators
    from swirl_fem_amd import switches
    bp, ip = self.boundary_op.facet_parts, self.interior_op.facet_parts
    if (bp is not None and ip is not None and
        switches.get('SFEM_LAYERED') != '0'):
      joint = operators.build_layer_plan(
          list(bp) + list(ip), mesh.num_elements, mesh.num_nodes,
          mesh.gridpoints_1d.num_points)
      if joint is not None and not any(
          q['geo_mode'] == operators._GEO_POINT for q in joint.parts):
        self._layers = joint
        self._halves = (joint.parts[:len(bp)], joint.parts[len(bp):])

  def layer_plan(self):
    return self._layers

  def apply_layered_with_dot(self, u, partials, per_wave=False):
    """The same in layered form (an extended vector owned by this object):
    boundary elements -> interface nodes made whole -> pack + post send/recv
    -> interior elements -> wait + unpack-add.  No atomics besides the
    neighbours' contributions, nothing cleared."""
    from swirl_fem_amd import _ops
    if per_wave:
      raise NotImplementedError('per-wave partial sums with two launches')
    op = self.boundary_op
    mesh = op.fespace.mesh
    if self._ext is None:
      self._ext = torch.zeros(self._layers.extent, dtype=op.fespace.dtype,
                              device=op.enc.device)
    ext, N = self._ext, self.num_nodes
    P = mesh.gridpoints_1d.num_points
    uu = u.to(op.fespace.dtype).contiguous()
    _ops.helmholtz_apply_layered(uu, ext, op.enc, self._halves[0], op.host,
                                 mesh.ndim, P, self.lambda0, self.lambda1,
                                 partials)
    _ops.fold_layers_at(ext, self.plan.interface_nodes(ext.device), N,
                        self._layers.layers)
    handle = comm.neighbor_exchange_start(ext[:N], self.plan, self.group)
    _ops.helmholtz_apply_layered(uu, ext, op.enc, self._halves[1], op.host,
                                 mesh.ndim, P, self.lambda0, self.lambda1,
                                 partials)
    comm.neighbor_exchange_finish(handle, ext[:N])
    return ext

  def _apply(self, u, partials):
    l0, l1 = self.lambda0, self.lambda1
    out = self.boundary_op.apply(u, l0, l1, dot_out=partials)
    handle = comm.neighbor_exchange_start(out, self.plan, self.group)
    self.interior_op.apply(u, l0, l1, out=out, zero=False, dot_out=partials)
    return comm.neighbor_exchange_finish(handle, out)

  def __call__(self, u):
    return self._apply(u, None)

  def apply_with_dot(self, u, partials):
    return self._apply(u, partials)


def make_runner(local_op, b_local, plan: comm.NeighborPlan, *, x0=None,
                tol=1e-5, atol=0.0, maxiter=None, group=None,
                assembled_rhs=False) -> cg_lib.CGRunner:
  """CGRunner for `QQ^T A_local x = QQ^T b_local` on this rank's partition.

  `b_local` is the unassembled local covector (as `local_covector` returns it)
  unless `assembled_rhs`; `x0`, if given, must be consistent.  `local_op` is a
  rank-local operator (wrapped in `PartitionedOperator`) or an
  `OverlappedHelmholtz`, which already returns the assembled result.
  """
  b = b_local if assembled_rhs else comm.neighbor_exchange_(
      b_local.clone(), plan, group)
  reduce_fn = lambda t: comm.all_reduce_sum_(t, group)
  A = (local_op if isinstance(local_op, OverlappedHelmholtz)
       else PartitionedOperator(local_op, plan, group))
  return cg_lib.CGRunner(
      A, b, x0, tol=tol, atol=atol,
      maxiter=maxiter, reduce_fn=reduce_fn,
      interface=plan.interface_weights(b.device, num_nodes=b.shape[0],
                                       group=group))


def cg(local_op, b_local, plan: comm.NeighborPlan, *, x0=None, tol=1e-5,
       atol=0.0, maxiter=None, group=None, assembled_rhs=False,
       check_every=16):
  """Partitioned CG; returns `(x, info)` like `linalg.cg.cg` (x consistent)."""
  run = make_runner(local_op, b_local, plan, x0=x0, tol=tol, atol=atol,
                    maxiter=maxiter, group=group, assembled_rhs=assembled_rhs)
  while run.issued < run.maxiter:
    for _ in range(min(check_every, run.maxiter - run.issued)):
      run.step()
    if run.done():
      break
  return run.x, run.info()
