"""Partitioned shared-DOF exchange: RCCL neighbour send/recv over xGMI.

Replaces the one collective on the reference's hot path: `lax.psum` of a dense
vector of ALL globally shared DOFs inside `gather_scatter.exchange`
(core/gather_scatter.py:247-248, indices built at :318-358).  A ring
all-reduce of that vector moves 2 (P-1)/P S values through every GPU and is
bound by one ~153 GB/s xGMI link; here each shared DOF travels only to the
ranks that hold a copy, as one packed message per neighbour, and the (up to 7)
point-to-point links of a GPU are used in parallel:

    pack (HIP gather)  ->  grouped ncclSend/ncclRecv  ->  unpack-add (HIP)

One process owns one partition (`torch.distributed`, backend "nccl" = RCCL on
ROCm; the CPU tests drive `exchange_buffers` over "gloo" and do the pack /
unpack halves with the oracle on their side).  The result is bit-for-bit QQ^T u: every rank
adds the *original* values of all other holders of a DOF to its own.
"""

from __future__ import annotations

import dataclasses
import os

import numpy as np
import torch
import torch.distributed as dist


# Set by tests that replace `exchange_buffers` with an in-process transport:
# the split exchange then goes through it instead of posting P2P operations.
_BLOCKING_TRANSPORT = False

# None: ranks are processes and talk through `torch.distributed` (RCCL / gloo).
# An `inprocess.ThreadWorld`: ranks are threads of this process that share one
# GPU (`bench.py --backend threads`: the 2 x 2 x 2 layout of `--gpus 8` on a
# one-GPU box, whose process limit rules out eight gloo ranks).
_TRANSPORT = None


def set_transport(world) -> None:
  global _TRANSPORT
  _TRANSPORT = world


def transport():
  return _TRANSPORT


def get_rank() -> int:
  if _TRANSPORT is not None:
    return _TRANSPORT.rank
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank()
  return int(os.environ.get('RANK', '0'))


def get_world_size() -> int:
  if _TRANSPORT is not None:
    return _TRANSPORT.world
  if dist.is_available() and dist.is_initialized():
    return dist.get_world_size()
  return int(os.environ.get('WORLD_SIZE', '1'))


def barrier() -> None:
  """All ranks (no-op for a single rank)."""
  if _TRANSPORT is not None:
    _TRANSPORT.barrier.wait()
  elif dist.is_available() and dist.is_initialized() and get_world_size() > 1:
    dist.barrier()


def all_reduce_max_(t: torch.Tensor) -> torch.Tensor:
  if _TRANSPORT is not None:
    return _TRANSPORT.all_reduce(t, torch.maximum)
  if dist.is_available() and dist.is_initialized() and get_world_size() > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return t


def all_gather(t: torch.Tensor) -> list:
  """[rank 0's t, rank 1's t, ...] on every rank."""
  if _TRANSPORT is not None:
    return _TRANSPORT.all_gather(t)
  if dist.is_available() and dist.is_initialized() and get_world_size() > 1:
    if t.is_cuda and dist.get_backend() == 'gloo':
      # gloo gathers host memory only (single-GPU rehearsals of the ranks)
      host = t.cpu()
      every = [torch.zeros_like(host) for _ in range(get_world_size())]
      dist.all_gather(every, host)
      return [e.to(t.device) for e in every]
    every = [torch.zeros_like(t) for _ in range(get_world_size())]
    dist.all_gather(every, t)
    return every
  return [t]


@dataclasses.dataclass(eq=False)
class NeighborPlan:
  """Per-neighbour lists of local positions of the DOFs shared with it.

  `neighbors[i]` is a rank; `indices[i]` the local node positions (host int32
  array) of the DOFs shared with that rank, ordered by global shared-DOF id so
  that both sides enumerate them identically.
  """
  rank: int
  neighbors: list
  indices: list
  # Periodic images held by this rank itself (a box that is periodic in a
  # direction with a single block): `local_gather[i]` is the position of an
  # image node, `local_unique[i]` its class and `local_rep[i]` the position of
  # the class representative.  Only representatives appear in `indices`.
  local_gather: np.ndarray | None = None
  local_unique: np.ndarray | None = None
  local_rep: np.ndarray | None = None
  _dev: dict = dataclasses.field(default_factory=dict, repr=False)

  @property
  def has_local_images(self) -> bool:
    return self.local_gather is not None and len(self.local_gather) > 0

  def local_device(self, device):
    """(gather int32, unique host array, gather int64, rep int64) on device."""
    key = ('local', str(device))
    if key not in self._dev:
      g = np.ascontiguousarray(self.local_gather, dtype=np.int32)
      self._dev[key] = (
          torch.as_tensor(g, dtype=torch.int32, device=device),
          np.ascontiguousarray(self.local_unique, dtype=np.int32),
          torch.as_tensor(g, dtype=torch.int64, device=device),
          torch.as_tensor(np.asarray(self.local_rep), dtype=torch.int64,
                          device=device))
    return self._dev[key]

  @classmethod
  def from_gather_indices(cls, gather_indices: np.ndarray,
                          rank: int) -> 'NeighborPlan':
    """From the reference-style `(P, S)` table (-1 = rank lacks the DOF)."""
    gi = np.asarray(gather_indices)
    mine = gi[rank] >= 0
    neighbors, indices = [], []
    for q in range(gi.shape[0]):
      if q == rank:
        continue
      both = mine & (gi[q] >= 0)
      if both.any():
        neighbors.append(q)
        indices.append(gi[rank, both].astype(np.int32))
    return cls(rank=rank, neighbors=neighbors, indices=indices)

  def device_indices(self, device):
    key = str(device)
    if key not in self._dev:
      self._dev[key] = [torch.as_tensor(ix, dtype=torch.int32, device=device)
                        for ix in self.indices]
    return self._dev[key]

  @property
  def num_shared(self) -> int:
    return int(sum(len(ix) for ix in self.indices))

  def concat_indices(self, device):
    """All per-neighbour lists back to back (int32, device) + their lengths."""
    key = ('cat', str(device))
    if key not in self._dev:
      cat = (np.concatenate(self.indices) if self.indices
             else np.zeros(0, np.int32))
      self._dev[key] = (torch.as_tensor(cat, dtype=torch.int32, device=device),
                        [len(ix) for ix in self.indices])
    return self._dev[key]

  def interface_weights(self, device, num_nodes=None, group=None):
    """`(idx, w)`: the nodes of this partition that have other holders (each
    once, int64) and `w = 1 - 1/m`, m = number of holders (other ranks and,
    with local periodic images, other images on this rank).  For vectors that
    are *consistent* (equal on all holders)  sum_ranks (a.b - sum_idx w a b)
    is the inner product over the unique global nodes."""
    key = ('w', str(device))
    if key not in self._dev:
      if self.has_local_images:
        # count the holders by exchanging ones (a collective call)
        if num_nodes is None:
          raise ValueError('interface_weights: num_nodes needed with local '
                           'periodic images')
        m = neighbor_exchange_(torch.ones(num_nodes, dtype=torch.float64,
                                          device=device), self, group)
        idx = torch.nonzero(m > 1.5).reshape(-1)
        self._dev[key] = (idx, 1.0 - 1.0 / m[idx])
      else:
        cat = (np.concatenate(self.indices) if self.indices
               else np.zeros(0, np.int32))
        idx, cnt = np.unique(cat[cat >= 0], return_counts=True)
        w = 1.0 - 1.0 / (1.0 + cnt)
        self._dev[key] = (
            torch.as_tensor(idx, dtype=torch.int64, device=device),
            torch.as_tensor(w, dtype=torch.float64, device=device))
    return self._dev[key]

  def interface_nodes(self, device) -> torch.Tensor:
    """Positions (int64, unique) of every node that takes part in the exchange:
    nodes shared with other ranks and local periodic images."""
    key = ('iface', str(device))
    if key not in self._dev:
      parts = [np.asarray(ix) for ix in self.indices]
      if self.has_local_images:
        parts += [np.asarray(self.local_gather), np.asarray(self.local_rep)]
      cat = np.concatenate(parts) if parts else np.zeros(0, np.int64)
      self._dev[key] = torch.as_tensor(np.unique(cat[cat >= 0]),
                                       dtype=torch.int64, device=device)
    return self._dev[key]


def exchange_buffers(plan: NeighborPlan, send_bufs, group=None,
                     recv_bufs=None):
  """Sends `send_bufs[i]` to `plan.neighbors[i]` and returns what they sent.

  One grouped batch of P2P ops (RCCL: ncclGroupStart ... ncclGroupEnd), so all
  neighbour links are driven concurrently.
  """
  if _TRANSPORT is not None:
    return _TRANSPORT.exchange(plan, send_bufs, recv_bufs)
  if recv_bufs is None:
    recv_bufs = [torch.empty_like(b) for b in send_bufs]
  if not plan.neighbors:
    return recv_bufs
  if send_bufs[0].is_cuda and dist.get_backend(group) == 'gloo':
    # gloo has no device-memory send/recv: single-GPU rehearsals of the
    # multi-rank path stage the (small) interface buffers through the host.
    host = exchange_buffers(plan, [b.cpu() for b in send_bufs], group=group)
    for rb, hb in zip(recv_bufs, host):
      rb.copy_(hb)
    return recv_bufs
  ops = []
  for q, sb, rb in zip(plan.neighbors, send_bufs, recv_bufs):
    ops.append(dist.P2POp(dist.isend, sb, q, group=group))
    ops.append(dist.P2POp(dist.irecv, rb, q, group=group))
  for req in dist.batch_isend_irecv(ops):
    req.wait()
  return recv_bufs


def neighbor_exchange(u: torch.Tensor, plan: NeighborPlan,
                      group=None) -> torch.Tensor:
  """QQ^T u for this rank's partition, as a new tensor (same memory layout as
  `u`).  Device tensors only: the pack / unpack halves are HIP kernels and
  nothing else."""
  return neighbor_exchange_(u.clone(), plan, group)


def _local_sum_(u, plan):
  """Step 1 with local periodic images: every image gets its class sum."""
  from swirl_fem_amd import _ops
  gi, ui, _, _ = plan.local_device(u.device)
  _ops.exchange_local(u, gi, ui, inplace=True)


def _local_broadcast_(u, plan):
  """Step 3: the representative's (now global) value goes to all images."""
  _, _, g64, rep = plan.local_device(u.device)
  u[g64] = u[rep]


def neighbor_exchange_(u: torch.Tensor, plan: NeighborPlan,
                       group=None) -> torch.Tensor:
  """In-place QQ^T on the interface nodes: one pack launch for all neighbours,
  one grouped send/recv, one atomic unpack-add.  `u` may be (N,), (N, nc)
  row-major or component-major; interior nodes are not touched (no clone).

  With local periodic images (see `NeighborPlan`): sum the images on this
  rank, exchange the class representatives with the other ranks, copy the
  result back to every image."""
  return neighbor_exchange_finish(neighbor_exchange_start(u, plan, group), u)


def neighbor_exchange_start(u: torch.Tensor, plan: NeighborPlan, group=None):
  """First half of `neighbor_exchange_`: packs the interface values of `u` and
  posts the grouped send/recv without waiting.  RCCL runs the transfers on its
  own stream, so kernels enqueued before `neighbor_exchange_finish` (the
  interior elements of an operator) overlap with them.  `u` must not change at
  the interface nodes in between."""
  from swirl_fem_amd import _ops
  if plan.has_local_images:
    _local_sum_(u, plan)
  if not plan.neighbors:
    return (None, None, [], None, plan)
  cat, sizes = plan.concat_indices(u.device)
  send = _ops.pack_strided(u, cat)
  recv = torch.empty_like(send)
  sends, recvs = list(torch.split(send, sizes)), list(torch.split(recv, sizes))
  if _BLOCKING_TRANSPORT or _TRANSPORT is not None or (
      send.is_cuda and dist.get_backend(group) == 'gloo'):
    exchange_buffers(plan, sends, group=group, recv_bufs=recvs)   # blocking
    return (recv, cat, [], send, plan)
  ops = []
  for q, sb, rb in zip(plan.neighbors, sends, recvs):
    ops.append(dist.P2POp(dist.isend, sb, q, group=group))
    ops.append(dist.P2POp(dist.irecv, rb, q, group=group))
  return (recv, cat, dist.batch_isend_irecv(ops), send, plan)


def neighbor_exchange_finish(handle, u: torch.Tensor) -> torch.Tensor:
  """Second half: waits for the transfers and adds the neighbours' values."""
  from swirl_fem_amd import _ops
  recv, cat, reqs, _send, plan = handle
  for req in reqs:
    req.wait()
  if recv is not None:
    _ops.unpack_add_atomic(recv, cat, u)
  if plan.has_local_images:
    _local_broadcast_(u, plan)
  return u


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
  """In-place sum over ranks (the CG scalars; one fused all-reduce)."""
  if _TRANSPORT is not None:
    return _TRANSPORT.all_reduce(t, torch.add)
  if dist.is_available() and dist.is_initialized() and get_world_size() > 1:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
  return t
