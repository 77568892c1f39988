"""Ranks as threads of one process that share one GPU.

A one-GPU box cannot run the eight ranks of `bench.py --gpus 8` as processes
(its process limit is six), and the 2 x 2 x 2 block layout is the only one with
nodes held by four and by eight ranks.  `ThreadWorld` lets the unchanged
partitioned path -- `blocks.build_block_partition`, `comm.neighbor_exchange*`,
`solver.OverlappedHelmholtz`, the consistent CG with its two scalar
all-reduces -- run with one thread per rank: only the transport underneath
(`comm.exchange_buffers`, the reductions, neighbour discovery) is replaced by
a barrier-synchronised mailbox.  All threads enqueue on the device's default
stream, so a buffer packed by one rank is complete before the kernel of the
rank that reads it runs.  Nothing here is timed meaningfully: the ranks'
kernels serialise on the one GPU.

    world = ThreadWorld(8)
    results = world.run(lambda rank: ...)      # comm.* now serve `rank`
"""

from __future__ import annotations

import threading
import traceback

import numpy as np
import torch

from swirl_fem_amd.distributed import comm


class ThreadWorld:
  """Transport of `comm` when ranks are threads (`comm.set_transport`)."""

  def __init__(self, world: int):
    self.world = int(world)
    self.barrier = threading.Barrier(self.world)
    self._box = {}
    self._local = threading.local()

  @property
  def rank(self) -> int:
    return self._local.rank

  # --- what `comm` calls -------------------------------------------------
  def exchange(self, plan, send_bufs, recv_bufs=None):
    me = self.rank
    for q, sb in zip(plan.neighbors, send_bufs):
      self._box[(me, q)] = sb
    self.barrier.wait()
    got = [self._box[(q, me)].clone() for q in plan.neighbors]
    self.barrier.wait()
    if recv_bufs is None:
      return got
    for rb, g in zip(recv_bufs, got):
      rb.copy_(g)
    return recv_bufs

  def all_reduce(self, t, op):
    me = self.rank
    self._box[('ar', me)] = t.clone()
    self.barrier.wait()
    total = self._box[('ar', 0)]
    for q in range(1, self.world):     # rank order: the same sum on all ranks
      total = op(total, self._box[('ar', q)])
    self.barrier.wait()
    t.copy_(total)
    return t

  def all_gather(self, t):
    me = self.rank
    self._box[('ag', me)] = t.clone()
    self.barrier.wait()
    every = [self._box[('ag', q)].clone() for q in range(self.world)]
    self.barrier.wait()
    return every

  def discover(self, global_keys):
    """The plan `discover.discover_neighbors` would route for: per other rank
    the local positions of the common keys, ordered by key."""
    me = self.rank
    keys = np.asarray(global_keys)
    valid = np.nonzero(keys >= 0)[0]       # negative = padding / non-members
    self._box[('keys', me)] = keys[valid]
    self.barrier.wait()
    neighbors, indices = [], []
    for q in range(self.world):
      if q == me:
        continue
      _, pos, _ = np.intersect1d(keys[valid], self._box[('keys', q)],
                                 return_indices=True)     # sorted by key
      if len(pos):
        neighbors.append(q)
        indices.append(valid[pos].astype(np.int32))
    self.barrier.wait()
    return comm.NeighborPlan(rank=me, neighbors=neighbors, indices=indices)

  # --- driver --------------------------------------------------------------
  def run(self, rank_main, timeout=1800.0):
    """Runs `rank_main(rank)` on one thread per rank with this world installed
    as `comm`'s transport; returns {rank: result}.  The first exception of any
    rank aborts the others' barriers and is re-raised."""
    results, errors = {}, []

    def body(rank):
      try:
        self._local.rank = rank
        results[rank] = rank_main(rank)
      except BaseException:           # pylint: disable=broad-except
        errors.append((rank, traceback.format_exc()))
        self.barrier.abort()

    previous = comm.transport()
    comm.set_transport(self)
    try:
      threads = [threading.Thread(target=body, args=(r,), daemon=True)
                 for r in range(self.world)]
      for t in threads:
        t.start()
      for t in threads:
        t.join(timeout=timeout)
      if any(t.is_alive() for t in threads):
        self.barrier.abort()
        raise RuntimeError('ThreadWorld: a rank did not finish in %.0f s'
                           % timeout)
    finally:
      comm.set_transport(previous)
    if errors:
      errors.sort()
      first = [e for e in errors if 'BrokenBarrierError' not in e[1]] or errors
      raise RuntimeError('rank %d failed:\n%s' % first[0])
    if torch.cuda.is_available():
      torch.cuda.synchronize()
    return results
