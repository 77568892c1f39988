"""Crystal router: a sparse, dynamic all-to-all built from log2(P) pairwise
exchanges (Fox et al., "Solving Problems on Concurrent Processors", 1988).

Same contract and the same divide-and-conquer schedule as the reference's
`swirl_fem/communication/crystal_router.py` (:36-110 the user-facing routine,
:238-372 the stage loop): at every stage a contiguous group of ranks splits
into a first (larger or equal) and a second half, partners are assigned by
reversing the index range of the group, and a rank hands its partner *all*
records whose target lies in the partner's half.  In an odd-sized group the
middle rank (last of the first half) sends to the first rank of the second half
and receives nothing during that stage.

The reference runs it inside `shard_map` on padded fixed-size buffers (`n` is
the dynamic length).  Here one process owns one rank (`torch.distributed`, RCCL
or gloo), messages have their true length and records are ordinary tensors, so
`n` is just the number of rows.  Used at setup time only (neighbour discovery
for unstructured partitions, `distributed/discover.py`) -- never inside the
Krylov iteration.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def _rank_world(group):
  return dist.get_rank(group), dist.get_world_size(group)


def _exchange(send_to, payload, recv_from, group):
  """Sends `payload` (list of tensors, same leading length) to `send_to` (or
  nobody if None) and receives one payload from every rank in `recv_from`."""
  device = payload[0].device
  ops, counts = [], []
  n_send = torch.tensor([payload[0].shape[0]], dtype=torch.int64, device=device)
  if send_to is not None:
    ops.append(dist.P2POp(dist.isend, n_send, send_to, group=group))
  for src in recv_from:
    counts.append(torch.zeros(1, dtype=torch.int64, device=device))
    ops.append(dist.P2POp(dist.irecv, counts[-1], src, group=group))
  if ops:
    for req in dist.batch_isend_irecv(ops):
      req.wait()
  ops, received = [], []
  if send_to is not None and int(n_send) > 0:
    for t in payload:
      ops.append(dist.P2POp(dist.isend, t.contiguous(), send_to, group=group))
  for src, cnt in zip(recv_from, counts):
    bufs = [torch.empty((int(cnt),) + tuple(t.shape[1:]), dtype=t.dtype,
                        device=device) for t in payload]
    received.append(bufs)
    if int(cnt) > 0:
      for b in bufs:
        ops.append(dist.P2POp(dist.irecv, b, src, group=group))
  if ops:
    for req in dist.batch_isend_irecv(ops):
      req.wait()
  return received


def crystal_router(n, data, target, return_source: bool = True, group=None):
  """Sends row `j` of every tensor in `data` to rank `target[j]`, j < n.

  Args:
    n: number of valid rows (None = all of them).
    data: a tensor or a list / tuple of tensors with the same leading length.
    target: (len,) integer tensor of destination ranks.
    return_source: also return the rank every received row came from.
  Returns:
    `(n_out, data_out[, source])`; the order of the received rows is
    unspecified, a second call `crystal_router(n_out, data_out, source)`
    returns the original rows up to ordering (reference :74-82).
  """
  single = isinstance(data, torch.Tensor)
  leaves = [data] if single else list(data)
  rank, world = _rank_world(group)
  n = target.shape[0] if n is None else int(n)
  target = target[:n].to(torch.int64)
  if n and (int(target.min()) < 0 or int(target.max()) >= world):
    raise ValueError('crystal_router: target rank out of range')
  leaves = [t[:n] for t in leaves]
  source = torch.full((n,), rank, dtype=torch.int64, device=target.device)
  records = [target, source] + leaves

  lo, hi = 0, world                    # my group: ranks lo .. hi-1
  while hi - lo > 1:
    size = hi - lo
    first = (size + 1) // 2            # the first half is the larger one
    mid = lo + first
    in_first = rank < mid
    # partner by reversing the group's index range
    partner = lo + hi - 1 - rank
    send_to, recv_from = partner, [partner]
    if size % 2 == 1:
      if rank == mid - 1:              # middle rank: sends, receives nothing
        send_to, recv_from = mid, []
      elif rank == mid:                # first of the second half: two senders
        recv_from = [partner, mid - 1]
    tgt = records[0]
    away = (tgt >= mid) if in_first else (tgt < mid)
    out_payload = [t[away] for t in records]
    records = [t[~away] for t in records]
    for bufs in _exchange(send_to, out_payload, recv_from, group):
      records = [torch.cat([a, b]) for a, b in zip(records, bufs)]
    lo, hi = (lo, mid) if in_first else (mid, hi)

  n_out = records[0].shape[0]
  assert bool((records[0] == rank).all())
  out = records[2] if single else type(data)(records[2:]) if isinstance(
      data, tuple) else records[2:]
  if return_source:
    return n_out, out, records[1]
  return n_out, out
