"""Neighbour discovery for arbitrary partitions, without a central table.

The reference derives the shared-DOF exchange from a dense `(P, S)` table built
on one host from the global mesh (`core/gather_scatter.py:318-358`).  For a
partition that only knows its own nodes -- their global ids, or any key that
identifies a DOF globally -- the same information follows from one round trip
through the crystal router (the role gslib's setup plays for Nek):

  1. every rank sends `(key, local position)` of its nodes to the key's home
     rank `key mod P`;
  2. the home rank sorts what it received and, for every key held by more than
     one rank, sends each holder the list of the other holders;
  3. every rank groups what comes back by neighbour rank and orders each list
     by key -- both sides of a pair see the same keys, hence the same order.

The result is the `NeighborPlan` consumed by `comm.neighbor_exchange*`.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from swirl_fem_amd.communication.crystal_router import crystal_router
from swirl_fem_amd.distributed.comm import NeighborPlan


def discover_neighbors(global_keys, group=None, device='cpu') -> NeighborPlan:
  """`global_keys`: (N_local,) int64, negative entries (padding) are ignored;
  a key may occur once per rank (images of one DOF on the same rank are the
  local periodic exchange's business and are skipped here).
  `device`: where the routing buffers live ('cpu' for gloo, a GPU for RCCL)."""
  from swirl_fem_amd.distributed import comm
  if comm.transport() is not None:      # ranks are threads of this process
    return comm.transport().discover(global_keys)
  rank, world = dist.get_rank(group), dist.get_world_size(group)
  keys = torch.as_tensor(np.asarray(global_keys), dtype=torch.int64,
                         device=device)
  pos = torch.nonzero(keys >= 0).reshape(-1)
  keys = keys[pos]
  # 1. to the home rank
  _, (k_in, p_in), src = crystal_router(None, [keys, pos], keys % world,
                                        group=group)
  # 2. holders of every key that has more than one
  order = torch.argsort(k_in * world + src)        # by key, then by holder
  k_s, p_s, r_s = k_in[order], p_in[order], src[order]
  if k_s.numel():
    _, inv, cnt = torch.unique_consecutive(k_s, return_inverse=True,
                                           return_counts=True)
    shared = cnt[inv] > 1
    k_s, p_s, r_s = k_s[shared], p_s[shared], r_s[shared]
  if k_s.numel():
    # all ordered pairs (holder a, holder b != a) within one key's group
    _, inv, cnt = torch.unique_consecutive(k_s, return_inverse=True,
                                           return_counts=True)
    size = cnt[inv]
    start = (torch.cumsum(cnt, 0) - cnt)[inv]
    a = torch.arange(k_s.numel(), device=device)
    ia, ib = [], []
    for shift in range(1, int(cnt.max())):
      ok = size > shift
      ia.append(a[ok])
      ib.append((start + (a - start + shift) % size)[ok])
    ia, ib = torch.cat(ia), torch.cat(ib)
    keep = r_s[ia] != r_s[ib]      # two images on one rank: a local matter
    ia, ib = ia[keep], ib[keep]
    # to holder a: (key, its local position, the other holder)
    back, dest = [k_s[ia], p_s[ia], r_s[ib]], r_s[ia]
  else:
    back, dest = [k_s, p_s, r_s], r_s
  # 3. back to the holders
  _, (k_b, p_b, other), _ = crystal_router(None, back, dest, group=group)
  neighbors, indices = [], []
  for q in torch.unique(other).tolist():
    sel = other == q
    kk, pp = k_b[sel], p_b[sel]
    o = torch.argsort(kk)
    neighbors.append(int(q))
    indices.append(pp[o].cpu().numpy().astype(np.int32))
  return NeighborPlan(rank=rank, neighbors=neighbors, indices=indices)
