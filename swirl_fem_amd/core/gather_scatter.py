"""Global<->local gather/scatter and the shared-DOF exchange QQ^T.

Mirrors `swirl_fem/core/gather_scatter.py` of the reference:
  device ops  : `gather` :121-127, `scatter` :130-133, `exchange` :189-261
  index builders (host, NumPy): `get_unique_node_indices` :136-160,
    `get_exchange_indices` :166-186 (+ `_unpartitioned` :284-315,
    `_partitioned` :318-358), `group_by_partitions` :369-396,
    `get_local_elements` :399-445.

The device ops run hand-written HIP kernels through the C-ABI
(`include/sfem.h`); the index builders are vectorised NumPy replacements of the
reference's Python dict/Counter loops and reproduce its numbering exactly
(pinned by `tests/golden/`).

Partitioned exchange: the reference all-reduces a dense vector of *all* S
globally shared DOFs (`lax.psum`, :247-248).  Here one process owns one
partition and the same QQ^T is carried by a neighbour exchange over RCCL
(`swirl_fem_amd/distributed/comm.py`): each shared DOF travels only to the
ranks that hold a copy.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse
import scipy.sparse.csgraph

# A dummy index used to denote missing entries.
SENTINEL = -1


# ----------------------------------------------------------------------------
# Device ops
# ----------------------------------------------------------------------------
def gather(u, indices, fill_value=SENTINEL):
  """`u[indices]` with SENTINEL entries replaced by `fill_value`."""
  from swirl_fem_amd import _ops
  if u.ndim != 1:
    raise ValueError(f'Expecting a rank-1 array. Got {tuple(u.shape)}')
  from swirl_fem_amd.core import autodiff
  if autodiff.needs_grad(u):
    return autodiff.gather(u, indices, float(fill_value))
  return _ops.gather(u, indices, float(fill_value))


def scatter(u, indices, num_nodes: int):
  """Direct-stiffness sum: out[k] = sum of u over entries with index k."""
  from swirl_fem_amd import _ops
  assert tuple(u.shape) == tuple(indices.shape), (
      f'Got: {tuple(u.shape)} v/s {tuple(indices.shape)}')
  from swirl_fem_amd.core import autodiff
  if autodiff.needs_grad(u):
    return autodiff.scatter_add(u, indices, int(num_nodes))
  return _ops.scatter_add(u, indices, int(num_nodes))


def exchange(u, gather_indices, unique_indices=None, axis_name=None, *,
             plan=None):
  """Applies QQ^T: sums nodal values over periodic images / partition copies.

  Args:
    u: `(num_nodes,)` device tensor.
    gather_indices: `(S,)` local positions of participating DOFs (SENTINEL
      where this partition does not hold the DOF), or None / empty.
    unique_indices: `(S,)` host array mapping each participating position to
      its unique DOF (unpartitioned + periodic case), or None.
    axis_name: non-None in the partitioned case.
    plan: `distributed.comm.NeighborPlan` of this rank (partitioned case).
  """
  from swirl_fem_amd import _ops
  if gather_indices is None or gather_indices.numel() == 0:
    return u
  if axis_name is None:
    if unique_indices is None:
      # Every participating DOF is its own class: QQ^T is the identity.
      return u.clone()
    from swirl_fem_amd.core import autodiff
    if autodiff.needs_grad(u):
      return autodiff.exchange_local(u, gather_indices, unique_indices)
    return _ops.exchange_local(u, gather_indices, unique_indices)
  from swirl_fem_amd.distributed import comm
  if plan is None:
    raise ValueError('a partitioned exchange needs the rank\'s NeighborPlan')
  if unique_indices is not None and not plan.has_local_images:
    # the reference stops here too (core/gather_scatter.py:352-353); the block
    # builder's plans carry the local images themselves
    raise NotImplementedError(
        'intra-partition periodicity combined with partitioning')
  return comm.neighbor_exchange(u, plan)


# ----------------------------------------------------------------------------
# Host index builders
# ----------------------------------------------------------------------------
def _get_periodic_mapping(periodic_links):
  """(sorted node ids, representative of each) for nodes in periodic links."""
  if periodic_links is None or len(periodic_links) == 0:
    return (np.zeros(0, dtype=np.int64),) * 2
  links = np.asarray(periodic_links)
  edges = np.transpose(links, axes=(0, 2, 1)).reshape(-1, 2).astype(np.int64)
  nodes, inv = np.unique(edges.reshape(-1), return_inverse=True)
  inv = inv.reshape(-1, 2)
  graph = scipy.sparse.coo_matrix(
      (np.ones(len(inv), dtype=np.int8), (inv[:, 0], inv[:, 1])),
      shape=(len(nodes), len(nodes)))
  ncomp, labels = scipy.sparse.csgraph.connected_components(
      graph, directed=False)
  rep = np.full(ncomp, np.iinfo(np.int64).max, dtype=np.int64)
  np.minimum.at(rep, labels, nodes)
  return nodes, rep[labels]


def get_unique_node_indices(node_indices: np.ndarray,
                            periodic_links) -> np.ndarray:
  """De-duplicates node ids: every periodic class maps to its minimum id."""
  if periodic_links is None or len(periodic_links) == 0:
    return node_indices
  nodes, reps = _get_periodic_mapping(periodic_links)
  flat = np.asarray(node_indices)
  pos = np.searchsorted(nodes, flat)
  pos_c = np.minimum(pos, len(nodes) - 1)
  hit = nodes[pos_c] == flat
  out = np.where(hit, reps[pos_c], flat)
  return out.astype(flat.dtype, copy=False)


def _shared_ranks(flat: np.ndarray):
  """Helper: per entry, rank of its id among ids occurring more than once.

  Returns (is_shared per entry, rank per entry, number of shared ids).
  """
  uniq, inv, counts = np.unique(flat, return_inverse=True, return_counts=True)
  shared = (counts > 1) & (uniq != SENTINEL)
  rank = np.cumsum(shared) - 1
  return shared[inv], rank[inv], int(shared.sum())


def _get_exchange_indices_unpartitioned(node_indices: np.ndarray):
  assert node_indices.ndim == 1, node_indices.ndim
  is_shared, rank, _ = _shared_ranks(node_indices)
  pos = np.nonzero(is_shared)[0]
  return pos.astype(np.int32), rank[pos].astype(np.int32)


def _get_exchange_indices_partitioned(node_indices: np.ndarray):
  assert node_indices.ndim == 2, node_indices.shape
  num_partitions, num_local = node_indices.shape
  is_shared, rank, num_shared = _shared_ranks(node_indices.reshape(-1))
  is_shared = is_shared.reshape(node_indices.shape)
  rank = rank.reshape(node_indices.shape)
  gather_indices = np.full((num_partitions, num_shared), SENTINEL,
                           dtype=np.int64)
  for p in range(num_partitions):
    pos = np.nonzero(is_shared[p])[0]
    r = rank[p, pos]
    if len(np.unique(r)) != len(r):
      # the same global DOF twice in one partition: intra-partition periodicity
      dup = node_indices[p, pos][np.argsort(r, kind='stable')]
      d = dup[1:][dup[1:] == dup[:-1]]
      raise NotImplementedError(
          f'Found node_idx={int(d[0])} occurring more than once in '
          f'partition_idx={p}')
    gather_indices[p, r] = pos
  return gather_indices, None


def get_exchange_indices(node_indices: np.ndarray):
  """Returns `(gather_indices, unique_indices)` for `exchange`."""
  node_indices = np.asarray(node_indices)
  if node_indices.ndim not in (1, 2):
    raise ValueError('node_indices must have ndim 1 or 2. Got '
                     f'{node_indices.ndim}')
  if node_indices.ndim == 2:
    return _get_exchange_indices_partitioned(node_indices)
  return _get_exchange_indices_unpartitioned(node_indices)


def _pad_evenly(rows):
  n = max(len(r) for r in rows)
  out = np.full((len(rows), n), SENTINEL, dtype=np.int64)
  for i, r in enumerate(rows):
    out[i, :len(r)] = r
  return out


def group_by_partitions(partitions: np.ndarray) -> np.ndarray:
  """`(P, n)` array of the element ids of each partition, SENTINEL padded."""
  partitions = np.asarray(partitions)
  assert partitions.ndim == 1, partitions.shape
  num_partitions = 1 + int(partitions.max())
  order = np.argsort(partitions, kind='stable')
  counts = np.bincount(partitions, minlength=num_partitions)
  rows = np.split(order, np.cumsum(counts)[:-1])
  return _pad_evenly(rows).astype(np.int32)


def get_local_elements(elements):
  """Renumbers per-partition elements from global to partition-local ids.

  Args:
    elements: `(P, ...)` global node ids (SENTINEL allowed).
  Returns:
    node_indices `(P, Nloc)`: local -> global id, ascending, SENTINEL padded;
    local_elements: same shape as `elements`.
  """
  elements = [np.asarray(e) for e in elements]
  uniques = [np.unique(e[e != SENTINEL]) for e in elements]
  local = []
  for e, u in zip(elements, uniques):
    if len(u) == 0:
      local.append(np.full(e.shape, SENTINEL, dtype=np.int64))
      continue
    pos = np.searchsorted(u, e)
    pos = np.minimum(pos, len(u) - 1)
    local.append(np.where(e != SENTINEL, pos, SENTINEL).astype(np.int64))
  return _pad_evenly(uniques), np.stack(local)
