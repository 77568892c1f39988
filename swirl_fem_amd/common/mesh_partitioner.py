"""Assigns every element of a `Premesh` to one of `num_partitions` partitions.

Same entry point as the reference's `common/mesh_partitioner.py:21-53`, which
hands the node-sharing graph of the elements to METIS (`pymetis`, absent
here).  This build needs no graph library: recursive coordinate bisection of
the element centroids.  Each cut splits a set of elements at the weighted
median along its longest extent, in proportion to the number of partitions on
either side, so partition sizes differ by at most one element and partitions
are compact boxes of elements -- which is what keeps the shared-DOF interface
(the RCCL neighbour exchange, `distributed/comm.py`) small.  The properties
the reference tests (`mesh_partitioner_test.py:38-82`) hold by construction:
ids in `[0, num_partitions)`, sizes within floor/ceil of the mean, contiguous
ranges on 1D meshes.
"""

from __future__ import annotations

import numpy as np

from swirl_fem_amd.core.premesh import Premesh


def _bisect(ids, centroids, first, count, out):
  if count == 1:
    out[ids] = first
    return
  left_parts = count // 2
  # elements on the left: proportional share, balanced to within one
  n_left = (len(ids) * left_parts + count - 1) // count
  n_left = min(max(n_left, min(left_parts, len(ids))), len(ids))
  x = centroids[ids]
  axis = int(np.argmax(x.max(axis=0) - x.min(axis=0))) if len(ids) else 0
  # stable order: ties broken by the other coordinates, then element id
  keys = [ids] + [x[:, d] for d in range(x.shape[1]) if d != axis] + [x[:, axis]]
  order = np.lexsort(keys)
  _bisect(ids[order[:n_left]], centroids, first, left_parts, out)
  _bisect(ids[order[n_left:]], centroids, first + left_parts,
          count - left_parts, out)


def partition(premesh: Premesh, num_partitions: int) -> Premesh:
  """Returns `premesh` with a partition id in `[0, num_partitions)` per
  element (`Premesh.partitions`); periodic links are ignored, as in the
  reference."""
  if num_partitions < 1:
    raise ValueError(f'num_partitions must be positive, got {num_partitions}')
  elements = np.asarray(premesh.elements)
  coords = np.asarray(premesh.node_coords, dtype=np.float64)
  valid = elements >= 0
  safe = np.where(valid, elements, 0)
  centroids = (coords[safe] * valid[..., None]).sum(axis=1) / np.maximum(
      valid.sum(axis=1), 1)[:, None]
  out = np.zeros(len(elements), dtype=np.int32)
  _bisect(np.arange(len(elements)), centroids, 0, int(num_partitions), out)
  return premesh.replace(partitions=out)
