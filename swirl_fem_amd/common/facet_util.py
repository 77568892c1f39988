"""Facets of tensor-product (quad / hex) elements.

Same vocabulary as the reference `swirl_fem/common/facet_util.py`
(`FacetDimType` :46-50, `slice_from_facet_type` :53-75, `get_facet_types`
:78-92, `get_orderings_mapping` :95-143): a d-cube has 3^d facets, each named
by a d-tuple saying whether an axis contributes its FIRST vertex, its LAST
vertex or its INNER nodes.  In addition this module exposes array-valued
tables (`orientation_table`) used by the vectorised mesh refiner.
"""

from __future__ import annotations

import enum
import functools
import itertools

import numpy as np


@enum.unique
class FacetDimType(enum.Enum):
  """Category of nodes included in a facet along a dimension."""
  FIRST = 'first'
  LAST = 'last'
  INNER = 'inner'


def slice_from_facet_type(facet_type, interior_nodes_only: bool):
  """Slice extracting a facet from an element index array of shape [k+1]*d."""
  inner = slice(1, -1) if interior_nodes_only else slice(None)
  table = {FacetDimType.FIRST: 0, FacetDimType.LAST: -1,
           FacetDimType.INNER: inner}
  return tuple(table[t] for t in facet_type)


def get_facet_types(ndim: int, facet_ndim: int | None = None):
  """All 3^ndim facet signatures (optionally only those of one dimension)."""
  facets = list(itertools.product(list(FacetDimType), repeat=ndim))
  if facet_ndim is None:
    return facets
  return [f for f in facets if f.count(FacetDimType.INNER) == facet_ndim]


def _orientations(ndim: int):
  """Yields (axis permutation, flipped axes) for the 2^d d! orientations."""
  for perm in itertools.permutations(range(ndim)):
    for r in range(ndim + 1):
      for axes in itertools.combinations(range(ndim), r):
        yield perm, axes


def get_orderings_mapping(ndim: int, num_points_1d: int):
  """Maps orderings of the 2^d vertices to orderings of the refined nodes."""
  source = np.arange(2 ** ndim, dtype=np.int32).reshape([2] * ndim)
  target = np.arange(num_points_1d ** ndim, dtype=np.int32).reshape(
      [num_points_1d] * ndim)
  orderings = {}
  for perm, axes in _orientations(ndim):
    key = tuple(np.flip(source.transpose(perm), axes).flatten().tolist())
    orderings[key] = np.flip(target.transpose(perm), axes).flatten()
  return orderings


@functools.lru_cache(maxsize=None)
def orientation_table(ndim: int, num_points_1d: int):
  """Array form of `get_orderings_mapping` for vectorised lookups.

  Returns `(codes, perms)`: `codes[i]` is the base-2^d encoding
  `sum(key[j] * (2^d)^j)` of the i-th vertex ordering (sorted ascending) and
  `perms[i]` the matching permutation of the `num_points_1d^d` refined nodes.
  """
  mapping = get_orderings_mapping(ndim, num_points_1d)
  base = 2 ** ndim
  weights = base ** np.arange(base, dtype=np.int64)
  keys = np.array(list(mapping.keys()), dtype=np.int64)
  codes = keys @ weights
  perms = np.stack([mapping[tuple(k)] for k in keys.tolist()]).astype(np.int64)
  order = np.argsort(codes)
  return codes[order], perms[order]
