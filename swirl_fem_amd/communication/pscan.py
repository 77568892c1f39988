"""Prefix scans and reductions over the ranks (reference
`swirl_fem/communication/pscan.py:243-290`): `pscan` is the EXCLUSIVE scan --
rank r receives op(x_0, ..., x_{r-1}), rank 0 the identity -- optionally with
the all-reduce as a second result.  The reference runs a fan-in / fan-out
tree inside `shard_map`; with one process per rank an all-gather of the (small)
setup-time operands followed by a local scan does the same in one collective.
"""

from __future__ import annotations

import torch
import torch.distributed as dist

_OPS = {
    'add': (torch.cumsum, lambda t: torch.zeros_like(t)),
    'maximum': (lambda t, dim: torch.cummax(t, dim=dim).values, None),
    'minimum': (lambda t, dim: torch.cummin(t, dim=dim).values, None),
    'multiply': (torch.cumprod, lambda t: torch.ones_like(t)),
}


def _identity(op, x):
  if op in ('add', 'multiply'):
    return _OPS[op][1](x)
  info = (torch.finfo(x.dtype) if x.dtype.is_floating_point
          else torch.iinfo(x.dtype))
  return torch.full_like(x, info.min if op == 'maximum' else info.max)


def pscan(x: torch.Tensor, op: str = 'add', reduction: bool = False, group=None):
  """Exclusive prefix scan of `x` over the ranks (same shape as `x`)."""
  if op not in _OPS:
    raise ValueError(f'unsupported scan operation {op!r}')
  world = dist.get_world_size(group)
  rank = dist.get_rank(group)
  parts = [torch.empty_like(x) for _ in range(world)]
  dist.all_gather(parts, x.contiguous(), group=group)
  stacked = torch.stack(parts)                       # (P,) + x.shape
  inclusive = _OPS[op][0](stacked, dim=0)
  scan = _identity(op, x) if rank == 0 else inclusive[rank - 1]
  if reduction:
    return scan, inclusive[-1]
  return scan


def preduce(x: torch.Tensor, op: str = 'add', group=None):
  """All-reduce of `x` over the ranks."""
  return pscan(x, op, reduction=True, group=group)[1]
