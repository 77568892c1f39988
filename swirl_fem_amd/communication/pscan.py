"""Prefix scans and reductions over the ranks (reference
`swirl_fem/communication/pscan.py:243-296`).

`pscan(x, op)` is the EXCLUSIVE scan -- rank r receives op(x_0, ..., x_{r-1}),
rank 0 the monoid's unit (`unit_table`, reference :42-51) -- optionally with the
all-reduce as a second result; `preduce` is the all-reduce.  `x` may be a
pytree (dict / list / tuple of tensors): the result has the same structure, as
if the function were mapped over the leaves (reference :225-241).

`op` is one of the reference's seven monoids, given by name ('add',
'multiply', 'maximum', 'minimum', 'bitwise_and', 'bitwise_or', 'bitwise_xor')
or by the torch function of that name (`torch.add`, `torch.maximum`,
`torch.bitwise_xor`, ...), where the reference takes `jnp.add` etc.

Substitution, stated: the reference runs a fan-in / fan-out tree of
point-to-point shuffles inside `shard_map` (:53-223, log2 P rounds).  With one
process per rank over RCCL the operands are small setup-time arrays (global
numbering offsets, counts), so ONE all-gather of all leaves followed by a
local scan over the P gathered copies replaces the 2 log2 P dependent rounds:
same results (the fold runs left to right over the ranks, the order of the
sequential definition), one collective launch instead of 2 log2 P.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def _dtype_range(dtype):
  if dtype == torch.bool:
    return False, True
  info = torch.finfo(dtype) if dtype.is_floating_point else torch.iinfo(dtype)
  return info.min, info.max


def _all_ones(dtype):
  return True if dtype == torch.bool else -1      # two's complement ~0


# name -> (binary op, unit of the monoid for a dtype)      reference :42-51
_MONOIDS = {
    'add': (torch.add, lambda t: 0),
    'multiply': (torch.mul, lambda t: 1),
    'maximum': (torch.maximum, lambda t: _dtype_range(t)[0]),
    'minimum': (torch.minimum, lambda t: _dtype_range(t)[1]),
    'bitwise_and': (torch.bitwise_and, _all_ones),
    'bitwise_or': (torch.bitwise_or, lambda t: 0),
    'bitwise_xor': (torch.bitwise_xor, lambda t: 0),
}
_ALIASES = {torch.add: 'add', torch.mul: 'multiply', torch.multiply: 'multiply',
            torch.maximum: 'maximum', torch.minimum: 'minimum',
            torch.bitwise_and: 'bitwise_and', torch.bitwise_or: 'bitwise_or',
            torch.bitwise_xor: 'bitwise_xor'}


def _resolve(op):
  name = op if isinstance(op, str) else _ALIASES.get(op)
  if name not in _MONOIDS:
    raise ValueError(f'unsupported scan operation {op!r}; expected one of '
                     f'{sorted(_MONOIDS)}')
  return name


def _flatten(x):
  """Leaves of a pytree (dict keys in sorted order) and the inverse map."""
  if isinstance(x, dict):
    keys = sorted(x)
    parts = [_flatten(x[k]) for k in keys]
    leaves = [l for p, _ in parts for l in p]

    def build(vals, parts=parts, keys=keys):
      out, at = {}, 0
      for k, (p, b) in zip(keys, parts):
        out[k] = b(vals[at:at + len(p)])
        at += len(p)
      return type(x)(out) if type(x) is not dict else out
    return leaves, build
  if isinstance(x, (list, tuple)):
    parts = [_flatten(v) for v in x]
    leaves = [l for p, _ in parts for l in p]

    def build(vals, parts=parts):
      out, at = [], 0
      for p, b in parts:
        out.append(b(vals[at:at + len(p)]))
        at += len(p)
      return type(x)(out)
    return leaves, build
  if not isinstance(x, torch.Tensor):
    raise TypeError(f'pscan operates on tensors, got {type(x)}')
  return [x], lambda vals: vals[0]


def _check(op, leaf):
  if op.startswith('bitwise') and leaf.dtype.is_floating_point:
    raise TypeError(f'{op} needs an integer or bool tensor, got {leaf.dtype}')
  if leaf.dtype.is_complex:
    raise TypeError('complex tensors are not supported')


def _gather_all(leaves, group):
  """(P,) + leaf.shape for every leaf: one all-gather per dtype group (bool
  travels as uint8; gloo and RCCL have no bool collectives)."""
  world = dist.get_world_size(group)
  out = [None] * len(leaves)
  by_dtype = {}
  for k, leaf in enumerate(leaves):
    by_dtype.setdefault((leaf.dtype, leaf.device), []).append(k)
  for (dtype, _), idx in by_dtype.items():
    wire = torch.uint8 if dtype == torch.bool else dtype
    flat = torch.cat([leaves[k].reshape(-1).to(wire) for k in idx])
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat.contiguous(), group=group)
    stacked = torch.stack(parts)                      # (P, total)
    at = 0
    for k in idx:
      n = leaves[k].numel()
      out[k] = stacked[:, at:at + n].reshape((world,) + tuple(
          leaves[k].shape)).to(dtype)
      at += n
  return out


def _scan_leaf(stacked, op, rank):
  """(exclusive scan at `rank`, reduction) of the (P, ...) copies."""
  fn, unit = _MONOIDS[op]
  acc = torch.full_like(stacked[0], unit(stacked.dtype))
  scan = acc
  for r in range(stacked.shape[0]):                   # left fold over the ranks
    if r == rank:
      scan = acc
    acc = fn(acc, stacked[r])
  return scan, acc


def pscan(x, op='add', axis_name=None, reduction: bool = False, group=None):
  """Exclusive prefix scan of `x` over the ranks (same structure as `x`).

  `axis_name` is accepted for signature parity with the reference (the mapped
  axis there; the process group plays that role here) and otherwise unused.
  With `reduction=True` returns `(scan, all_reduce)`.
  """
  op = _resolve(op)
  leaves, build = _flatten(x)
  for leaf in leaves:
    _check(op, leaf)
  if not leaves:
    return (x, x) if reduction else x
  rank = dist.get_rank(group)
  results = [_scan_leaf(s, op, rank) for s in _gather_all(leaves, group)]
  scan = build([r[0] for r in results])
  if reduction:
    return scan, build([r[1] for r in results])
  return scan


def preduce(x, op='add', axis_name=None, group=None):
  """All-reduce of `x` over the ranks with the monoid `op` (reference
  :272-296; 'add' / 'maximum' / 'minimum' map to the native collective)."""
  op = _resolve(op)
  leaves, build = _flatten(x)
  for leaf in leaves:
    _check(op, leaf)
  native = {'add': dist.ReduceOp.SUM, 'maximum': dist.ReduceOp.MAX,
            'minimum': dist.ReduceOp.MIN}
  if op in native and all(l.dtype != torch.bool for l in leaves):
    out = []
    for leaf in leaves:
      t = leaf.clone().contiguous()
      dist.all_reduce(t, op=native[op], group=group)
      out.append(t)
    return build(out)
  return pscan(x, op, reduction=True, group=group)[1]
