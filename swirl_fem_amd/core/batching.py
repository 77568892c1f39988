"""`vmap` over an ensemble for callers of the path.

The reference trains its closure model on an ensemble of trajectories by
`jax.vmap`-ing the gather / scatter and the whole solver step over a leading
batch axis (niles/train.py:232, :262-264).  The kernels of this build take one
field per call (an ensemble shares nothing but the operator data, and every
solve has its own iteration count), so the batch axis is walked on the host:
same results as a vmapped call, every item on the same stream, autograd
through each item (`linalg.cg.symmetric_solve`, `core/autodiff.py`).

For the solver step itself there is a batched path that needs no host loop:
`StokesSEM.ensemble(B)` (navier_stokes/navier_stokes.py) runs the B members as
one mesh of B copies -- operators launched once for all members, one CG
recurrence per member in the kernels (`linalg/cg_ensemble.py`) -- at a third
of this loop's cost for 8 members of the Kolmogorov generator, autograd
included.  This `vmap` stays for arbitrary functions of one member.
"""

from __future__ import annotations

import torch


def _map_tree(fn, tree):
  if isinstance(tree, (tuple, list)):
    return type(tree)(_map_tree(fn, t) for t in tree)
  if isinstance(tree, dict):
    return {k: _map_tree(fn, v) for k, v in tree.items()}
  return fn(tree)


def _leaves(tree):
  if isinstance(tree, (tuple, list)):
    return [x for t in tree for x in _leaves(t)]
  if isinstance(tree, dict):
    return [x for t in tree.values() for x in _leaves(t)]
  return [tree]


def vmap(fn, in_axes=0):
  """`jax.vmap(fn, in_axes)` for tensors and pytrees of tensors (tuples,
  lists, dicts): the mapped axis of every argument whose `in_axes` entry is 0
  is walked item by item and the results are stacked along a new leading axis
  (non-tensor leaves of the result -- iteration counts, status strings -- are
  returned as lists).  `in_axes`: 0, or one entry (0 / None) per argument.
  """
  def mapped(*args):
    axes = [in_axes] * len(args) if not isinstance(in_axes, (tuple, list)) \
        else list(in_axes)
    if len(axes) != len(args):
      raise ValueError(f'in_axes has {len(axes)} entries for {len(args)} '
                       'arguments')
    sizes = {int(leaf.shape[0]) for a, ax in zip(args, axes) if ax == 0
             for leaf in _leaves(a) if isinstance(leaf, torch.Tensor)}
    if len(sizes) != 1:
      raise ValueError(f'mapped axes have inconsistent sizes {sorted(sizes)}')
    outs = []
    for b in range(sizes.pop()):
      item = [a if ax is None else _map_tree(
          lambda t: t[b] if isinstance(t, torch.Tensor) else t, a)
              for a, ax in zip(args, axes)]
      outs.append(fn(*item))
    return _stack(outs)
  return mapped


def _stack(outs):
  first = outs[0]
  if isinstance(first, (tuple, list)):
    return type(first)(_stack([o[k] for o in outs]) for k in range(len(first)))
  if isinstance(first, dict):
    return {k: _stack([o[k] for o in outs]) for k in first}
  if isinstance(first, torch.Tensor):
    return torch.stack(outs)
  return list(outs)
