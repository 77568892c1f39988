"""Quadrature-point expressions: the build's stand-in for JAX tracing.

The reference evaluates a form such as
`lambda x: jnp.vdot(grad(u)(x), grad(v)(x))` under a double `vmap` over
quadrature points and elements and obtains the operator action with
`jax.linear_transpose` (core/fespace.py:121-167, :458-471).  There is no tracer
here.  Instead a q-function called on `x` returns a `QExpr` that carries the
values at *all* quadrature points of *all* elements at once:

  * a concrete `QExpr` wraps a device tensor `(E, Q) + shape`;
  * a linear `QExpr` (derived from the placeholder q-function, the one whose
    nodal values are `None`) records how to pull a cotangent back to the
    placeholder's value / gradient slots.

Pointwise arithmetic (`*`, `+`, indexing, `torch.vdot`, `torch.einsum`,
`torch.trace`, `torch.stack`, elementwise functions ...) works on both kinds
through `__torch_function__`, so forms are written with `torch.*` exactly where
the reference uses `jnp.*`.  Pulling back the quadrature weights through a
linear `QExpr` yields the coefficient fields `(c0, c1)` consumed by the
transposed basis kernel (`sfem_basis_eval_t`).
"""

from __future__ import annotations

import numbers
import re

import torch

_UNARY = {torch.sin, torch.cos, torch.tan, torch.exp, torch.log, torch.sqrt,
          torch.tanh, torch.abs, torch.square, torch.sinh, torch.cosh,
          torch.neg, torch.reciprocal, torch.sigmoid}


def _is_q(x):
  return isinstance(x, QExpr)


def _bshape(val, nb=2):
  return tuple(val.shape[nb:])


def _expand_spec(spec: str):
  """'i,ij->j'  ->  '...i,...ij->...j' (batch dims ride in the ellipsis)."""
  spec = spec.replace(' ', '')
  if '->' in spec:
    ins, out = spec.split('->')
  else:
    ins = spec
    letters = re.sub(r'[^a-zA-Z]', '', spec)
    out = ''.join(sorted(c for c in set(letters) if letters.count(c) == 1))
  ins = ins.split(',')
  return ['...' + s for s in ins], '...' + out


def _sum_to_shape(t, shape, nb=2):
  """Reduces broadcast value dims of `t (E,Q,*big)` down to `shape`."""
  extra = t.dim() - nb - len(shape)
  if extra > 0:
    t = t.sum(dim=tuple(range(nb, nb + extra)))
  for i, s in enumerate(shape):
    if s == 1 and t.shape[nb + i] != 1:
      t = t.sum(dim=nb + i, keepdim=True)
  return t


class QExpr:
  """Values (or a linear functional of the placeholder) at quadrature points."""

  __array_priority__ = 1000

  def __init__(self, val=None, shape=None, pullback=None):
    self.val = val                  # (E, Q) + shape, concrete only
    self.pullback = pullback        # ct (E,Q)+shape -> (c0, c1), linear only
    self.shape = tuple(shape if shape is not None else _bshape(val))

  # ----------------------------------------------------------------- helpers
  @property
  def is_linear(self):
    return self.pullback is not None

  @property
  def ndim(self):
    return len(self.shape)

  def __len__(self):
    return self.shape[0]

  def __iter__(self):
    for i in range(self.shape[0]):
      yield self[i]

  @staticmethod
  def _lift(other, like):
    """Constant (python number / plain tensor) -> broadcastable tensor."""
    if isinstance(other, numbers.Number):
      return other
    if isinstance(other, torch.Tensor):
      return other
    raise TypeError(type(other))

  # -------------------------------------------------------------- arithmetic
  def _mul_const(self, a, a_shape, batched):
    """self * a, `a` concrete: a number, a constant tensor of value shape
    `a_shape`, or (batched) a per-point tensor `(E, Q) + a_shape`."""
    if not self.is_linear:
      if batched:
        return QExpr(_align(self.val, self.shape, a_shape) *
                     _align(a, a_shape, self.shape))
      return QExpr(self.val * a)
    out_shape = tuple(torch.broadcast_shapes(self.shape, tuple(a_shape)))
    pb, my_shape = self.pullback, self.shape
    nd = len(out_shape)

    def pullback(ct):
      aa = a
      if batched:
        aa = a.reshape(a.shape[:2] + (1,) * (nd - len(a_shape)) +
                       tuple(a_shape))
      t = ct * aa
      pad = nd - len(my_shape)
      t = _sum_to_shape(t, (1,) * pad + tuple(my_shape))
      return pb(t.reshape(t.shape[:2] + tuple(my_shape)))

    return QExpr(shape=out_shape, pullback=pullback)

  def __mul__(self, other):
    if _is_q(other):
      if other.is_linear and self.is_linear:
        raise ValueError('form is not linear in the placeholder function')
      if other.is_linear:
        return other._mul_const(self.val, self.shape, True)
      return self._mul_const(other.val, other.shape, True)
    if isinstance(other, torch.Tensor) and other.dim() > 0:
      return self._mul_const(other, tuple(other.shape), False)
    return self._mul_const(other, (), False)

  __rmul__ = __mul__

  def __truediv__(self, other):
    if _is_q(other):
      if other.is_linear:
        raise ValueError('cannot divide by the placeholder function')
      return self * QExpr(1.0 / other.val)
    return self * (1.0 / other)

  def __rtruediv__(self, other):
    if self.is_linear:
      raise ValueError('cannot divide by the placeholder function')
    return QExpr(other / self.val)

  def __neg__(self):
    return self * (-1.0)

  def __add__(self, other):
    if isinstance(other, numbers.Number) and other == 0:
      return self
    if not _is_q(other):
      if self.is_linear:
        raise ValueError('affine (non-linear) form in the placeholder')
      return QExpr(self.val + other)
    if self.is_linear != other.is_linear:
      raise ValueError('affine (non-linear) form in the placeholder')
    if not self.is_linear:
      a, b = _align_pair(self, other)
      return QExpr(a + b)
    if self.shape != other.shape:
      raise ValueError(f'shape mismatch {self.shape} vs {other.shape}')
    pa, pb = self.pullback, other.pullback

    def pullback(ct):
      return _add_pairs(pa(ct), pb(ct))

    return QExpr(shape=self.shape, pullback=pullback)

  __radd__ = __add__

  def __sub__(self, other):
    return self + (-other if _is_q(other) else -other)

  def __rsub__(self, other):
    return (-self) + other

  def __pow__(self, k):
    if self.is_linear:
      raise ValueError('form is not linear in the placeholder function')
    return QExpr(self.val ** k)

  def __getitem__(self, idx):
    if not isinstance(idx, tuple):
      idx = (idx,)
    full = (slice(None), slice(None)) + idx
    if not self.is_linear:
      return QExpr(self.val[full])
    pb, my_shape = self.pullback, self.shape
    out_shape = tuple(torch.empty(my_shape)[idx].shape)

    def pullback(ct):
      z = ct.new_zeros(ct.shape[:2] + tuple(my_shape))
      z[full] = ct
      return pb(z)

    return QExpr(shape=out_shape, pullback=pullback)

  def sum(self):
    """Sum over the value dims (per quadrature point)."""
    if not self.is_linear:
      if not self.shape:
        return self
      return QExpr(self.val.sum(dim=tuple(range(2, 2 + len(self.shape)))))
    pb, my_shape = self.pullback, self.shape

    def pullback(ct):
      ct = ct.reshape(ct.shape[:2] + (1,) * len(my_shape))
      return pb(ct.expand(ct.shape[:2] + tuple(my_shape)))

    return QExpr(shape=(), pullback=pullback)

  # -------------------------------------------------------- torch dispatching
  @classmethod
  def __torch_function__(cls, func, types, args=(), kwargs=None):
    kwargs = kwargs or {}
    if func in (torch.vdot, torch.dot, torch.inner):
      return vdot(args[0], args[1])
    if func is torch.einsum:
      return einsum(args[0], *args[1:])
    if func is torch.trace:
      return trace(args[0])
    if func is torch.stack:
      return stack(list(args[0]))
    if func is torch.sum:
      return args[0].sum()
    if func in (torch.mul, torch.multiply):
      return _as_q(args[0]) * args[1] if _is_q(args[0]) else args[1] * args[0]
    if func is torch.add:
      return args[0] + args[1]
    if func in (torch.sub, torch.subtract):
      return args[0] - args[1]
    if func in (torch.div, torch.true_divide):
      return args[0] / args[1]
    if func is torch.pow:
      return args[0] ** args[1]
    if func in _UNARY:
      x = args[0]
      if x.is_linear:
        raise ValueError('form is not linear in the placeholder function')
      return QExpr(func(x.val))
    return NotImplemented


def _as_q(x):
  return x


def _align(val, shape, other_shape):
  """Reshapes concrete `val (E,Q)+shape` for broadcasting against other."""
  nd = max(len(shape), len(other_shape))
  return val.reshape(val.shape[:2] + (1,) * (nd - len(shape)) + tuple(shape))


def _align_pair(a: QExpr, b: QExpr):
  return _align(a.val, a.shape, b.shape), _align(b.val, b.shape, a.shape)


def _add_pairs(p, q):
  out = []
  for a, b in zip(p, q):
    out.append(b if a is None else (a if b is None else a + b))
  return tuple(out)


def _pointwise_einsum(spec, *vals):
  """`torch.einsum` for batched *tiny* contractions (value dims <= 4, batch =
  elements x quadrature points).  torch lowers those to batched GEMMs with
  3x3 matrices, which run far below HBM speed; here the operands are broadcast
  over the joint index space and reduced elementwise instead."""
  ins, out = spec.split('->')
  ins = ins.split(',')
  letters = []
  for s_ in ins:
    for c in s_[3:]:
      if c not in letters:
        letters.append(c)
  sizes = {}
  for s_, v in zip(ins, vals):
    for c, n in zip(s_[3:], v.shape[v.dim() - len(s_[3:]):]):
      sizes[c] = n
  small = all(n <= 4 for n in sizes.values())
  joint = 1
  for n in sizes.values():
    joint *= n
  same_batch = all(
      v.shape[:v.dim() - len(s_[3:])] == vals[0].shape[:vals[0].dim() -
                                                       len(ins[0][3:])]
      for s_, v in zip(ins, vals))
  if (len(vals) < 2 or not small or joint > 64 or not same_batch or
      any(len(set(s_[3:])) != len(s_[3:]) for s_ in ins)):
    return torch.einsum(spec, *vals)
  nb = vals[0].dim() - len(ins[0][3:])
  prod = None
  for s_, v in zip(ins, vals):
    own = s_[3:]
    order = [own.index(c) for c in letters if c in own]
    v = v.permute(*range(nb), *[nb + k for k in order])
    shape = list(v.shape[:nb]) + [sizes[c] if c in own else 1 for c in letters]
    v = v.reshape(shape)
    prod = v if prod is None else prod * v
  keep = out[3:]
  red = [nb + k for k, c in enumerate(letters) if c not in keep]
  if red:
    prod = prod.sum(dim=red)
  left = [c for c in letters if c in keep]
  prod = prod.expand(*prod.shape[:nb], *[sizes[c] for c in left])
  return prod.permute(*range(nb), *[nb + left.index(c) for c in keep])


# ----------------------------------------------------------------- functions
def einsum(spec, *operands):
  """Pointwise einsum over value dims; at most one linear operand."""
  ins, out = _expand_spec(spec)
  lin = [i for i, o in enumerate(operands) if _is_q(o) and o.is_linear]
  if len(lin) > 1:
    raise ValueError('form is not linear in the placeholder function')
  vals = [o.val if _is_q(o) else o for o in operands]
  if not lin:
    return QExpr(_pointwise_einsum(','.join(ins) + '->' + out, *vals))
  p = lin[0]
  op = operands[p]
  others_spec = [s for i, s in enumerate(ins) if i != p]
  others = [v for i, v in enumerate(vals) if i != p]
  # every index of the linear operand must survive somewhere to be transposed
  rest = ''.join(others_spec) + out
  missing = [c for c in ins[p][3:] if c not in rest]
  if missing:
    raise NotImplementedError(
        f'einsum index {missing} of the placeholder operand is summed alone')
  t_spec = ','.join([out] + others_spec) + '->' + ins[p]
  out_letters = out[3:]
  dims = {}
  for s, o in zip(ins, operands):
    shp = o.shape if _is_q(o) else tuple(o.shape)
    for c, n in zip(s[3:], shp[len(shp) - len(s[3:]):]):
      dims[c] = n
  out_shape = tuple(dims[c] for c in out_letters)
  pb = op.pullback

  def pullback(ct):
    return pb(_pointwise_einsum(t_spec, ct, *others))

  return QExpr(shape=out_shape, pullback=pullback)


def vdot(a, b):
  """Sum of elementwise products over all value dims (jnp.vdot semantics)."""
  if not _is_q(a) and not _is_q(b):
    return torch.vdot(a, b)
  return (a * b).sum() if _is_q(a) else (b * a).sum()


def trace(a: QExpr):
  if not a.is_linear:
    return QExpr(torch.diagonal(a.val, dim1=-2, dim2=-1).sum(-1))
  pb, shp = a.pullback, a.shape

  def pullback(ct):
    eye = torch.eye(shp[-1], dtype=ct.dtype, device=ct.device)
    return pb(ct[..., None, None] * eye)

  return QExpr(shape=(), pullback=pullback)


def stack(items):
  """Stacks per-point values along a new leading value axis."""
  items = list(items)
  if any(_is_q(i) and i.is_linear for i in items):
    raise NotImplementedError('stack of placeholder-dependent values')
  ref = next(i for i in items if _is_q(i))
  vals = []
  for it in items:
    if _is_q(it):
      vals.append(it.val)
    else:
      vals.append(torch.as_tensor(it, dtype=ref.val.dtype,
                                  device=ref.val.device).expand(ref.val.shape))
  return QExpr(torch.stack(vals, dim=2))
