"""Autograd through the operators and through a Navier-Stokes step.

The reference gets this from JAX (linear_transpose of the local operators,
custom_linear_solve(symmetric=True) for both solves, navier_stokes.py:436-452);
niles/train.py:227-293 differentiates a rolled-out solver with respect to a
learned forcing.  Here every HIP primitive carries its transpose
(core/autodiff.py): checked against finite differences (fp64) and against the
fused forward kernels.
"""
import numpy as np
import pytest
import torch

from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
from swirl_fem_amd.navier_stokes.navier_stokes import BCType, StokesSEM

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _sem(ndim=2, n=2, order=4, periodic=(), bcs=None):
  pm = unit_cube_mesh(n, ndim=ndim, periodic_dims=periodic)
  if bcs is None:
    bcs = {} if len(periodic) == ndim else {
        'boundary': (BCType.DIRICHLET, 0.0)}
  return StokesSEM.create(pm, bcs, order=order, device=DEV)


def _rand(shape, seed, grad=False):
  g = torch.Generator(device=DEV).manual_seed(seed)
  t = torch.randn(shape, dtype=torch.float64, device=DEV, generator=g)
  return t.requires_grad_(grad)


def _directional(fn, x, direction, eps=1e-6):
  with torch.no_grad():
    return (fn(x + eps * direction) - fn(x - eps * direction)) / (2 * eps)


@pytest.mark.parametrize('ndim,periodic', [(2, ()), (2, (0, 1)), (3, (2,))])
def test_operator_cotangents_match_finite_differences(ndim, periodic):
  sem = _sem(ndim=ndim, periodic=periodic, order=4 if ndim == 2 else 3)
  Nv = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  ops = {
      'A': (sem.A, (Nv, ndim)), 'B': (sem.B, (Nv, ndim)),
      'Bi': (sem.Bi, (Nv, ndim)), 'C': (sem.C, (Nv, ndim)),
      'D': (sem.D, (Nv, ndim)), 'Dt': (sem.Dt, (Np,)),
      'H': (lambda u: sem.H(u, 3.0, 0.1), (Nv, ndim)),
      'E': (lambda p: sem.E(p, dt=1e-2, time_order=2), (Np,)),
      'filter': (lambda u: sem.filter(u, alpha=0.3), (Nv, ndim)),
  }
  for k, (name, (op, shape)) in enumerate(ops.items()):
    x = _rand(shape, 10 + k, grad=True)
    y = op(x)
    w = _rand(tuple(y.shape), 50 + k)
    (g,) = torch.autograd.grad((y * w).sum(), x)
    # the differentiable composition equals the fused forward kernels
    with torch.no_grad():
      y_fused = op(x.detach())
    assert float((y.detach() - y_fused).abs().max()) <= 1e-10 * max(
        1.0, float(y_fused.abs().max())), name
    for s in range(3):
      d = _rand(shape, 100 + 7 * k + s)
      fd = float(_directional(lambda t: (op(t) * w).sum(), x.detach(), d,
                              eps=1e-5 if name == 'C' else 1e-3))
      an = float((g * d).sum())
      assert abs(fd - an) <= 1e-7 * max(1.0, abs(an)), (name, fd, an)


def test_linear_operators_are_transposed_exactly():
  """<D u, q> = <u, D^T q>, A and E symmetric: with autograd on one side and
  the fused kernels on the other."""
  sem = _sem(periodic=(0, 1))
  Nv = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  u, q = _rand((Nv, 2), 1, grad=True), _rand((Np,), 2)
  (g,) = torch.autograd.grad((sem.D(u) * q).sum(), u)
  with torch.no_grad():
    want = sem.Dt(q)            # fused D^T (masked: nothing to mask here)
  assert float((g - want).abs().max()) < 1e-12 * float(want.abs().max())
  v = _rand((Nv, 2), 3)
  (g,) = torch.autograd.grad((sem.A(u) * v).sum(), u)
  with torch.no_grad():
    want = sem.A(v)
  assert float((g - want).abs().max()) < 1e-11 * float(want.abs().max())


@pytest.mark.parametrize('periodic', [(0, 1), ()])
def test_gradient_through_a_navier_stokes_step(periodic):
  """d loss / d forcing and d loss / d u_prev through `navier_stokes_step`
  (extrapolated convection, Helmholtz solve, filter, pressure solve,
  projection) against central differences."""
  sem = _sem(n=2, order=4, periodic=periodic)
  x = sem.velocity.mesh.node_coords
  Nv = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  mask = sem.velocity.interior_mask
  two_pi = 2 * np.pi
  u0 = mask * torch.stack([torch.sin(two_pi * x[:, 0]) * torch.cos(two_pi * x[:, 1]),
                           -torch.cos(two_pi * x[:, 0]) * torch.sin(two_pi * x[:, 1])],
                          dim=-1)
  p0 = torch.zeros(Np, dtype=torch.float64, device=DEV)
  w = _rand((Nv, 2), 5)
  kw = dict(reynolds=50.0, dt=1e-2, time_order=2, tol=1e-13, atol=0.0)

  def loss(forcing, u_prev):
    us, ps = (u0, u_prev), (p0, p0)
    Cus = tuple(sem.C(u) for u in us)
    u, p, Cu, _ = navier_stokes_step(sem, us, ps, Cus, forcing=forcing, **kw)
    u2, _, _, _ = navier_stokes_step(sem, (u_prev, u), (p0, p), (Cus[1], Cu),
                                     forcing=forcing, **kw)
    return (u2 * w).sum() + 0.5 * (u * u).sum()

  f = (0.3 * _rand((Nv, 2), 6)).requires_grad_(True)
  up = (u0 + 0.05 * mask * _rand((Nv, 2), 7)).requires_grad_(True)
  gf, gu = torch.autograd.grad(loss(f, up), (f, up))
  assert float(gf.abs().max()) > 0 and float(gu.abs().max()) > 0
  for s in range(3):
    d = _rand((Nv, 2), 20 + s)
    fd = float(_directional(lambda t: loss(t, up.detach()), f.detach(), d,
                            eps=1e-4))
    an = float((gf * d).sum())
    assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), ('forcing', fd, an)
    d = mask * _rand((Nv, 2), 30 + s)
    fd = float(_directional(lambda t: loss(f.detach(), t), up.detach(), d,
                            eps=1e-5))
    an = float((gu * d).sum())
    assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), ('u_prev', fd, an)


def test_gather_scatter_exchange_rules():
  from swirl_fem_amd.core import autodiff
  sem = _sem(periodic=(0, 1))
  mesh = sem.velocity.mesh
  u = _rand((mesh.num_nodes,), 1, grad=True)
  ul = mesh.gather(u)
  assert ul.requires_grad
  w = _rand(tuple(ul.shape), 2)
  (g,) = torch.autograd.grad((ul * w).sum(), u)
  with torch.no_grad():
    assert torch.allclose(g, mesh.scatter(w), rtol=0, atol=1e-13)
  wl = _rand(tuple(mesh.elements.shape), 3, grad=True)
  v = _rand((mesh.num_nodes,), 4)
  (g,) = torch.autograd.grad((mesh.scatter(wl) * v).sum(), wl)
  with torch.no_grad():
    assert torch.allclose(g, mesh.gather(v), rtol=0, atol=0)
  (g,) = torch.autograd.grad((mesh.exchange(u) * v).sum(), u)
  with torch.no_grad():
    assert torch.allclose(g, mesh.exchange(v), rtol=0, atol=1e-13)
  assert not autodiff.needs_grad(u.detach())


def test_linear_operators_stay_on_the_fused_kernels_under_autograd(monkeypatch):
  """A, B_local-based pieces, H, D, D^T and E differentiate through the FUSED
  kernels (`core/autodiff.py`: each carries its transpose); only the quadratic
  convection term needs the generic q-function path.  The generic path is
  switched off here, so a cotangent that still arrives came from the fused
  rules; it must equal the one of the generic path."""
  sem = _sem(ndim=3, n=2, order=3)
  Nv = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  ops = {'A': (sem.A, (Nv, 3)), 'H': (lambda u: sem.H(u, 2.0, 0.3), (Nv, 3)),
         'D': (sem.D, (Nv, 3)), 'Dt': (sem.Dt, (Np,)),
         'E': (lambda p: sem.E(p, dt=1e-2, time_order=2), (Np,)),
         'A_local': (sem.velocity.A_local, (sem.velocity.mesh.num_elements,
                                            64, 3))}
  generic = {}
  from swirl_fem_amd.core import autodiff
  real_needs_grad = autodiff.needs_grad
  for k, (name, (op, shape)) in enumerate(ops.items()):
    x = _rand(shape, 300 + k, grad=True)
    w = _rand(tuple(op(x.detach()).shape), 400 + k)
    generic[name] = (x, w)
  fused_grads = {}
  for name, (op, _) in ops.items():
    x, w = generic[name]
    (fused_grads[name],) = torch.autograd.grad((op(x) * w).sum(), x)

  def boom(*a, **k):
    raise AssertionError('generic q-function path used')
  for space in (sem.velocity.vspace, sem.pressure.pspace):
    monkeypatch.setattr(type(space), 'local_covector', boom)
  for name, (op, _) in ops.items():
    x, w = generic[name]
    (g,) = torch.autograd.grad((op(x) * w).sum(), x)   # no generic call
    # (same kernels both times; the atomics may sum in another order)
    assert float((g - fused_grads[name]).abs().max()) <= 1e-12 * float(
        g.abs().max())
    # transpose check: <w, op(v)> == <g, v> for a linear op
    v = _rand(tuple(x.shape), 500)
    with torch.no_grad():
      lhs = float((op(v) * w).sum())
    rhs = float((g * v).sum())
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), 1.0), name


def test_vmap_over_an_ensemble_matches_item_by_item():
  """The reference vmaps gather / scatter and the solver step over an ensemble
  (niles/train.py:232, :262-264); `core.batching.vmap` gives the same call
  shape: stacked results equal the item-by-item calls, gradients flow to a
  batched forcing through every item's solves."""
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core.batching import vmap
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
  sem = StokesSEM.create(unit_cube_mesh(3, ndim=2, periodic_dims=(0, 1)), {},
                         order=4, device=DEV)
  mesh = sem.velocity.mesh
  g = torch.Generator(device=DEV).manual_seed(3)
  B = 3
  u0 = 0.1 * torch.randn(B, mesh.num_nodes, 2, dtype=torch.float64, device=DEV,
                         generator=g)
  u0 = vmap(sem.velocity.exchange)(u0) / vmap(sem.velocity.exchange)(
      torch.ones_like(u0))
  # gather / scatter over the ensemble
  loc = vmap(sem.velocity.gather)(u0)
  assert loc.shape == (B, mesh.num_elements, mesh.num_nodes_per_element, 2)
  assert torch.equal(loc[1], sem.velocity.gather(u0[1]))
  back = vmap(sem.velocity.scatter)(loc)
  assert torch.equal(back[2], sem.velocity.scatter(loc[2]))
  p0 = torch.zeros(B, sem.pressure.pspace.mesh.num_nodes, dtype=torch.float64,
                   device=DEV)
  forcing = (0.05 * torch.randn(B, mesh.num_nodes, 2, dtype=torch.float64,
                                device=DEV, generator=g)).requires_grad_(True)

  def one(u, p, f):
    us, ps = (u, u), (p, p)
    cu = sem.C(u)
    un, pn, cn, aux = navier_stokes_step(
        sem, us, ps, (cu, cu), reynolds=50.0, dt=1e-2, time_order=2,
        forcing=f, tol=1e-12, atol=0.0)
    return un, pn, aux['u_star_info']['num_iterations']

  un, pn, iters = vmap(one)(u0, p0, forcing)
  assert un.shape == u0.shape and pn.shape == p0.shape and len(iters) == B
  for b in range(B):
    ub, pb, _ = one(u0[b], p0[b], forcing[b].detach())
    assert float((un[b] - ub).abs().max()) <= 1e-12 * float(ub.abs().max())
    assert float((pn[b] - pb).abs().max()) <= 1e-10 * max(
        1.0, float(pb.abs().max()))
  # d/d forcing of a functional of the ensemble: item b only sees forcing[b]
  w = torch.randn(un.shape, dtype=un.dtype, device=DEV, generator=g)
  (un * w).sum().backward()
  gb = forcing.grad
  f2 = forcing.detach()[1].clone().requires_grad_(True)
  u1, _, _ = one(u0[1], p0[1], f2)
  (u1 * w[1]).sum().backward()
  assert float((gb[1] - f2.grad).abs().max()) <= 1e-10 * float(
      f2.grad.abs().max())
  # in_axes = None keeps an argument whole
  same = vmap(lambda u, s: s * u, in_axes=(0, None))(u0, torch.tensor(
      2.0, dtype=torch.float64, device=DEV))
  assert torch.equal(same, 2.0 * u0)
