"""Ensembles on the device (the reference's `jax.vmap` of its solver step,
niles/train.py:232, :262-264): B members as B disjoint copies of the mesh,
operators launched once for all, one CG recurrence per member
(`StokesSEM.ensemble`, `linalg/cg_ensemble.py`, csrc/sfem_cg_ensemble.hip).
Every member must equal its own single solve."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device('cuda', 0)


def _sem(n=4, order=5, periodic=(0, 1), bcs=None):
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
  return StokesSEM.create(unit_cube_mesh(n, ndim=2, periodic_dims=periodic),
                          bcs or {}, order=order, device=DEV)


def _fields(sem, B, seed):
  g = torch.Generator(device=DEV).manual_seed(seed)
  N = sem.velocity.mesh.num_nodes
  u = 0.1 * torch.randn(B, N, 2, dtype=torch.float64, device=DEV, generator=g)
  # consistent on periodic images
  ex = lambda t: torch.stack([sem.velocity.exchange(v) for v in t])
  return ex(u) / ex(torch.ones_like(u)), g


def test_operators_of_an_ensemble_are_the_members_operators():
  sem = _sem()
  B = 3
  ens = sem.ensemble(B)
  assert ens.members == B and sem.ensemble(1) is sem
  u, g = _fields(sem, B, 1)
  uf = ens.flatten(u)
  assert uf.shape == (B * sem.velocity.mesh.num_nodes, 2)
  assert torch.equal(ens.unflatten(uf), u)
  Np = sem.pressure.pspace.mesh.num_nodes
  p = torch.randn(B, Np, dtype=torch.float64, device=DEV, generator=g)
  for name, batched, fn_e, fn_s in [
      ('C', u, ens.C, sem.C), ('D', u, ens.D, sem.D), ('B', u, ens.B, sem.B),
      ('Dt', p, ens.Dt, sem.Dt), ('filter', u, ens.filter, sem.filter),
      ('H', u, lambda v: ens.H(v, 30.0, 0.02), lambda v: sem.H(v, 30.0, 0.02)),
      ('E', p, lambda v: ens.E(v, dt=1e-2, time_order=2),
       lambda v: sem.E(v, dt=1e-2, time_order=2))]:
    got = ens.unflatten(fn_e(ens.flatten(batched)))
    for b in range(B):
      want = fn_s(batched[b])
      scale = max(float(want.abs().max()), 1e-30)
      assert float((got[b] - want).abs().max()) <= 1e-12 * scale, (name, b)


@pytest.mark.parametrize('graph', [False, True])
def test_ensemble_cg_is_one_recurrence_per_member(graph):
  """Members of very different difficulty: each stops at its own iteration
  with its own single-solve iterate."""
  from swirl_fem_amd.linalg.cg import cg
  from swirl_fem_amd.linalg.cg_ensemble import cg_ensemble
  sem = _sem(n=5, order=6)
  B = 4
  ens = sem.ensemble(B)
  u, _ = _fields(sem, B, 2)
  rhs = torch.stack([sem.B(v) for v in u])
  rhs[1] *= 1e-3                      # relative stop: same count, scaled x
  rhs[2] = 0.0                        # converged before the first iteration
  rhs[3] = sem.B(torch.ones_like(u[3]))        # smooth: a handful of iterations
  H_e = lambda v: ens.H(v, 25.0, 0.01)
  H_s = lambda v: sem.H(v, 25.0, 0.01)
  x, info = cg_ensemble(H_e, ens.flatten(rhs), B, M=ens.velocity.exchange,
                        tol=1e-10, graph=graph, check_every=4)
  x = ens.unflatten(x)
  assert info['status'] == 'converged'
  counts = []
  for b in range(B):
    xb, ib = cg(H_s, rhs[b], M=sem.velocity.exchange, tol=1e-10)
    counts.append(ib['num_iterations'])
    assert abs(info['member_iterations'][b] - ib['num_iterations']) <= 1, b
    scale = max(float(xb.abs().max()), 1e-30)
    assert float((x[b] - xb).abs().max()) <= 1e-9 * scale, b
  assert info['member_iterations'][2] == 0
  assert len(set(counts)) > 1         # the members did stop at different counts
  assert info['num_iterations'] == max(info['member_iterations'])
  # a second solve (the inner products are fixed-order sums; the operator's
  # scatter still adds in arrival order)
  x2, info2 = cg_ensemble(H_e, ens.flatten(rhs), B, M=ens.velocity.exchange,
                          tol=1e-10, graph=graph, check_every=4)
  assert float((ens.unflatten(x2) - x).abs().max()) <= 1e-11 * float(
      x.abs().max())
  assert all(abs(a - b) <= 1 for a, b in zip(info2['member_iterations'],
                                             info['member_iterations']))


@pytest.mark.parametrize('pc', [None, 'unfused', 'schwarz'])
def test_ensemble_step_equals_the_members_steps(pc, monkeypatch):
  """Three flows, three steps of the Kolmogorov generator's step: every
  member of the ensemble step equals its own single step (default: the mean
  projection of the pressure solve folded into the vector updates, per member;
  'unfused': z = M r stored)."""
  if pc == 'unfused':
    monkeypatch.setenv('SFEM_FUSED_MEAN', '0')
    pc = None
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  from swirl_fem_amd.niles.datagen import datagen
  sem = _sem(n=6, order=5)
  B = 3
  ens = sem.ensemble(B)
  x = sem.velocity.mesh.node_coords
  amp = [1.0, 0.4, 1.7]
  u0 = torch.stack([a * datagen.u_init_fn(x) for a in amp])
  Np = sem.pressure.pspace.mesh.num_nodes
  p0 = torch.zeros(B, Np, dtype=torch.float64, device=DEV)
  kw = dict(reynolds=200.0, dt=2e-3, time_order=2, tol=1e-11, atol=0.0,
            pressure_preconditioner=pc)

  def run(s, u, p, steps=3):
    us, ps = (u, u), (p, p)
    c = s.C(u)
    Cus = (c, c)
    its = []
    for _ in range(steps):
      f = datagen.forcing(s.velocity.mesh.node_coords, us[-1], 0.1)
      un, pn, cn, aux = navier_stokes_step(s, us, ps, Cus, forcing=f, **kw)
      us, ps, Cus = us[1:] + (un,), ps[1:] + (pn,), Cus[1:] + (cn,)
      its.append(aux)
    return us[-1], ps[-1], its

  ue, pe, aux_e = run(ens, ens.flatten(u0), ens.flatten(p0))
  ue, pe = ens.unflatten(ue), ens.unflatten(pe)
  for b in range(B):
    ub, pb, aux_b = run(sem, u0[b], p0[b])
    assert float((ue[b] - ub).abs().max()) <= 1e-9 * float(ub.abs().max()), b
    assert float((pe[b] - pb).abs().max()) <= 1e-8 * max(
        1.0, float(pb.abs().max())), b
    for k in range(3):
      for which in ('u_star_info', 'dp_info'):
        got = aux_e[k][which]['member_iterations'][b]
        want = aux_b[k][which]['num_iterations']
        assert abs(got - want) <= 2, (b, k, which, got, want)


def test_ensemble_with_dirichlet_walls():
  """The lid-driven cavity as an ensemble of two lids."""
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  from swirl_fem_amd.navier_stokes.navier_stokes import BCType
  sem = _sem(n=4, order=5, periodic=(),
             bcs={'boundary': (BCType.DIRICHLET, 0.0)})
  B = 2
  ens = sem.ensemble(B)
  x = sem.velocity.mesh.node_coords
  lid = (x[:, 1] > 1.0 - 1e-12).to(x.dtype)
  prof = torch.stack([lid * 16 * x[:, 0] ** 2 * (1 - x[:, 0]) ** 2,
                      torch.zeros_like(lid)], dim=-1)
  ub = torch.stack([prof, -0.5 * prof])
  Np = sem.pressure.pspace.mesh.num_nodes
  p0 = torch.zeros(B, Np, dtype=torch.float64, device=DEV)
  kw = dict(reynolds=100.0, dt=1e-3, time_order=2, tol=1e-11, atol=0.0)

  def one(s, u, p, u_b):
    c = s.C(u)
    return navier_stokes_step(s, (u, u), (p, p), (c, c), u_boundary=u_b,
                              **kw)[:2]

  ue, pe = one(ens, ens.flatten(ub), ens.flatten(p0), ens.flatten(ub))
  ue, pe = ens.unflatten(ue), ens.unflatten(pe)
  for b in range(B):
    u1, p1 = one(sem, ub[b], p0[b], ub[b])
    assert float((ue[b] - u1).abs().max()) <= 1e-9 * float(u1.abs().max())
    assert float((pe[b] - p1).abs().max()) <= 1e-8 * max(
        1.0, float(p1.abs().max()))


def test_gradients_through_an_ensemble_step():
  """d/d forcing of a functional of the ensemble step: member b only sees
  forcing[b], and its gradient is the single step's (the reference vmaps a
  differentiable step, niles/train.py:262-264)."""
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  sem = _sem(n=3, order=4)
  B = 3
  ens = sem.ensemble(B)
  u0, g = _fields(sem, B, 5)
  N = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  p0 = torch.zeros(B, Np, dtype=torch.float64, device=DEV)
  forcing = 0.05 * torch.randn(B, N, 2, dtype=torch.float64, device=DEV,
                               generator=g)
  w = torch.randn(B, N, 2, dtype=torch.float64, device=DEV, generator=g)
  kw = dict(reynolds=50.0, dt=1e-2, time_order=2, tol=1e-12, atol=0.0)

  def one(s, u, p, f):
    cu = s.C(u)
    return navier_stokes_step(s, (u, u), (p, p), (cu, cu), forcing=f, **kw)[0]

  fe = forcing.clone().requires_grad_(True)
  ue = ens.unflatten(one(ens, ens.flatten(u0), ens.flatten(p0),
                         ens.flatten(fe)))
  (ue * w).sum().backward()
  for b in range(B):
    fb = forcing[b].clone().requires_grad_(True)
    ub = one(sem, u0[b], p0[b], fb)
    assert float((ue[b].detach() - ub.detach()).abs().max()) <= 1e-10 * float(
        ub.detach().abs().max())
    (ub * w[b]).sum().backward()
    assert float((fe.grad[b] - fb.grad).abs().max()) <= 1e-9 * float(
        fb.grad.abs().max()), b


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_ensemble_kernels_against_a_host_recurrence(dtype):
  """The `sfem_ens_*` kernels on diagonal systems (A = diag(d), the same for
  every member) against the recurrence of linalg/cg.py:60-97 written out on the
  host in float64: per-member alpha, beta, stop test and counts; float32
  vectors with float64 scalars."""
  from swirl_fem_amd.linalg.cg_ensemble import cg_ensemble
  rng = np.random.default_rng(7)
  B, n = 5, 3001                      # (not a multiple of the group count)
  d = np.linspace(1.0, 50.0, n) ** 1.5
  rhs = rng.standard_normal((B, n))
  rhs[1] *= 1e4
  rhs[3] = 0.0
  rhs[4, 10:] = 0.0                   # few distinct eigenvalues: stops early
  tol = 1e-9 if dtype == torch.float64 else 1e-4
  dd = torch.as_tensor(d, dtype=dtype, device=DEV).repeat(B)
  b = torch.as_tensor(rhs, dtype=dtype, device=DEV).reshape(-1)
  x, info = cg_ensemble(lambda v: dd * v, b, B, tol=tol, maxiter=400,
                        check_every=7)
  x = x.reshape(B, n).double().cpu().numpy()
  d32 = dd[:n].double().cpu().numpy()
  r32 = b.reshape(B, n).double().cpu().numpy()
  for m in range(B):
    xs = np.zeros(n)
    r = r32[m].copy()
    p = r.copy()
    gamma = r @ r
    thresh = tol * tol * (r32[m] @ r32[m])
    k = 0
    while gamma > thresh and k < 400:
      ap = d32 * p
      alpha = gamma / (p @ ap)
      xs += alpha * p
      r -= alpha * ap
      gnew = r @ r
      p = r + (gnew / gamma) * p
      gamma = gnew
      k += 1
    slack = 0 if dtype == torch.float64 else 2
    assert abs(info['member_iterations'][m] - k) <= slack, (m, k, info)
    scale = max(np.abs(xs).max(), 1e-300)
    err = np.abs(x[m] - xs).max() / scale
    assert err <= (1e-10 if dtype == torch.float64 else 2e-3), (m, err)
  assert info['member_iterations'][3] == 0
  assert info['member_status'] == ['converged'] * B
  # maxiter: every running member stops there and says so
  x2, info2 = cg_ensemble(lambda v: dd * v, b, B, tol=1e-30, maxiter=5)
  assert info2['member_iterations'][0] == 5
  assert info2['member_status'][0] == 'maxiter'
  assert info2['member_status'][3] == 'converged'
  assert info2['status'] == 'maxiter'


@pytest.mark.parametrize('case', ['periodic', 'walls', 'jittered3d'])
def test_layered_pressure_operator_equals_the_atomic_one(case, monkeypatch):
  """`StokesDivGrad.e_layered` (D^T with a position per writer, sums by the
  class kernel, no atomics) against the default E, on the meshes that run the
  index-row kernels: 2D periodic, 2D with Dirichlet walls, 3D at an order
  outside the facet kernels."""
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.navier_stokes.navier_stokes import BCType, StokesSEM
  if case == 'periodic':
    sem = _sem(n=5, order=6)
  elif case == 'walls':
    sem = _sem(n=4, order=5, periodic=(),
               bcs={'boundary': (BCType.DIRICHLET, 0.0)})
  else:
    sem = StokesSEM.create(unit_cube_mesh(3, ndim=3, periodic_dims=(0,)),
                           {'boundary': (BCType.DIRICHLET, 0.0)}, order=4,
                           device=DEV)
  op = sem._divgrad()
  assert op is not None and op.supports_layered_e()
  g = torch.Generator(device=DEV).manual_seed(11)
  Np = sem.pressure.pspace.mesh.num_nodes
  p = torch.randn(Np, dtype=torch.float64, device=DEV, generator=g)
  monkeypatch.setenv('SFEM_STOKES_LAYERED', '0')
  want = sem.E(p, dt=1e-2, time_order=2)
  monkeypatch.setenv('SFEM_STOKES_LAYERED', '1')
  got = sem.E(p, dt=1e-2, time_order=2)
  again = sem.E(p, dt=1e-2, time_order=2)
  scale = float(want.abs().max())
  assert float((got - want).abs().max()) <= 1e-12 * scale
  assert torch.equal(got, again)          # fixed order of the sums
  # with the fused p . E p
  from swirl_fem_amd import _lib
  parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device=DEV)
  sem.E(p, dt=1e-2, time_order=2, dot_out=parts)
  assert abs(float(parts.sum()) - float(p @ want)) <= 1e-11 * abs(
      float(p @ want))
  # an ensemble of this mesh
  ens = sem.ensemble(2)
  pe = torch.cat([p, -0.5 * p])
  ge = ens.unflatten(ens.E(pe, dt=1e-2, time_order=2))
  assert float((ge[0] - want).abs().max()) <= 1e-12 * scale
  assert float((ge[1] + 0.5 * want).abs().max()) <= 1e-12 * scale


def test_ensemble_cg_from_an_initial_guess_and_with_projection():
  """`x0` per member (the successive-right-hand-side projection of the
  stepper hands one to the pressure solve) and the stepper's
  `pressure_projection` on an ensemble (one basis per member): same answer,
  no more iterations."""
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  from swirl_fem_amd.linalg.cg_ensemble import cg_ensemble
  from swirl_fem_amd.niles.datagen import datagen
  sem = _sem(n=5, order=5)
  B = 3
  ens = sem.ensemble(B)
  u, g = _fields(sem, B, 8)
  rhs = ens.flatten(torch.stack([sem.B(v) for v in u]))
  H = lambda v: ens.H(v, 40.0, 0.02)
  x, info = cg_ensemble(H, rhs, B, M=ens.velocity.exchange, tol=1e-11)
  # start two members at the solution, one at zero
  x0 = ens.unflatten(x.clone())
  x0[1] = 0.0
  x1, info1 = cg_ensemble(H, rhs, B, x0=ens.flatten(x0),
                          M=ens.velocity.exchange, tol=1e-11)
  assert float((x1 - x).abs().max()) <= 1e-9 * float(x.abs().max())
  assert info1['member_iterations'][0] <= 1
  assert info1['member_iterations'][2] <= 1
  assert abs(info1['member_iterations'][1] - info['member_iterations'][1]) <= 1
  # the stepper with its projection of the pressure solve onto earlier
  # increments
  xx = sem.velocity.mesh.node_coords
  u0 = ens.flatten(torch.stack([a * datagen.u_init_fn(xx)
                                for a in (1.0, 0.6, 1.4)]))
  p0 = torch.zeros(ens.pressure.pspace.mesh.num_nodes, dtype=torch.float64,
                   device=DEV)

  def run(projection):
    s = sem.ensemble(B)          # (fresh solver state)
    us, ps = (u0, u0), (p0, p0)
    c = s.C(u0)
    Cus, its = (c, c), []
    for _ in range(5):
      f = datagen.forcing(s.velocity.mesh.node_coords, us[-1], 0.1)
      un, pn, cn, aux = navier_stokes_step(
          s, us, ps, Cus, reynolds=200.0, dt=2e-3, time_order=2, forcing=f,
          tol=1e-10, atol=0.0, pressure_projection=projection)
      us, ps, Cus = us[1:] + (un,), ps[1:] + (pn,), Cus[1:] + (cn,)
      its.append(aux['dp_info']['num_iterations'])
    return us[-1], its

  ua, ia = run(0)
  ub, ib = run(4)
  assert float((ub - ua).abs().max()) <= 1e-7 * float(ua.abs().max())
  # (a start-up from rest gains little in its first steps: non-regression
  # here, the algebra of the projection is pinned in test_gpu_stokes.py)
  assert ib[0] == ia[0] and all(b <= a + 2 for a, b in zip(ia, ib)), (ia, ib)
  # member-wise: one member of the ensemble alone gives the same counts
  s1 = sem.ensemble(1)
  assert s1 is sem
