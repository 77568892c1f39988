"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle.

fp64 tolerance 1e-10 relative, fp32 1e-5 relative (BASELINE.json north_star).
Every test here needs a real MI355X: run with `pytest -m gpu`.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sfem_oracle as O
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import gather_scatter
from swirl_fem_amd.core.fespace import FiniteElementSpace, div, grad
from swirl_fem_amd.core.interpolation import (Nodes1D, NodeType, Quadrature1D)
from swirl_fem_amd.core.mesh import Mesh
from swirl_fem_amd.core.mesh_refiner import refine_premesh
from tests.fp32util import F32Rng, f32_mesh, tolerance
from swirl_fem_amd.core.premesh import Premesh

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
NT = {'gll': NodeType.GAUSS_LOBATTO_LEGENDRE, 'gl': NodeType.GAUSS_LEGENDRE,
      'nc': NodeType.NEWTON_COTES}
TOL = {torch.float64: 1e-10, torch.float32: 1e-5}


def relerr(a, b):
  a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
  b = np.asarray(b)
  assert a.shape == b.shape, (a.shape, b.shape)
  return np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


def dev(x, dtype=None):
  t = torch.as_tensor(np.ascontiguousarray(x), device=DEV)
  return t if dtype is None else t.to(dtype)


def make_case(ndim, n, P, jitter=0.1, seed=0, periodic=(), scramble=False):
  """Deformed structured mesh refined to GLL P; returns (refined premesh)."""
  rng = np.random.default_rng(seed)
  pm = unit_cube_mesh(n, ndim=ndim, periodic_dims=periodic)
  if jitter and not periodic:
    h = 1.0 / n
    pm = pm.replace(node_coords=pm.node_coords + jitter * h *
                    rng.uniform(-1, 1, pm.node_coords.shape))
  if scramble:
    pm = pm.replace(elements=pm.elements[rng.permutation(pm.num_elements)])
  return refine_premesh(pm, Nodes1D.create(P, NT['gll']))


def spaces(rp, P, q, qt, dtype=torch.float64):
  # (fp32: node coordinates the kernels can hold, tests/fp32util.py)
  rp = f32_mesh(rp, dtype)
  mesh = rp.finalize(device=DEV, dtype=dtype)
  quad = Quadrature1D.create(q, NT[qt])
  fes = FiniteElementSpace.create(mesh, quad)
  ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (q, qt))
  return mesh, fes, ofes


# ---------------------------------------------------------- gather / scatter
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_gather_scatter_sentinel(dtype):
  rng = np.random.default_rng(1)
  N, E, n = 1000, 300, 27
  idx = rng.integers(0, N, (E, n)).astype(np.int32)
  idx[rng.random((E, n)) < 0.1] = -1
  idx[-1, :] = -1                      # a fully padded element
  u = rng.standard_normal(N)
  got = gather_scatter.gather(dev(u, dtype), dev(idx), fill_value=0.)
  assert relerr(got, O.gather(u, idx, 0.)) < TOL[dtype]
  got = gather_scatter.gather(dev(u, dtype), dev(idx))   # fill = SENTINEL
  assert relerr(got, O.gather(u, idx)) < TOL[dtype]
  ul = rng.standard_normal((E, n))
  got = gather_scatter.scatter(dev(ul, dtype), dev(idx), N)
  assert relerr(got, O.scatter(ul, idx, N)) < 10 * TOL[dtype]
  with pytest.raises(ValueError):
    gather_scatter.gather(dev(ul, dtype), dev(idx))


def test_gather_scatter_empty_and_cpu_rejected():
  idx = torch.zeros((0, 8), dtype=torch.int32, device=DEV)
  u = torch.zeros(5, dtype=torch.float64, device=DEV)
  assert gather_scatter.gather(u, idx).shape == (0, 8)
  out = gather_scatter.scatter(torch.zeros((0, 8), dtype=torch.float64,
                                           device=DEV), idx, 5)
  assert out.shape == (5,) and float(out.abs().max()) == 0.0
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    gather_scatter.gather(torch.zeros(5, dtype=torch.float64),
                          torch.zeros(3, dtype=torch.int32))


def test_scatter_csr_deterministic():
  from swirl_fem_amd import _ops
  rp = make_case(3, 3, 4)
  mesh = rp.finalize(device=DEV)
  rng = np.random.default_rng(2)
  ul = rng.standard_normal(rp.elements.shape)
  off, slots = mesh.assembly_plan().csr()
  a = _ops.scatter_csr(dev(ul), off, slots, mesh.num_nodes)
  b = _ops.scatter_csr(dev(ul), off, slots, mesh.num_nodes)
  assert torch.equal(a, b)
  assert relerr(a, O.scatter(ul, rp.elements, mesh.num_nodes)) < 1e-13


def test_mesh_gather_scatter_element_coords():
  rp = make_case(2, 4, 3)
  mesh = rp.finalize(device=DEV)
  u = np.random.default_rng(3).standard_normal(mesh.num_nodes)
  assert relerr(mesh.gather(dev(u)), u[rp.elements]) == 0.0
  assert relerr(mesh.element_coords(), rp.node_coords[rp.elements]) == 0.0
  with pytest.raises(ValueError):
    mesh.gather(dev(u[:-1]))


# ------------------------------------------------------------------ exchange
def test_exchange_reference_known_answers():
  # core/gather_scatter_test.py:50-130 and core/premesh_test.py:79-176
  ni = gather_scatter.get_unique_node_indices(
      np.arange(3, dtype=np.int32), np.array([[[0], [2]]]))
  gi, ui = gather_scatter.get_exchange_indices(ni)
  out = gather_scatter.exchange(dev(np.array([1., 2., 3.])), dev(gi), ui)
  np.testing.assert_allclose(out.cpu().numpy(), [4., 2., 4.])
  links = np.array([[[0, 1], [6, 7]], [[1, 2], [7, 8]], [[0, 3], [2, 5]],
                    [[3, 6], [5, 8]]], dtype=np.int32)
  ni = gather_scatter.get_unique_node_indices(np.arange(9, dtype=np.int32),
                                              links)
  gi, ui = gather_scatter.get_exchange_indices(ni)
  out = gather_scatter.exchange(dev(np.arange(9.)), dev(gi), ui)
  np.testing.assert_allclose(out.cpu().numpy(),
                             [16., 8., 16., 8., 4., 8., 16., 8., 16.])
  # premesh_test.py:128-176
  coords = np.array([[0, 0], [0, 1], [1, 0], [1, 1], [0, 2], [2, 1]], float)
  elements = np.array([[0, 1, 2, 3], [2, 3, 4, 5]], dtype=np.int32)
  pm = Premesh.create(coords, elements,
                      periodic_links=np.array([[[0, 1], [4, 5]]], np.int32))
  mesh = pm.finalize(device=DEV)
  out = mesh.exchange(dev(1. + np.arange(6.)))
  np.testing.assert_allclose(out.cpu().numpy(), [6, 8, 3, 4, 6, 8])
  pm = Premesh.create(coords, elements, periodic_links=np.array(
      [[[0, 1], [4, 5]], [[0, 2], [1, 3]], [[2, 4], [3, 5]]], np.int32))
  out = pm.finalize(device=DEV).exchange(dev(np.ones(6)))
  np.testing.assert_allclose(out.cpu().numpy(), [4, 4, 2, 2, 4, 4])
  # no-op cases
  pm = Premesh.create(coords, elements)
  out = pm.finalize(device=DEV).exchange(dev(np.arange(6.)))
  np.testing.assert_allclose(out.cpu().numpy(), np.arange(6.))


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_exchange_classes_layouts_and_workspace_form(dtype):
  """`sfem_exchange_classes` (one launch, in place, every layout) against the
  oracle and against the C-ABI's workspace form `sfem_exchange_local`."""
  from swirl_fem_amd import _ops
  from swirl_fem_amd.core import layout
  rp = make_case(3, 3, 4, periodic=(0, 1, 2))
  mesh = rp.finalize(device=DEV)
  arrs = rp.finalize_all()
  gi, ui = arrs['exchange_gather_indices'], arrs['exchange_unique_indices']
  u = np.random.default_rng(9).standard_normal((mesh.num_nodes, 3))
  ref = np.stack([O.exchange_unpartitioned(u[:, k], gi, ui)
                  for k in range(3)], axis=-1)
  tol = 1e-14 if dtype == torch.float64 else 1e-6
  ud, gd = dev(u, dtype), dev(gi)
  assert relerr(_ops.exchange_local(ud[:, 0].contiguous(), gd, ui),
                ref[:, 0]) < tol
  assert relerr(_ops.exchange_local(ud, gd, ui), ref) < tol
  cm = layout.component_major(ud)
  out = _ops.exchange_local(cm, gd, ui)
  assert layout.is_component_major(out) and relerr(out, ref) < tol
  w = ud.clone()
  assert _ops.exchange_local(w, gd, ui, inplace=True) is w
  assert relerr(w, ref) < tol
  assert relerr(_ops.exchange_local_atomic(ud, gd, ui), ref) < tol
  # bitwise reproducible (member order), unlike the atomic form
  assert torch.equal(_ops.exchange_local(ud, gd, ui),
                     _ops.exchange_local(ud, gd, ui))


@pytest.mark.parametrize('dtype,n', [(torch.float64, 5), (torch.float64, 70001),
                                     (torch.float32, 4099),
                                     (torch.float64, 3000000)])
def test_subtract_weighted_mean(dtype, n):
  """The pressure nullspace projection w - (b.w / 1.b) 1 (reference
  navier_stokes.py:73-78 with b = B 1)."""
  from swirl_fem_amd import _lib, _ops
  g = torch.Generator(device=DEV).manual_seed(n)
  w = torch.randn(n, dtype=dtype, device=DEV, generator=g)
  b = torch.rand(n, dtype=dtype, device=DEV, generator=g) + 0.5
  total = float(b.double().sum())
  partials = torch.full((_lib.SFEM_DOT_SLOTS,), float('nan'),
                        dtype=torch.float64, device=DEV)   # need not be cleared
  out = _ops.subtract_weighted_mean(w, b, total, partials)
  want = w.double() - torch.dot(b.double(), w.double()) / total
  assert relerr(out.double(), want.cpu().numpy()) < TOL[dtype]
  assert abs(float(torch.dot(b.double(), out.double())) / total) < (
      1e-12 if dtype == torch.float64 else 1e-5)
  # ... and w . out on the side (the r . z of the pressure CG)
  scal = torch.zeros(16, dtype=torch.float64, device=DEV)
  scal[3] = 2.5
  out2 = _ops.subtract_weighted_mean(w, b, total, partials,
                                     dot_result=(scal, 3))
  assert torch.equal(out2, out)
  wz = float(torch.dot(w.double(), out.double()))
  assert abs(float(scal[3]) - 2.5 - wz) < (1e-11 if dtype == torch.float64
                                           else 1e-4) * max(abs(wz), 1.0)
  assert _ops.subtract_weighted_mean(w, b, total, partials, out=w) is w
  assert relerr(w.double(), want.cpu().numpy()) < TOL[dtype]


@pytest.mark.parametrize('ndim,per', [(2, (0,)), (2, (0, 1)), (3, (0, 1, 2))])
def test_exchange_periodic_refined(ndim, per):
  rp = make_case(ndim, 3, 4, periodic=per)
  mesh = rp.finalize(device=DEV)
  arrs = rp.finalize_all()
  u = np.random.default_rng(4).standard_normal(mesh.num_nodes)
  ref = O.exchange_unpartitioned(u, arrs['exchange_gather_indices'],
                                 arrs['exchange_unique_indices'])
  assert relerr(mesh.exchange(dev(u)), ref) < 1e-14
  # idempotence up to multiplicity: exchanging a constant counts the images
  ones = mesh.exchange(dev(np.ones(mesh.num_nodes)))
  ref1 = O.exchange_unpartitioned(np.ones(mesh.num_nodes),
                                  arrs['exchange_gather_indices'],
                                  arrs['exchange_unique_indices'])
  assert relerr(ones, ref1) == 0.0


def test_partition_pack_unpack_kernels_emulated_ranks():
  """The HIP halves of the RCCL neighbour exchange (sfem_pack /
  sfem_unpack_add) on several block partitions living on one GPU: buffers are
  swapped by hand instead of ncclSend/ncclRecv; the result must equal QQ^T."""
  from swirl_fem_amd import _ops
  from swirl_fem_amd.distributed import blocks
  for grid, n, P in [((2, 2, 1), 2, 4), ((2, 1, 1), 3, 3)]:
    world = int(np.prod(grid))
    parts = [blocks.build_block_partition(n, P, grid, r, device=DEV)
             for r in range(world)]
    rng = np.random.default_rng(5)
    for nc in (1, 3):
      us = [rng.standard_normal((p.mesh.num_nodes, nc)) for p in parts]
      us = [u[:, 0] if nc == 1 else u for u in us]
      dus = [dev(u) for u in us]
      send = [[_ops.pack(du, ix) for ix in p.plan.device_indices(DEV)]
              for du, p in zip(dus, parts)]
      outs = []
      for r, p in enumerate(parts):
        out = dus[r].clone()
        for q, ix in zip(p.plan.neighbors, p.plan.device_indices(DEV)):
          j = parts[q].plan.neighbors.index(r)
          _ops.unpack_add(send[q][j], ix, out)
        outs.append(out.cpu().numpy())
      tot = {}
      for p, u in zip(parts, us):
        xc = np.round(p.mesh.node_coords.cpu().numpy() * 1e7).astype(np.int64)
        for k, v in zip(map(tuple, xc), u):
          tot[k] = tot.get(k, 0.0) + v
      for p, out in zip(parts, outs):
        xc = np.round(p.mesh.node_coords.cpu().numpy() * 1e7).astype(np.int64)
        ref = np.array([tot[k] for k in map(tuple, xc)])
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13)
    # the partitioned operator: unassembled local apply + exchange == the
    # single-mesh operator restricted to the block (interior Dirichlet aside)
    assert parts[0].mesh.axis_name == 'blocks'
    assert parts[0].mesh.neighbor_plan is parts[0].plan


# ---------------------------------------------------- geometry and basis eval
CASES = [  # ndim, n, P, q, quadrature type
    (1, 5, 4, 5, 'gl'), (2, 3, 4, 5, 'gl'), (2, 3, 5, 5, 'gll'),
    (2, 2, 6, 8, 'gll'), (3, 2, 3, 4, 'gl'), (3, 2, 4, 4, 'gll'),
    (3, 2, 4, 6, 'gll'), (3, 2, 6, 4, 'gll'),
]


@pytest.mark.parametrize('ndim,n,P,q,qt', CASES)
def test_geometric_factors(ndim, n, P, q, qt):
  rp = make_case(ndim, n, P, seed=5)
  _, fes, ofes = spaces(rp, P, q, qt)
  assert relerr(fes.invjacs, ofes.invjacs) < 1e-10
  assert relerr(fes.jacdets, ofes.jacdets) < 1e-10
  assert relerr(fes.quad_coords, ofes.quad_coords) < 1e-12


def test_signed_jacobian_of_reflected_element():
  # a mirrored element has negative det J and the reference keeps the sign
  coords = np.array([[1., 0], [1, 1], [0, 0], [0, 1]])
  mesh = Mesh.create(coords, np.arange(4).reshape(1, 4), device=DEV)
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create(2, NT['gl']))
  ofes = O.FESpace(coords, np.arange(4).reshape(1, 4), (2, 'nc'), (2, 'gl'))
  assert float(fes.jacdets.max()) < 0
  assert relerr(fes.jacdets, ofes.jacdets) < 1e-13


@pytest.mark.parametrize('ndim,n,P,q,qt', CASES)
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_qfunction_values_and_gradients(ndim, n, P, q, qt, dtype):
  rp = make_case(ndim, n, P, seed=6)
  mesh, fes, ofes = spaces(rp, P, q, qt, dtype)
  rng = np.random.default_rng(7)
  tol = TOL[dtype] * (1 if dtype == torch.float64 else 30)
  us = rng.standard_normal(rp.elements.shape)
  f = fes.scalar_function(dev(us, dtype))
  assert relerr(f(None).val, ofes.value(us)) < tol
  assert relerr(grad(f)(None).val, ofes.grad(us)) < tol
  uv = rng.standard_normal(rp.elements.shape + (ndim,))
  fv = fes.vector_function(dev(uv, dtype))
  assert relerr(fv(None).val, ofes.value(uv)) < tol
  assert relerr(grad(fv)(None).val, ofes.grad(uv)) < tol
  ig = fes.integrate(lambda x: torch.vdot(grad(f)(x), grad(f)(x)))
  ref = ofes.integrate(np.einsum('mqj,mqj->mq', ofes.grad(us), ofes.grad(us)))
  assert abs(float(ig) - ref) < 30 * tol * abs(ref)


# ------------------------------------------------------- forms / local_covector
@pytest.mark.parametrize('ndim,n,P,q,qt', CASES[1:])
def test_local_covector_forms(ndim, n, P, q, qt):
  rp = make_case(ndim, n, P, seed=8)
  mesh, fes, ofes = spaces(rp, P, q, qt)
  rng = np.random.default_rng(9)
  us = rng.standard_normal(rp.elements.shape)
  uv = rng.standard_normal(rp.elements.shape + (ndim,))
  wv = rng.standard_normal(rp.elements.shape + (ndim,))

  u, v = fes.scalar_function(dev(us)), fes.scalar_function(None)
  mass = fes.local_covector(lambda a, b: lambda x: a(x) * b(x), (u, v))
  assert relerr(mass, ofes.mass_local(us)) < 1e-10
  stiff = fes.local_covector(
      lambda a, b: lambda x: torch.vdot(grad(a)(x), grad(b)(x)), (u, v))
  assert relerr(stiff, ofes.stiffness_local(us)) < 1e-10

  U, V = fes.vector_function(dev(uv)), fes.vector_function(None)
  vmass = fes.local_covector(
      lambda a, b: lambda x: torch.vdot(a(x), b(x)), (U, V))
  assert relerr(vmass, ofes.mass_local(uv)) < 1e-10
  vstiff = fes.local_covector(
      lambda a, b: lambda x: torch.einsum('ij,ij->', grad(a)(x), grad(b)(x)),
      (U, V))
  assert relerr(vstiff, ofes.stiffness_local(uv)) < 1e-10
  W = fes.vector_function(dev(wv))
  conv = fes.local_covector(
      lambda a, w, b: lambda x: torch.einsum('i,ij,j->', a(x), grad(w)(x),
                                             b(x)), (U, W, V))
  assert relerr(conv, ofes.convection_local(uv, wv)) < 1e-10
  # placeholder in the first slot (symmetric form) gives the same covector
  stiff2 = fes.local_covector(
      lambda a, b: lambda x: torch.vdot(grad(a)(x), grad(b)(x)), (v, u))
  assert relerr(stiff2, ofes.stiffness_local(us)) < 1e-10
  with pytest.raises(ValueError):
    fes.local_covector(lambda a, b: lambda x: a(x) * b(x), (u, u))
  with pytest.raises(ValueError):
    fes.scalar_function(dev(us[:, :-1]))


@pytest.mark.parametrize('ndim,n,P', [(2, 3, 6), (3, 2, 5)])
def test_divergence_forms_mixed_spaces(ndim, n, P):
  # P_N - P_{N-2}: velocity GLL(P), pressure GL(P-2), shared GLL(P) quadrature
  # (navier_stokes.py:117-121, :279-282, :313-329)
  rng = np.random.default_rng(10)
  pm = unit_cube_mesh(n, ndim=ndim)
  pm = pm.replace(node_coords=pm.node_coords + 0.05 *
                  rng.uniform(-1, 1, pm.node_coords.shape))
  vgrid, pgrid = Nodes1D.create(P, NT['gll']), Nodes1D.create(P - 2, NT['gl'])
  rv, rq = refine_premesh(pm, vgrid), refine_premesh(pm, pgrid)
  quad = Quadrature1D.create(P, NT['gll'])
  vsp = FiniteElementSpace.create(rv.finalize(device=DEV), quad)
  psp = FiniteElementSpace.create(rq.finalize(device=DEV), quad)
  ov = O.FESpace(rv.node_coords, rv.elements, (P, 'gll'), (P, 'gll'))
  op = O.FESpace(rq.node_coords, rq.elements, (P - 2, 'gl'), (P, 'gll'))
  b = lambda vf, qf: lambda x: div(vf)(x) * qf(x)
  uv = rng.standard_normal(rv.elements.shape + (ndim,))
  pl = rng.standard_normal(rq.elements.shape)
  d_local = psp.local_covector(
      b, (vsp.vector_function(dev(uv)), psp.scalar_function(None)))
  assert relerr(d_local, O.div_local(ov, op, uv)) < 1e-10
  dt_local = vsp.local_covector(
      b, (vsp.vector_function(None), psp.scalar_function(dev(pl))))
  assert relerr(dt_local, O.div_t_local(ov, op, pl)) < 1e-10
  # adjointness: <D u, p> == <u, D^T p>
  lhs = float((d_local * dev(pl)).sum())
  rhs = float((dt_local * dev(uv)).sum())
  assert abs(lhs - rhs) < 1e-11 * abs(lhs)


def test_fespace_reference_known_answers():
  # core/fespace_test.py:128-180 (skewed quad) and :73-126 (single elements)
  coords = np.array([[0, 0], [0, 1], [1, 0], [1, 2]], dtype=np.float64)
  mesh = Mesh.create(coords, np.arange(4).reshape(1, 4), device=DEV)
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create(2, NT['gl']))
  xe = mesh.element_coords()
  f = fes.scalar_function(2 * xe[..., 0] - xe[..., 1] + 1)
  assert float(fes.integrate(lambda x: grad(f)(x)[0])) == pytest.approx(3.0)
  assert float(fes.integrate(lambda x: grad(f)(x)[1])) == pytest.approx(-1.5)
  fv = fes.vector_function(torch.stack(
      [2 * xe[..., 0] - xe[..., 1], 3 * xe[..., 1]], dim=-1))
  assert float(fes.integrate(div(fv))) == pytest.approx(7.5)
  for ndim in (1, 2, 3):
    for order in (1, 2, 3, 4):
      n = (order + 1) ** ndim
      c1 = np.linspace(0, 1, order + 1)
      cc = np.stack(np.meshgrid(*([c1] * ndim), indexing='ij'),
                    axis=-1).reshape(n, ndim)
      m = Mesh.create(cc, np.arange(n).reshape(1, n), device=DEV)
      fs = FiniteElementSpace.create(m, Quadrature1D.create(order + 1,
                                                            NT['gl']))
      fn = lambda x: sum(x[i] ** order for i in range(ndim))
      assert float(fs.integrate(fn)) == pytest.approx(ndim / (1 + order))
      assert float(fs.integrate(lambda x: grad(fn)(x)[0])) == pytest.approx(1.)
      ec = m.element_coords()
      nodal = fs.scalar_function((ec ** order).sum(-1))
      assert float(fs.integrate(nodal)) == pytest.approx(ndim / (1 + order))
      assert float(fs.integrate(lambda x: grad(nodal)(x)[0])) == pytest.approx(
          1.)
  assert float(fes.integrate(lambda x: 1.)) == pytest.approx(1.5)


# ------------------------------------------------- fused Helmholtz operator
def _helmholtz_ref(ofes, u, l0, l1, dirichlet):
  ul = ofes.gather(u)
  loc = 0.0
  if l0:
    loc = loc + l0 * ofes.mass_local(ul)
  if l1:
    loc = loc + l1 * ofes.stiffness_local(ul)
  out = ofes.scatter(loc)
  if dirichlet is not None:
    keep = 1.0 - dirichlet.astype(np.float64)
    out = out * (keep if out.ndim == 1 else keep[:, None])
  return out


@pytest.mark.parametrize('ndim,n,P', [
    (2, 4, 2), (2, 3, 3), (2, 3, 4), (2, 3, 5), (2, 2, 7), (2, 2, 8),
    (2, 2, 11), (2, 2, 12), (3, 3, 2), (3, 2, 3), (3, 3, 4), (3, 2, 5),
    (3, 2, 6), (3, 2, 7), (3, 3, 8), (3, 2, 9), (3, 2, 10), (3, 1, 11),
    (3, 2, 12)])
@pytest.mark.parametrize('geometry', ['auto', 'stored'])
def test_fused_helmholtz_fp64(ndim, n, P, geometry):
  rp = make_case(ndim, n, P, seed=11, scramble=True)
  mesh, fes, ofes = spaces(rp, P, P, 'gll')
  rng = np.random.default_rng(12)
  u = rng.standard_normal(mesh.num_nodes)
  bmask = mesh.physical_masks['boundary'].cpu().numpy()
  op_free = fes.helmholtz_operator(None, geometry)
  op_bc = fes.helmholtz_operator(mesh.physical_masks['boundary'], geometry)
  if geometry == 'auto':     # vertex-jittered elements are multilinear
    assert op_free.num_multilinear == mesh.num_elements
  else:
    assert op_free.num_curved == mesh.num_elements
  for l0, l1, op, msk in [(0., 1., op_free, None), (1., 0., op_free, None),
                          (0.7, 1.3, op_bc, bmask)]:
    got = op.apply(dev(u), l0, l1)
    assert relerr(got, _helmholtz_ref(ofes, u, l0, l1, msk)) < 1e-10, (l0, l1)
  ul = rng.standard_normal(rp.elements.shape)
  got = op_free.apply_local(dev(ul), 0.3, 2.0)
  ref = 0.3 * ofes.mass_local(ul) + 2.0 * ofes.stiffness_local(ul)
  assert relerr(got, ref) < 1e-10
  # constants are in the nullspace of the stiffness operator
  z = op_free.apply(dev(np.ones(mesh.num_nodes)), 0., 1.)
  assert float(z.abs().max()) < 1e-9 * float(np.abs(
      _helmholtz_ref(ofes, u, 0., 1., None)).max())


@pytest.mark.parametrize('name,ndim,P', [('cube.msh', 3, 4),
                                         ('periodic_cube.msh', 3, 3),
                                         ('kovasznay.msh', 2, 6)])
def test_fused_helmholtz_on_gmsh_meshes(name, ndim, P):
  """Reference data files (swirl_fem/testdata/*.msh) through the native Gmsh
  reader, the refiner and the fused operator, against the oracle."""
  from swirl_fem_amd.common import mesh_reader
  pm = mesh_reader.read(os.path.join(os.path.dirname(__file__), 'golden',
                                     'msh', name), ndim=ndim)
  rp = refine_premesh(pm, Nodes1D.create(P, NT['gll']))
  mesh, fes, ofes = spaces(rp, P, P, 'gll')
  rng = np.random.default_rng(23)
  x = rp.node_coords
  lo, hi = x.min(axis=0), x.max(axis=0)
  dirichlet = (np.isclose(x[:, 0], lo[0]) | np.isclose(x[:, 0], hi[0]))
  u = rng.standard_normal(mesh.num_nodes)
  op = fes.helmholtz_operator(dev(dirichlet))
  ref = _helmholtz_ref(ofes, u, 0.3, 1.0, dirichlet)
  assert relerr(op.apply(dev(u), 0.3, 1.0), ref) < 1e-10
  assert op.num_affine + op.num_multilinear == mesh.num_elements


@pytest.mark.parametrize('ndim,n,P', [(2, 3, 6), (3, 2, 4), (3, 2, 8),
                                      (3, 1, 12)])
def test_fused_helmholtz_fp32_and_vector(ndim, n, P):
  rp = make_case(ndim, n, P, seed=13)
  rng = F32Rng(14)
  for dtype in (torch.float32, torch.float64):
    mesh, fes, ofes = spaces(rp, P, P, 'gll', dtype)
    bmask = mesh.physical_masks['boundary'].cpu().numpy()
    tol = tolerance(dtype, P)
    for geometry in ('auto', 'stored'):
      op = fes.helmholtz_operator(mesh.physical_masks['boundary'], geometry)
      for nc in (1, 2, 3):
        u = rng.standard_normal((mesh.num_nodes, nc))
        uu = u[:, 0] if nc == 1 else u
        got = op.apply(dev(uu, dtype), 0.5, 1.5)
        assert relerr(got, _helmholtz_ref(ofes, uu, 0.5, 1.5, bmask)) < tol
      ul = rng.standard_normal(rp.elements.shape + (ndim,))
      got = op.apply_local(dev(ul, dtype), 0.0, 1.0)
      assert relerr(got, ofes.stiffness_local(ul)) < tol
      # component-major storage seen as (N, nc) / (E, n, nc) views
      u = rng.standard_normal((mesh.num_nodes, 3))
      ucm = dev(u.T.copy(), dtype).t()
      assert not ucm.is_contiguous() and ucm.shape == (mesh.num_nodes, 3)
      got = op.apply(ucm, 0.5, 1.5)
      assert got.stride() == ucm.stride()
      assert relerr(got, _helmholtz_ref(ofes, u, 0.5, 1.5, bmask)) < tol
      ulcm = dev(np.moveaxis(ul, -1, 0).copy(), dtype).movedim(0, -1)
      got = op.apply_local(ulcm, 0.0, 1.0)
      assert relerr(got, ofes.stiffness_local(ul)) < tol


@pytest.mark.parametrize('mode', ['structured', 'sheared', 'jittered',
                                  'mixed'])
def test_matrix_core_helmholtz_p11_fp32(mode, monkeypatch):
  """`helmholtz_mfma_p12_kernel` (p = 11, fp32: the 12 x 12 contractions as
  v_mfma_f32_16x16x4_f32 tiles) vs the oracle and vs the vector-ALU kernel of
  the same operator: affine and multilinear elements, mass on / off, pure
  mass, Dirichlet mask, fused u . A u, element lists (a mesh that also has
  curved elements keeps those on the vector-ALU kernel)."""
  from swirl_fem_amd import _lib
  P, n = 12, 2 if mode != 'mixed' else 3
  rng = F32Rng(71)
  pm = unit_cube_mesh(n, ndim=3)
  x = pm.node_coords.copy()
  if mode == 'sheared':
    x = x @ (np.eye(3) + 0.25 * rng.uniform(-1, 1, (3, 3))).T + 0.2
  if mode in ('jittered', 'mixed'):
    x = x + 0.08 / n * rng.uniform(-1, 1, x.shape)
  rp = refine_premesh(pm.replace(node_coords=x), Nodes1D.create(P, NT['gll']))
  if mode == 'mixed':           # bend the nodes of the first layer of elements
    xc = rp.node_coords.copy()
    inside = (xc[:, 0] > 1e-9) & (xc[:, 0] < 1 / 3 - 1e-9)   # first layer only
    xc[:, 2] += 0.02 * np.sin(3 * np.pi * xc[:, 0]) * inside * (
        xc[:, 2] * (1 - xc[:, 2]))
    rp = rp.replace(node_coords=xc)
  mesh, fes, ofes = spaces(rp, P, P, 'gll', torch.float32)
  bmask = mesh.physical_masks['boundary'].cpu().numpy()
  # the matrix-core kernel is an opt-in of the INDEX-ROW path (since round 3
  # box / affine elements of refiner meshes default to the facet kernels)
  monkeypatch.setenv('SFEM_FACET', '0')
  op = fes.helmholtz_operator(mesh.physical_masks['boundary'])
  assert op.facet_parts is None
  if mode in ('structured', 'sheared'):
    assert op.num_affine == mesh.num_elements
  if mode == 'mixed':
    assert op.num_curved > 0 and op.num_multilinear + op.num_affine > 0, (
        op.num_curved, op.num_multilinear, op.num_affine)
  monkeypatch.setenv('SFEM_MFMA', '1')
  assert 'helmholtz_mfma_p12_kernel' in op.kernel_name(0.5, 1.0)
  u = rng.standard_normal(mesh.num_nodes)
  ud = dev(u, torch.float32)
  for l0, l1 in ((0.0, 1.0), (0.7, 1.2), (1.0, 0.0)):
    ref = _helmholtz_ref(ofes, u, l0, l1, bmask)
    monkeypatch.setenv('SFEM_MFMA', '1')
    parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device=DEV)
    got = op.apply(ud, l0, l1, dot_out=parts)
    assert relerr(got, ref) < tolerance(torch.float32, P), (mode, l0, l1)
    want, scale = float((u * ref).sum()), float(np.abs(u * ref).sum())
    assert abs(float(parts.sum()) - want) <= 3e-4 * scale
    monkeypatch.setenv('SFEM_MFMA', '0')
    assert 'helmholtz_kernel<float, 12' in op.kernel_name(l0, l1)
    valu = op.apply(ud, l0, l1)
    assert relerr(valu, ref) < tolerance(torch.float32, P)
    assert relerr(got, valu.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('ndim,n,P', [(2, 4, 5), (3, 3, 4), (3, 3, 8)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_fused_helmholtz_geometry_kinds(ndim, n, P, dtype):
  """Affine / multilinear elements evaluate their factors in registers,
  curved elements read stored factors; a mesh may mix all three."""
  rng = F32Rng(19)
  tol = TOL[dtype]
  for mode in ('structured', 'sheared', 'vertex', 'curved', 'mixed'):
    pm = unit_cube_mesh(n, ndim=ndim)
    x = pm.node_coords.copy()
    if mode == 'sheared':       # affine map of the whole mesh: still affine
      A = np.eye(ndim) + 0.3 * rng.uniform(-1, 1, (ndim, ndim))
      x = x @ A.T + 0.1
    if mode in ('vertex', 'mixed'):   # move one interior vertex: multilinear
      centre = np.argmin(((x - 0.5) ** 2).sum(-1))
      x[centre] += 0.1 / n
    rp = refine_premesh(pm.replace(node_coords=x),
                        Nodes1D.create(P, NT['gll']))
    if mode in ('curved', 'mixed'):   # bend the high-order nodes themselves
      xc = rp.node_coords.copy()
      bump = 0.03 * np.sin(np.pi * xc[:, 0]) * np.sin(2 * np.pi * xc[:, 1])
      if mode == 'mixed':
        bump = bump * (xc[:, 0] < 1.0 / n + 1e-9)   # first layer of elements
      xc[:, -1] += bump * np.prod(xc * (1 - xc), axis=1) * 4 ** ndim
      rp = rp.replace(node_coords=xc)
    mesh, fes, ofes = spaces(rp, P, P, 'gll', dtype)
    bmask = mesh.physical_masks['boundary'].cpu().numpy()
    ops = {g: fes.helmholtz_operator(mesh.physical_masks['boundary'], g)
           for g in ('auto', 'multilinear', 'stored')}
    E, op = mesh.num_elements, ops['auto']
    assert op.num_affine + op.num_multilinear + op.num_curved == E
    assert ops['stored'].num_curved == E
    if dtype == torch.float64:
      if mode in ('structured', 'sheared'):
        assert op.num_affine == E and ops['multilinear'].num_multilinear == E
      if mode == 'vertex':
        assert op.num_multilinear == 2 ** ndim and op.num_curved == 0
      if mode == 'curved':
        assert op.num_curved > 0
      if mode == 'mixed':
        assert min(op.num_affine, op.num_multilinear, op.num_curved) > 0
    for nc in (1, ndim):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      ref = _helmholtz_ref(ofes, uu, 0.4, 1.1, bmask)
      for g, o in ops.items():
        assert relerr(o.apply(dev(uu, dtype), 0.4, 1.1), ref) < tol, (mode, g)
    # pure stiffness through the ASSEMBLED kernels: lambda0 = 0 selects the
    # MASS=false instantiations, with and without the fused u . A u, scalar
    # and component-major.  (n = 3 runs unchained here; the instantiation
    # bench.py times is pinned at its chain length by tests/test_gpu_facet.py::
    # test_headline_chain_instantiation_matches_oracle and, at full size, by
    # test_config2_properties_full_size[0.0-64].)
    from swirl_fem_amd import _lib
    for nc in (1, ndim):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      ref = _helmholtz_ref(ofes, uu, 0.0, 1.0, bmask)
      for g, o in ops.items():
        ud = dev(uu, dtype) if nc == 1 else dev(u.T.copy(), dtype).t()
        assert relerr(o.apply(ud, 0.0, 1.0), ref) < tol, (mode, g, nc)
        parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                            device=ud.device)
        got = o.apply(ud, 0.0, 1.0, dot_out=parts)
        assert relerr(got, ref) < tol, (mode, g, nc, 'dot')
        # u . (mask * A u): the Dirichlet rows of `ref` are zero already
        want, scale = float((uu * ref).sum()), float(np.abs(uu * ref).sum())
        assert abs(float(parts.sum()) - want) <= 10 * tol * scale, (mode, g)
    ul = rng.standard_normal(rp.elements.shape)
    ref = 0.2 * ofes.mass_local(ul) + ofes.stiffness_local(ul)
    for g, o in ops.items():
      assert relerr(o.apply_local(dev(ul, dtype), 0.2, 1.0), ref) < tol, (mode,
                                                                          g)
      assert relerr(o.apply_local(dev(ul, dtype), 1.0, 0.0),
                    ofes.mass_local(ul)) < tol, (mode, g)


@pytest.mark.parametrize('n,P,scramble', [(4, 4, False), (3, 5, True),
                                          (3, 6, False), (2, 7, True),
                                          (4, 8, False), (3, 8, True)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_fused_helmholtz_cluster_assembly(n, P, scramble, dtype):
  """Cluster assembly (`helmholtz_cluster_kernel`: shared nodes of 8 elements
  summed in LDS, atomics on the cluster surfaces only) vs the oracle, next to
  the one-atomic-per-slot and the coloured assembly of the same operator:
  every geometry kind, mass on / off, 1 and 3 components (row- and
  component-major), fused u . A u, odd element counts (clusters with empty
  places) and scrambled element order (clusters found by bisection)."""
  from swirl_fem_amd import _lib
  from swirl_fem_amd.core import operators
  rp = make_case(3, n, P, seed=41 + P, scramble=scramble)
  rng = F32Rng(43)
  mesh, fes, ofes = spaces(rp, P, P, 'gll', dtype)
  bmask = mesh.physical_masks['boundary'].cpu().numpy()
  tol = TOL[dtype]
  for geometry in ('auto', 'stored'):
    ops = {a: operators.HelmholtzOperator.create(
        fes, mesh.physical_masks['boundary'], geometry, a)
           for a in ('cluster', 'atomic')}
    assert all(p.get('cluster') is not None for p in ops['cluster'].parts)
    assert all(p.get('cluster') is None for p in ops['atomic'].parts)
    plan = ops['cluster'].parts[0]['cluster']
    assert plan.num_complete > 0
    assert plan.num_surface > 0 or plan.num_clusters == 1
    assert 'helmholtz_cluster_kernel' in ops['cluster'].kernel_name()
    for l0, l1 in ((0.0, 1.0), (0.6, 1.3), (1.0, 0.0)):
      for nc in (1, 3):
        u = rng.standard_normal((mesh.num_nodes, nc))
        uu = u[:, 0] if nc == 1 else u
        ref = _helmholtz_ref(ofes, uu, l0, l1, bmask)
        for layout in ('rows', 'components'):
          if nc == 1 and layout == 'components':
            continue
          ud = (dev(uu, dtype) if layout == 'rows'
                else dev(u.T.copy(), dtype).t())
          got = {a: o.apply(ud, l0, l1) for a, o in ops.items()}
          for a, g in got.items():
            assert relerr(g, ref) < tol, (geometry, a, l0, nc, layout)
          assert relerr(got['cluster'], got['atomic'].cpu().numpy()) < (
              1e-12 if dtype == torch.float64 else 1e-5)
          parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                              device=ud.device)
          g = ops['cluster'].apply(ud, l0, l1, dot_out=parts)
          assert relerr(g, ref) < tol
          want, scale = float((uu * ref).sum()), float(np.abs(uu * ref).sum())
          assert abs(float(parts.sum()) - want) <= 10 * tol * scale
  # the two halves of a split operator cluster their own elements
  op = operators.HelmholtzOperator.create(fes, mesh.physical_masks['boundary'],
                                          assembly='cluster')
  pick = torch.as_tensor(rng.random(mesh.num_elements) < 0.3, device=DEV)
  a, b = op.split(pick)
  u = rng.standard_normal(mesh.num_nodes)
  out = a.apply(dev(u, dtype), 0.2, 1.0)
  b.apply(dev(u, dtype), 0.2, 1.0, out=out, zero=False)
  assert relerr(out, _helmholtz_ref(ofes, u, 0.2, 1.0, bmask)) < tol


def test_cluster_assembly_on_partition_padded_elements():
  """Uneven partitions pad the element list with all -1 rows
  (premesh_test.py:309-317): such elements join no cluster."""
  from swirl_fem_amd.core import operators
  P = 4
  pm = unit_cube_mesh(3, ndim=3)
  pm = pm.replace(partitions=(np.arange(27) >= 10).astype(np.int32))  # 10 | 17
  rp = refine_premesh(pm, Nodes1D.create(P, NT['gll']))
  rng = np.random.default_rng(5)
  for rank in (0, 1):
    mesh = rp.finalize('i', rank=rank, device=DEV)
    el = mesh.elements.cpu().numpy()
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(rp.gridpoints_1d))
    ops = {a: operators.HelmholtzOperator.create(fes, None, 'auto', a)
           for a in ('cluster', 'atomic')}
    u = dev(rng.standard_normal(mesh.num_nodes))
    got = {a: o.apply(u, 0.5, 1.0) for a, o in ops.items()}
    assert relerr(got['cluster'], got['atomic'].cpu().numpy()) < 1e-12
    real = int((el >= 0).any(axis=1).sum())
    assert real == (10, 17)[rank] and len(el) == 17
    plan = ops['cluster'].parts[0]['cluster']
    assert int((plan.elems >= 0).sum()) == real


@pytest.mark.parametrize('ndim,n,P,scramble', [(3, 4, 4, False), (3, 3, 5, True),
                                               (2, 6, 4, True), (3, 2, 8, False)])
def test_fused_helmholtz_colored_assembly(ndim, n, P, scramble):
  """One launch per conflict-free colour class: same operator, no atomics,
  bitwise reproducible."""
  rp = make_case(ndim, n, P, seed=23, scramble=scramble)
  mesh, fes, ofes = spaces(rp, P, P, 'gll')
  bmask = mesh.physical_masks['boundary'].cpu().numpy()
  plan = mesh.assembly_plan()
  colors, ncol, first = plan.coloring()
  assert 2 ** ndim <= ncol <= 4 * 2 ** ndim
  # every node has exactly one first toucher; classes are conflict free
  el = mesh.elements.to(torch.int64)
  cnt = torch.zeros(mesh.num_nodes, dtype=torch.int64, device=DEV)
  cnt.index_add_(0, el[first], torch.ones_like(el[first]))
  assert int(cnt.min()) == 1 and int(cnt.max()) == 1
  for c in range(ncol):
    nodes = el[colors == c].reshape(-1)
    assert nodes.unique().numel() == nodes.numel()
  rng = np.random.default_rng(24)
  for geometry in ('auto', 'stored'):
    op_c = fes.helmholtz_operator(mesh.physical_masks['boundary'], geometry,
                                  'colored')
    op_a = fes.helmholtz_operator(mesh.physical_masks['boundary'], geometry,
                                  'atomic')
    for nc in (1, 3):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      ref = _helmholtz_ref(ofes, uu, 0.6, 1.2, bmask)
      # poison the output buffer: every entry must be written
      out = torch.full_like(dev(uu), float('nan'))
      got = op_c.apply(dev(uu), 0.6, 1.2, out=out)
      assert relerr(got, ref) < 1e-10
      again = op_c.apply(dev(uu), 0.6, 1.2)
      assert torch.equal(got, again)
      assert relerr(op_a.apply(dev(uu), 0.6, 1.2), ref) < 1e-10
  # CG with the coloured operator (fused dot) matches the oracle
  from swirl_fem_amd.linalg.cg import cg
  b = (1.0 - bmask) * rng.standard_normal(mesh.num_nodes)
  op_c = fes.helmholtz_operator(mesh.physical_masks['boundary'], 'auto',
                                'colored')
  xo, io = O.cg(lambda x: _helmholtz_ref(ofes, x, 0.1, 1.0, bmask), b,
                tol=1e-10)
  xg, ig = cg(op_c.linear_operator(0.1, 1.0), dev(b), tol=1e-10)
  assert ig['num_iterations'] == io['num_iterations']
  assert relerr(xg, xo) < 1e-8


def test_fused_helmholtz_padded_elements_and_errors():
  # partition-style padding: trailing elements with all -1 connectivity
  rp = make_case(3, 2, 4, seed=15)
  elements = np.concatenate([rp.elements, np.full((3, 64), -1, np.int32)])
  mesh = Mesh.create(rp.node_coords, elements,
                     gridpoints_1d=rp.gridpoints_1d, device=DEV)
  fes = FiniteElementSpace.create(
      mesh, Quadrature1D.create_from_nodes_1d(rp.gridpoints_1d))
  ofes = O.FESpace(rp.node_coords, rp.elements, (4, 'gll'), (4, 'gll'))
  u = np.random.default_rng(16).standard_normal(mesh.num_nodes)
  got = fes.helmholtz_operator(None).apply(dev(u), 0.2, 1.0)
  assert torch.isfinite(got).all()
  assert relerr(got, _helmholtz_ref(ofes, u, 0.2, 1.0, None)) < 1e-10
  # quadrature != nodes: the two-grid operator, also with padded elements
  fes2 = FiniteElementSpace.create(mesh, Quadrature1D.create(5, NT['gl']))
  ofes2 = O.FESpace(rp.node_coords, rp.elements, (4, 'gll'), (5, 'gl'))
  got2 = fes2.helmholtz_operator(None).apply(dev(u), 0.2, 1.0)
  assert relerr(got2, _helmholtz_ref(ofes2, u, 0.2, 1.0, None)) < 1e-10
  # more quadrature points than the kernels are compiled for
  fes3 = FiniteElementSpace.create(mesh, Quadrature1D.create(13, NT['gl']))
  with pytest.raises(NotImplementedError):
    fes3.helmholtz_operator(None)
  with pytest.raises(ValueError):
    fes.helmholtz_operator(None).apply(dev(u[:-1]))


# ----------------------------------------------------------------------- CG
@pytest.mark.parametrize('ndim,n,P,q,qt', [(2, 4, 4, 5, 'gl'), (2, 3, 2, 3, 'gl'),
                                           (3, 2, 4, 6, 'gl'), (3, 2, 6, 8, 'gll'),
                                           (3, 2, 3, 5, 'gl')])
def test_two_grid_helmholtz_matches_oracle(ndim, n, P, q, qt):
  """Quadrature != nodes (the Poisson example's Gauss rule, the convection
  space's over-integration): interpolate -> fused kernel on the quadrature
  grid -> transposed interpolation, vs the oracle's dense forms."""
  from swirl_fem_amd.core import operators
  rp = make_case(ndim, n, P, jitter=0.15, seed=21, scramble=True)
  mesh, fes, ofes = spaces(rp, P, q, qt)
  assert not fes.is_collocated and operators.supports_two_grid(fes) is None
  bm = mesh.physical_masks['boundary']
  bmask = bm.cpu().numpy()
  rng = np.random.default_rng(22)
  for geometry in ('auto', 'stored'):
    op = fes.helmholtz_operator(bm, geometry)
    assert isinstance(op, operators.TwoGridHelmholtzOperator)
    if geometry == 'auto':
      assert {p['geo_mode'] for p in op.parts} == {3}
    for nc in (1, ndim):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      for l0, l1 in ((0.0, 1.0), (1.0, 0.0), (0.4, 1.3)):
        ref = _helmholtz_ref(ofes, uu, l0, l1, bmask)
        assert relerr(op.apply(dev(uu), l0, l1), ref) < 1e-10, (geometry, l0)
  ul = rng.standard_normal(rp.elements.shape)
  ref = 0.3 * ofes.mass_local(ul) + ofes.stiffness_local(ul)
  assert relerr(fes.helmholtz_operator(None).apply_local(dev(ul), 0.3, 1.0),
                ref) < 1e-10


def test_cg_reference_known_answers():
  from swirl_fem_amd.linalg.cg import cg
  b = dev(np.arange(9.0).reshape(3, 3))
  x, info = cg(lambda x: 2 * x, b)
  np.testing.assert_allclose(x.cpu().numpy(), np.arange(9.).reshape(3, 3) / 2)
  assert info['num_iterations'] == 1
  A = lambda x: {'a': x['a'] + 0.5 * x['b'], 'b': 0.5 * x['a'] + x['b']}
  x, _ = cg(A, {'a': dev(np.array([1.0])), 'b': dev(np.array([-4.0]))})
  assert float(x['a']) == pytest.approx(4.0, abs=1e-6)
  assert float(x['b']) == pytest.approx(-6.0, abs=1e-6)
  A = lambda x: torch.stack([2 * x[0], 0 * x[1]])
  M = lambda x: torch.stack([x[0], 0 * x[1]])
  x, _ = cg(A, dev(1 + np.arange(2.0)), M=M)
  np.testing.assert_allclose(x.cpu().numpy(), [0.5, 0.])
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    cg(lambda x: x, torch.ones(3))


def test_cg_matches_oracle_iterates():
  from swirl_fem_amd.linalg.cg import cg
  rng = np.random.default_rng(17)
  n = 200
  Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  Amat = Q @ np.diag(np.linspace(1, 50, n)) @ Q.T
  b = rng.standard_normal(n)
  Ad = dev(Amat)
  for tol, maxiter in [(1e-5, None), (1e-12, None), (1e-12, 7)]:
    xo, io = O.cg(lambda x: Amat @ x, b, tol=tol, maxiter=maxiter)
    xg, ig = cg(lambda x: Ad @ x, dev(b), tol=tol, maxiter=maxiter,
                check_every=5)
    assert ig['num_iterations'] == io['num_iterations']
    assert relerr(xg, xo) < 1e-9
    assert float(ig['residual']) == pytest.approx(io['residual'], rel=1e-6)
  # preconditioned: Jacobi
  dinv = 1.0 / np.diag(Amat)
  xo, io = O.cg(lambda x: Amat @ x, b, tol=1e-10, M=lambda r: dinv * r)
  dd = dev(dinv)
  xg, ig = cg(lambda x: Ad @ x, dev(b), tol=1e-10, M=lambda r: dd * r)
  assert ig['num_iterations'] == io['num_iterations']
  assert relerr(xg, xo) < 1e-9
  # zero right-hand side: no iterations
  xg, ig = cg(lambda x: Ad @ x, dev(np.zeros(n)))
  assert ig['num_iterations'] == 0 and float(xg.abs().max()) == 0.0


def test_cg_reports_breakdown_instead_of_convergence():
  """The reference's stop rule `gamma > atol2` (linalg/cg.py:68-73) reads a
  negative r.Mr as converged and divides by any p.Ap; here the solve stops and
  `info['status']` says why.  SPD solves are unaffected ('converged' /
  'maxiter', same iterates as the oracle: test_cg_matches_oracle_iterates)."""
  from swirl_fem_amd.linalg.cg import cg
  b = dev(1.0 + np.arange(6.0))
  x, info = cg(lambda x: 3 * x, b)
  assert info['status'] == 'converged' and info['num_iterations'] == 1
  x, info = cg(lambda x: 3 * x + 0 * x.sum(), b, tol=0.0, maxiter=0)
  assert info['status'] == 'maxiter' and info['num_iterations'] == 0
  # negative definite operator: the reference divides by any p.Ap (cg.py:
  # 78-79) and converges; so does this solve, with the oracle's iterates
  x0 = dev(np.full(6, 0.25))
  Aneg = -np.diag([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
  x, info = cg(lambda x: dev(Aneg) @ x, b, x0=x0, tol=1e-12)
  xo, io = O.cg(lambda v: Aneg @ v, b.cpu().numpy(), x0=x0.cpu().numpy(),
                tol=1e-12)
  assert info['status'] == 'converged'
  assert info['num_iterations'] == io['num_iterations']
  assert relerr(x, xo) < 1e-12
  # p.Ap exactly zero: the solve stops before the update; x stays x0
  x, info = cg(lambda x: 0 * x, b, x0=x0)
  assert info['status'] == 'breakdown_pAp' and info['num_iterations'] == 0
  np.testing.assert_array_equal(x.cpu().numpy(), x0.cpu().numpy())
  # indefinite "preconditioner": r.Mr < 0 before the first iteration
  x, info = cg(lambda x: 2 * x, b, M=lambda r: -r)
  assert info['status'] == 'breakdown_gamma' and info['num_iterations'] == 0
  assert float(info['residual']) < 0
  # ... and one that turns negative after an iteration (M flips the sign of
  # the second residual): the iteration is counted, then the solve stops
  A = dev(np.diag([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))
  calls = []
  def flipping(r):
    calls.append(1)
    return r if len(calls) == 1 else -r
  x, info = cg(lambda x: A @ x, b, M=flipping, tol=1e-12)
  assert info['status'] == 'breakdown_gamma' and info['num_iterations'] == 1
  # NaN from the operator
  x, info = cg(lambda x: x * float('nan'), b)
  assert info['status'].startswith('breakdown')
  # the fused-dot path (one scalar launch per iteration) has the same guards
  rp = make_case(3, 2, 4, seed=5)
  mesh, fes, _ = spaces(rp, 4, 4, 'gll')
  op = fes.helmholtz_operator(mesh.physical_masks['boundary'])
  rhs = dev(np.random.default_rng(1).standard_normal(mesh.num_nodes)) * (
      ~mesh.physical_masks['boundary'])
  x, info = cg(op.linear_operator(0.0, 1.0), rhs, tol=1e-10)
  assert info['status'] == 'converged'
  xm, infom = cg(op.linear_operator(0.0, -1.0), rhs, tol=1e-10)   # -A
  assert infom['status'] == 'converged'
  assert infom['num_iterations'] == info['num_iterations']
  assert float((xm + x).abs().max()) <= 1e-12 * float(x.abs().max())
  x, info = cg(op.linear_operator(0.0, 0.0), rhs, tol=1e-10)      # zero
  assert info['status'] == 'breakdown_pAp' and info['num_iterations'] == 0
  assert float(x.abs().max()) == 0.0


def test_cg_with_fused_operator_dot():
  """The operator's in-kernel u.A(u) equals the separate dot; CG iterates and
  iteration counts with it match the oracle's CG on the same operator."""
  from swirl_fem_amd import _lib
  from swirl_fem_amd.linalg.cg import cg
  for ndim, n, P in [(3, 3, 4), (2, 5, 6), (3, 2, 8)]:
    rp = make_case(ndim, n, P, seed=21)
    mesh, fes, ofes = spaces(rp, P, P, 'gll')
    bmask = mesh.physical_masks['boundary'].cpu().numpy()
    op = fes.helmholtz_operator(mesh.physical_masks['boundary'])
    rng = np.random.default_rng(22)
    for nc in (1, ndim):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device=DEV)
      Au = op.apply(dev(uu), 0.3, 1.0, dot_out=parts)
      ref = _helmholtz_ref(ofes, uu, 0.3, 1.0, bmask)
      assert relerr(Au, ref) < 1e-10
      assert float(parts.sum()) == pytest.approx(float(np.vdot(uu, ref)),
                                                 rel=1e-11)
    interior = 1.0 - bmask
    b = interior * rng.standard_normal(mesh.num_nodes)
    A = op.linear_operator(0.2, 1.0)
    assert relerr(A(dev(b)), _helmholtz_ref(ofes, b, 0.2, 1.0, bmask)) < 1e-10
    for tol in (1e-6, 1e-12):
      xo, io = O.cg(lambda x: _helmholtz_ref(ofes, x, 0.2, 1.0, bmask), b,
                    tol=tol)
      xg, ig = cg(A, dev(b), tol=tol, check_every=7)
      assert ig['num_iterations'] == io['num_iterations']
      assert relerr(xg, xo) < 1e-8
      xs, is_ = cg(lambda x: op.apply(x, 0.2, 1.0), dev(b), tol=tol)
      assert is_['num_iterations'] == io['num_iterations']


# ------------------------------------------------------------------ Poisson
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_cg_update_groupings_agree(dtype):
  """9-pass (xr, p) and 8-pass (r, xp) groupings of the CG vector updates are
  the same arithmetic; odd lengths exercise the scalar tails."""
  from swirl_fem_amd import _lib, _ops
  g = torch.Generator(device=DEV).manual_seed(3)
  for n in (1, 7, 1000, 100003):
    v = [torch.randn(n, dtype=dtype, device=DEV, generator=g) for _ in range(4)]
    s = torch.zeros(_lib.SFEM_CG_NSCALARS, dtype=torch.float64, device=DEV)
    s[0], s[1] = 2.5, 1.7                      # gamma, p.Ap -> alpha = gamma / pAp
    # fuse 1: r.r by atomics on the named slot; 2: spread over the 64 partial
    # slots behind the named scalars (what the single-GPU solver uses)
    for fuse in (1, 2, 0):
      x1, r1, p1, ap = (t.clone() for t in v)
      x2, r2, p2 = x1.clone(), r1.clone(), p1.clone()
      s1, s2 = s.clone(), s.clone()
      _ops.cg_update_xr(x1, r1, p1, ap, s1, min(fuse, 1))
      _ops.cg_update_r(r2, ap, s2, fuse)
      if not fuse:                              # gamma_new from a separate dot
        s1[2] = s2[2] = float((r1.double() ** 2).sum())
      _ops.cg_update_p(p1, r1, s1)
      _ops.cg_update_xp(x2, p2, r2, s2)
      tol = 1e-14 if dtype == torch.float64 else 1e-6
      assert torch.equal(r1, r2) and torch.equal(x1, x2)
      # beta = gamma_new / gamma: the fused r.r is summed by atomics, so its
      # last bits (and with them p) depend on the arrival order
      assert float((p1 - p2).abs().max()) <= 100 * tol * float(p1.abs().max())
      alpha = 2.5 / 1.7
      assert float((x1 - (v[0] + alpha * v[2])).abs().max()) < tol * 10
      rr = float((r1.double() ** 2).sum())
      assert abs(float(s1[2]) - rr) < 1e-9 * n
      gamma_new = float(s2[2]) + float(s2[_lib.SFEM_CG_NSCALARS_NAMED:].sum())
      assert abs(gamma_new - rr) < 1e-9 * n
      assert (float(s2[2]) == 0.0) == (fuse == 2)
      # closing the iteration folds the partial sums into gamma and clears them
      _ops.cg_scalars(s2, 1, 10 ** 9, 0.0, 0.0)
      assert abs(float(s2[0]) - rr) < 1e-9 * n
      assert float(s2[_lib.SFEM_CG_NSCALARS_NAMED:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_cg_mean_projection_updates(dtype):
  """sfem_cg_update_r_mean / _xp_mean (z = r - (w.r / total) 1 never stored)
  against the same arithmetic in torch, two iterations in a row (the sums of
  the two parities), odd lengths for the scalar tails."""
  from swirl_fem_amd import _lib, _ops
  g = torch.Generator(device=DEV).manual_seed(5)
  tol = 1e-13 if dtype == torch.float64 else 2e-5
  for n in (2, 7, 1000, 100003):     # (n = 1: the projection leaves nothing)
    x, r, p = (torch.randn(n, dtype=dtype, device=DEV, generator=g)
               for _ in range(3))
    w = torch.rand(n, dtype=dtype, device=DEV, generator=g) + 0.5
    total = float(w.double().sum())
    s = torch.zeros(_lib.SFEM_CG_NSCALARS, dtype=torch.float64, device=DEV)
    sums = torch.zeros(_lib.SFEM_CG_MEAN_SUMS, dtype=torch.float64, device=DEV)
    gamma = 2.5
    s[0] = gamma
    xr, rr, pr = (t.double().clone() for t in (x, r, p))
    for it in range(3):
      ap = torch.randn(n, dtype=dtype, device=DEV, generator=g)
      pap = 1.7 + it
      s[1] = pap
      _ops.cg_scalars(s, 0, 10 ** 9, 0.0, 0.0)        # alpha = gamma / p.Ap
      _ops.cg_update_r_mean(r, ap, w, s, sums)
      _ops.cg_update_xp_mean(x, p, r, s, sums, total)
      alpha = gamma / pap
      rr = rr - alpha * ap.double()
      c = float((w.double() * rr).sum()) / total
      z = rr - c
      gamma_new = float((rr * z).sum())
      xr = xr + alpha * pr
      pr = z + (gamma_new / gamma) * pr
      scale = lambda t: float(t.abs().max()) + 1e-300
      assert float((r.double() - rr).abs().max()) <= tol * scale(rr)
      assert float((x.double() - xr).abs().max()) <= tol * scale(xr)
      assert float((p.double() - pr).abs().max()) <= 50 * tol * (
          scale(pr) + scale(rr)), (n, it)
      assert abs(float(s[12]) - c) <= tol * (abs(c) + 1e-300) * n ** 0.5 + 1e-300
      _ops.cg_scalars(s, 1, 10 ** 9, 0.0, 0.0)        # close: gamma <- r.z
      assert abs(float(s[0]) - gamma_new) <= 100 * tol * float((rr ** 2).sum())
      assert float(s[11]) == 0.0
      gamma = float(s[0])
      if dtype == torch.float32:        # continue from what the kernels hold
        xr, rr, pr = (t.double().clone() for t in (x, r, p))


def test_cg_folds_the_mean_projection(monkeypatch):
  """A preconditioner offering `mean_projection()` gives the same iterates,
  iteration count and solution as the same projection applied as a separate
  M(r), on the (singular) Neumann stiffness operator; with and without the
  operator's fused p . Ap and as a graph replay."""
  from swirl_fem_amd.linalg.cg import cg
  rp = make_case(3, 3, 5, seed=31)
  mesh, fes, _ = spaces(rp, 5, 5, 'gll')
  op = fes.helmholtz_operator(None)
  w = op.apply(torch.ones(mesh.num_nodes, dtype=torch.float64, device=DEV),
               1.0, 0.0)                              # B 1
  total = float(w.sum())

  class Projection:
    def __call__(self, r):
      return r - (torch.dot(w, r) / total)

    def mean_projection(self):
      return w, total

  class Plain:
    def __call__(self, r):
      return r - (torch.dot(w, r) / total)

  rng = np.random.default_rng(32)
  b = dev(rng.standard_normal(mesh.num_nodes))
  b = b - b.mean()                                     # compatible right side
  fused_dot = op.linear_operator(0.0, 1.0)
  assert hasattr(fused_dot, 'apply_with_dot')
  plain_op = lambda x: op.apply(x, 0.0, 1.0)
  for A in (fused_dot, plain_op):
    for tol in (1e-6, 1e-11):
      x0, i0 = cg(A, b, tol=tol, M=Plain())
      x1, i1 = cg(A, b, tol=tol, M=Projection(), check_every=5)
      assert i1['status'] == i0['status'] == 'converged'
      # the same recurrence up to rounding: at tol = 1e-11 (r.z ~ 1e-19 of
      # b.b) the last iterations may differ by one
      assert abs(i1['num_iterations'] - i0['num_iterations']) <= 2, (tol, i0,
                                                                     i1)
      # (r . z = r.r - c 1.r instead of a sum over r_i z_i: rounding-level
      # differences, amplified by the conditioning of the Neumann operator)
      close = 100 * tol + 1e-7
      assert relerr(x1, x0.cpu().numpy()) < close
      if i1['num_iterations'] == i0['num_iterations']:
        assert 0.5 < float(i1['residual']) / float(i0['residual']) < 2.0
      x2, i2 = cg(A, b, tol=tol, M=Projection(), graph=True)
      assert abs(i2['num_iterations'] - i0['num_iterations']) <= 2
      assert relerr(x2, x0.cpu().numpy()) < close
  monkeypatch.setenv('SFEM_FUSED_MEAN', '0')
  x3, i3 = cg(fused_dot, b, tol=1e-11, M=Projection())
  assert abs(i3['num_iterations'] - i0['num_iterations']) <= 2


def test_cg_runner_reuse_keeps_the_recorded_iteration():
  """`cg(..., graph=True, workspace=ws, key=k)`: the second and third solve
  with the same key restart the kept runner (no new recording) and return what
  a fresh solve returns; a different stopping rule or shape builds a new one."""
  from swirl_fem_amd.linalg import cg as cgmod
  from swirl_fem_amd.linalg.cg import cg
  rp = make_case(3, 3, 5, seed=41)
  mesh, fes, _ = spaces(rp, 5, 5, 'gll')
  bmask = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bmask)
  A = op.linear_operator(0.3, 1.0)
  interior = (~bmask).to(torch.float64)
  rng = np.random.default_rng(42)
  bs = [interior * dev(rng.standard_normal(mesh.num_nodes)) for _ in range(3)]
  ws = {}
  captures = []
  orig = cgmod.CGRunner.capture
  cgmod.CGRunner.capture = lambda self: (captures.append(1), orig(self))[1]
  try:
    for k, b in enumerate(bs):
      x0 = None if k < 2 else 0.5 * bs[0]
      want, iw = cg(A, b, x0, tol=1e-9)
      got, ig = cg(A, b, x0, tol=1e-9, graph=True, workspace=ws, key='A')
      assert ig['status'] == 'converged'
      assert ig['num_iterations'] == iw['num_iterations']
      assert relerr(got, want.cpu().numpy()) < 1e-9
      assert len(ws) == 1 and len(captures) == 1, (k, len(ws), captures)
    first = ws['A']
    assert got.data_ptr() != first.x.data_ptr()      # a copy: x is reused
    # another stopping rule: a new runner under the same key
    cg(A, bs[0], tol=1e-6, graph=True, workspace=ws, key='A')
    assert ws['A'] is not first and len(captures) == 2
    # without a key nothing is kept
    cg(A, bs[0], tol=1e-6, graph=True, workspace=ws)
    assert len(ws) == 1 and len(captures) == 3
    # a stepper with ever-changing coefficients does not pile up states
    for k in range(cgmod.MAX_KEPT_RUNNERS + 3):
      cg(A, bs[0], tol=1e-3, graph=True, workspace=ws, key=('dt', k))
    assert len(ws) == cgmod.MAX_KEPT_RUNNERS and 'A' not in ws
    # least recently USED goes first: a hit moves its key to the end
    keys = list(ws)
    cg(A, bs[0], tol=1e-3, graph=True, workspace=ws, key=keys[0])
    assert list(ws)[-1] == keys[0] and len(ws) == cgmod.MAX_KEPT_RUNNERS
    cg(A, bs[0], tol=1e-3, graph=True, workspace=ws, key=('dt', 'new'))
    assert keys[0] in ws and keys[1] not in ws
    # a kept runner takes a right-hand side of another memory layout: a
    # component-major (N, 3) field after a row-major one (same shape)
    from swirl_fem_amd.core import layout
    A3 = op.linear_operator(0.3, 1.0)
    b_rows = (interior[:, None] * dev(rng.standard_normal((mesh.num_nodes, 3))
                                      )).contiguous()
    b_cm = layout.component_major(b_rows.clone())
    assert b_rows.stride() != b_cm.stride() and b_rows.shape == b_cm.shape
    ws3 = {}
    x_cm, _ = cg(A3, b_cm, tol=1e-9, graph=True, workspace=ws3, key='v')
    kept = ws3['v']
    # a dense row-major b is repacked into the kept (component-major) state ...
    x_rows, _ = cg(A3, 2.0 * b_rows, tol=1e-9, graph=True, workspace=ws3,
                   key='v')
    assert ws3['v'] is kept                 # restarted, not rebuilt
    assert relerr(x_rows, 2.0 * x_cm.cpu().numpy()) < 1e-8
    # ... a strided b of another layout gets a state of its own
    ws4 = {}
    cg(A3, b_rows, tol=1e-9, graph=True, workspace=ws4, key='v')
    kept = ws4['v']
    x2, _ = cg(A3, b_cm, tol=1e-9, graph=True, workspace=ws4, key='v')
    assert ws4['v'] is not kept
    assert relerr(x2, x_cm.cpu().numpy()) < 1e-8
  finally:
    cgmod.CGRunner.capture = orig


def test_symmetric_solve_is_differentiable_in_b():
  """d/db of <w, A^-1 b> = A^-1 w (adjoint solve with the same operator)."""
  from swirl_fem_amd.linalg.cg import cg, symmetric_solve
  rp = make_case(2, 4, 4, jitter=0.1, seed=6)
  mesh, fes, _ = spaces(rp, 4, 4, 'gll')
  bm = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bm)
  A = lambda u: op.apply(u, 0.7, 1.0)
  rng = np.random.default_rng(7)
  keep = (~bm).to(torch.float64)
  b = (dev(rng.standard_normal(mesh.num_nodes)) * keep).requires_grad_(True)
  w = dev(rng.standard_normal(mesh.num_nodes)) * keep
  x = symmetric_solve(A, b, tol=1e-13, maxiter=2000)
  (x * w).sum().backward()
  want, _ = cg(A, w, tol=1e-13, maxiter=2000)
  assert float((b.grad - want).abs().max()) < 1e-9 * float(want.abs().max())


def test_cg_graph_replay_matches_eager():
  """One HIP graph launch per iteration == the eager launch sequence."""
  from swirl_fem_amd.linalg.cg import cg, CGRunner
  rp = make_case(3, 3, 5, jitter=0.1, seed=4)
  mesh, fes, _ = spaces(rp, 5, 5, 'gll')
  bm = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bm)
  rng = np.random.default_rng(8)
  b = dev(rng.standard_normal(mesh.num_nodes)) * (~bm)
  for A, M in ((op.linear_operator(0.5, 1.0), None),
               (lambda u: op.apply(u, 0.5, 1.0), lambda r: 0.5 * r)):
    x0, i0 = cg(A, b, tol=1e-12, maxiter=500, M=M)
    x1, i1 = cg(A, b, tol=1e-12, maxiter=500, M=M, graph=True)
    assert i0['num_iterations'] == i1['num_iterations'] > 10
    assert float((x0 - x1).abs().max()) < 1e-12 * float(x0.abs().max())
  run = CGRunner(op.linear_operator(0.5, 1.0), b, tol=1e-12, maxiter=500)
  assert run.capture() and run._graph is not None
  # an operator that synchronises cannot be captured: the solve stays eager
  def syncing(u):
    float(u[0])
    return op.apply(u, 0.5, 1.0)
  x3, i3 = cg(lambda u: op.apply(u, 0.5, 1.0), b, tol=1e-12, maxiter=500)
  x2, i2 = cg(syncing, b, tol=1e-12, maxiter=500, graph=True)
  assert i2['num_iterations'] == i3['num_iterations']
  assert float((x3 - x2).abs().max()) < 1e-12 * float(x3.abs().max())


def test_cg_single_scalar_launch_matches_the_three_phase_sequence():
  """With a fused p.Ap and no all-reduce an iteration has ONE scalar launch
  (phase 5: closes the previous iteration, then alpha); the bookkeeping seen
  by the host (`info`, `done`) must equal the three-phase sequence step by
  step, including the step at which convergence is flagged and maxiter."""
  from swirl_fem_amd.linalg.cg import CGRunner
  rp = make_case(2, 4, 5, jitter=0.1, seed=5)          # 2D: deterministic sums
  mesh, fes, _ = spaces(rp, 5, 5, 'gll')
  bm = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bm, assembly='colored')   # bitwise reproducible
  b = dev(np.random.default_rng(3).standard_normal(mesh.num_nodes)) * (~bm)
  A = op.linear_operator(0.3, 1.0)
  for maxiter in (10 ** 6, 7):
    one = CGRunner(A, b, tol=1e-9, maxiter=maxiter)
    three = CGRunner(A, b, tol=1e-9, maxiter=maxiter, reduce_fn=lambda t: t)
    assert one.fused_dot and three.fused_dot
    for step in range(200):
      one.step(); three.step()
      if step % 3 == 0 or step > 40:                    # polls interleave
        i1, i3 = one.info(), three.info()
        assert i1['num_iterations'] == i3['num_iterations']
        assert abs(float(i1['residual']) - float(i3['residual'])) <= 1e-13 * abs(
            float(i3['residual']))
        assert one.done() == three.done()
      if one.done() and three.done():
        break
    assert one.done() and three.done()
    assert one.info()['num_iterations'] == three.info()['num_iterations'] == (
        7 if maxiter == 7 else one.info()['num_iterations'])
    assert float((one.x - three.x).abs().max()) <= 1e-13 * float(
        three.x.abs().max())


def test_poisson_config1_matches_oracle_and_series():
  """BASELINE config 1: 2D Poisson, 16x16 quads on [-1,1]^2, p=3."""
  from swirl_fem_amd.examples.poisson import BCType, solve_poisson
  P = 4
  pm = unit_cube_mesh(16, ndim=2, a=-1.0, b=1.0)
  rp = refine_premesh(pm, Nodes1D.create(P, NT['gll']))
  mesh = rp.finalize(device=DEV)
  assert mesh.num_nodes == 2401
  f = np.ones(mesh.num_nodes)
  u, info = solve_poisson(mesh, dev(f), {'boundary': (BCType.DIRICHLET, 0.)},
                          rtol=1e-12, return_info=True)
  bmask = mesh.physical_masks['boundary'].cpu().numpy()
  uo, io = O.solve_poisson(rp.node_coords, rp.elements, (P, 'gll'), bmask, f,
                           rtol=1e-12, return_ops=True)[:2]
  assert relerr(u, uo) < 1e-9
  assert abs(info['num_iterations'] - io['num_iterations']) <= 2
  x = rp.node_coords
  s = np.zeros(len(x))
  for k in range(1, 40, 2):
    s += (1 / (k ** 3 * np.sinh(k * np.pi))) * np.sin(
        k * np.pi * (1 + x[:, 0]) / 2) * (
            np.sinh(k * np.pi * (1 - x[:, 1]) / 2) +
            np.sinh(k * np.pi * (1 + x[:, 1]) / 2))
  exact = (1 - x[:, 0] ** 2) / 2 - (16 / np.pi ** 3) * s
  assert np.abs(u.cpu().numpy() - exact).max() < 2e-5
  # default tolerance path of the reference (rtol 1e-5)
  u5 = solve_poisson(mesh, dev(f), {'boundary': (BCType.DIRICHLET, 0.)})
  assert np.abs(u5.cpu().numpy() - exact).max() < 1e-4
  with pytest.raises(NotImplementedError):
    solve_poisson(mesh, dev(f), {'boundary': (BCType.DIRICHLET, 1.)})


def test_poisson_reference_1d_and_circle():
  from swirl_fem_amd.examples.poisson import BCType, solve_poisson
  nn = 33
  coords = np.linspace(0, 1, nn).reshape(nn, 1)
  elements = np.array([[i, i + 1] for i in range(32)])
  mesh = Premesh.create(coords, elements, physical_groups={
      'boundary': np.array([[0, nn - 1]], np.int32)}).finalize(device=DEV)
  bc = {'boundary': (BCType.DIRICHLET, 0.)}
  u = solve_poisson(mesh, dev(np.ones(nn)), bc)
  np.testing.assert_allclose(u.cpu().numpy(),
                             .5 * (coords[:, 0] - coords[:, 0] ** 2),
                             rtol=1e-6, atol=1e-12)
  u = solve_poisson(mesh, dev(6 * coords[:, 0]), bc)
  np.testing.assert_allclose(u.cpu().numpy(), coords[:, 0] - coords[:, 0] ** 3,
                             rtol=1e-6, atol=1e-12)
  pm = unit_cube_mesh(32, ndim=2, a=-1.0, b=1.0)
  x = pm.node_coords
  r2 = 1 / np.sqrt(2)
  xc = np.stack([
      x[:, 0] * (np.cos(np.pi * x[:, 1] / 4) - r2) + np.sin(np.pi * x[:, 0] / 4),
      x[:, 1] * (np.cos(np.pi * x[:, 0] / 4) - r2) + np.sin(np.pi * x[:, 1] / 4)],
                axis=-1)
  mesh = pm.replace(node_coords=xc).finalize(device=DEV)
  u = solve_poisson(mesh, dev(np.ones(mesh.num_nodes)), bc)
  np.testing.assert_allclose(u.cpu().numpy(), .25 * (1 - (xc ** 2).sum(-1)),
                             rtol=1e-6, atol=1e-4)


# -------------------------------------------- full-size properties (config 2)
@pytest.mark.parametrize('n', [int(os.environ.get('SFEM_TEST_FULL_N', '64'))])
@pytest.mark.parametrize('jitter', [0.2, 0.0])
def test_config2_properties_full_size(n, jitter):
  """3D p=7 Laplacian at bench scale: symmetry, nullspace, linearity,
  fused == generic path, assembled == deterministic assembly == stored
  factors.  `jitter = 0.2`: deformed mesh, multilinear chain kernel;
  `jitter = 0`: the Cartesian mesh bench.py times, i.e. the box chain kernel
  `helmholtz_chain_kernel<double, 8, BoxElem<double, 8, false>>` on segments
  of 8 (asserted by name and by the chain list)."""
  P = 8
  rng = np.random.default_rng(18)
  pm = unit_cube_mesh(n, ndim=3)
  h = 1.0 / n
  if jitter:
    pm = pm.replace(node_coords=pm.node_coords + jitter * h *
                    rng.uniform(-1, 1, pm.node_coords.shape))
  grid = Nodes1D.create(P, NT['gll'])
  mesh = refine_premesh(pm, grid).finalize(device=DEV)
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
  op = fes.helmholtz_operator(None)
  from swirl_fem_amd.core import operators
  assert len(op.facet_parts) == 1
  elem = ('sfem::FacetElem<double, 8, 3, false>' if jitter
          else 'sfem::BoxElem<double, 8, false>')
  if operators.chain_segment_length(mesh.num_elements) > 1:
    assert op.kernel_name(0.0, 1.0) == (
        'sfem::helmholtz_chain_kernel<double, 8, %s, ' % elem)
    off = op.facet_parts[0]['chains'][0]
    seg = operators.chain_segment_length(mesh.num_elements)
    assert int((off[1:] - off[:-1]).max()) == min(seg, n)
    if n >= 64:
      assert seg == 8 and bool(((off[1:] - off[:-1]) == 8).all())
  g = torch.Generator(device=DEV).manual_seed(0)
  u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=DEV, generator=g)
  v = torch.randn(mesh.num_nodes, dtype=torch.float64, device=DEV, generator=g)
  Au, Av = op.apply(u), op.apply(v)
  scale = float(Au.abs().max())
  # symmetric
  assert abs(float(torch.dot(Au, v) - torch.dot(u, Av))) < 1e-9 * abs(
      float(torch.dot(Au, v)))
  # constants in the nullspace, linearity
  assert float(op.apply(torch.ones_like(u)).abs().max()) < 1e-10 * scale
  lin = op.apply(2.0 * u - 3.0 * v) - (2.0 * Au - 3.0 * Av)
  assert float(lin.abs().max()) < 1e-11 * scale
  # fused kernel == generic (basis_eval / basis_eval_t) path == CSR assembly
  uf = fes.scalar_function(mesh.gather(u))
  loc = fes.local_covector(
      lambda a, b: lambda x: torch.vdot(grad(a)(x), grad(b)(x)),
      (uf, fes.scalar_function(None)))
  gen = mesh.scatter(loc)
  assert float((gen - Au).abs().max()) < 1e-10 * scale
  from swirl_fem_amd import _ops
  off, slots = mesh.assembly_plan().csr()
  det = _ops.scatter_csr(loc, off, slots, mesh.num_nodes)
  assert float((det - Au).abs().max()) < 1e-10 * scale
  loc2 = op.apply_local(mesh.gather(u), 0.0, 1.0)
  assert float((loc2 - loc).abs().max()) < 1e-10 * float(loc.abs().max())
  del loc, loc2, gen, det, uf
  # stored-factor path == on-the-fly geometry; fused p.Ap == dot(p, Ap)
  op_s = fes.helmholtz_operator(None, 'stored')
  assert float((op_s.apply(u) - Au).abs().max()) < 1e-10 * scale
  from swirl_fem_amd import _lib
  parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device=DEV)
  op.apply(u, dot_out=parts)
  ref = float(torch.dot(u, Au))
  assert abs(float(parts.sum()) - ref) < 1e-10 * abs(ref)
  # a few CG iterations reduce the energy-norm error monotonically: solve
  # A x = A x* on the Dirichlet problem and watch |x - x*|_A
  bm = mesh.physical_masks['boundary']
  op_d = fes.helmholtz_operator(bm)
  assert op_d.kernel_name(0.0, 1.0) == op.kernel_name(0.0, 1.0)
  # ... with the Dirichlet rows: chain kernel == index-row kernel
  os.environ['SFEM_FACET'] = '0'
  try:
    op_rows = operators.HelmholtzOperator.create(fes, bm)   # (not the cached)
  finally:
    del os.environ['SFEM_FACET']
  assert op_rows.facet_parts is None
  assert float((op_rows.apply(u) - op_d.apply(u)).abs().max()) < 1e-11 * scale
  del op_rows
  xs = u * (~bm)
  from swirl_fem_amd.linalg.cg import CGRunner
  run = CGRunner(op_d.linear_operator(0.0, 1.0), op_d.apply(xs), tol=0.0,
                 maxiter=10 ** 6)
  errs = []
  for _ in range(6):
    run.step()
    e = run.x - xs
    errs.append(float(torch.dot(e, op_d.apply(e))))
  assert all(b < a for a, b in zip(errs, errs[1:])), errs


@pytest.mark.parametrize('n', [int(os.environ.get('SFEM_TEST_P11_N', '64'))])
def test_config5_properties_at_scale(n):
  """p = 11 fp32 Helmholtz on an n^3 block, by default config 5's full per-GPU
  block of 64^3 elements (350 M DOFs, 20 s on one MI355X): symmetry, constants in the nullspace of the stiffness
  part, on-the-fly geometry == stored factors, fused p.Ap, within the fp32
  tolerance of the north star (1e-5 relative, scaled by the operator norm)."""
  from swirl_fem_amd import _lib
  from swirl_fem_amd.distributed import blocks
  P = 12
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=DEV,
                                      dtype=torch.float32)
  mesh = part.mesh
  grid = Nodes1D.create(P, NT['gll'])
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
  op = fes.helmholtz_operator(None)
  g = torch.Generator(device=DEV).manual_seed(5)
  u = torch.randn(mesh.num_nodes, dtype=torch.float32, device=DEV, generator=g)
  v = torch.randn(mesh.num_nodes, dtype=torch.float32, device=DEV, generator=g)
  Hu, Hv = op.apply(u, 0.5, 1.0), op.apply(v, 0.5, 1.0)
  scale = float(Hu.abs().max())
  a, b = float(torch.dot(Hu.double(), v.double())), float(
      torch.dot(u.double(), Hv.double()))
  assert abs(a - b) < 1e-5 * float(Hu.double().norm() * v.double().norm())
  ones = torch.ones_like(u)
  assert float(op.apply(ones, 0.0, 1.0).abs().max()) < 2e-5 * scale
  # stored factors come from fp32 coordinate differences over elements of
  # width 1/n: their own rounding (eps |x| / h) is what separates the two paths
  op_s = fes.helmholtz_operator(None, 'stored')
  assert float((op_s.apply(u, 0.5, 1.0) - Hu).abs().max()) < 2e-5 * n * scale
  parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64, device=DEV)
  op.apply(u, 0.5, 1.0, dot_out=parts)
  ref = float(torch.dot(u.double(), Hu.double()))
  assert abs(float(parts.sum()) - ref) < 1e-5 * abs(ref)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('m', [4, 2, 3])
def test_cg_lazy_solution_update_is_bitwise_the_same(dtype, m, monkeypatch):
  """`sfem_cg_update_xp_lazy`: x takes its terms every m-th iteration only,
  in the order and with the roundings of `x += alpha p` (linalg/cg.py:80) --
  bit for bit the x of the iteration-by-iteration update after ANY number of
  iterations (the pending terms are added when x is read), same iteration
  count, also across `restart` and when somebody peeks at x mid-batch."""
  from swirl_fem_amd.linalg import cg as cg_mod
  # (two separate solves are only bitwise comparable when nothing in them
  # depends on the arrival order of atomics: a P = 6 mesh takes the layered
  # assembly and the stored partial sums of `CGRunner.det`)
  monkeypatch.setenv('SFEM_CHAIN_LEN', '3')
  rp = make_case(3, 3, 6, seed=3)
  mesh, fes, _ = spaces(rp, 6, 6, 'gll', dtype)
  bmask = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bmask)
  A = op.linear_operator(0.3, 1.0)
  g = torch.Generator(device=DEV).manual_seed(2)
  b = torch.randn(mesh.num_nodes, dtype=dtype, device=DEV, generator=g) * ~bmask
  monkeypatch.setenv('SFEM_LAZY_X', '0')
  plain = cg_mod.CGRunner(A, b, tol=0.0, maxiter=10 ** 6)
  assert plain.lazy is None and plain.vector_passes == 8
  assert plain.layered is not None and plain.det is not None
  xs = []
  for _ in range(11):
    plain.step()
    xs.append(plain.x.clone())
  monkeypatch.setenv('SFEM_LAZY_X', str(m))
  monkeypatch.setenv('SFEM_LAZY_X_MIN_MB', '0')
  for k in range(1, 12):          # k iterations, x read once at the end
    run = cg_mod.CGRunner(A, b, tol=0.0, maxiter=10 ** 6)
    assert run.lazy is not None and run.lazy[0].shape[0] == m
    assert abs(run.vector_passes - (3 + (4 * m + 1) / m)) < 1e-12
    for _ in range(k):
      run.step()
    assert torch.equal(run.x, xs[k - 1]), k
    assert torch.equal(run.x, xs[k - 1])            # reading twice adds nothing
  run = cg_mod.CGRunner(A, b, tol=0.0, maxiter=10 ** 6)
  for k in range(11):             # peeking every iteration
    run.step()
    assert torch.equal(run.x, xs[k]), k
    assert torch.equal(run.p, plain.p) or k < 10
  # whole solves, and the state reused for a second right-hand side
  tol = 1e-9 if dtype == torch.float64 else 1e-5
  ws = {}
  x1, i1 = cg_mod.cg(A, b, tol=tol)
  monkeypatch.setenv('SFEM_LAZY_X', '0')
  x0, i0 = cg_mod.cg(A, b, tol=tol)
  assert torch.equal(x1, x0) and i1['num_iterations'] == i0['num_iterations']
  assert i1['status'] == 'converged'
  b2 = torch.randn(mesh.num_nodes, dtype=dtype, device=DEV, generator=g) * ~bmask
  y0, j0 = cg_mod.cg(A, b2, tol=tol)
  monkeypatch.setenv('SFEM_LAZY_X', str(m))
  run = cg_mod.CGRunner(A, b, tol=tol)
  while not run.done():
    for _ in range(5):
      run.step()
  assert torch.equal(run.x, x0)
  run.restart(b2)
  while not run.done():
    for _ in range(7):
      run.step()
  assert torch.equal(run.x, y0)
  assert run.info()['num_iterations'] == j0['num_iterations']
  # graph replay keeps fixed operands: the ring is dropped before recording
  run = cg_mod.CGRunner(A, b, tol=tol)
  assert run.lazy is not None and run.capture() and run.lazy is None
  while not run.done():
    run.step()
  assert torch.equal(run.x, x0)
