"""Pins the CPU oracle (oracle/sfem_oracle.py).

(1) bit-for-bit against arrays produced by the reference's own NumPy code
    (tests/golden/*.npz, see tests/golden/make_golden.py);
(2) against the exact / analytic known answers held by the reference's tests:
    core/gather_scatter_test.py:50-263, core/premesh_test.py:79-176,243-376,
    core/fespace_test.py:57-242, core/interpolation_test.py:264-281,
    examples/poisson_test.py:95-183, linalg/cg_test.py:26-50.
"""
import os
import re

import numpy as np
import pytest
import scipy.integrate

from oracle import sfem_oracle as O


@pytest.fixture(scope='module')
def g1d(golden_dir):
  return np.load(os.path.join(golden_dir, 'interp1d.npz'))


# ---------------------------------------------------------------- (1) goldens
def test_1d_nodes_weights_matrices_bitexact(g1d):
  checked = 0
  for key in g1d.files:
    m = re.match(r'(nc|gl|gll)(\d+)_(nodes|weights|bary|D)$', key)
    if m:
      nt, p, what = m.group(1), int(m.group(2)), m.group(3)
      mine = {'nodes': lambda: O.nodes_1d(p, nt),
              'weights': lambda: O.quadrature_weights(p, nt),
              'bary': lambda: O.barycentric_weights(p, nt),
              'D': lambda: O.differentiation_matrix_1d(O.nodes_1d(p, nt), nt)
              }[what]()
      np.testing.assert_array_equal(mine, g1d[key], err_msg=key)
      checked += 1
    m = re.match(r'I_(nc|gl|gll)(\d+)_(nc|gl|gll)(\d+)$', key)
    if m:
      gn, gp, en, ep = m.group(1), int(m.group(2)), m.group(3), int(m.group(4))
      mine = O.interpolation_matrix_1d(O.nodes_1d(gp, gn), gn,
                                       O.nodes_1d(ep, en))
      np.testing.assert_array_equal(mine, g1d[key], err_msg=key)
      checked += 1
  assert checked > 200


def test_kron_matrices_bitexact(g1d):
  for (d, gp, ep) in [(2, 3, 4), (3, 3, 3), (3, 2, 3)]:
    it = O.Interpolator(d, gp, 'gll', ep, 'gl')
    np.testing.assert_array_equal(it.interpolation_matrix(),
                                  g1d[f'kron_M_d{d}_gll{gp}_gl{ep}'])
    np.testing.assert_array_equal(it.interpolation_matrix_grad(),
                                  g1d[f'kron_G_d{d}_gll{gp}_gl{ep}'])
  np.testing.assert_array_equal(
      O.weights_nd(O.quadrature_weights(4, 'gll'), 3), g1d['weights_nd_gll4_d3'])


def test_bdf_ext_coeffs(g1d):
  for k in range(1, 5):
    np.testing.assert_array_equal(O.extk_coeffs(k), g1d[f'ext{k}_M'])
    np.testing.assert_array_equal(O.bdfk_coeffs(k), g1d[f'bdf{k}_G'])
  # interpolation_test.py:264-281 known values
  np.testing.assert_allclose(O.bdfk_coeffs(1), [-1, 1], atol=1e-12)
  np.testing.assert_allclose(O.bdfk_coeffs(2), [1 / 2, -2, 3 / 2], atol=1e-12)
  np.testing.assert_allclose(O.bdfk_coeffs(3), [-1 / 3, 3 / 2, -3, 11 / 6],
                             atol=1e-12)
  np.testing.assert_allclose(O.bdfk_coeffs(4),
                             [1 / 4, -4 / 3, 3, -4, 25 / 12], atol=1e-12)
  np.testing.assert_allclose(O.extk_coeffs(1), [-1, 2], atol=1e-12)
  np.testing.assert_allclose(O.extk_coeffs(2), [1, -3, 3], atol=1e-12)


def test_index_builders_match_reference(golden_dir):
  g = np.load(os.path.join(golden_dir, 'meshes.npz'))
  names = sorted({k.split('/')[0] for k in g.files})
  checked = 0
  for name in names:
    if name + '/final/node_indices' not in g.files:
      continue
    ref_links = (g[name + '/refined/periodic_links']
                 if name + '/refined/periodic_links' in g.files else None)
    if name + '/final/local_nodes' in g.files:
      base = g[name + '/final/local_nodes']
    else:
      base = np.arange(len(g[name + '/refined/node_coords']), dtype=np.int32)
    ni = O.get_unique_node_indices(base, ref_links)
    np.testing.assert_array_equal(ni, g[name + '/final/node_indices'], name)
    gi, ui = O.get_exchange_indices(ni)
    np.testing.assert_array_equal(gi, g[name + '/final/gather_indices'], name)
    if ui is not None:
      np.testing.assert_array_equal(ui, g[name + '/final/unique_indices'], name)
    checked += 1
  assert checked >= 15


# ---------------------------------------- (2a) gather_scatter_test.py answers
def test_exchange_noop_and_periodic():
  gi, ui = O.get_exchange_indices(np.arange(3, dtype=np.int32))
  assert gi.shape == (0,) and ui.shape == (0,) and gi.dtype == np.int32
  np.testing.assert_array_equal(
      O.exchange_unpartitioned(np.arange(3.), gi, ui), np.arange(3.))
  ni = O.get_unique_node_indices(np.arange(3, dtype=np.int32),
                                 np.array([[[0], [2]]]))
  gi, ui = O.get_exchange_indices(ni)
  assert gi.shape == (2,) and ui.shape == (2,)
  np.testing.assert_allclose(
      O.exchange_unpartitioned(np.array([1., 2., 3.]), gi, ui), [4., 2., 4.])


def test_exchange_doubly_periodic():
  links = np.array([[[0, 1], [6, 7]], [[1, 2], [7, 8]], [[0, 3], [2, 5]],
                    [[3, 6], [5, 8]]], dtype=np.int32)
  ni = O.get_unique_node_indices(np.arange(9, dtype=np.int32), links)
  gi, ui = O.get_exchange_indices(ni)
  assert sorted(np.bincount(gi)[np.bincount(gi) > 0]) == [1] * 8
  assert sorted(np.bincount(ui)) == [2, 2, 4]
  np.testing.assert_allclose(
      O.exchange_unpartitioned(np.arange(9.), gi, ui),
      [16., 8., 16., 8., 4., 8., 16., 8., 16.])


PART_NI = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 8]], dtype=np.int32)


def test_partitioned_gather_indices_and_exchange():
  gi, ui = O.get_exchange_indices(PART_NI)
  np.testing.assert_array_equal(
      gi, [[2, -1, -1], [0, 2, -1], [-1, 0, 2], [-1, -1, 0]])
  assert ui is None
  u = np.arange(12.).reshape(4, 3)
  np.testing.assert_array_equal(
      O.exchange_partitioned(u, gi),
      [[0, 1, 5], [5, 4, 11], [11, 7, 17], [17, 10, 11]])


def test_partitioned_periodic():
  ni = O.get_unique_node_indices(PART_NI, np.array([[[0], [8]]]))
  gi, ui = O.get_exchange_indices(ni)
  np.testing.assert_array_equal(
      gi, [[0, 2, -1, -1], [-1, 0, 2, -1], [-1, -1, 0, 2], [2, -1, -1, 0]])
  u = np.arange(12.).reshape(4, 3)
  np.testing.assert_array_equal(
      O.exchange_partitioned(u, gi),
      [[11, 1, 5], [5, 4, 11], [11, 7, 17], [17, 10, 11]])


def test_partitioned_doubly_periodic_and_not_implemented():
  ni = np.array([[0, 1, 3, 4], [1, 2, 4, 5], [3, 4, 6, 7], [4, 5, 7, 8]],
                dtype=np.int32)
  links = np.array([[[0, 1], [6, 7]], [[1, 2], [7, 8]], [[0, 3], [2, 5]],
                    [[3, 6], [5, 8]]], dtype=np.int32)
  gi, _ = O.get_exchange_indices(O.get_unique_node_indices(ni, links))
  np.testing.assert_array_equal(O.exchange_partitioned(np.ones((4, 4)), gi),
                                4 * np.ones((4, 4)))
  ni = np.array([[0, 1, 5, 6], [1, 2, 6, 7], [2, 3, 7, 8], [3, 4, 8, 9]],
                dtype=np.int32)
  links = np.array([[[0, 1], [5, 6]], [[1, 2], [6, 7]], [[2, 3], [7, 8]],
                    [[3, 4], [8, 9]]], dtype=np.int32)
  with pytest.raises(NotImplementedError, match='more than once'):
    O.get_exchange_indices(O.get_unique_node_indices(ni, links))


def test_premesh_partitioned_exchange_with_padding():
  # premesh_test.py:286-317: node_indices with -1 padding
  ni = np.array([[0, 1, 2], [2, 3, 4], [4, 5, -1], [5, 6, -1]])
  gi, _ = O.get_exchange_indices(ni)
  u = np.arange(7.)
  u_p = np.stack([O.gather(u, row, 0.) for row in ni])
  np.testing.assert_array_equal(
      u_p, [[0, 1, 2], [2, 3, 4], [4, 5, 0], [5, 6, 0]])
  np.testing.assert_array_equal(
      O.exchange_partitioned(u_p, gi),
      [[0, 1, 4], [4, 3, 8], [8, 10, 0], [10, 6, 0]])
  # premesh_test.py:319-345: gather with padded (all -1) elements
  local_elements = np.array([[[0, 1], [1, 2]], [[0, 1], [1, 2]],
                             [[0, 1], [-1, -1]], [[0, 1], [-1, -1]]])
  u_local = np.stack([O.gather(u_p[p], local_elements[p], 0.)
                      for p in range(4)])
  np.testing.assert_array_equal(
      u_local, [[[0, 1], [1, 2]], [[2, 3], [3, 4]], [[4, 5], [0, 0]],
                [[5, 6], [0, 0]]])


def test_scatter_sentinel():
  idx = np.array([[0, 1], [1, -1]])
  out = O.scatter(np.array([[1., 2.], [3., 4.]]), idx, 3)
  np.testing.assert_array_equal(out, [1., 5., 0.])


# -------------------------------------------------- (2b) fespace_test answers
def _single_element(ndim, order):
  n = (order + 1) ** ndim
  c1 = np.linspace(0, 1, order + 1)
  coords = np.stack(np.meshgrid(*([c1] * ndim), indexing='ij'),
                    axis=-1).reshape(n, ndim)
  return coords, np.arange(n).reshape(1, n)


@pytest.mark.parametrize('ndim', [1, 2, 3])
@pytest.mark.parametrize('order', [1, 2, 3, 4])
def test_integrate_single_element(ndim, order):
  coords, elements = _single_element(ndim, order)
  fes = O.FESpace(coords, elements, (order + 1, 'nc'), (order + 1, 'gl'))
  f = lambda x: sum(x[..., i] ** order for i in range(ndim))
  u_local = f(coords[elements])
  assert fes.integrate(fes.value(u_local)) == pytest.approx(ndim / (1 + order),
                                                            abs=1e-7)
  assert fes.integrate(f(fes.quad_coords)) == pytest.approx(
      ndim / (1 + order), abs=1e-7)
  assert fes.integrate(fes.grad(u_local)[..., 0]) == pytest.approx(1., abs=1e-7)


def test_generic_quad_grad_div():
  coords = np.array([[0, 0], [0, 1], [1, 0], [1, 2]], dtype=np.float64)
  elements = np.arange(4).reshape(1, 4)
  fes = O.FESpace(coords, elements, (2, 'nc'), (2, 'gl'))
  xe = coords[elements]
  g = fes.grad(2 * xe[..., 0] - xe[..., 1] + 1)
  assert fes.integrate(g[..., 0]) == pytest.approx(3.0, abs=1e-12)
  assert fes.integrate(g[..., 1]) == pytest.approx(-1.5, abs=1e-12)
  v = np.stack([2 * xe[..., 0] - xe[..., 1], 3 * xe[..., 1]], axis=-1)
  div = np.einsum('mqjj->mq', fes.grad(v))
  assert fes.integrate(div) == pytest.approx(7.5, abs=1e-12)


def test_unit_interval_nodal():
  for order in (1, 2, 3):
    ne = 8
    nn = 1 + order * ne
    coords = np.linspace(0, 1, nn).reshape(nn, 1)
    elements = np.array([list(range(order * i, order * (i + 1) + 1))
                         for i in range(ne)])
    fes = O.FESpace(coords, elements, (order + 1, 'nc'), (order + 1, 'gl'))
    xe = coords[elements][..., 0]
    assert fes.integrate(fes.value(np.ones_like(xe))) == pytest.approx(1.)
    assert fes.integrate(fes.value(xe ** order)) == pytest.approx(
        1 / (order + 1))
  # fespace_test.py:57-71
  coords = np.linspace(0, 1, 9).reshape(9, 1)
  elements = np.array([[i, i + 1] for i in range(8)])
  fes = O.FESpace(coords, elements, (2, 'nc'), (4, 'gl'))
  assert fes.integrate(1 + 5 * fes.quad_coords[..., 0]) == pytest.approx(3.5)


@pytest.mark.parametrize('order', [1, 2])
def test_covector_single_element_2d(order):
  coords, elements = _single_element(2, order)
  fes = O.FESpace(coords, elements, (order + 1, 'nc'), (order + 1, 'gl'))
  f = lambda x: (x[0] + 2 * x[1]) ** (order // 2)
  g = lambda x: (3 * x[0] - x[1]) ** ((order + 1) // 2)
  xe = coords[elements]
  f_local = f([xe[..., 0], xe[..., 1]]) * np.ones(xe.shape[:-1])
  g_local = g([xe[..., 0], xe[..., 1]]) * np.ones(xe.shape[:-1])
  cov = fes.mass_local(f_local)
  expected, _ = scipy.integrate.dblquad(
      lambda y, x: f([x, y]) * g([x, y]), 0, 1, lambda x: 0, lambda x: 1)
  got = np.vdot(fes.scatter(cov), fes.scatter(g_local))
  assert got == pytest.approx(expected, abs=1e-7)


def test_covector_is_the_transpose_of_integrate():
  """covector(c0, c1) . v == integrate(c0 v + c1 . grad v) for random data."""
  rng = np.random.default_rng(0)
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  for ndim, p, q, qt in [(2, 4, 5, 'gl'), (3, 3, 3, 'gll'), (3, 3, 5, 'gll')]:
    pm = unit_cube_mesh(2, ndim=ndim)
    pm = pm.replace(node_coords=pm.node_coords +
                    0.1 * rng.standard_normal(pm.node_coords.shape))
    rp = refine_premesh(pm, Nodes1D.create(p, NodeType.GAUSS_LOBATTO_LEGENDRE))
    fes = O.FESpace(rp.node_coords, rp.elements, (p, 'gll'), (q, qt))
    E, n, Q = fes.num_elements, fes.n, fes.Q
    v = rng.standard_normal((E, n))
    c0 = rng.standard_normal((E, Q))
    c1 = rng.standard_normal((E, Q, ndim))
    lhs = np.vdot(fes.covector(c0, c1), v)
    rhs = fes.integrate(c0 * fes.value(v) +
                        np.einsum('mqj,mqj->mq', c1, fes.grad(v)))
    assert lhs == pytest.approx(rhs, rel=1e-12)
    vv = rng.standard_normal((E, n, ndim))
    c0 = rng.standard_normal((E, Q, ndim))
    c1 = rng.standard_normal((E, Q, ndim, ndim))
    lhs = np.vdot(fes.covector(c0, c1), vv)
    rhs = fes.integrate(np.einsum('mqk,mqk->mq', c0, fes.value(vv)) +
                        np.einsum('mqjk,mqjk->mq', c1, fes.grad(vv)))
    assert lhs == pytest.approx(rhs, rel=1e-12)


# ------------------------------------------------------------ (2c) cg_test.py
def test_cg_known_answers():
  b = np.arange(9.0).reshape(3, 3)
  x, info = O.cg(lambda x: 2 * x, b)
  np.testing.assert_allclose(x, b / 2)
  A = lambda x: {'a': x['a'] + 0.5 * x['b'], 'b': 0.5 * x['a'] + x['b']}
  x, _ = O.cg(A, {'a': np.asarray(1.0), 'b': np.asarray(-4.0)})
  assert x['a'] == pytest.approx(4.0, abs=1e-6)
  assert x['b'] == pytest.approx(-6.0, abs=1e-6)
  A = lambda x: np.array([2 * x[0], 0 * x[1]])
  M = lambda x: np.array([x[0], 0.])
  x, _ = O.cg(A, 1 + np.arange(2.0), M=M)
  np.testing.assert_allclose(x, [0.5, 0.])


# ------------------------------------------------------ (2d) poisson_test.py
def _line_mesh(ne):
  nn = ne + 1
  coords = np.linspace(0, 1, nn).reshape(nn, 1)
  elements = np.array([[i, i + 1] for i in range(ne)])
  mask = np.zeros(nn, dtype=bool)
  mask[[0, nn - 1]] = True
  return coords, elements, mask


def test_poisson_1d():
  coords, elements, mask = _line_mesh(32)
  u = O.solve_poisson(coords, elements, (2, 'nc'), mask, np.ones(33))
  np.testing.assert_allclose(u, .5 * (coords[:, 0] - coords[:, 0] ** 2),
                             rtol=1e-6, atol=1e-12)
  u = O.solve_poisson(coords, elements, (2, 'nc'), mask, 6 * coords[:, 0])
  np.testing.assert_allclose(u, coords[:, 0] - coords[:, 0] ** 3, rtol=1e-6,
                             atol=1e-12)
  coords, elements, mask = _line_mesh(128)
  f = -.5 * np.pi ** 2 * np.cos(np.pi * coords[:, 0])
  u = O.solve_poisson(coords, elements, (2, 'nc'), np.zeros(129, bool), f,
                      rtol=1e-7)
  np.testing.assert_allclose(u, np.sin(.5 * np.pi * coords[:, 0]) ** 2 - .5,
                             rtol=1e-4, atol=1e-5)


def _series(x, num_terms=5):
  s = np.zeros_like(x[..., 0])
  for k in range(1, 2 * num_terms, 2):
    s += (1 / (k ** 3 * np.sinh(k * np.pi))) * np.sin(
        k * np.pi * (1 + x[..., 0]) / 2) * (
            np.sinh(k * np.pi * (1 - x[..., 1]) / 2) +
            np.sinh(k * np.pi * (1 + x[..., 1]) / 2))
  return (1 - x[..., 0] ** 2) / 2 - (16 / np.pi ** 3) * s


def _square(n):
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  pm = unit_cube_mesh(n, ndim=2, a=-1.0, b=1.0)
  mask = np.zeros(pm.num_nodes, bool)
  mask[np.unique(pm.physical_groups['boundary'])] = True
  return pm, mask


def test_poisson_square_series():
  pm, mask = _square(32)
  u = O.solve_poisson(pm.node_coords, pm.elements, (2, 'nc'), mask,
                      np.ones(pm.num_nodes))
  np.testing.assert_allclose(u, _series(pm.node_coords), rtol=1e-6, atol=1e-3)


def test_poisson_unit_circle():
  pm, mask = _square(32)
  x = pm.node_coords
  r2 = 1 / np.sqrt(2)
  xc = np.stack([
      x[:, 0] * (np.cos(np.pi * x[:, 1] / 4) - r2) + np.sin(np.pi * x[:, 0] / 4),
      x[:, 1] * (np.cos(np.pi * x[:, 0] / 4) - r2) + np.sin(np.pi * x[:, 1] / 4)],
                axis=-1)
  u = O.solve_poisson(xc, pm.elements, (2, 'nc'), mask, np.ones(pm.num_nodes))
  np.testing.assert_allclose(u, .25 * (1 - np.sum(xc ** 2, axis=-1)),
                             rtol=1e-6, atol=1e-4)


def test_cpu_reference_matches_oracle():
  """bench.py's multi-threaded CPU baseline computes what the oracle computes."""
  import torch
  from oracle import cpu_reference
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  rng = np.random.default_rng(2)
  P = 5
  pm = unit_cube_mesh(3, ndim=3)
  pm = pm.replace(node_coords=pm.node_coords + 0.04 * rng.uniform(
      -1, 1, pm.node_coords.shape))
  rp = refine_premesh(pm, Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE))
  mask = np.zeros(rp.num_nodes)
  mask[np.unique(rp.physical_groups['boundary'])] = 1.0
  ref = cpu_reference.StiffnessCG(rp.node_coords, rp.elements, P, mask)
  fes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
  A = lambda u: (1 - mask) * fes.scatter(fes.stiffness_local(fes.gather(u)))
  u = rng.standard_normal(rp.num_nodes)
  got = ref.apply(torch.from_numpy(u)).numpy()
  assert np.abs(got - A(u)).max() < 1e-13 * np.abs(A(u)).max()
  got = ref.apply_sum_factorised(torch.from_numpy(u)).numpy()
  assert np.abs(got - A(u)).max() < 1e-13 * np.abs(A(u)).max()
  b = (1 - mask) * u
  x, k, _ = ref.cg_iterations(torch.from_numpy(b), iters=25)
  xo, _ = O.cg(A, b, tol=0.0, maxiter=25)
  assert k == 25
  assert np.abs(x.numpy() - xo).max() < 1e-10 * np.abs(xo).max()
