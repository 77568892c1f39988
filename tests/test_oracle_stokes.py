"""Pins the oracle's Stokes operators with the analytic answers of the
reference's navier_stokes_test.py:79-358 (decaying Stokes vortices)."""
import numpy as np
import pytest

from oracle import sfem_oracle as O
from tests import stokes_case as SC

ORDER, K, DT = 7, 3, 1e-3


@pytest.fixture(scope='module')
def sem():
  pm = SC.make_premesh()
  v, p = SC.staged_meshes(pm, ORDER)
  assert len(pm.node_coords) == 100 and len(pm.elements) == 81
  return O.StokesOracle(v, p, ORDER, v['physical_masks']['boundary']), v, p


def _states(v, p, n):
  return list(zip(*[SC.reference_soln(v['node_coords'], p['node_coords'],
                                      t=i * DT) for i in range(n)]))


def test_momentum_divergence_bdf(sem):
  s, v, p = sem
  u, pr = SC.reference_soln(v['node_coords'], p['node_coords'], t=0.0)
  _, sigma = SC.soln_params()
  err = s.vexchange(s.B(sigma * u) + s.A(u) - s.Dt(pr))
  assert np.abs(err).max() < 1e-7
  assert np.abs(s.D(u)).max() < 1e-10
  us, ps = _states(v, p, K + 1)
  du_dt = (1 / DT) * sum(c * x for c, x in zip(O.bdfk_coeffs(K), us))
  err = s.vexchange(s.B(du_dt) + s.A(us[-1]) - s.Dt(ps[-1]))
  assert np.abs(err).max() < 1e-7


def test_fractional_step_identities(sem):
  s, v, p = sem
  us, ps = _states(v, p, K + 1)
  us, u = us[:-1], us[-1]
  ps, pr = ps[:-1], ps[-1]
  ext = O.extk_coeffs(1)
  p_ext = sum(ext[-i] * ps[-i] for i in range(1, len(ext) + 1))
  bdf = O.bdfk_coeffs(K)
  f = -(1 / DT) * sum(c * x for c, x in zip(bdf[:-1], us))
  b = s.B(f) + s.Dt(p_ext)
  H = lambda w: (bdf[-1] / DT) * s.B(w) + s.A(w)
  Q = lambda w: (DT / bdf[-1]) * s.Bi(w)
  dp = pr - p_ext
  assert np.abs(s.vexchange(H(u) - s.Dt(dp) - b)).max() < 1e-7
  assert np.abs(s.vexchange(H(u) - H(Q(s.Dt(dp))) - b)).max() < 10 * DT ** 2
  u_star = u - Q(s.Dt(dp))
  assert np.abs(s.vexchange(H(u_star) - b)).max() < 10 * DT ** 2
  u_cg, _ = O.cg(H, b, M=s.vexchange, tol=1e-15)
  assert np.abs(s.vexchange(H(u_cg) - b)).max() < 1e-12
  assert np.abs(u_cg - u + Q(s.Dt(dp))).max() < 5 * DT ** 2


def test_one_step(sem):
  s, v, p = sem
  us, ps = _states(v, p, K + 1)
  u, pr, aux = s.stokes_one_step(us[:-1], ps[:-1], 0, 1, DT, K, alpha=0.05,
                                 tol=1e-12, atol=1e-12)
  assert np.abs(u - us[-1]).max() < 5 * DT ** 2
  assert np.abs(pr - ps[-1]).max() < 50 * DT ** 2
  assert aux['u_star_info']['residual'] < 1e-7
  assert aux['dp_info']['residual'] < 1e-7
