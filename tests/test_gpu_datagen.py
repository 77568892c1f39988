"""The Kolmogorov-flow generator (reference niles/datagen/datagen.py) on a
small mesh: snapshot layout, the forced step against the plain driver step,
file round trip."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def test_kolmogorov_cycle_snapshots(tmp_path):
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  from swirl_fem_amd.niles.datagen import datagen
  cfg = datagen.DatagenConfig(resolution=4, order=5, num_cycles=2,
                              num_steps_per_cycle=20, dt=1e-3,
                              reynolds_number=200.0, tol=1e-10, atol=0.0)
  us, ps, cfls, paths = datagen.run_simulation(cfg, str(tmp_path), device=DEV)
  assert len(us) == len(ps) == cfg.time_order and len(cfls) == 2
  assert all(0.0 < c < 1.0 for c in cfls)
  names = sorted(os.listdir(tmp_path))
  stem = 'kolmogorov_flow_grid_4_order_5_step_'
  assert [n.rsplit('.', 1)[0] for n in names] == [stem + '0_20', stem + '20_40']
  assert [os.path.join(tmp_path, n) for n in names] == sorted(paths)

  assert all(p.endswith('.hdf5') for p in paths)       # the reference's format
  with open(sorted(paths)[0], 'rb') as fh:
    assert fh.read(8) == b'\x89HDF\r\n\x1a\n'            # HDF5 signature
  load = datagen.read_snapshots

  sem = datagen.create_sem(cfg, DEV)
  Nv = sem.velocity.mesh.num_nodes
  Np = sem.pressure.pspace.mesh.num_nodes
  first, second = load(sorted(paths)[0]), load(sorted(paths)[1])
  for k, d in enumerate((first, second)):
    assert sorted(d) == ['p', 't', 'u']
    assert d['t'].shape == (3,) and d['u'].shape == (3, Nv, 2)
    assert d['p'].shape == (3, Np)
    np.testing.assert_allclose(d['t'], (20 * k + np.array([0, 10, 20])) * cfg.dt,
                               rtol=0, atol=1e-12)
  # a cycle starts from the state the previous one ended with
  np.testing.assert_array_equal(first['u'][-1], second['u'][0])
  np.testing.assert_array_equal(second['u'][-1], us[-1].cpu().numpy())

  # the generator's step is the driver step with the Kolmogorov body force
  x = sem.velocity.mesh.node_coords
  u0 = datagen.u_init_fn(x)
  np.testing.assert_array_equal(first['u'][0], u0.cpu().numpy())
  p0 = torch.zeros(Np, dtype=u0.dtype, device=DEV)
  hist_u, hist_p = (u0,) * 3, (p0,) * 3
  Cus = tuple(sem.C(u) for u in hist_u)
  for _ in range(10):
    f = datagen.forcing(x, hist_u[-1], cfg.drag_coeff)
    u, p, Cu, _ = navier_stokes_step(
        sem, hist_u, hist_p, Cus, reynolds=cfg.reynolds_number, dt=cfg.dt,
        time_order=3, forcing=f, tol=cfg.tol, atol=cfg.atol)
    hist_u, hist_p, Cus = hist_u[1:] + (u,), hist_p[1:] + (p,), Cus[1:] + (Cu,)
  np.testing.assert_allclose(first['u'][1], u.cpu().numpy(), rtol=0, atol=1e-9)
  np.testing.assert_allclose(first['p'][1], p.cpu().numpy(), rtol=0, atol=1e-7)
  # incompressible, and the forcing keeps the flow alive
  assert float(sem.D(us[-1]).abs().max()) < 1e-6
  assert 0.1 < float(us[-1].abs().max()) < 2.0


def test_compute_dx_is_the_gll_end_spacing():
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  from swirl_fem_amd.niles.datagen import datagen
  cfg = datagen.DatagenConfig(resolution=3, order=6)
  sem = datagen.create_sem(cfg, DEV)
  nodes = np.asarray(Nodes1D.create(7, NodeType.GAUSS_LOBATTO_LEGENDRE).node_values)
  want = (nodes[1] - nodes[0]) / 2 / 3          # reference interval [-1, 1]
  assert datagen.compute_dx(sem.velocity.mesh) == pytest.approx(want, rel=1e-5)
