"""Shared fixture: the Stokes decaying-vortex case of the reference's tests
(navier_stokes/navier_stokes_test.py:39-73): [-1,1] x [-pi,pi], 9x9 elements,
periodic in y, order 7, analytic eigen-solution of the Stokes operator."""
import numpy as np
import scipy.optimize

from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
from swirl_fem_amd.core.mesh_refiner import refine_premesh


def make_premesh(n=9):
  pm = unit_cube_mesh(n, ndim=2, periodic_dims=(1,))
  x = pm.node_coords
  return pm.replace(node_coords=np.stack(
      [2 * x[:, 0] - 1, 2 * np.pi * x[:, 1] - np.pi], axis=-1))


def soln_params(k=1., viscosity=1.):
  mu = scipy.optimize.newton(lambda x: k * np.tanh(k) + x * np.tan(x), np.pi)
  return mu, -viscosity * (k ** 2 + mu ** 2)


def reference_soln(vcoords, pcoords, t, k=1., viscosity=1.):
  mu, sigma = soln_params(k, viscosity)
  f = lambda x: np.cos(mu) * np.cosh(k * x) - np.cosh(k) * np.cos(mu * x)
  g = lambda x: (1j / k) * (k * np.cos(mu) * np.sinh(k * x) +
                            mu * np.cosh(k) * np.sin(mu * x))
  h = lambda x: -(sigma / k) * np.cos(mu) * np.sinh(k * x)
  lead = lambda x: np.exp(sigma * t) * np.exp(1j * k * x[:, 1])
  u = np.real(lead(vcoords)[:, None] * np.stack(
      [f(vcoords[:, 0]), g(vcoords[:, 0])], axis=-1))
  p = np.real(lead(pcoords) * h(pcoords[:, 0]))
  return u, p


def staged_meshes(premesh, order):
  """Refined + finalised host arrays for velocity (GLL) and pressure (GL)."""
  v = refine_premesh(premesh, Nodes1D.create(
      order + 1, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize_all()
  p = refine_premesh(premesh, Nodes1D.create(
      order - 1, NodeType.GAUSS_LEGENDRE)).finalize_all()
  return v, p
