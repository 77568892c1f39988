"""Incompressible Navier-Stokes drivers on top of `StokesSEM`.

The reference ships one driver, the 2D periodic Kolmogorov-flow generator
`swirl_fem/niles/datagen/datagen.py`; its time step (:90-102) is
`navier_stokes_step` below: EXT_{k-1} extrapolation of the over-integrated
convection term, body force through the mass matrix, one `stokes_one_step`
(Helmholtz PCG + pressure PCG), convection of the new velocity.  The two
BASELINE configurations that have no reference driver are assembled from the
same operators:

  * `lid_driven_cavity` (config 3): 2D, Dirichlet walls, moving lid imposed
    through `u_boundary` (navier_stokes.py:353, :433-440);
  * `taylor_green` (config 4): 3D triply periodic box [0, 2 pi]^3.
"""

from __future__ import annotations

import time

import numpy as np
import torch

from swirl_fem_amd import switches
from swirl_fem_amd.common.premesh_commons import box_mesh, unit_cube_mesh
from swirl_fem_amd.navier_stokes import navier_stokes
from swirl_fem_amd.navier_stokes.navier_stokes import BCType, StokesSEM


def navier_stokes_step(sem: StokesSEM, us, ps, Cus, *, reynolds: float,
                       dt: float, time_order: int, forcing=None,
                       u_boundary=None, tol=1e-5, atol=1e-4, alpha=0.05,
                       pressure_projection=None, pressure_preconditioner=None,
                       velocity_preconditioner=None):
  """One BDFk/EXT(k-1) step (datagen.py:90-102).

  Args:
    us, ps, Cus: histories (oldest first) of velocity, pressure and
      convection C(u); `time_order` entries each.
    forcing: optional nodal body force (N, d).
  Returns:
    (u, p, C(u), aux)
  """
  k_ext = max(time_order - 1, 1)
  ext = navier_stokes.extk_coeffs(k=k_ext)
  Cu = sum(float(ext[-i]) * Cus[-i] for i in range(1, len(ext) + 1))
  f = -Cu
  if forcing is not None:
    f = f + sem.B(forcing)
  # beyond the reference: a preconditioner by name for its
  # `pressure_preconditioner` hook (None / 'projection' = the reference's
  # nullspace projection; 'schwarz': navier_stokes/pressure_preconditioner.py)
  if pressure_preconditioner is None:
    pressure_preconditioner = switches.get('SFEM_PRESSURE_PC')
  if isinstance(pressure_preconditioner, str):
    from swirl_fem_amd.navier_stokes import pressure_preconditioner as pc
    pressure_preconditioner = pc.make_pressure_preconditioner(
        sem, pressure_preconditioner, dt, time_order)
  u, p, aux = sem.stokes_one_step(us, ps, f, mu=1.0 / reynolds, dt=dt,
                                  time_order=time_order, alpha=alpha,
                                  pressure_preconditioner=
                                  pressure_preconditioner,
                                  u_boundary=u_boundary, tol=tol, atol=atol,
                                  pressure_projection=pressure_projection,
                                  velocity_preconditioner=
                                  velocity_preconditioner)
  return u, p, sem.C(u), aux


class _StepTimer:
  """Optional wall-clock record of a driver run: `profile['setup_s']` and one
  entry of `profile['step_s']` per time step (device-synchronised)."""

  def __init__(self, profile, device):
    self.profile, self.device = profile, device
    self.t0 = time.perf_counter()
    if profile is not None:
      profile['step_s'] = []

  def _sync(self):
    if torch.cuda.is_available():
      torch.cuda.synchronize(self.device)

  def setup_done(self):
    if self.profile is not None:
      self._sync()
      self.profile['setup_s'] = time.perf_counter() - self.t0
      self.t0 = time.perf_counter()

  def step_done(self):
    if self.profile is not None:
      self._sync()
      now = time.perf_counter()
      self.profile['step_s'].append(now - self.t0)
      self.t0 = now


def _histories(sem, u0, p0, time_order):
  us = tuple(u0 for _ in range(time_order))
  ps = tuple(p0 for _ in range(time_order))
  c0 = sem.C(u0)
  return us, ps, tuple(c0 for _ in range(time_order))


def lid_driven_cavity(n=8, order=5, reynolds=100.0, dt=1e-3, steps=10,
                      time_order=3, device=None, premesh=None, tol=1e-8,
                      profile=None, pressure_projection=None,
                      pressure_preconditioner=None):
  """2D lid-driven cavity on [0,1]^2; returns (sem, u, p, diagnostics)."""
  timer = _StepTimer(profile, device)
  pm = premesh if premesh is not None else unit_cube_mesh(n, ndim=2)
  sem = StokesSEM.create(pm, {'boundary': (BCType.DIRICHLET, 0.0)},
                         order=order, device=device)
  x = sem.velocity.mesh.node_coords
  # regularised lid u = (16 x^2 (1-x)^2, 0) on y = 1, zero on the other walls
  lid = (x[:, 1] > 1.0 - 1e-12).to(x.dtype)
  u_b = torch.stack([lid * 16 * x[:, 0] ** 2 * (1 - x[:, 0]) ** 2,
                     torch.zeros_like(lid)], dim=-1)
  u0 = u_b.clone()
  p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype,
                   device=x.device)
  us, ps, Cus = _histories(sem, u0, p0, time_order)
  iters = []
  timer.setup_done()
  for _ in range(steps):
    u, p, Cu, aux = navier_stokes_step(
        sem, us, ps, Cus, reynolds=reynolds, dt=dt, time_order=time_order,
        u_boundary=u_b, tol=tol, atol=0.0,
        pressure_projection=pressure_projection,
        pressure_preconditioner=pressure_preconditioner)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    timer.step_done()
    iters.append((aux['u_star_info']['num_iterations'],
                  aux['dp_info']['num_iterations']))
  diag = {'cg_iterations': iters,
          'max_divergence': float(sem.D(us[-1]).abs().max()),
          'kinetic_energy': float(0.5 * (sem.velocity_mass_diag *
                                         us[-1] ** 2).sum())}
  return sem, us[-1], ps[-1], diag


def taylor_green(n=4, order=3, reynolds=100.0, dt=1e-2, steps=5, time_order=3,
                 device=None, tol=1e-8, profile=None, pressure_projection=None,
                 pressure_preconditioner=None):
  """3D Taylor-Green vortex on the periodic box [0, 2 pi]^3 (`n` elements
  per direction, or one count per direction)."""
  timer = _StepTimer(profile, device)
  ns = (n,) * 3 if np.isscalar(n) else tuple(n)
  pm = box_mesh(ns, (0.0,) * 3, (2 * np.pi,) * 3, periodic_dims=(0, 1, 2))
  sem = StokesSEM.create(pm, {}, order=order, device=device)
  x = sem.velocity.mesh.node_coords
  u0 = torch.stack([torch.sin(x[:, 0]) * torch.cos(x[:, 1]) * torch.cos(x[:, 2]),
                    -torch.cos(x[:, 0]) * torch.sin(x[:, 1]) * torch.cos(x[:, 2]),
                    torch.zeros_like(x[:, 0])], dim=-1)
  p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype,
                   device=x.device)
  us, ps, Cus = _histories(sem, u0, p0, time_order)
  # unassembled mass diagonal: the images of a periodic node each carry their
  # share, so the plain sum is the integral
  w = sem.velocity_mass_diag
  energy = [float(0.5 * (w * u0 ** 2).sum())]
  iters = []
  timer.setup_done()
  for _ in range(steps):
    u, p, Cu, aux = navier_stokes_step(
        sem, us, ps, Cus, reynolds=reynolds, dt=dt, time_order=time_order,
        tol=tol, atol=0.0, pressure_projection=pressure_projection,
        pressure_preconditioner=pressure_preconditioner)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    timer.step_done()
    energy.append(float(0.5 * (w * u ** 2).sum()))
    iters.append((aux['u_star_info']['num_iterations'],
                  aux['dp_info']['num_iterations']))
  diag = {'kinetic_energy': energy, 'cg_iterations': iters,
          'max_divergence': float(sem.D(us[-1]).abs().max())}
  return sem, us[-1], ps[-1], diag


def taylor_green_blocks(n=4, order=3, block_grid=(2, 2, 2), rank=None,
                        reynolds=100.0, dt=1e-2, steps=5, time_order=3,
                        device=None, tol=1e-8, profile=None,
                        pressure_projection=None, pressure_preconditioner=None):
  """BASELINE config 4: the 3D Taylor-Green vortex on the triply periodic box
  [0, 2 pi]^3, one `n^3`-element block per rank (`block_grid` ranks, launched
  with torch.distributed; 2 x 2 x 2 blocks of 64^3 elements are the 128^3
  mesh).  Every rank builds only its own block (`distributed/blocks.py`); the
  shared-DOF exchange runs over RCCL neighbour send/recv.

  Returns (sem, u, p, diagnostics) with this rank's part of the fields.
  """
  from swirl_fem_amd.distributed import blocks, comm
  timer = _StepTimer(profile, device)
  if rank is None:
    rank = comm.get_rank()
  part = blocks.build_block_partition(
      n, order + 1, block_grid, rank, device=device, lo=0.0, hi=2 * np.pi,
      periodic_dims=tuple(range(len(block_grid))))
  sem = StokesSEM.create(part.premesh, {}, order=order, device=device,
                         neighbor_plan=part.plan)
  x = sem.velocity.mesh.node_coords
  u0 = torch.stack([torch.sin(x[:, 0]) * torch.cos(x[:, 1]) * torch.cos(x[:, 2]),
                    -torch.cos(x[:, 0]) * torch.sin(x[:, 1]) * torch.cos(x[:, 2]),
                    torch.zeros_like(x[:, 0])], dim=-1)
  p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype,
                   device=x.device)
  us, ps, Cus = _histories(sem, u0, p0, time_order)
  # unassembled mass diagonal: the holders of a node each carry their share
  w = sem.velocity_mass_diag
  energy = [float(sem._global_sum(0.5 * (w * u0 ** 2).sum().reshape(1)))]
  iters = []
  timer.setup_done()
  for _ in range(steps):
    u, p, Cu, aux = navier_stokes_step(
        sem, us, ps, Cus, reynolds=reynolds, dt=dt, time_order=time_order,
        tol=tol, atol=0.0, pressure_projection=pressure_projection,
        pressure_preconditioner=pressure_preconditioner)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    energy.append(float(sem._global_sum(0.5 * (w * u ** 2).sum().reshape(1))))
    iters.append((aux['u_star_info']['num_iterations'],
                  aux['dp_info']['num_iterations']))
    timer.step_done()
  diag = {'kinetic_energy': energy, 'cg_iterations': iters}
  return sem, us[-1], ps[-1], diag
