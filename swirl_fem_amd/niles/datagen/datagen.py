"""Kolmogorov-flow dataset generator on the MI355X operators.

Same program as the reference's `swirl_fem/niles/datagen/datagen.py`: the 2D
doubly periodic unit square, velocity order 8, BDF3 / EXT2, forcing
`(sin(2 pi k y), 0) - drag * u` through the mass matrix (:64-71, :90-102),
cycles of `num_steps_per_cycle` steps with a snapshot every tenth step, one
file per cycle holding the datasets `t (S,)`, `u (S, N, 2)`, `p (S, Np)`
(:105-165).  The constants of the reference (:46-53) are the defaults of
`DatagenConfig`; the time step itself is
`examples.navier_stokes_driver.navier_stokes_step`.

Snapshots are HDF5 files with the reference's layout (datasets 't', 'u', 'p'
at the root; same names and shapes), written through `h5py` when it is
importable and through the HDF5 C library otherwise (`h5lite`).
"""

from __future__ import annotations

import argparse
import dataclasses
import logging
import os
import time

import numpy as np
import torch

from swirl_fem_amd.common import premesh_commons
from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
from swirl_fem_amd.navier_stokes import navier_stokes

# pylint: disable=invalid-name


@dataclasses.dataclass(frozen=True)
class DatagenConfig:
  """Module constants of the reference (datagen.py:46-53)."""
  resolution: int = 64
  order: int = 8
  time_order: int = 3
  reynolds_number: float = 20000.0
  num_cycles: int = 500
  num_steps_per_cycle: int = 500
  dt: float = 1e-4
  drag_coeff: float = 0.1
  snapshot_every: int = 10
  tol: float = 1e-5
  atol: float = 1e-4


def u_init_fn(x: torch.Tensor) -> torch.Tensor:
  """Initial velocity of the Kolmogorov flow at nodes `x (N, 2)` (:56-61)."""
  l = 2.0
  a, b = 2 * l * np.pi * x[:, 0], 2 * l * np.pi * x[:, 1]
  return torch.stack([torch.cos(a) * torch.sin(b),
                      -torch.sin(a) * torch.cos(b)], dim=-1)


def forcing(x: torch.Tensor, u: torch.Tensor,
            drag_coeff: float = 0.1) -> torch.Tensor:
  """Kolmogorov forcing with linear drag at every node (:64-71)."""
  k = 4.0
  f = torch.stack([torch.sin(2 * np.pi * k * x[:, 1]),
                   torch.zeros_like(x[:, 1])], dim=-1)
  return f - drag_coeff * u


def compute_dx(mesh) -> float:
  """Smallest distance between two nodes of one element (:74-85; the
  reference loops over the elements on the host)."""
  x = mesh.element_coords().to(torch.float32)              # (E, n, d)
  best = float('inf')
  chunk = max(1, (1 << 24) // (x.shape[1] ** 2))
  for s in range(0, x.shape[0], chunk):
    d = torch.cdist(x[s:s + chunk], x[s:s + chunk],
                    compute_mode='donot_use_mm_for_euclid_dist')
    d.diagonal(dim1=-2, dim2=-1).fill_(float('inf'))
    best = min(best, float(d.min()))
  return best


def _solve_one_step(sem, us, ps, Cus, cfg: DatagenConfig):
  """One step of the forced Navier-Stokes system (:88-102)."""
  f = forcing(sem.velocity.mesh.node_coords, us[-1], cfg.drag_coeff)
  u, p, Cu, _ = navier_stokes_step(
      sem, us, ps, Cus, reynolds=cfg.reynolds_number, dt=cfg.dt,
      time_order=cfg.time_order, forcing=f, tol=cfg.tol, atol=cfg.atol)
  return u, p, Cu


def write_snapshots(path_stem: str, dataset: dict, format: str = 'hdf5') -> str:
  """Writes `{'t', 'u', 'p'}` as `<path_stem>.hdf5`, the reference's container
  (:127-165: one dataset per key at the root of the file); returns the path.

  HDF5 goes through `h5py` when it is importable, else through the HDF5 C
  library (`h5lite`, ctypes).  If neither is present this raises -- a dataset
  silently written in another container would break its readers.
  `format='npz'` asks for a NumPy archive explicitly.
  """
  if format == 'npz':
    path = path_stem + '.npz'
    np.savez(path, **dataset)
    return path
  if format != 'hdf5':
    raise ValueError(f'unknown snapshot format {format!r}')
  path = path_stem + '.hdf5'
  try:
    import h5py                      # pylint: disable=import-outside-toplevel
  except ImportError:
    h5py = None
  if h5py is not None:
    with h5py.File(path, 'w') as f:
      for k, v in dataset.items():
        f[k] = v
    return path
  from swirl_fem_amd.niles.datagen import h5lite
  if not h5lite.available():
    raise ImportError(
        'writing HDF5 snapshots needs h5py or the HDF5 C library (libhdf5); '
        "neither was found -- pass format='npz' for a NumPy archive instead")
  h5lite.write(path, {k: np.asarray(v) for k, v in dataset.items()})
  return path


def read_snapshots(path: str) -> dict:
  """Reads a file written by `write_snapshots` back into NumPy arrays."""
  if path.endswith('.npz'):
    with np.load(path) as z:
      return {k: z[k] for k in z.files}
  try:
    import h5py                      # pylint: disable=import-outside-toplevel
  except ImportError:
    from swirl_fem_amd.niles.datagen import h5lite
    return h5lite.read(path)
  with h5py.File(path, 'r') as f:
    return {k: np.asarray(f[k]) for k in f}


def one_cycle(sem, start_step: int, num_steps: int, us, ps, *,
              cfg: DatagenConfig = DatagenConfig(), workdir: str | None = None):
  """Runs `num_steps` steps from the histories `us`, `ps` (oldest first) and
  writes the cycle's snapshots (:105-165).

  Returns `(us, ps, dataset, path)`; `path` is None without a `workdir`.
  """
  t = start_step * cfg.dt
  dataset = {'t': [t], 'u': [us[-1].cpu().numpy()],
             'p': [ps[-1].cpu().numpy()]}
  start_time = time.time()
  Cus = tuple(sem.C(u) for u in us)
  for step_idx in range(1, num_steps + 1):
    t += cfg.dt
    u, p, Cu = _solve_one_step(sem, us, ps, Cus, cfg)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    if step_idx % cfg.snapshot_every == 0:
      dataset['t'].append(t)
      dataset['u'].append(u.cpu().numpy())
      dataset['p'].append(p.cpu().numpy())
  logging.info('one cycle walltime %f seconds', time.time() - start_time)
  dataset = {k: np.stack(v) for k, v in dataset.items()}
  path = None
  if workdir is not None:
    os.makedirs(workdir, exist_ok=True)
    path = write_snapshots(os.path.join(workdir, (
        f'kolmogorov_flow_grid_{cfg.resolution}_order_{cfg.order}'
        f'_step_{start_step}_{start_step + num_steps}')), dataset)
    logging.info('wrote %s', path)
  return us, ps, dataset, path


def create_sem(cfg: DatagenConfig, device=None):
  premesh = premesh_commons.unit_cube_mesh(cfg.resolution, ndim=2,
                                           periodic_dims=(0, 1))
  return navier_stokes.StokesSEM.create(premesh, boundary_conditions={},
                                        order=cfg.order, device=device)


def run_simulation(cfg: DatagenConfig = DatagenConfig(), workdir=None,
                   device=None):
  """Runs all cycles (:168-199); returns the final histories and the CFL
  number after every cycle."""
  sem = create_sem(cfg, device)
  mesh = sem.velocity.mesh
  mesh_dx = compute_dx(mesh)
  logging.info('Created mesh with %d nodes and %d elements, dx %f',
               mesh.num_nodes, mesh.num_elements, mesh_dx)
  u_init = u_init_fn(mesh.node_coords)
  p_init = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=u_init.dtype,
                       device=u_init.device)
  us = tuple(u_init for _ in range(cfg.time_order))
  ps = tuple(p_init for _ in range(cfg.time_order))
  cfls, paths = [], []
  for cycle_idx in range(cfg.num_cycles):
    us, ps, _, path = one_cycle(
        sem, cycle_idx * cfg.num_steps_per_cycle, cfg.num_steps_per_cycle,
        us, ps, cfg=cfg, workdir=workdir)
    paths.append(path)
    cfls.append(float(us[-1].max()) * cfg.dt / mesh_dx)
    logging.info('At cycle %d, CFL number: %f', cycle_idx, cfls[-1])
  return us, ps, cfls, paths


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
  ap.add_argument('--workdir', required=True)
  for f in dataclasses.fields(DatagenConfig):
    ap.add_argument('--' + f.name.replace('_', '-'), type=type(f.default),
                    default=f.default)
  args = ap.parse_args(argv)
  logging.basicConfig(level=logging.INFO)
  cfg = DatagenConfig(**{f.name: getattr(args, f.name)
                         for f in dataclasses.fields(DatagenConfig)})
  run_simulation(cfg, args.workdir, device=torch.device('cuda', 0))


if __name__ == '__main__':
  main()
