"""Navier-Stokes step timing on one MI355X (BASELINE configs 3 and 4 shapes).

Prints one JSON line per case: wall time per step (setup excluded, first step
excluded as warm-up), CG iteration counts, DOFs.  Not the contract bench
(`bench.py` is); evidence for the caller rows of SURVEY 8 (a14).
  python scripts/bench_ns.py [cavity] [tgv16] [tgv32] [tgv64] [kolmogorov]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd import switches
from swirl_fem_amd.examples import navier_stokes_driver as drv

CASES = {
    'cavity': dict(fn='lid_driven_cavity', kw=dict(n=32, order=5, reynolds=100.0, dt=1e-3, steps=8, tol=1e-8),
                   name='2D lid-driven cavity, 32x32 quads, p=5 (config 3 shape)'),
    'tgv16': dict(fn='taylor_green', kw=dict(n=16, order=7, reynolds=1600.0, dt=1e-3, steps=4, tol=1e-6),
                  name='3D Taylor-Green, 16^3 hexes, p=7, triply periodic'),
    'tgv32': dict(fn='taylor_green', kw=dict(n=32, order=7, reynolds=1600.0, dt=1e-3, steps=4, tol=1e-6),
                  name='3D Taylor-Green, 32^3 hexes, p=7, triply periodic (1/8 of a config-4 GPU block)'),
    'tgv64': dict(fn='taylor_green', kw=dict(n=64, order=7, reynolds=1600.0, dt=1e-3, steps=3, tol=1e-6),
                  name='3D Taylor-Green, 64^3 hexes, p=7, triply periodic (one config-4 GPU block)'),
}


def kolmogorov(steps=60, warm=10):
  """The reference generator's own configuration (niles/datagen/datagen.py:
  64 x 64 quads, order 8, Re 20000, dt 1e-4, BDF3/EXT2, tol 1e-5 / atol 1e-4)."""
  import time
  from swirl_fem_amd.niles.datagen import datagen
  cfg = datagen.DatagenConfig()
  t0 = time.perf_counter()
  sem = datagen.create_sem(cfg, torch.device('cuda', 0))
  x = sem.velocity.mesh.node_coords
  u0 = datagen.u_init_fn(x)
  p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=u0.dtype, device=u0.device)
  us, ps = (u0,) * 3, (p0,) * 3
  Cus = tuple(sem.C(u) for u in us)
  torch.cuda.synchronize(); setup = time.perf_counter() - t0
  times = []
  for _ in range(steps):
    t1 = time.perf_counter()
    u, p, Cu = datagen._solve_one_step(sem, us, ps, Cus, cfg)
    us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t1)
  print(json.dumps({
      'case': '2D Kolmogorov flow generator, 64x64 quads, order 8, Re 20000, dt 1e-4 (reference datagen.py constants)',
      'ms_per_step': 1e3 * float(np.mean(times[warm:])), 'steps_timed': steps - warm,
      'first_step_ms': 1e3 * times[0], 'setup_s': setup,
      'velocity_dofs': 2 * sem.velocity.mesh.num_nodes, 'pressure_dofs': sem.pressure.pspace.mesh.num_nodes,
      'max_divergence': float(sem.D(us[-1]).abs().max()), 'max_velocity': float(us[-1].abs().max()),
      'dtype': 'f64', 'hip_graphs': os.environ.get('SFEM_GRAPHS', '1') != '0',
      'peak_memory_gb': torch.cuda.max_memory_allocated() / 1e9,
      'switches': switches.active()}), flush=True)


for key in (sys.argv[1:] or ['cavity', 'tgv16']):
  if key == 'kolmogorov':
    kolmogorov()
    continue
  c = CASES[key]; prof = {}
  if os.environ.get('STEPS'):       # longer runs (SFEM_PRESSURE_PROJECTION needs history)
    c['kw']['steps'] = int(os.environ['STEPS'])
  sem, u, p, diag = getattr(drv, c['fn'])(device='cuda:0', profile=prof, **c['kw'])
  steps = prof['step_s'][1:]
  nv = sem.velocity.mesh.num_nodes; d = sem.velocity.mesh.ndim
  print(json.dumps({
      'case': c['name'], 'ms_per_step': 1e3 * float(np.mean(steps)), 'steps_timed': len(steps),
      'first_step_ms': 1e3 * prof['step_s'][0], 'setup_s': prof['setup_s'],
      'velocity_dofs': nv * d, 'pressure_dofs': sem.pressure.pspace.mesh.num_nodes,
      'cg_iterations_helmholtz_pressure': diag['cg_iterations'][1:],
      'max_divergence': diag['max_divergence'], 'dtype': 'f64',
      'hip_graphs': os.environ.get('SFEM_GRAPHS', '1') != '0',
      'switches': switches.active(),
      'step_ms': [round(1e3 * t, 2) for t in prof['step_s']],
      'peak_memory_gb': torch.cuda.max_memory_allocated() / 1e9}), flush=True)
  del sem, u, p; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
