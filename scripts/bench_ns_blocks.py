"""BASELINE config 4's layout on ONE GPU: the triply periodic Taylor-Green box as 2 x 2 x 2 rank-local blocks, the
eight ranks as threads (`distributed/inprocess.py`; their kernels serialise on the one device, so the time per step is
the SUM over the ranks, not what eight GPUs would take).  What it shows: iteration counts of the partitioned solves
with the reference's solver settings and with the opt-in ones (Schwarz pressure preconditioner with the coarse level
by FFT after an all-gather, mass preconditioner of the velocity solve).
  N=16 STEPS=3 python scripts/bench_ns_blocks.py            (SFEM_PRESSURE_PC=schwarz SFEM_VELOCITY_PC=mass)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd import switches
from swirl_fem_amd.distributed import inprocess
from swirl_fem_amd.examples import navier_stokes_driver as drv

n = int(os.environ.get('N', '16')); steps = int(os.environ.get('STEPS', '3'))
grid = tuple(int(v) for v in os.environ.get('GRID', '2,2,2').split(','))
dev = torch.device('cuda', 0)


def rank_main(rank):
  prof = {}
  sem, u, p, diag = drv.taylor_green_blocks(n=n, order=7, block_grid=grid, rank=rank, reynolds=1600.0, dt=1e-3,
                                            steps=steps, time_order=3, device=dev, tol=1e-6, profile=prof)
  return diag['cg_iterations'], prof['step_s'], diag['kinetic_energy'][-1]

world = inprocess.ThreadWorld(int(np.prod(grid)))
t0 = time.perf_counter()
out = world.run(rank_main)
its, step_s, energy = out[0]
print(json.dumps({
    'case': f'3D Taylor-Green, p=7, {grid[0]}x{grid[1]}x{grid[2]} blocks of {n}^3 hexes, {world.world} ranks as threads on one GPU',
    'cg_iterations_helmholtz_pressure': its, 'ms_per_step_all_ranks_serialised': [1e3 * s for s in step_s],
    'kinetic_energy': energy, 'wall_s': time.perf_counter() - t0, 'switches': switches.active()}), flush=True)
