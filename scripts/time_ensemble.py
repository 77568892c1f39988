"""Step time of the Kolmogorov generator (64 x 64 quads, order 8: the
reference's datagen.py constants) for one flow and for an ensemble of B flows
run as ONE StokesSEM on B copies of the mesh (`StokesSEM.ensemble`).
  B=8 STEPS=40 python scripts/time_ensemble.py          (SFEM_PRESSURE_PC=schwarz)
Prints one JSON line; `members_equal` is the largest relative difference
between a member of the ensemble and the same flow stepped alone.
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd import switches
from swirl_fem_amd.niles.datagen import datagen

B = int(os.environ.get('B', '8')); STEPS = int(os.environ.get('STEPS', '40'))
WARM = int(os.environ.get('WARM', '10'))
dev = torch.device('cuda', 0)
cfg = datagen.DatagenConfig(resolution=int(os.environ.get('RES', '64')),
                            order=int(os.environ.get('ORDER', '8')),
                            tol=float(os.environ.get('TOL', '1e-5')),
                            atol=float(os.environ.get('ATOL', '1e-4')))
sem = datagen.create_sem(cfg, dev)
ens = sem.ensemble(B)
x = sem.velocity.mesh.node_coords
amps = 1.0 + 0.05 * torch.arange(B, dtype=x.dtype, device=dev)
u0 = amps[:, None, None] * datagen.u_init_fn(x)[None]
p0 = torch.zeros(B, sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype, device=dev)


def start(s, u, p):
  us, ps = (u,) * 3, (p,) * 3
  return dict(sem=s, us=us, ps=ps, Cus=tuple(s.C(v) for v in us), iters=[])


def advance(m, steps):
  times = []
  for _ in range(steps):
    t0 = time.perf_counter()
    u, p, Cu = datagen._solve_one_step(m['sem'], m['us'], m['ps'], m['Cus'], cfg)
    m['us'], m['ps'], m['Cus'] = m['us'][1:] + (u,), m['ps'][1:] + (p,), m['Cus'][1:] + (Cu,)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
  return times

single = [start(sem, u0[b], p0[b]) for b in range(B)]
t_single = advance(single[0], WARM + STEPS)[WARM:]
for m in single[1:]:
  advance(m, WARM + STEPS)
e = start(ens, ens.flatten(u0), ens.flatten(p0))
t_ens = advance(e, WARM + STEPS)[WARM:]
ue = ens.unflatten(e['us'][-1])
diff = max(float((ue[b] - single[b]['us'][-1]).abs().max() /
                 single[b]['us'][-1].abs().max()) for b in range(B))
print(json.dumps({
    'case': f'2D Kolmogorov flow generator, {cfg.resolution}x{cfg.resolution} quads, order {cfg.order}, '
            f'tol {cfg.tol} / atol {cfg.atol}',
    'members': B, 'single_ms_per_step': 1e3 * float(np.mean(t_single)),
    'ensemble_ms_per_step': 1e3 * float(np.mean(t_ens)),
    'ensemble_over_single': float(np.mean(t_ens) / np.mean(t_single)),
    'members_equal': diff, 'steps_timed': STEPS,
    'switches': switches.active()}), flush=True)
