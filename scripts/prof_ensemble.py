"""rocprofv3 driver: steps of the Kolmogorov generator as an ensemble of B flows
(`StokesSEM.ensemble`).  B=8 STEPS=10 rocprofv3 --kernel-trace --stats -- python3 scripts/prof_ensemble.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.niles.datagen import datagen
B = int(os.environ.get('B', '8')); STEPS = int(os.environ.get('STEPS', '10'))
dev = torch.device('cuda', 0)
cfg = datagen.DatagenConfig()
sem = datagen.create_sem(cfg, dev)
ens = sem.ensemble(B) if B > 1 else sem
x = sem.velocity.mesh.node_coords
amps = 1.0 + 0.05 * torch.arange(B, dtype=x.dtype, device=dev)
u = (amps[:, None, None] * datagen.u_init_fn(x)[None]).reshape(-1, 2)
p = torch.zeros(ens.pressure.pspace.mesh.num_nodes, dtype=x.dtype, device=dev)
us, ps = (u,) * 3, (p,) * 3
Cus = tuple(ens.C(v) for v in us)
its = []
for _ in range(STEPS):
  f = datagen.forcing(ens.velocity.mesh.node_coords, us[-1], cfg.drag_coeff)
  from swirl_fem_amd.examples.navier_stokes_driver import navier_stokes_step
  un, pn, cn, aux = navier_stokes_step(ens, us, ps, Cus, reynolds=cfg.reynolds_number, dt=cfg.dt,
                                       time_order=cfg.time_order, forcing=f, tol=cfg.tol, atol=cfg.atol)
  us, ps, Cus = us[1:] + (un,), ps[1:] + (pn,), Cus[1:] + (cn,)
  its.append((aux['u_star_info']['num_iterations'], aux['dp_info']['num_iterations']))
torch.cuda.synchronize()
print('iterations (velocity, pressure) per step:', its)
