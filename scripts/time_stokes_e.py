"""The two kernels of E = D Q D^T on the triply periodic n^3 box (config 4 block), timed one by one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
n = int(os.environ.get('N', '64'))
dev = torch.device('cuda', 0)
per = (0, 1, 2) if os.environ.get('PERIODIC', '1') == '1' else ()
sem = StokesSEM.create(unit_cube_mesh(n, ndim=3, a=0.0, b=2 * np.pi, periodic_dims=per),
                       {} if per else {'boundary': (1, 0.0)}, order=7, device=dev)
op = sem._divgrad()
Nv, Np = sem.velocity.mesh.num_nodes, sem.pressure.pspace.mesh.num_nodes
p = torch.randn(Np, dtype=torch.float64, device=dev)
scale = torch.rand(Nv, dtype=torch.float64, device=dev) + 0.5
w = layout.empty_component_major((Nv, 3), p.dtype, dev)
def t(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
parts = op._parts_for(w)
print('facet parts' if op.facet_parts is not None and parts is op.facet_parts else 'index rows',
      [(q['geo_mode'], 'facet' if 'facet_table' in q else 'rows',
        (q['chains'][0].numel() - 1) if 'chains' in q else None) for q in parts])
print('n=%d periodic=%s  grad_t(scale) %.3f  exchange %.3f  div %.3f  div(dot) %.3f ms' % (
    n, bool(per), t(lambda: op.grad_t(p, out=w, scale=scale)),
    t(lambda: sem.velocity.exchange(w, inplace=True)), t(lambda: op.div(w)),
    t(lambda: op.div(w, dot_with=p, dot_out=torch.zeros(1024, dtype=torch.float64, device=dev)))))
