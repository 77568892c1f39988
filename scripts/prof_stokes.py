"""Driver for rocprofv3: the fused Stokes divergence / pressure-gradient
kernels (navier_stokes.py:313-338) on the config-4 GPU block: n^3 elements,
p = 7 velocity / P - 2 Gauss pressure, component-major velocity, the per-node Q
of E folded into the pressure gradient and p . E p into the divergence (the two
kernels of one application of E).
Prints one JSON line with HIP-event times and the bytes each launch must move.
env: N (64), P (8), REPS (10)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core import layout, operators
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.core.mesh_refiner import refine_premesh
n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
reps = int(os.environ.get('REPS', '10'))
dev = 'cuda:0'
pm = unit_cube_mesh(n, ndim=3)
quad = Quadrature1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
vsp = FiniteElementSpace.create(refine_premesh(pm, Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize(device=dev), quad)
psp = FiniteElementSpace.create(refine_premesh(pm, Nodes1D.create(P - 2, NodeType.GAUSS_LEGENDRE)).finalize(device=dev), quad)
N, E = vsp.mesh.num_nodes, vsp.mesh.num_elements
nn, npp, s = P ** 3, (P - 2) ** 3, 8
op = operators.StokesDivGrad.create(vsp, psp, vsp.mesh.physical_masks['boundary'], 'auto')
u = layout.component_major(torch.randn(N, 3, dtype=torch.float64, device=dev))
scale = torch.rand(N, dtype=torch.float64, device=dev) + 0.5
p = torch.randn(psp.mesh.num_nodes, dtype=torch.float64, device=dev)
pout, out = torch.empty_like(p), torch.empty_like(u)
so = 0 if op.shared_order is None else 2 * op.shared_order.shape[1]
# bytes a launch has to move (affine / multilinear geometry: 24 reals per
# element); connectivity: 432 B of facet table + 4 B of chain list per element,
# or the index row (+ the sorted slot list for the scatter)
facet = op.facet_parts is not None and all('facet_table' in q for q in op.facet_parts)
conn_div = 436 * E if facet else 4 * E * nn
conn_grad = 436 * E if facet else (4 * nn + so) * E
# geometry: 24 map coefficients per element, of which the box kernels read 3
box = facet and all(q['geo_mode'] == operators._GEO_BOX for q in op.facet_parts)
geo = (3 if box else 24) * s * E
# the pair as E = D Q D^T issues it (navier_stokes.py:340-348): the per-node Q
# rides in grad_t (one factor per node), div reads the three components and
# accumulates p . (D w) for the pressure CG.  PAIR=old: round 2's split
# (scale in div, none in grad_t)
old_pair = os.environ.get('PAIR', 'e') == 'old'
must = {
    'stokes_div': conn_div + 3 * s * N + (s * N if old_pair else s * E * npp) + geo + s * E * npp,
    'stokes_grad_t': conn_grad + s * E * npp + geo + 3 * s * N + (0 if old_pair else s * N),
}
dots = torch.zeros(1024, dtype=torch.float64, device=dev)
if old_pair:
  run_div = lambda: op.div(u, scale=scale, out=pout)
  run_grad = lambda: op.grad_t(p, out=out)
else:
  run_div = lambda: op.div(u, out=pout, dot_with=p, dot_out=dots)
  run_grad = lambda: op.grad_t(p, out=out, scale=scale)
def timeit(fn):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sum(a.elapsed_time(b) for a, b in ev) / reps
res = {'n': n, 'P': P, 'geometry': 'box' if box else 'affine / multilinear', 'connectivity': 'facet table + chains' if facet else 'index rows', 'velocity_dofs': 3 * N, 'pressure_dofs': E * npp,
       'pair': 'round 2 (scale in div)' if old_pair else 'as E issues it (Q in grad_t, p . Dw in div)',
       'stokes_div': {'ms': timeit(run_div), 'bytes_must_move': must['stokes_div']},
       'stokes_grad_t': {'ms': timeit(run_grad), 'bytes_must_move': must['stokes_grad_t']}}
for k in ('stokes_div', 'stokes_grad_t'):
  res[k]['frac_of_8TBs'] = res[k]['bytes_must_move'] / (res[k]['ms'] * 1e-3) / 8e12
print(json.dumps(res))
