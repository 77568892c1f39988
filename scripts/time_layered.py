"""Layered vs atomic assembly on the config-2 mesh: the apply alone, the
`r -= alpha Ap` update alone, and whole CG iterations (HIP events, medians).
env: N (64), P (8), REPS (30), DTYPE (f64), JITTER, GEOMETRY (auto)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd import _lib, _ops
from swirl_fem_amd.core import operators
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.linalg.cg import CGRunner
n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
reps = int(os.environ.get('REPS', '30'))
dt = torch.float64 if os.environ.get('DTYPE', 'f64') == 'f64' else torch.float32
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt,
                                    jitter=float(os.environ.get('JITTER', '0')))
mesh = part.mesh
grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
op = operators.HelmholtzOperator.create(
    fes, mesh.physical_masks.get('boundary'), os.environ.get('GEOMETRY', 'auto'))
plan = op.layer_plan()
N = mesh.num_nodes
res = {'N': N, 'plan': None if plan is None else {
    'layers': plan.layers, 'extent': plan.extent, 'written': plan.written},
       'kernel': op.kernel_name(), 'kernel_layered':
       None if plan is None else op.kernel_name(layered=True)}


def timed(fn):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(reps)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return round(ts[len(ts) // 2], 4)


u = torch.randn(N, dtype=dt, device=dev)
out = torch.empty_like(u)
res['apply_atomic_ms'] = timed(lambda: op.apply(u, 0.0, 1.0, out=out))
if plan is not None:
  ext = op.new_extended()
  res['apply_layered_ms'] = timed(lambda: op.apply_layered(u, ext, 0.0, 1.0))
  scal = torch.zeros(_lib.SFEM_CG_NSCALARS, dtype=torch.float64, device=dev)
  scal[0] = 1.0; scal[1] = 3.0
  r = torch.randn(N, dtype=dt, device=dev)
  res['update_r_ms'] = timed(lambda: _ops.cg_update_r(r, out, scal, 2))
  res['update_r_layered_ms'] = timed(
      lambda: _ops.cg_update_r_layered(r, ext, plan.layers, scal, 2,
                                       masks=plan.masks))
  res['update_r_layered_unmasked_ms'] = timed(
      lambda: _ops.cg_update_r_layered(r, ext, plan.layers, scal, 2))
  res['layer_values_read'] = plan.read
  res['fold_ms'] = timed(lambda: _ops.fold_layers(ext, N, plan.layers))
b = torch.randn(N, dtype=dt, device=dev) * ~mesh.physical_masks['boundary']
for name, env in (('cg_layered_ms', '1'), ('cg_atomic_ms', '0')):
  os.environ['SFEM_LAYERED'] = env
  o2 = operators.HelmholtzOperator.create(
      fes, mesh.physical_masks.get('boundary'),
      os.environ.get('GEOMETRY', 'auto'))
  run = CGRunner(o2.linear_operator(0.0, 1.0), b, tol=0.0, maxiter=10 ** 9)
  res[name.replace('_ms', '_on')] = run.layered is not None
  for _ in range(10):
    run.step()
  torch.cuda.synchronize()
  a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(50):
    run.step()
  bb.record(); torch.cuda.synchronize()
  res[name] = round(a.elapsed_time(bb) / 50, 4)
  del run, o2
print(json.dumps(res))
