"""Timing experiments on the fused apply (run on the GPU box)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd import _ops

n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, jitter=float(os.environ.get('JITTER', '0')), tile=int(os.environ.get('TILE', '0')))
mesh = part.mesh
grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
op = fes.helmholtz_operator(mesh.physical_masks.get('boundary'))
u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev)
out = torch.empty_like(u)
E, nn = mesh.elements.shape
alg = 4 * E * nn + 8 * mesh.num_nodes * 2 + 48 * E * nn
print('shared nodes', mesh.assembly_plan().num_shared, 'zero range', op.zero_range, 'N', mesh.num_nodes)

def timeit(label, fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s0.record()
  for _ in range(reps): fn()
  s1.record(); torch.cuda.synchronize()
  ms = s0.elapsed_time(s1) / reps
  print(f'{label:40s} {ms:8.3f} ms  {alg/ms/1e6:8.1f} GB/s alg')
  return ms

bm = mesh.physical_masks.get('boundary')
ops = {g: fes.helmholtz_operator(bm, g) for g in ('auto', 'multilinear', 'stored')}
for g, o in ops.items():
  print(g, 'affine', o.num_affine, 'multilinear', o.num_multilinear, 'curved', o.num_curved)
for rep in range(2):
  for g, o in ops.items():
    timeit(f'kernel geometry={g}', lambda o=o: o.apply(u, 0.0, 1.0, out=out, zero=False))
for g, o in ops.items():
  timeit(f'helmholtz geometry={g}', lambda o=o: o.apply(u, 0.7, 1.0, out=out, zero=False))
  timeit(f'mass geometry={g}', lambda o=o: o.apply(u, 1.0, 0.0, out=out, zero=False))
import time
t0 = time.time(); opc = fes.helmholtz_operator(bm, 'auto', 'colored'); torch.cuda.synchronize()
print('colouring setup s', time.time() - t0, 'launches', len(opc.parts), 'colors', mesh.assembly_plan().coloring()[1])
ops['auto-colored'] = opc
for rep in range(2):
  timeit('apply auto colored', lambda: opc.apply(u, 0.0, 1.0, out=out))
  timeit('apply auto atomic', lambda: ops['auto'].apply(u, 0.0, 1.0, out=out))
opsc = fes.helmholtz_operator(bm, 'stored', 'colored')
timeit('apply stored colored', lambda: opsc.apply(u, 0.0, 1.0, out=out))
timeit('apply stored atomic', lambda: ops['stored'].apply(u, 0.0, 1.0, out=out))
u3 = torch.randn(mesh.num_nodes, 3, dtype=torch.float64, device=dev)
o3 = torch.empty_like(u3)
for g, o in ops.items():
  ms = timeit(f'3-component stiffness geometry={g}', lambda o=o: o.apply(u3, 0.0, 1.0, out=o3, zero=False))
ucm = torch.randn(3, mesh.num_nodes, dtype=torch.float64, device=dev).t()
ocm = torch.empty_like(ucm)
for g in ('auto', 'stored'):
  timeit(f'3-component, component-major, geometry={g}', lambda o=ops[g]: o.apply(ucm, 0.0, 1.0, out=ocm, zero=False))
a = torch.empty(mesh.num_nodes * 10, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
ms = timeit('torch copy 7.2GB+7.2GB', lambda: b.copy_(a))
print('copy GB/s', 2 * a.numel() * 8 / ms / 1e6)
