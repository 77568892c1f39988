"""Times the fused apply (HIP events) on the config-2 mesh; one line of JSON.
env: N (64), P (8), REPS (30), GEOMETRY (auto), DTYPE (f64), MASS (0), JITTER"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from swirl_fem_amd.distributed import blocks
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
n = int(os.environ.get('N', '64')); P = int(os.environ.get('P', '8'))
reps = int(os.environ.get('REPS', '30'))
dt = torch.float64 if os.environ.get('DTYPE', 'f64') == 'f64' else torch.float32
dev = torch.device('cuda:0')
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, dtype=dt,
                                    jitter=float(os.environ.get('JITTER', '0')))
mesh = part.mesh
grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
res = {'lib': os.path.basename(os.environ.get('SFEM_LIB', 'default')),
       'tag': os.environ.get('TAG', '')}
mass = float(os.environ.get('MASS', '0'))
for geometry in os.environ.get('GEOMETRY', 'auto').split(','):
  op = fes.helmholtz_operator(mesh.physical_masks.get('boundary'), geometry)
  u = torch.randn(mesh.num_nodes, dtype=dt, device=dev)
  out = torch.empty_like(u)
  for _ in range(5):
    op.apply(u, mass, 1.0, out=out)
  torch.cuda.synchronize()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(reps)]
  for a, b in ev:
    a.record(); op.apply(u, mass, 1.0, out=out); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  res[geometry] = {'median_ms': round(ts[len(ts) // 2], 4), 'min_ms': round(ts[0], 4),
                   'kernel': op.kernel_name(mass, 1.0)}
  del op
print(json.dumps(res))
