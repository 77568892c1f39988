"""E = D Q D^T on index-row kernels: D^T with atomics (default launches of rounds 1-3) against one position per
writer + class sums (`StokesDivGrad.e_layered`).  3D Taylor-Green meshes at orders outside the facet kernels.
  python scripts/time_layered_e.py [n order]..."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swirl_fem_amd.common.premesh_commons import box_mesh
from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
dev = torch.device('cuda', 0)
args = [int(a) for a in sys.argv[1:]] or [24, 4, 16, 9]
for n, order in zip(args[::2], args[1::2]):
  pm = box_mesh((n,) * 3, (0.0,) * 3, (2 * np.pi,) * 3, periodic_dims=(0, 1, 2))
  sem = StokesSEM.create(pm, {}, order=order, device=dev)
  op = sem._divgrad()
  p = torch.randn(sem.pressure.pspace.mesh.num_nodes, dtype=torch.float64, device=dev)
  row = {'mesh': f'{n}^3 hexes, order {order}, periodic', 'elements': n ** 3, 'index_rows': op.facet_parts is None}
  ref = None
  for lay in ('0', '1'):
    os.environ['SFEM_STOKES_LAYERED'] = lay
    for _ in range(3):
      out = sem.E(p, dt=1e-2, time_order=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
      out = sem.E(p, dt=1e-2, time_order=2)
    torch.cuda.synchronize()
    row['ms_layered' if lay == '1' else 'ms_atomic'] = 1e3 * (time.perf_counter() - t0) / 20
    if ref is None: ref = out
    else: row['max_rel_diff'] = float((out - ref).abs().max() / ref.abs().max())
  print(json.dumps(row), flush=True)
