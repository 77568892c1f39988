"""Values-only sfem_basis_eval / sfem_basis_eval_t: generic kernels (SFEM_INTERP=0) against the
compile-time-sized interpolation (csrc/sfem_interp.h).  python scripts/time_interp.py [E]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd import _ops
dev = torch.device('cuda', 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for d, P, q, nc in [(3, 8, 10, 3), (3, 8, 7, 3), (3, 7, 8, 3), (2, 9, 11, 2)]:
  EE = E if d == 3 else 8 * E
  i1 = torch.randn(q, P, dtype=torch.float64, device=dev); g1 = torch.randn_like(i1)
  u = torch.randn(EE, P ** d, nc, dtype=torch.float64, device=dev)
  c0 = torch.randn(EE, q ** d, nc, dtype=torch.float64, device=dev)
  w = torch.rand(EE, q ** d, dtype=torch.float64, device=dev)
  row = {'ndim': d, 'P': P, 'q': q, 'nc': nc, 'elements': EE,
         'bytes_fwd_gb': 8e-9 * EE * nc * (P ** d + q ** d), 'bytes_t_gb': 8e-9 * EE * (nc * (P ** d + q ** d) + q ** d)}
  for v in ('0', '1'):
    os.environ['SFEM_INTERP'] = v
    for name, fn in (('fwd', lambda: _ops.basis_eval(u, i1, g1, None, d, P, q, False, True, False)),
                     ('t', lambda: _ops.basis_eval_t(c0, None, i1, g1, None, w, d, P, q, nc, False))):
      fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(5): fn()
      torch.cuda.synchronize()
      row[f'{name}_ms_' + ('generic' if v == '0' else 'sized')] = 1e3 * (time.perf_counter() - t0) / 5
  print(json.dumps(row), flush=True)
