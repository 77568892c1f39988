"""Calibration: achievable HBM rates of simple access patterns on this box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd import _ops
from swirl_fem_amd.distributed import blocks
dev = torch.device('cuda:0')
def timeit(label, fn, nbytes, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s0.record()
  for _ in range(reps): fn()
  s1.record(); torch.cuda.synchronize()
  ms = s0.elapsed_time(s1) / reps
  print(f'{label:44s} {ms:8.3f} ms  {nbytes/ms/1e6:8.1f} GB/s')
n = 800_000_000
a = torch.randn(n, dtype=torch.float64, device=dev)
b = torch.randn(n, dtype=torch.float64, device=dev)
res = torch.zeros(4, dtype=torch.float64, device=dev)
timeit('sfem_dot (read 2 x 6.4 GB)', lambda: _ops.dot(a, b, res, 0), 16 * n)
timeit('sfem_axpby y=ax+by (read 2, write 1)', lambda: _ops.axpby(1.0, a, 0.5, b), 24 * n)
timeit('sfem_axpby y=ax (read 1, write 1)', lambda: _ops.axpby(1.0, a, 0.0, b), 16 * n)
timeit('torch copy_', lambda: b.copy_(a), 16 * n)
timeit('torch sum', lambda: a.sum(), 8 * n)
timeit('torch memset (zero_)', lambda: b.zero_(), 8 * n)
del a, b
part = blocks.build_block_partition(64, 8, (1, 1, 1), 0, device=dev)
mesh = part.mesh
u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev)
E, nn = mesh.elements.shape
ul = _ops.gather(u, mesh.elements, 0.0)
timeit('sfem_gather (idx + u + write E-vector)', lambda: _ops.gather(u, mesh.elements, 0.0), 4*E*nn + 8*mesh.num_nodes + 8*E*nn)
timeit('sfem_scatter_add (memset + idx + read + atomics)', lambda: _ops.scatter_add(ul, mesh.elements, mesh.num_nodes), 4*E*nn + 8*mesh.num_nodes*2 + 8*E*nn)
