"""Probe: does a buffer written (or read) by one kernel stay in the Infinity
Cache for the next kernel?  Times a read of `size` MB right after touching it,
against the same read after 2 GB of unrelated traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd import _ops
dev = torch.device('cuda', 0)
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float64, device=dev)   # 2 GB
res = torch.zeros(4, dtype=torch.float64, device=dev)
def timed(fn):
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record(); fn(); b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
  n = mb * 1024 * 1024 // 8
  x = torch.randn(n, dtype=torch.float64, device=dev)
  y = torch.randn(n, dtype=torch.float64, device=dev)
  out = []
  for prep in ('cold', 'after write', 'after read'):
    ts = []
    for _ in range(5):
      flush.fill_(1.0); torch.cuda.synchronize()
      if prep == 'after write':
        _ops.axpby(1.0, y, 0.0, x)          # x <- y   (writes x)
      elif prep == 'after read':
        _ops.dot(x, x, res, 1)
      torch.cuda.synchronize()
      ts.append(timed(lambda: _ops.dot(x, x, res, 0)))
    t = sorted(ts)[len(ts) // 2]
    out.append(f'{prep}: {t*1e3:7.1f} us = {2 * mb / 1024 / (t * 1e-3):6.2f} TB/s'.replace('2 *', ''))
  print(f'{mb:5d} MB  ' + '   '.join(out), flush=True)
