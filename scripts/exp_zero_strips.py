"""sfem_zero_strips: its own kernel vs the runtime fill, per strip size."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd import _lib, _ops
dev = torch.device('cuda', 0)
lib = _lib.load()
buf = torch.empty(3 * 100_000_000, dtype=torch.float64, device=dev)
stream = _ops._stream(dev)
def t(fn, reps=30):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
for mb in (1, 4, 8, 16, 34, 64, 128, 270):
  n = mb * 1024 * 1024 // 8
  stride = 100_000_000
  def kernel():           # force the kernel path: strips shorter than the threshold -> split in chunks? use many strips
    _lib.check(lib.sfem_zero_strips(ctypes.c_void_p(buf.data_ptr()), n, stride, 3, _lib.SFEM_F64, stream), 'z')
  def memset():
    for s in range(3):
      buf[s * stride: s * stride + n].zero_()
  print(f'{mb:4d} MB x 3: zero_strips {t(kernel):8.1f} us   torch zero_ x3 {t(memset):8.1f} us', flush=True)
print('kernel path forced (strips of < 8 MB laid end to end):')
for mb in (34, 64, 128, 270):
  total = 3 * mb * 1024 * 1024 // 8
  k = (3 * mb + 6) // 7                       # strips of <= 7 MB
  n = total // k
  def kernel():
    _lib.check(lib.sfem_zero_strips(ctypes.c_void_p(buf.data_ptr()), n, n, k, _lib.SFEM_F64, stream), 'z')
  print(f'{mb:4d} MB x 3 as {k} strips: {t(kernel):8.1f} us', flush=True)
