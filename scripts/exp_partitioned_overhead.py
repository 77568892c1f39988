"""Cost of the partitioned CG machinery on ONE GPU: a one-rank RCCL world whose
rank exchanges three faces of its 64^3 block with itself (3 x 449^2 interface
nodes, about what one of 8 GPUs exchanges), against the plain single-GPU CG."""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.distributed import blocks, comm, solver
from swirl_fem_amd.linalg.cg import CGRunner
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
with socket.socket() as s:
  s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
n, P = int(os.environ.get('N', '64')), 8
part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev)
mesh = part.mesh
fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)))
op = fes.helmholtz_operator(mesh.physical_masks['boundary'])
x = mesh.node_coords
ids = [torch.nonzero((x[:, d] - 1.0).abs() < 1e-12).reshape(-1).cpu().numpy().astype(np.int32) for d in range(3)]
iface = np.unique(np.concatenate(ids))
plan = comm.NeighborPlan(rank=0, neighbors=[0], indices=[iface])
b = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev) * (~mesh.physical_masks['boundary'])
def timeit(run, steps=50, warm=10):
  for _ in range(warm): run.step()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): run.step()
  th = time.perf_counter() - t0          # host time to enqueue
  torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps, 1e3 * th / steps
plain = CGRunner(op.linear_operator(0.0, 1.0), b, tol=0.0, maxiter=10 ** 9)
print('plain CG            %.3f ms / iteration (host enqueue %.3f)' % timeit(plain))
for name, A in (('overlapped', solver.OverlappedHelmholtz(op, plan, 0.0, 1.0)), ('not overlapped', op.linear_operator(0.0, 1.0))):
  run = solver.make_runner(A, b, plan, tol=0.0, atol=0.0, maxiter=10 ** 9)
  print('partitioned, %-14s %.3f ms / iteration (host enqueue %.3f), %d interface nodes' % ((name,) + timeit(run) + (len(iface),)))
dist.destroy_process_group()
