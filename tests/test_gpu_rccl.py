"""The RCCL code path on real hardware with ONE rank: a communicator of size 1
whose rank exchanges with itself.  RCCL refuses two ranks per GPU, so the
multi-rank tests (test_gpu_partitioned.py) run over gloo; here the same calls
-- all-reduce of the CG scalars on views, grouped isend/irecv, the split
start / finish of the overlapped exchange, stream ordering against the compute
kernels -- go through RCCL itself.

A self-neighbour plan makes every listed node its own second holder.  With the
interface weights (1 - 1/2 on those nodes) the consistent-vector CG then solves
(I + S) A x = (I + S) b in the weighted inner product, i.e. A x = b: the answer
must equal the plain single-GPU solve (the iteration count need not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.fixture(scope='module')
def rccl():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  try:
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=DEV)
  except Exception as e:     # pylint: disable=broad-except
    pytest.skip(f'RCCL communicator could not be created here: {e}')
  yield
  dist.destroy_process_group()


def test_exchange_and_reductions_over_rccl(rccl):
  from swirl_fem_amd.core import layout
  from swirl_fem_amd.distributed import comm
  idx = np.array([1, 3, 5, 6], dtype=np.int32)
  plan = comm.NeighborPlan(rank=0, neighbors=[0], indices=[idx])
  u = torch.arange(8, dtype=torch.float64, device=DEV)
  want = u.clone()
  want[torch.as_tensor(idx, device=DEV, dtype=torch.int64)] *= 2
  assert torch.equal(comm.neighbor_exchange(u, plan), want)
  assert torch.equal(comm.neighbor_exchange_(u.clone(), plan), want)
  h = comm.neighbor_exchange_start(u, plan)
  v = u.clone()
  assert torch.equal(comm.neighbor_exchange_finish(h, v), want)
  u3 = torch.randn(8, 3, dtype=torch.float32, device=DEV)
  w3 = u3.clone()
  w3[torch.as_tensor(idx, device=DEV, dtype=torch.int64)] *= 2
  assert torch.equal(comm.neighbor_exchange_(layout.component_major(u3), plan),
                     w3)
  s = torch.arange(16, dtype=torch.float64, device=DEV)
  comm.all_reduce_sum_(s[2:3])                       # a view, as CG passes it
  assert float(s[2]) == 2.0


def test_partitioned_cg_over_rccl_equals_plain_cg(rccl):
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  from swirl_fem_amd.distributed import blocks, comm, solver
  from swirl_fem_amd.linalg.cg import cg
  P = 5
  part = blocks.build_block_partition(4, P, (1, 1, 1), 0, device=DEV,
                                      jitter=0.1)
  mesh = part.mesh
  nodes = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
  fes = FiniteElementSpace.create(mesh,
                                  Quadrature1D.create_from_nodes_1d(nodes))
  bm = mesh.physical_masks['boundary']
  op = fes.helmholtz_operator(bm)
  A = op.linear_operator(0.2, 1.0)
  g = torch.Generator(device=DEV).manual_seed(4)
  b = torch.randn(mesh.num_nodes, dtype=torch.float64, device=DEV,
                  generator=g) * (~bm)
  x_ref, info_ref = cg(A, b, tol=1e-12, maxiter=2000)
  # interface = the nodes of the plane x ~ 0.5 (a fake cut through the block)
  x0 = mesh.node_coords[:, 0]
  cut = torch.nonzero((x0 - 0.5).abs() < 0.13).reshape(-1)
  assert 0 < cut.numel() < mesh.num_nodes
  plan = comm.NeighborPlan(rank=0, neighbors=[0],
                           indices=[cut.to(torch.int32).cpu().numpy()])
  for local_op in (A, solver.OverlappedHelmholtz(op, plan, 0.2, 1.0)):
    x, info = solver.cg(local_op, b, plan, tol=1e-12, maxiter=2000,
                        assembled_rhs=False)
    # b was unassembled: the solver doubled it on the cut, as A is doubled
    assert float((x - x_ref).abs().max()) < 1e-9 * float(x_ref.abs().max())
    # (I + S) acts as a preconditioner here, so the iteration count differs
    # from the plain solve; in a real partition the copies live on different
    # ranks and the iterates are those of CG on the global operator
    assert 0 < info['num_iterations'] < 2000
  ov = solver.OverlappedHelmholtz(op, plan, 0.2, 1.0)
  assert 0 < ov.num_boundary_elements < mesh.num_elements


def test_router_pscan_and_discovery_over_rccl(rccl):
  """`communication/` and `distributed/discover.py` through RCCL itself (one
  rank): the sparse all-to-all degenerates to a permutation to self, the scans
  to their units, and the discovered plan of a mesh that is periodic onto
  itself is empty of remote neighbours -- every call runs on device tensors
  over the nccl backend (the multi-rank behaviour is covered over gloo in
  tests/test_distributed_cpu.py)."""
  from swirl_fem_amd.communication.crystal_router import crystal_router
  from swirl_fem_amd.communication.pscan import preduce, pscan
  from swirl_fem_amd.distributed import blocks
  from swirl_fem_amd.distributed.discover import discover_neighbors
  g = torch.Generator(device=DEV).manual_seed(3)
  n = 37
  payload = torch.randint(0, 10 ** 6, (n, 3), device=DEV, generator=g)
  tag = torch.arange(n, device=DEV)
  target = torch.zeros(n, dtype=torch.int64, device=DEV)
  n_out, (pay_o, tag_o), src = crystal_router(None, [payload, tag], target)
  assert n_out == n and bool((src == 0).all())
  order = torch.argsort(tag_o)
  assert torch.equal(pay_o[order], payload) and pay_o.is_cuda
  x = torch.tensor([5, 7], dtype=torch.int64, device=DEV)
  ex, tot = pscan(x, 'add', reduction=True)
  assert ex.tolist() == [0, 0] and tot.tolist() == [5, 7] and ex.is_cuda
  for name, unit in (('multiply', 1), ('bitwise_and', -1), ('bitwise_xor', 0),
                     ('minimum', torch.iinfo(torch.int64).max)):
    ex, tot = pscan(x, name, reduction=True)
    assert ex.tolist() == [unit, unit] and tot.tolist() == [5, 7], name
  tree = {'a': torch.tensor([1.5], dtype=torch.float64, device=DEV),
          'b': [torch.tensor([True, False], device=DEV)]}
  red = preduce(tree, 'maximum')
  assert float(red['a']) == 1.5 and red['b'][0].tolist() == [True, False]
  assert float(preduce(tree['a'], torch.add)) == 1.5
  part = blocks.build_block_partition(2, 3, (1, 1, 1), 0, device=DEV)
  plan = discover_neighbors(part.global_keys, device=DEV)
  assert plan.neighbors == [] and plan.num_shared == 0
