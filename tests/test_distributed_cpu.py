"""World-size-2 (and 4) gloo tests of the partitioned exchange on CPU.

One process per partition, `torch.distributed` with the gloo backend.  The
product's communication code (`NeighborPlan`, `exchange_buffers`,
`all_reduce_sum_`) runs unchanged; the pack / unpack-add halves around it are
HIP kernels in the product (`neighbor_exchange`, GPU tests), so here the test
does them itself with the oracle's gather / scatter.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sfem_oracle as O
from swirl_fem_amd.distributed import blocks, comm


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _oracle_pack(u, idx):
  return torch.from_numpy(O.gather(u.numpy(), idx.numpy(), 0.))


def _oracle_unpack_add(buf, idx, u):
  out = u.numpy()
  np.add.at(out, idx.numpy(), buf.numpy())
  return u


def _worker(rank, world, port, grid, n, P, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    part = blocks.build_block_partition(n, P, grid, rank, device='cpu')
    assert comm.get_rank() == rank and comm.get_world_size() == world
    rng = np.random.default_rng(100 + rank)
    u = torch.from_numpy(rng.standard_normal(part.mesh.num_nodes))
    idx = [torch.as_tensor(ix) for ix in part.plan.indices]
    recv = comm.exchange_buffers(part.plan, [_oracle_pack(u, ix) for ix in idx])
    out = u.clone()
    for rb, ix in zip(recv, idx):
      _oracle_unpack_add(rb, ix, out)
    # Mesh.exchange routes to the same call in the partitioned case
    assert part.mesh.axis_name == 'blocks'
    assert part.mesh.neighbor_plan is part.plan
    # global inner product of a dual (unassembled) with a primal (assembled)
    # vector = plain local dot + all-reduce   
    s = torch.tensor([float(torch.dot(u, out))], dtype=torch.float64)
    comm.all_reduce_sum_(s)
    results[rank] = (u.numpy(), out.numpy(),
                     part.mesh.node_coords.numpy().copy(), float(s))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('grid,n,P', [((2, 1, 1), 2, 3), ((2, 2, 1), 2, 4),
                                      ((2, 1), 3, 4),
                                      # non-cubic blocks (strong scaling)
                                      ((2, 1, 1), (1, 2, 2), 3),
                                      ((2, 2), (2, 3), 4)])
def test_neighbor_exchange_gloo(grid, n, P):
  world = int(np.prod(grid))
  port = _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, grid, n, P, results), nprocs=world,
             join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  # expected QQ^T: sum of all copies of a node, identified by coordinates
  tot = {}
  for r in range(world):
    u, _, xc, _ = res[r]
    for k, v in zip(map(tuple, np.round(xc * 1e7).astype(np.int64)), u):
      tot[k] = tot.get(k, 0.0) + v
  n_shared = 0
  for r in range(world):
    u, out, xc, _ = res[r]
    ref = np.array([tot[k] for k in map(tuple,
                                        np.round(xc * 1e7).astype(np.int64))])
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13)
    n_shared += int((np.abs(out - u) > 0).sum())
  assert n_shared > 0
  # u^T (QQ^T u) summed over ranks equals the squared norm of the assembled
  # vector counted once per unique node
  expected = sum(v * v for v in tot.values())
  for r in range(world):
    assert res[r][3] == pytest.approx(expected, rel=1e-12)


def test_plan_matches_reference_style_gather_indices():
  """Neighbour lists from the reference's dense (P, S) table == block plan."""
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  pm = unit_cube_mesh(4, ndim=2, partitions=np.arange(4).reshape(2, 2))
  rp = refine_premesh(pm, Nodes1D.create(3, NodeType.GAUSS_LOBATTO_LEGENDRE))
  arrs = rp.finalize_all('i')
  gi = arrs['exchange_gather_indices']
  us = np.random.default_rng(0).standard_normal(arrs['node_indices'].shape)
  ref = O.exchange_partitioned(us, gi)
  plans = [comm.NeighborPlan.from_gather_indices(gi, r) for r in range(4)]
  for r, plan in enumerate(plans):
    out = us[r].copy()
    for q, ix in zip(plan.neighbors, plan.indices):
      j = plans[q].neighbors.index(r)
      out[ix] += us[q][plans[q].indices[j]]
    np.testing.assert_allclose(out, ref[r], rtol=0, atol=1e-14)
    assert plan.num_shared == sum(len(i) for i in plan.indices)


def _router_worker(rank, world, port, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from swirl_fem_amd.communication.crystal_router import crystal_router
    from swirl_fem_amd.communication.pscan import preduce, pscan
    from swirl_fem_amd.distributed.discover import discover_neighbors
    rng = np.random.default_rng(50 + rank)
    n = int(rng.integers(0, 40))
    target = torch.from_numpy(rng.integers(0, world, n))
    payload = torch.from_numpy(rng.integers(0, 10 ** 6, (n, 3)))
    tag = torch.full((n,), rank, dtype=torch.int64)
    n_out, (pay_o, tag_o), src = crystal_router(None, [payload, tag], target)
    assert n_out == len(src) and bool((tag_o == src).all())
    # round trip: everything returns to its sender
    n_back, (pay_b, tag_b), src_b = crystal_router(n_out, [pay_o, tag_o], src)
    assert n_back == n and bool((tag_b == rank).all())
    a = np.sort(pay_b.numpy().view([('', pay_b.numpy().dtype)] * 3), axis=0)
    b = np.sort(payload.numpy().view([('', payload.numpy().dtype)] * 3), axis=0)
    assert np.array_equal(a, b)
    # scans
    x = torch.tensor([rank + 1, 10 * (rank + 1)], dtype=torch.int64)
    ex, tot = pscan(x, 'add', reduction=True)
    assert ex.tolist() == [sum(range(1, rank + 1)),
                           10 * sum(range(1, rank + 1))]
    assert tot.tolist() == [sum(range(1, world + 1)),
                            10 * sum(range(1, world + 1))]
    assert int(pscan(torch.tensor([rank]), 'maximum')) == (
        rank - 1 if rank else torch.iinfo(torch.int64).min)
    assert int(preduce(torch.tensor([rank]), 'maximum')) == world - 1
    # the reference's seven monoids (pscan.py:42-51), by name and by function
    ranks = np.arange(world)
    vals = (3 * ranks + 1).astype(np.int64)           # 1, 4, 7, ...
    mine = torch.tensor([int(vals[rank]), int(vals[rank]) ^ 5])
    both = np.stack([vals, vals ^ 5], axis=1)
    for name, fn, unit in (('multiply', np.multiply, 1),
                           ('minimum', np.minimum, np.iinfo(np.int64).max),
                           ('bitwise_and', np.bitwise_and, -1),
                           ('bitwise_or', np.bitwise_or, 0),
                           ('bitwise_xor', np.bitwise_xor, 0)):
      want = np.full(2, unit, dtype=np.int64)
      for r in range(rank):
        want = fn(want, both[r])
      total = np.full(2, unit, dtype=np.int64)
      for r in range(world):
        total = fn(total, both[r])
      ex, tot = pscan(mine, name, reduction=True)
      assert ex.tolist() == want.tolist(), (name, ex, want)
      assert tot.tolist() == total.tolist(), name
      assert preduce(mine, getattr(torch, name)).tolist() == total.tolist()
    flags = torch.tensor([rank % 2 == 0, rank == 1])
    assert pscan(flags, 'bitwise_or').tolist() == [rank > 0, rank > 1]
    assert preduce(flags, torch.bitwise_and).tolist() == [world == 1, False]
    # pytrees: mapped over the leaves, structure preserved (pscan.py:225-241)
    tree = {'n': torch.tensor([rank + 1]),
            'x': (torch.tensor([0.5 * (rank + 1)], dtype=torch.float64),
                  [torch.tensor([[rank, 1]], dtype=torch.int32)])}
    ex, tot = pscan(tree, torch.add, axis_name='parts', reduction=True)
    tri = rank * (rank + 1) // 2
    assert int(ex['n']) == tri and float(ex['x'][0]) == 0.5 * tri
    assert ex['x'][1][0].tolist() == [[rank * (rank - 1) // 2, rank]]
    assert ex['x'][1][0].dtype == torch.int32 and isinstance(ex['x'], tuple)
    assert int(tot['n']) == world * (world + 1) // 2
    assert preduce(tree, 'maximum')['x'][1][0].tolist() == [[world - 1, 1]]
    with pytest.raises(ValueError):
      pscan(mine, 'subtract')
    with pytest.raises(TypeError):
      pscan(torch.tensor([1.0]), 'bitwise_or')
    # neighbour discovery == the block builder's lattice-based plan
    grid = {2: (2, 1, 1), 3: (3, 1), 4: (2, 2, 1), 5: (5, 1)}[world]
    part = blocks.build_block_partition(2, 3, grid, rank, device='cpu')
    plan = discover_neighbors(part.global_keys)
    assert plan.neighbors == part.plan.neighbors, (plan.neighbors,
                                                   part.plan.neighbors)
    for i1, i2 in zip(plan.indices, part.plan.indices):
      np.testing.assert_array_equal(i1, i2)
    # ... and the reference-style (P, S) table of a partitioned Gmsh mesh
    from swirl_fem_amd.common import mesh_partitioner, mesh_reader
    from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
    from swirl_fem_amd.core.mesh_refiner import refine_premesh
    pm = mesh_reader.read(os.path.join(os.path.dirname(__file__), 'golden',
                                       'msh', 'cube.msh'), ndim=3)
    rp = refine_premesh(mesh_partitioner.partition(pm, world),
                        Nodes1D.create(3, NodeType.GAUSS_LOBATTO_LEGENDRE))
    arrs = rp.finalize_all('parts')
    central = comm.NeighborPlan.from_gather_indices(
        arrs['exchange_gather_indices'], rank)
    found = discover_neighbors(arrs['global_node_ids'][rank])
    assert found.neighbors == central.neighbors
    for i1, i2 in zip(found.indices, central.indices):
      np.testing.assert_array_equal(i1, i2)
    results[rank] = (n, n_out, target.numpy().tolist())
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4, 5])
def test_crystal_router_pscan_and_discovery(world):
  """communication/: sparse all-to-all (odd group sizes included), exclusive
  scan, and the neighbour plan discovered through the router equals the one the
  block builder derives from lattice coordinates."""
  port = _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_router_worker, args=(world, port, results), nprocs=world,
             join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  # conservation: rank q received exactly what was addressed to it
  for q in range(world):
    assert res[q][1] == sum(t.count(q) for _, _, t in res.values())


def _periodic_worker(rank, world, port, grid, per, n, P, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    part = blocks.build_block_partition(n, P, grid, rank, device='cpu',
                                        periodic_dims=per)
    rng = np.random.default_rng(200 + rank)
    u = torch.from_numpy(rng.standard_normal(part.mesh.num_nodes))
    plan = part.plan
    out = u.clone()
    if plan.has_local_images:           # 1. sum the images held by this rank
      sums = np.zeros(int(plan.local_unique.max()) + 1)
      np.add.at(sums, plan.local_unique, out.numpy()[plan.local_gather])
      out.numpy()[plan.local_gather] = sums[plan.local_unique]
    idx = [torch.as_tensor(ix) for ix in plan.indices]
    recv = comm.exchange_buffers(plan, [_oracle_pack(out, ix) for ix in idx])
    for rb, ix in zip(recv, idx):       # 2. representatives across ranks
      _oracle_unpack_add(rb, ix, out)
    if plan.has_local_images:           # 3. back to every image
      out.numpy()[plan.local_gather] = out.numpy()[plan.local_rep]
    # the Dirichlet mask is exactly the faces of the non-periodic directions
    x = part.mesh.node_coords.numpy()
    on_wall = np.zeros(len(x), dtype=bool)
    for d in range(len(grid)):
      if d not in per:
        on_wall |= (np.abs(x[:, d]) < 1e-12) | (np.abs(x[:, d] - 1.0) < 1e-12)
    mask = part.mesh.physical_masks.get('boundary')
    np.testing.assert_array_equal(
        on_wall, np.zeros(len(x), bool) if mask is None else mask.numpy())
    has_boundary = 'boundary' in part.mesh.physical_masks
    results[rank] = (part.global_keys, u.numpy(), out.numpy(),
                     part.plan.neighbors, has_boundary, part.num_global_nodes)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('grid,per,n,P', [((2, 1, 1), (0,), 2, 3),
                                          ((2, 2, 1), (0, 1), 2, 3),
                                          ((2, 2), (0, 1), (2, 3), 4),
                                          # single block along a periodic
                                          # direction: local images as well
                                          ((2, 1, 1), (0, 1, 2), 2, 3),
                                          ((2, 1), (0, 1), 3, 4),
                                          ((2, 2, 1), (0, 1, 2), 2, 3),
                                          # ... next to a Dirichlet direction
                                          ((2, 1), (1,), 3, 4),
                                          ((1, 2, 1), (0, 1), 2, 3)])
def test_periodic_block_partitions(grid, per, n, P):
  """Blocks of a box that is periodic across the partition cuts (config 4's
  situation): the neighbour plan comes from the router-based discovery on
  periodic lattice keys; QQ^T sums every image of a node exactly once."""
  world = int(np.prod(grid))
  port = _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_periodic_worker, args=(world, port, grid, per, n, P, results),
             nprocs=world, join=True)
    res = dict(results)
  tot = {}
  self_periodic = any(grid[d] == 1 for d in per)
  for keys, u, _, _, _, _ in res.values():
    assert self_periodic or len(np.unique(keys)) == len(keys)
    for k, v in zip(keys.tolist(), u):
      tot[k] = tot.get(k, 0.0) + v
  assert len(tot) == res[0][5]                    # unique periodic nodes
  for r, (keys, u, out, neighbors, has_boundary, _) in res.items():
    np.testing.assert_allclose(out, [tot[k] for k in keys.tolist()],
                               rtol=0, atol=1e-13)
    assert has_boundary == (len(per) < len(grid))
  if grid == (2, 1, 1):
    assert res[0][3] == [1] and res[1][3] == [0]  # one neighbour, met twice


def test_block_plans_of_the_8_gpu_grid_are_symmetric():
  """The 2 x 2 x 2 block grid of `bench.py --gpus 8` (no process group needed:
  the non-periodic plans are built without communication): every pair of
  ranks lists the same global nodes in the same order."""
  grid = (2, 2, 2)
  parts = [blocks.build_block_partition(3, 4, grid, r, device='cpu')
           for r in range(8)]
  for r, p in enumerate(parts):
    assert p.plan.neighbors == [q for q in range(8) if q != r]
    assert p.mesh.axis_name == 'blocks'
    for q, ix in zip(p.plan.neighbors, p.plan.indices):
      j = parts[q].plan.neighbors.index(r)
      np.testing.assert_array_equal(
          p.global_keys[ix], parts[q].global_keys[parts[q].plan.indices[j]])
  # every global node is held by someone; holders of interior nodes are unique
  allk = np.concatenate([p.global_keys for p in parts])
  assert len(np.unique(allk)) == parts[0].num_global_nodes


def _run_bench(*flags, timeout=600, **extra_env):
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items()
         if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  env.update(extra_env)
  res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *flags],
                       env=env, capture_output=True, text=True,
                       timeout=timeout)
  lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
  return res, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2 ...` as a plain command (no torchrun, no
  WORLD_SIZE): the parent spawns the ranks, they meet over gloo, build their
  blocks and exchange the interface; rank 0 prints exactly one JSON line."""
  res, line = _run_bench('--gpus', '2', '--backend', 'gloo', '--elems', '4',
                         '--dry-run')
  assert res.returncode == 0, res.stderr[-2000:]
  assert line['n_gpus'] == 2 and line['config']['world_size_seen'] == 2
  assert line['config']['blocks'] == '2x1x1'
  assert line['config']['dof_count_via_exchange_ok'] is True
  assert line['config']['interface_values_sent_rank0'] == 29 * 29
  assert sum(l.startswith('{') for l in res.stdout.splitlines()) == 1


def test_bench_rehearses_the_eight_rank_layout():
  """`python bench.py --gpus 8 --dry-run`: the 2 x 2 x 2 block grid of the
  driver's scaling run (7 neighbours per rank: 3 faces, 3 edges, 1 corner),
  eight gloo ranks on the CPU."""
  res, line = _run_bench('--gpus', '8', '--backend', 'gloo', '--elems', '2',
                         '--p', '3', '--dry-run')
  assert res.returncode == 0, res.stderr[-2000:]
  assert line['n_gpus'] == 8 and line['config']['world_size_seen'] == 8
  assert line['config']['blocks'] == '2x2x2'
  assert line['config']['dof_count_via_exchange_ok'] is True
  # rank 0 holds the corner block: 3 faces of 7^2, minus overlaps counted per
  # neighbour -> faces 3 * 49, edges 3 * 7, corner 1 values sent
  assert line['config']['interface_values_sent_rank0'] == 3 * 49 + 3 * 7 + 1
  assert line['config']['dofs_global'] == 13 ** 3


def test_bench_refuses_a_world_size_that_differs_from_gpus():
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
  res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--dry-run'], env=env,
                       capture_output=True, text=True, timeout=300)
  assert res.returncode != 0
  assert 'WORLD_SIZE=4 but --gpus 2' in (res.stderr + res.stdout)


@pytest.mark.gpu
def test_bench_two_ranks_on_the_gpu_as_a_plain_command():
  """The real thing on the GPU box: two ranks sharing the visible GPU (gloo
  transport, because RCCL refuses two ranks per device), kernels running."""
  res, line = _run_bench('--gpus', '2', '--backend', 'gloo', '--elems', '8',
                         '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                         '--no-general')
  assert res.returncode == 0, res.stderr[-2000:]
  assert line['n_gpus'] == 2 and line['value'] > 0
  assert line['config']['world_size_seen'] == 2
  assert len(line['config']['per_rank']['apply_ms']) == 2
  assert line['roofline']['frac'] <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('flags,blocks', [
    (('--elems', '16'), '2x2x2'),
    (('--elems', '16', '--periodic'), '2x2x2'),
    (('--scaling', 'strong', '--elems', '32'), '2x2x2')])
def test_bench_eight_ranks_with_real_kernels_on_one_gpu(flags, blocks):
  """The whole `--gpus 8` bench path with real kernels, on the one GPU of the
  box: 2 x 2 x 2 rank-local blocks (7 neighbours per rank, nodes held by 4 and
  8 ranks), `OverlappedHelmholtz`, the consistent CG with its two scalar
  all-reduces, the per-rank gather and the JSON line -- as eight THREADS of
  one process (`--backend threads`, `distributed/inprocess.py`), because the
  box allows six processes on its card.  `--verify` solves A x = A x* for a
  manufactured x* through the timed operator and solver on all ranks.
  Reference: core/premesh.py:170-222, core/gather_scatter.py:318-358."""
  # (SFEM_LAZY_X_MIN_MB=0: the lazy x update that the 64^3 blocks of the
  # real run take, on these small ones)
  res, line = _run_bench('--gpus', '8', '--backend', 'threads', *flags,
                         '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                         '--no-general', '--verify', timeout=900,
                         SFEM_LAZY_X_MIN_MB='0')
  assert line['config']['switches'] == {'SFEM_LAZY_X_MIN_MB': '0'}
  assert res.returncode == 0, res.stderr[-3000:]
  assert sum(l.startswith('{') for l in res.stdout.splitlines()) == 1
  cfg = line['config']
  assert line['n_gpus'] == 8 and cfg['world_size_seen'] == 8
  assert cfg['blocks'] == blocks
  assert line['value'] is not None and line['value'] > 0
  for key in ('apply_ms', 'exchange_ms', 'interface_values'):
    assert len(cfg['per_rank'][key]) == 8, key
  assert min(cfg['per_rank']['interface_values']) > 0
  assert cfg['verify']['status'] == 'converged', cfg['verify']
  assert cfg['verify']['rel_err_vs_manufactured'] < 1e-8, cfg['verify']
  # the ranks' solver ran the layered assembly over its boundary / interior
  # halves (interface nodes folded before the exchange) and the lazy x update
  assert cfg['assembly'].startswith('layered'), cfg['assembly']
  assert cfg['solver']['lazy_x'] is not None
  if '--periodic' in flags:
    assert cfg['periodic_dims'] == [0, 1, 2]
    assert cfg['dofs_global'] == (2 * 16 * 7) ** 3
  elif '--scaling' in flags:
    assert cfg['dofs_global'] == (32 * 7 + 1) ** 3 and line['scaling'] == 'strong'
  else:
    assert cfg['dofs_global'] == (2 * 16 * 7 + 1) ** 3


@pytest.mark.gpu
def test_bench_four_gloo_ranks_verify_on_the_gpu():
  """Four PROCESS ranks (2 x 2 x 1 blocks) sharing the GPU over gloo, the
  launch path of the driver's scaling run, with the manufactured-solution
  check on the result."""
  res, line = _run_bench('--gpus', '4', '--backend', 'gloo', '--elems', '8',
                         '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                         '--no-general', '--verify', timeout=900)
  assert res.returncode == 0, res.stderr[-3000:]
  cfg = line['config']
  assert line['n_gpus'] == 4 and cfg['world_size_seen'] == 4
  assert cfg['blocks'] == '2x2x1' and len(cfg['per_rank']['apply_ms']) == 4
  assert cfg['verify']['status'] == 'converged', cfg['verify']
  assert cfg['verify']['rel_err_vs_manufactured'] < 1e-8, cfg['verify']


@pytest.mark.parametrize('periodic', [(), (0, 1, 2)])
def test_thread_world_transport(periodic):
  """`distributed/inprocess.py`: eight threads as the ranks of the 2 x 2 x 2
  grid.  Block plans (on the periodic box: through the transport's neighbour
  discovery) are symmetric pair by pair, an exchange of ones counts every
  global node once, reductions and gathers agree on all ranks."""
  import torch
  from swirl_fem_amd.distributed import blocks, comm, inprocess
  grid, n, P = (2, 2, 2), 2, 3
  world = inprocess.ThreadWorld(8)

  def rank_main(rank):
    assert comm.get_rank() == rank and comm.get_world_size() == 8
    part = blocks.build_block_partition(n, P, grid, rank, device='cpu',
                                        periodic_dims=periodic)
    ones = torch.ones(part.mesh.num_nodes, dtype=torch.float64)
    idx = [torch.as_tensor(ix, dtype=torch.int64) for ix in part.plan.indices]
    recv = comm.exchange_buffers(part.plan, [ones[ix] * (rank + 1)
                                             for ix in idx])
    holders = ones.clone()
    for q, rb, ix in zip(part.plan.neighbors, recv, idx):
      assert bool((rb == q + 1).all())
      holders.index_add_(0, ix, torch.ones_like(rb))
    total = comm.all_reduce_sum_(torch.tensor([float((1 / holders).sum())],
                                              dtype=torch.float64))
    top = comm.all_reduce_max_(torch.tensor([float(rank)]))
    every = comm.all_gather(torch.tensor([rank]))
    comm.barrier()
    return part, float(total), float(top), [int(t) for t in every]

  res = world.run(rank_main)
  assert comm.transport() is None
  parts = [res[r][0] for r in range(8)]
  for r in range(8):
    assert abs(res[r][1] - parts[0].num_global_nodes) < 1e-9
    assert res[r][2] == 7.0 and res[r][3] == list(range(8))
    assert len(parts[r].plan.neighbors) == 7
    for q, ix in zip(parts[r].plan.neighbors, parts[r].plan.indices):
      j = parts[q].plan.neighbors.index(r)
      assert np.array_equal(parts[r].global_keys[ix] if not periodic else
                            np.sort(parts[r].global_keys[ix]),
                            parts[q].global_keys[parts[q].plan.indices[j]]
                            if not periodic else
                            np.sort(parts[q].global_keys[parts[q].plan.indices[j]]))


def test_thread_world_reports_the_failing_rank():
  from swirl_fem_amd.distributed import comm, inprocess
  world = inprocess.ThreadWorld(3)

  def rank_main(rank):
    if rank == 1:
      raise ValueError('boom on rank one')
    comm.barrier()

  with pytest.raises(RuntimeError, match='rank 1 failed(.|\n)*boom'):
    world.run(rank_main)
  assert comm.transport() is None
