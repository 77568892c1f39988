"""Multi-rank solves on ONE MI355X: 2 and 4 processes share cuda:0.

RCCL refuses two ranks on one device, so these rehearsals use the gloo backend
(interface buffers staged through the host, `comm.exchange_buffers`); every
kernel on the path -- fused apply with p.Ap, strided pack, atomic unpack-add,
CG updates -- is the HIP one.  Checked: the consistent-vector CG
(`distributed/solver.py`) against a manufactured solution, against the
reference's partitioned convention (unassembled A, M = exchange,
navier_stokes.py:436-438) and for agreement of the replicated scalars.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, grid, n, P, dtype_name, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from swirl_fem_amd.core.fespace import FiniteElementSpace
    from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
    from swirl_fem_amd.core.interpolation import Quadrature1D
    from swirl_fem_amd.distributed import blocks, comm, solver
    from swirl_fem_amd.linalg.cg import cg
    dtype = getattr(torch, dtype_name)
    dev = torch.device('cuda', 0)
    part = blocks.build_block_partition(n, P, grid, rank, device=dev,
                                        dtype=dtype, jitter=0.15)
    mesh = part.mesh
    nodes = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
    fes = FiniteElementSpace.create(mesh,
                                    Quadrature1D.create_from_nodes_1d(nodes))
    op = fes.helmholtz_operator(mesh.physical_masks['boundary'])
    A = op.linear_operator(0.3, 1.0)
    x = mesh.node_coords
    x_true = torch.ones_like(x[:, 0])
    for d in range(x.shape[1]):
      x_true = x_true * torch.sin(np.pi * x[:, d]) * (1.0 + 0.3 * x[:, d])
    x_true = x_true * (~mesh.physical_masks['boundary']).to(dtype)
    b_local = A(x_true)                                  # unassembled
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    xs, info = solver.cg(A, b_local, part.plan, tol=tol, maxiter=2000)
    # boundary elements first, exchange overlapped with the interior elements
    Ao = solver.OverlappedHelmholtz(op, part.plan, 0.3, 1.0)
    assert 0 < Ao.num_boundary_elements <= mesh.num_elements
    xo, info_o = solver.cg(Ao, b_local, part.plan, tol=tol, maxiter=2000)
    w_ref = comm.neighbor_exchange_(A(x_true), part.plan)
    xr, info_r = cg(A, b_local, tol=tol, maxiter=2000, M=mesh.exchange,
                    reduce_fn=part.reduce_sum_)
    # in-place single-launch exchange == reference-style exchange
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    v = torch.randn(mesh.num_nodes, 3, dtype=dtype, device=dev, generator=g)
    e1 = mesh.exchange(v)
    e2 = comm.neighbor_exchange_(v.clone(), part.plan)
    from swirl_fem_amd.core import layout
    e3 = comm.neighbor_exchange_(layout.component_major(v), part.plan)
    results[rank] = dict(
        err=float((xs - x_true).abs().max() / x_true.abs().max()),
        err_ref=float((xs - xr).abs().max() / x_true.abs().max()),
        its=info['num_iterations'], its_ref=info_r['num_iterations'],
        its_o=info_o['num_iterations'],
        err_o=float((xo - xs).abs().max() / x_true.abs().max()),
        app_o=float((Ao(x_true) - w_ref).abs().max() / w_ref.abs().max()),
        res=float(info['residual']),
        ex12=float((e1 - e2).abs().max()), ex13=float((e1 - e3).abs().max()),
        shared=part.plan.num_shared)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('grid,n,P,dtype_name,lazy', [
    ((2, 1, 1), 3, 4, 'float64', False),
    ((2, 2, 1), 2, 5, 'float64', False),
    ((2, 1, 1), 2, 8, 'float32', False),
    # the lazy solution update (vectors of 256 MB and more: every 64^3 block
    # of the scaling run) next to the all-reduces and the interface correction
    ((2, 2, 1), 2, 5, 'float64', True),
    ((2, 1, 1), 2, 8, 'float32', True),
])
def test_partitioned_cg_on_one_gpu(grid, n, P, dtype_name, lazy, monkeypatch):
  if lazy:        # (the rank processes inherit the environment)
    monkeypatch.setenv('SFEM_LAZY_X_MIN_MB', '0')
  world = int(np.prod(grid))
  port = _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, grid, n, P, dtype_name, results),
             nprocs=world, join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  f64 = dtype_name == 'float64'
  for r in range(world):
    assert res[r]['shared'] > 0
    assert res[r]['err'] < (1e-9 if f64 else 2e-4), res[r]
    assert res[r]['err_ref'] < (1e-9 if f64 else 2e-4), res[r]
    assert res[r]['err_o'] < (1e-9 if f64 else 2e-4), res[r]
    assert res[r]['app_o'] < (1e-12 if f64 else 1e-5), res[r]
    assert abs(res[r]['its_o'] - res[r]['its']) <= 1, res[r]
    assert res[r]['ex12'] < (1e-13 if f64 else 1e-5), res[r]
    assert res[r]['ex13'] < (1e-13 if f64 else 1e-5), res[r]
    # replicated scalars: every rank stops at the same iteration
    assert res[r]['its'] == res[0]['its']
    assert res[r]['res'] == res[0]['res']
    # (the two conventions normalise the tolerance by different b-norms)
    assert abs(res[r]['its'] - res[r]['its_ref']) <= 5


def _gmsh_worker(rank, world, port, P, source, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from swirl_fem_amd.common import mesh_partitioner, mesh_reader
    from swirl_fem_amd.core.fespace import FiniteElementSpace
    from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
    from swirl_fem_amd.core.interpolation import Quadrature1D
    from swirl_fem_amd.core.mesh_refiner import refine_premesh
    from swirl_fem_amd.distributed import solver
    from swirl_fem_amd.linalg.cg import cg
    dev = torch.device('cuda', 0)
    path = os.path.join(os.path.dirname(__file__), 'golden', 'msh', 'cube.msh')
    nodes = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
    quad = Quadrature1D.create_from_nodes_1d(nodes)
    if source == 'gmsh':
      pm = mesh_reader.read(path, ndim=3)
      rp = refine_premesh(mesh_partitioner.partition(pm, world), nodes)
    else:
      # periodic in the partitioned direction: the two ranks meet twice, at
      # the cut and through the wrap-around (config 4's situation)
      from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
      pm = unit_cube_mesh(4, ndim=3, periodic_dims=(0,))
      rp = refine_premesh(pm.replace(partitions=np.repeat(
          np.arange(world), pm.num_elements // world).astype(np.int32)), nodes)
    mesh = rp.finalize('parts', rank=rank, device=dev)
    arrays = rp.finalize_all('parts')
    gids = arrays['global_node_ids'][rank]                  # -1 = padding
    real = torch.as_tensor(gids >= 0, device=dev)
    x = mesh.node_coords
    ax = 0 if source == 'gmsh' else 1          # Dirichlet planes
    dirichlet = ((x[:, ax] < 1e-9) | (x[:, ax] > 1 - 1e-9)) | ~real
    fes = FiniteElementSpace.create(mesh, quad)
    op = fes.helmholtz_operator(dirichlet)
    A = op.linear_operator(0.5, 1.0)
    # the same problem on the whole mesh, one rank
    gmesh = refine_premesh(pm, nodes).finalize(device=dev)
    gx = gmesh.node_coords
    gdir = (gx[:, ax] < 1e-9) | (gx[:, ax] > 1 - 1e-9)
    gop = FiniteElementSpace.create(gmesh, quad).helmholtz_operator(gdir)
    f = torch.sin(2 * np.pi * gx[:, 0]) * torch.cos(2 * gx[:, 1]) + gx[:, 2]
    # reference convention: unassembled operator, M = exchange (periodic)
    xg, _ = cg(lambda u: gop.apply(u, 0.5, 1.0), gop.apply(f * ~gdir, 1.0, 0.0),
               tol=1e-12, maxiter=3000, M=gmesh.exchange)
    # partitioned: unassembled local load vector of the same forcing
    ids = torch.as_tensor(np.where(gids >= 0, gids, 0), device=dev)
    f_loc = f[ids] * ~dirichlet
    free = fes.helmholtz_operator(None)
    b_loc = free.apply(f_loc, 1.0, 0.0) * ~dirichlet
    xl, info = solver.cg(A, b_loc, mesh.neighbor_plan, tol=1e-12, maxiter=3000)
    err = float(((xl - xg[ids]) * real).abs().max() / xg.abs().max())
    results[rank] = dict(err=err, its=info['num_iterations'],
                         elems=int((mesh.elements[:, 0] >= 0).sum()),
                         shared=mesh.neighbor_plan.num_shared)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world,P,source', [(2, 4, 'gmsh'), (3, 3, 'gmsh'),
                                            (2, 4, 'periodic_x')])
def test_gmsh_partitioned_solve_matches_single_rank(world, P, source):
  """Gmsh file -> native reader -> coordinate-bisection partitioner -> refiner
  -> reference-style (P, S) exchange table -> neighbour plan -> partitioned CG
  on the GPU; equals the one-rank solve of the same mesh (uneven partitions
  pad their element and node arrays with -1)."""
  port = _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_gmsh_worker, args=(world, port, P, source, results),
             nprocs=world, join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  assert sum(r['elems'] for r in res.values()) == 64
  for r in range(world):
    assert res[r]['shared'] > 0
    assert res[r]['err'] < 1e-9, res[r]
    assert res[r]['its'] == res[0]['its']


def _ns_worker(rank, world, port, order, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
    from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
    from swirl_fem_amd.core.mesh_refiner import refine_premesh
    from swirl_fem_amd.examples.navier_stokes_driver import (
        _histories, navier_stokes_step)
    from swirl_fem_amd.navier_stokes.navier_stokes import BCType, StokesSEM
    dev = torch.device('cuda', 0)
    pm = unit_cube_mesh(4, ndim=3, periodic_dims=(0,))
    parts = np.repeat(np.arange(world), pm.num_elements // world).astype(
        np.int32)
    bcs = {'boundary': (BCType.DIRICHLET, 0.0)}

    def run(sem):
      x = sem.velocity.mesh.node_coords
      wall = torch.sin(np.pi * x[:, 1]) * torch.sin(np.pi * x[:, 2])
      u0 = torch.stack([torch.sin(2 * np.pi * x[:, 0]) * wall,
                        torch.cos(2 * np.pi * x[:, 0]) * wall * x[:, 1],
                        0.5 * wall], dim=-1)
      p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype,
                       device=dev)
      us, ps, Cus = _histories(sem, u0, p0, 2)
      for _ in range(2):
        u, p, Cu, aux = navier_stokes_step(
            sem, us, ps, Cus, reynolds=50.0, dt=1e-2, time_order=2,
            tol=1e-11, atol=0.0)
        us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
      return u, p, aux

    sem_p = StokesSEM.create(pm.replace(partitions=parts), bcs, order,
                             device=dev, axis_name='parts', rank=rank)
    assert sem_p.is_partitioned and sem_p._divgrad() is not None
    u_p, p_p, aux = run(sem_p)
    sem_g = StokesSEM.create(pm, bcs, order, device=dev)
    u_g, p_g, _ = run(sem_g)
    gl = NodeType.GAUSS_LEGENDRE
    gll = NodeType.GAUSS_LOBATTO_LEGENDRE
    pp = pm.replace(partitions=parts)
    vid = refine_premesh(pp, Nodes1D.create(order + 1, gll)).finalize_all(
        'parts')['global_node_ids'][rank]
    pid = refine_premesh(pp, Nodes1D.create(order - 1, gl)).finalize_all(
        'parts')['global_node_ids'][rank]
    vt = torch.as_tensor(np.where(vid >= 0, vid, 0), device=dev)
    pt = torch.as_tensor(np.where(pid >= 0, pid, 0), device=dev)
    vreal = torch.as_tensor(vid >= 0, device=dev)[:, None]
    preal = torch.as_tensor(pid >= 0, device=dev)
    results[rank] = dict(
        eu=float(((u_p - u_g[vt]) * vreal).abs().max() / u_g.abs().max()),
        ep=float(((p_p - p_g[pt]) * preal).abs().max() / p_g.abs().max()),
        its=(aux['u_star_info']['num_iterations'],
             aux['dp_info']['num_iterations']))
  finally:
    dist.destroy_process_group()


def test_partitioned_navier_stokes_matches_single_rank():
  """Two ranks, mesh periodic across the cut: two BDF2/EXT1 Navier-Stokes steps
  (fused H, D, D^T, E, C, filter; all-reduced inner products) equal the
  one-rank run."""
  world, port = 2, _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_ns_worker, args=(world, port, 4, results), nprocs=world,
             join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  for r in range(world):
    assert res[r]['eu'] < 1e-8 and res[r]['ep'] < 1e-7, res[r]
    assert res[r]['its'] == res[0]['its']


def _ns_blocks_worker(rank, world, port, order, results):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from swirl_fem_amd.common.premesh_commons import box_mesh
    from swirl_fem_amd.distributed import blocks
    from swirl_fem_amd.examples.navier_stokes_driver import (
        _histories, navier_stokes_step)
    from swirl_fem_amd.navier_stokes.navier_stokes import BCType, StokesSEM
    dev = torch.device('cuda', 0)
    bcs = {'boundary': (BCType.DIRICHLET, 0.0)}

    def run(sem):
      x = sem.velocity.mesh.node_coords
      wall = torch.sin(np.pi * x[:, 1]) * torch.sin(np.pi * x[:, 2])
      u0 = torch.stack([torch.sin(2 * np.pi * x[:, 0]) * wall,
                        torch.cos(2 * np.pi * x[:, 0]) * wall * x[:, 1],
                        0.5 * wall], dim=-1)
      p0 = torch.zeros(sem.pressure.pspace.mesh.num_nodes, dtype=x.dtype,
                       device=dev)
      us, ps, Cus = _histories(sem, u0, p0, 2)
      for _ in range(2):
        u, p, Cu, aux = navier_stokes_step(
            sem, us, ps, Cus, reynolds=50.0, dt=1e-2, time_order=2,
            tol=1e-11, atol=0.0)
        us, ps, Cus = us[1:] + (u,), ps[1:] + (p,), Cus[1:] + (Cu,)
      return u, p

    # this rank's block only: 2 x 2 x 2 elements of the 4 x 2 x 2 box that is
    # periodic in x (the two blocks meet at the cut and through the wrap)
    part = blocks.build_block_partition(2, order + 1, (2, 1, 1), rank,
                                        device=dev, periodic_dims=(0,))
    sem_b = StokesSEM.create(part.premesh, bcs, order, device=dev,
                             neighbor_plan=part.plan)
    assert sem_b.is_partitioned
    u_b, p_b = run(sem_b)
    # the same flow on the whole box, one rank
    sem_g = StokesSEM.create(box_mesh((4, 2, 2), (0, 0, 0), (1, 1, 1),
                                      periodic_dims=(0,)), bcs, order,
                             device=dev)
    u_g, p_g = run(sem_g)

    def match(xl, xg):
      """Index of the global point with the coordinates of each local one
      (x taken modulo the period)."""
      key = lambda x: torch.round(
          torch.stack([x[:, 0] % 1.0, x[:, 1], x[:, 2]], 1) * 1e6).long()
      kl, kg = key(xl), key(xg)
      kl[:, 0] %= 10 ** 6
      kg[:, 0] %= 10 ** 6
      flat = lambda k: (k[:, 0] * 2000003 + k[:, 1]) * 2000003 + k[:, 2]
      fg, order_g = torch.sort(flat(kg))
      pos = torch.searchsorted(fg, flat(kl)).clamp(max=len(fg) - 1)
      assert bool((fg[pos] == flat(kl)).all())
      return order_g[pos]

    iv = match(sem_b.velocity.mesh.node_coords, sem_g.velocity.mesh.node_coords)
    ip = match(sem_b.pressure.pspace.mesh.node_coords,
               sem_g.pressure.pspace.mesh.node_coords)
    results[rank] = dict(
        eu=float((u_b - u_g[iv]).abs().max() / u_g.abs().max()),
        ep=float((p_b - p_g[ip]).abs().max() / p_g.abs().max()))
  finally:
    dist.destroy_process_group()


def test_navier_stokes_on_rank_local_blocks():
  """StokesSEM built from each rank's own block premesh + the block builder's
  neighbour plan (no global mesh anywhere) == the one-rank run on the whole
  periodic box."""
  world, port = 2, _free_port()
  with mp.Manager() as mgr:
    results = mgr.dict()
    mp.spawn(_ns_blocks_worker, args=(world, port, 4, results), nprocs=world,
             join=True)
    res = dict(results)
  assert sorted(res) == list(range(world))
  for r in range(world):
    assert res[r]['eu'] < 1e-8 and res[r]['ep'] < 1e-7, res[r]
