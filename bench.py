"""Headline benchmark: CG-iteration throughput of the 3D p=7 Laplacian.

    python bench.py --gpus N --steps K --warmup W

A "step" is one CG iteration (cg.py:75-86: operator apply + the fused vector
kernels) on the Dirichlet Laplacian of BASELINE config 2: a 64^3 structured hex
mesh on [0,1]^3 refined to GLL p=7 nodes (N = 449^3 = 90.5 M DOFs), fp64,
synthetic right-hand side, everything resident in HBM before the timed region.
For N > 1 every rank owns one 64^3 block of a (px,py,pz)-blocked mesh (weak
scaling; 8 GPUs = the 128^3, 2 M element mesh) and the shared-DOF exchange runs
over RCCL.

Prints ONE JSON line (rank 0): metric GDOF/s per CG iteration (whole job), the
roofline of the dominant kernel (fused gather-apply-scatter, HBM-bound) and the
CPU baseline (oracle = reference algorithm restated, timed on the host cores on
a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes_per_apply(E, n, N, sizeof=8, ngeo=6):
  """SURVEY 8(d) stored-factor model: idx 4 E n + u s N + geo 6 s E Q + out s N."""
  return 4 * E * n + sizeof * N + ngeo * sizeof * E * n + sizeof * N


# Single-GPU time of ONE CG iteration on the fixed meshes the N > 1 runs
# split (`--scaling strong`) or assemble (`--scaling weak`: N blocks of
# --elems^3), measured on one MI355X and committed under profiles/: the first
# real scaling record can then be read as the speed-up north_star asks for.
STRONG_REF = {   # (global elements per direction, p, dtype) -> (ms, source)
    (128, 7, 'f64'): (12.4, 'profiles/r03_bench_n128_single_gpu.json'),
}


TRAFFIC_FILE = 'profiles/traffic_r04.json'


def kernel_source_hash():
  """sha256 over the kernel sources (csrc/*.h, *.hip, include/sfem.h): the
  profile in TRAFFIC_FILE counts only for the build it was taken on."""
  import glob
  import hashlib
  h = hashlib.sha256()
  files = sorted(glob.glob(os.path.join(ROOT, 'swirl_fem_amd', 'csrc', '*.h')) +
                 glob.glob(os.path.join(ROOT, 'swirl_fem_amd', 'csrc', '*.hip')) +
                 [os.path.join(ROOT, 'include', 'sfem.h')])
  for f in files:
    h.update(os.path.basename(f).encode())
    with open(f, 'rb') as fh:
      h.update(fh.read())
  return h.hexdigest()[:16]


def measured_traffic(n, p, dtype, geometry, jitter):
  """(HBM bytes per launch of the dominant kernel, source hash of the build
  they were measured on) from the rocprofv3 PMC passes of this exact command
  (`scripts/collect_profiles3.py` writes TRAFFIC_FILE), or (None, None) --
  also when the kernel sources have changed since (`kernel_source_hash`).
  PMC counters cannot be read from inside the process, so the figure is a
  profile of an earlier run of the same command on the same sources."""
  key = 'n%d_p%d_%s_%s' % (n, p, dtype, 'stored' if geometry == 'stored'
                           else 'auto')
  if jitter:
    key += '_jitter'
  try:
    with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
      entry = json.load(f).get(key)
  except (OSError, ValueError):
    return None, None
  if entry is None or entry.get('src_hash') != kernel_source_hash():
    return None, None
  return entry['bytes'], entry.get('src_hash')


def block_grid(world):
  return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(
      world, (world, 1, 1))


def cpu_baseline(P, budget_s=12.0, ne=32):
  """Times the reference algorithm on this host's cores: dense Kronecker
  element matrices, 9+1 stored geometric arrays, un-fused CG
  (`oracle/cpu_reference.py`, the oracle's algorithm with the element-batch
  contractions as multi-threaded torch-CPU GEMMs), on a bounded sample of the
  same Dirichlet Laplacian: 32^3 elements ("32^3 if 64^3 does not
  fit"; the reference's array layout needs 25 GB at 64^3), as many CG
  iterations as fit `budget_s` seconds (at least 3)."""
  import torch
  from oracle import cpu_reference
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  rp = refine_premesh(unit_cube_mesh(ne, ndim=3),
                      Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE))
  mask = np.zeros(rp.num_nodes)
  mask[np.unique(rp.physical_groups['boundary'])] = 1.0
  # the GPU box shares its host: a rank gets about 16 cores
  saved = torch.get_num_threads()
  torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
  try:
    ref = cpu_reference.StiffnessCG(rp.node_coords, rp.elements, P, mask)
    rng = np.random.default_rng(0)
    b = torch.from_numpy((1.0 - mask) * rng.standard_normal(rp.num_nodes))
    _, iters, el = ref.cg_iterations(b, budget_s=budget_s)
    # "also report the sum-factorised CPU variant" (the same CG
    # with the element gradient contracted axis by axis)
    _, it_sf, el_sf = ref.cg_iterations(b, budget_s=budget_s / 2,
                                        sum_factorised=True)
    threads = torch.get_num_threads()
  finally:
    torch.set_num_threads(saved)
  return {
      'value': rp.num_nodes * iters / el / 1e9, 'unit': 'GDOF/s',
      'cores': threads, 'kind': 'port',
      'sum_factorised': {
          'value': rp.num_nodes * it_sf / el_sf / 1e9, 'unit': 'GDOF/s',
          'sample': f'same mesh and CG, element gradient by three 1D '
                    f'contractions (torch-CPU einsum, {threads} threads): '
                    f'{it_sf} iterations in {el_sf:.1f} s'},
      'sample': f'{ne}^3 hex elements p={P - 1} ({rp.num_nodes} DOFs), {iters} '
                f'CG iterations in {el:.1f} s of the reference algorithm '
                f'(dense-Kronecker element matrices as torch-CPU GEMMs on '
                f'{threads} threads, stored J^-1 / detJ, '
                f'un-fused CG; a restatement, not JAX)',
  }


def _free_port():
  import socket
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    return sock.getsockname()[1]


def launch_ranks(n):
  """Runs this script as `n` ranks (one process per GPU) and returns the exit
  code: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` with a
  local rendezvous, the command line passed through.  The parent has not
  initialised the GPU (it has not even imported torch)."""
  import subprocess
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.abspath(__file__)
         ] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
  return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
  """Everything of the N-rank bench path that needs no GPU: rendezvous (gloo),
  this rank's block and neighbour plan, one QQ^T exchange of the interface
  values and the scalar all-reduce, on CPU tensors."""
  import torch
  import torch.distributed as dist
  from swirl_fem_amd.distributed import blocks, comm
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
  grid_b = block_grid(world)
  periodic_dims = (0, 1, 2) if args.periodic and world > 1 else ()
  part = blocks.build_block_partition(args.n, args.p + 1, grid_b, rank,
                                      device='cpu',
                                      periodic_dims=periodic_dims)
  ones = torch.ones(part.mesh.num_nodes, dtype=torch.float64)
  holders = ones.clone()
  sent = 0
  if world > 1:
    idx = [torch.as_tensor(ix, dtype=torch.int64) for ix in part.plan.indices]
    recv = comm.exchange_buffers(part.plan, [ones[ix] for ix in idx])
    for rb, ix in zip(recv, idx):
      holders.index_add_(0, ix, rb)
      sent += ix.numel()
  # every DOF counted once: sum over ranks of sum_i 1 / holders_i = N_global
  total = torch.tensor([float((1.0 / holders).sum())], dtype=torch.float64)
  if world > 1:
    comm.all_reduce_sum_(total)
  ok = abs(float(total) - part.num_global_nodes) < 1e-6 * part.num_global_nodes
  if rank == 0:
    print(json.dumps({
        'metric': 'GDOF/s per CG iteration, 3D p=%d Laplacian' % args.p,
        'value': None, 'unit': 'GDOF/s', 'n_gpus': world, 'dry_run': True,
        'config': {'workload': 'launch rehearsal on CPU: %d^3 elements per '
                               'rank, p=%d' % (args.n, args.p),
                   'blocks': 'x'.join(map(str, grid_b)),
                   'backend': 'gloo', 'world_size_seen':
                       dist.get_world_size() if world > 1 else 1,
                   'dofs_global': part.num_global_nodes,
                   'interface_values_sent_rank0': sent,
                   'dof_count_via_exchange_ok': bool(ok)}}))
  if world > 1:
    dist.destroy_process_group()
  if not ok:
    raise SystemExit('dry run: the exchanged holder counts do not add up')
  return 0


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--n', '--elems', dest='n', type=int, default=64, help='elements per dim per GPU')
  ap.add_argument('--p', type=int, default=7, help='polynomial order')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-general', action='store_true',
                  help='skip timing the stored-factor kernel beside the default')
  ap.add_argument('--geometry', default='auto',
                  choices=['auto', 'multilinear', 'stored'],
                  help="'auto': affine / multilinear elements evaluate their "
                       "geometric factors in registers; 'stored': 6 factors "
                       "per point are read for every element")
  ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
  ap.add_argument('--mass-coeff', type=float, default=0.0,
                  help='Helmholtz operator mass_coeff * B + A (0 = Laplacian)')
  ap.add_argument('--tile', type=int, default=0,
                  help='visit elements in tile^3 blocks (0 = lexicographic)')
  ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                  help="'weak' (default): --elems^3 elements per GPU; 'strong': "
                       'a fixed --elems^3 mesh split over the GPUs (blocks of '
                       'elems/px x elems/py x elems/pz elements)')
  ap.add_argument('--periodic', action='store_true',
                  help='N>1: the triply periodic box of BASELINE config 4 '
                       '(N=8: 2x2x2 blocks; along a direction with one block '
                       'the images are summed on the rank itself); solves the '
                       'Helmholtz problem B + A instead of the singular A')
  ap.add_argument('--backend', default='nccl',
                  choices=['nccl', 'gloo', 'threads'],
                  help="'gloo' rehearses the N>1 path with all ranks on the "
                       'visible GPU(s) (interface buffers staged via host); '
                       "'threads' runs the N ranks as threads of ONE process "
                       'on one GPU (distributed/inprocess.py: the 2x2x2 layout '
                       'of --gpus 8 on a one-GPU box, whose process limit '
                       'rules out eight gloo ranks); neither is a measurement')
  ap.add_argument('--verify', action='store_true',
                  help='after the timed iterations, solve A x = A x* for a '
                       'manufactured x* through the same operator and solver '
                       'and report max |x - x*| / max |x*| over all ranks in '
                       'config.verify')
  ap.add_argument('--partitioned', default='consistent',
                  choices=['consistent', 'reference'],
                  help="N>1 CG formulation: consistent vectors with the "
                       "exchange inside A (default) or the reference's "
                       'unassembled A with M = exchange')
  ap.add_argument('--no-overlap', action='store_true',
                  help='N>1: do not overlap the interface exchange with the '
                       'interior elements')
  ap.add_argument('--jitter', type=float, default=0.0,
                  help='smooth mesh deformation amplitude (fraction of h)')
  ap.add_argument('--graph', action='store_true',
                  help='N=1: replay each CG iteration as one HIP graph launch')
  ap.add_argument('--repeats', type=int, default=4,
                  help='further timed batches of --steps iterations after the '
                       'one `value` is computed from; their ms per step go to '
                       'config.repeat_ms_per_step (box-to-box and run-to-run '
                       'spread of a 30 ms measurement)')
  ap.add_argument('--dry-run', action='store_true',
                  help='rehearse launch, rendezvous, block build and one '
                       'interface exchange on CPU tensors (gloo) and print a '
                       'line with value null; nothing is timed and no kernel '
                       'runs (the kernels have no CPU path)')
  args = ap.parse_args()

  if args.backend == 'threads' and args.gpus > 1:
    if args.dry_run:
      raise SystemExit('bench.py: --dry-run rehearses process ranks; '
                       '--backend threads has none')
    return run_threads(args)
  world_env = os.environ.get('WORLD_SIZE')
  if world_env is None and args.gpus > 1:
    # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh
    # child processes, BEFORE this process touches the GPU (it never does)
    raise SystemExit(launch_ranks(args.gpus))
  world = int(world_env or '1')
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit(f'bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; '
                     'the block grid and the reported rate follow the ranks '
                     'that actually run')
  if args.dry_run:
    return dry_run(args, world, rank)

  import torch
  import torch.distributed as dist

  if not torch.cuda.is_available():
    raise SystemExit('bench.py: no GPU visible; the kernels have no CPU path '
                     '(use --dry-run to rehearse the launch on CPU)')
  if args.backend == 'gloo':
    local_rank %= torch.cuda.device_count()
  if local_rank >= torch.cuda.device_count():
    print(f'bench.py: rank {rank} of {world} needs GPU {local_rank}, but this '
          f'process sees {torch.cuda.device_count()} device(s) '
          f'(HIP_VISIBLE_DEVICES={os.environ.get("HIP_VISIBLE_DEVICES")}); '
          'one rank per GPU over RCCL -- `--backend gloo` lets ranks share a '
          'device for a rehearsal', file=sys.stderr, flush=True)
    raise SystemExit(3)
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    import datetime
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    try:
      # a rendezvous that does not complete in two minutes is a broken launch:
      # fail with the rank and device instead of sitting out the driver's limit
      if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=device,
                                timeout=datetime.timedelta(seconds=120))
      else:
        dist.init_process_group('gloo', rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=120))
    except Exception as exc:    # pylint: disable=broad-except
      print(f'bench.py: rank {rank}/{world} on {device} '
            f'({args.backend}, MASTER_ADDR={os.environ.get("MASTER_ADDR")}, '
            f'MASTER_PORT={os.environ.get("MASTER_PORT")}) could not join the '
            f'process group: {exc!r}', file=sys.stderr, flush=True)
      raise SystemExit(3)
  rank_main(args, world, rank, device)
  if world > 1:
    dist.destroy_process_group()


def run_threads(args):
  """`--backend threads`: the N ranks as threads of this process on cuda:0."""
  import torch
  from swirl_fem_amd.distributed import inprocess
  if not torch.cuda.is_available():
    raise SystemExit('bench.py: no GPU visible; the kernels have no CPU path')
  torch.cuda.set_device(0)
  device = torch.device('cuda', 0)
  world = inprocess.ThreadWorld(args.gpus)
  world.run(lambda rank: rank_main(args, args.gpus, rank, device))
  return 0


def manufactured_solution(mesh, periodic):
  """x*: smooth, zero on the Dirichlet boundary of the unit cube (or periodic
  on it), a function of the coordinates -- so equal on every holder of a node."""
  import torch
  x = mesh.node_coords
  k = 2.0 * np.pi if periodic else np.pi
  xs = torch.sin(k * x[:, 0]) * torch.sin(k * x[:, 1]) * torch.sin(k * x[:, 2])
  if periodic:
    xs = xs + torch.cos(k * x[:, 0])
  return xs


def rank_main(args, world, rank, device):
  """One rank of the benchmark (a process, or a thread of `run_threads`)."""
  import torch
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  from swirl_fem_amd import switches
  from swirl_fem_amd.distributed import comm
  from swirl_fem_amd.linalg.cg import CGRunner

  P = args.p + 1
  grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
  t_setup = time.perf_counter()
  from swirl_fem_amd.distributed import blocks
  tdtype = torch.float64 if args.dtype == 'f64' else torch.float32
  sizeof = 8 if args.dtype == 'f64' else 4
  grid_b = block_grid(world)
  if args.scaling == 'strong':
    if any(args.n % g for g in grid_b):
      raise SystemExit(f'--elems {args.n} is not divisible by the block grid '
                       f'{grid_b}')
    block_n = tuple(args.n // g for g in grid_b)
  else:
    block_n = args.n
  periodic_dims = (0, 1, 2) if args.periodic and world > 1 else ()
  if len(periodic_dims) == 3 and args.mass_coeff == 0.0:
    args.mass_coeff = 1.0          # no Dirichlet boundary left: A is singular
  part = blocks.build_block_partition(block_n, P, grid_b, rank,
                                      device=device, jitter=args.jitter,
                                      dtype=tdtype, tile=args.tile,
                                      periodic_dims=periodic_dims)
  mesh = part.mesh
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
  op = fes.helmholtz_operator(mesh.physical_masks.get('boundary'),
                              geometry=args.geometry)
  setup_s = time.perf_counter() - t_setup

  N_local = mesh.num_nodes
  E, n = mesh.elements.shape
  N_global = part.num_global_nodes

  g = torch.Generator(device=device).manual_seed(1234 + rank)
  b = torch.randn(N_local, dtype=tdtype, device=device, generator=g)
  if 'boundary' in mesh.physical_masks:
    b = b * (~mesh.physical_masks['boundary']).to(b.dtype)
  # Partitioned CG follows the reference's solver convention
  # (navier_stokes.py:436-438): A returns the *unassembled* local
  # result, the preconditioner slot carries QQ^T (M = exchange), so plain local
  # dots + one all-reduce are the global inner products.
  out_buf = torch.empty_like(b)

  # the operator also hands CG its p.Ap (accumulated in the scatter stage)
  A = op.linear_operator(args.mass_coeff, 1.0)

  if world > 1 and args.partitioned == 'consistent':
    # consistent vectors: exchange inside A on the interface nodes only, fused
    # p.Ap, interface-corrected r.r (distributed/solver.py)
    from swirl_fem_amd.distributed import solver
    if not args.no_overlap:
      # partition-boundary elements first; the interface exchange then runs
      # on RCCL's stream while the interior elements compute
      A = solver.OverlappedHelmholtz(op, part.plan, args.mass_coeff, 1.0)
    run = solver.make_runner(A, b, part.plan, tol=0.0, atol=0.0,
                             maxiter=10 ** 9)
  elif world > 1:
    run = CGRunner(A, b, tol=0.0, atol=0.0, maxiter=10 ** 9, M=mesh.exchange,
                   reduce_fn=part.reduce_sum_)
  else:
    run = CGRunner(A, b, tol=0.0, atol=0.0, maxiter=10 ** 9)
    if args.graph:
      captured = run.capture()
      if rank == 0 and not captured:
        print('bench.py: graph capture failed, running eagerly', file=sys.stderr)

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      comm.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    run.step()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    run.step()
  barrier()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
    elapsed = float(comm.all_reduce_max_(tt).item())
  ms_per_step = 1e3 * elapsed / args.steps
  value = N_global / (elapsed / args.steps) / 1e9
  repeat_ms = []
  for _ in range(max(0, args.repeats)):
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      run.step()
    barrier()
    tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64,
                      device=device)
    if world > 1:
      tt = comm.all_reduce_max_(tt)
    repeat_ms.append(1e3 * float(tt.item()) / args.steps)

  # ---- roofline of the dominant kernel: HIP events around ONE apply as the
  # solver issues it -- the fused kernel and whatever it needs around it (the
  # clearing of the atomically accumulated node range, when there is one)
  u = run.p
  from swirl_fem_amd import _ops

  def time_apply(operator):
    for _ in range(3):
      operator.apply(u, args.mass_coeff, 1.0, out=out_buf)
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True))
             for _ in range(args.steps)]
    for e0, e1 in pairs:
      e0.record()
      operator.apply(u, args.mass_coeff, 1.0, out=out_buf)
      e1.record()
    torch.cuda.synchronize()
    return float(np.mean([x.elapsed_time(y) for x, y in pairs]))

  # what the solver issues: with layered assembly (`CGRunner.layered`) the
  # apply stores into an extended vector, nothing is cleared, no atomics
  layered = getattr(run, 'layered', None)
  if layered is not None:
    class _LayeredApply:       # same call shape as `HelmholtzOperator.apply`
      def __init__(self, operator):
        self.op, self.ext = operator, operator.new_extended()
      def apply(self, v, l0, l1, out=None):
        return self.op.apply_layered(v, self.ext, l0, l1)
    solver_apply = _LayeredApply(op)
  else:
    solver_apply = op
  kern_ms = time_apply(solver_apply)
  # back-to-back applies (no event between them)
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
      enable_timing=True)
  s0.record()
  for _ in range(args.steps):
    solver_apply.apply(u, args.mass_coeff, 1.0, out=out_buf)
  s1.record()
  torch.cuda.synchronize()
  apply_ms = s0.elapsed_time(s1) / args.steps
  if layered is not None:
    del solver_apply
  per_rank = None
  if world > 1:
    # per-rank figures: local apply (above) and the interface exchange alone
    from swirl_fem_amd.distributed import comm
    w = torch.zeros_like(b)
    for _ in range(3):
      comm.neighbor_exchange_(w, part.plan)
    barrier()
    s0.record()
    for _ in range(args.steps):
      comm.neighbor_exchange_(w, part.plan)
    s1.record()
    torch.cuda.synchronize()
    exchange_ms = s0.elapsed_time(s1) / args.steps
    mine = torch.tensor([apply_ms, exchange_ms, float(part.plan.num_shared)],
                        dtype=torch.float64, device=device)
    every = comm.all_gather(mine)
    per_rank = {'apply_ms': [float(t[0]) for t in every],
                'exchange_ms': [float(t[1]) for t in every],
                'interface_values': [int(t[2]) for t in every]}
    del w
  # the same mesh through the general-geometry path (6 stored factors per
  # point are READ): the kernel whose traffic the stored-factor model describes
  general = None
  if world == 1 and args.geometry == 'auto' and not args.no_general:
    op_g = fes.helmholtz_operator(mesh.physical_masks.get('boundary'),
                                  geometry='stored')
    general = time_apply(op_g)
    stored_bytes = op_g.bytes_per_apply(args.mass_coeff)
    stored_kernel = op_g.kernel_name(args.mass_coeff, 1.0)
    del op_g
  # device stream figure beside the vendor peak: y = a x + b y over
  # N-vectors, 3 passes
  xs, ys = torch.randn_like(b), torch.randn_like(b)
  for _ in range(3):
    _ops.axpby(1.0, xs, 0.5, ys)
  s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
      enable_timing=True)
  s0.record()
  for _ in range(20):
    _ops.axpby(1.0, xs, 0.5, ys)
  s1.record()
  torch.cuda.synchronize()
  stream_gbs = 3 * xs.numel() * sizeof / (s0.elapsed_time(s1) / 20 * 1e-3) / 1e9
  del xs, ys
  ngeo = 7 if args.mass_coeff else 6
  model_bytes = algorithmic_bytes_per_apply(E, n, N_local, sizeof=sizeof,
                                            ngeo=ngeo)
  kernel_bytes = op.bytes_per_apply(args.mass_coeff,
                                    layered=layered is not None)
  layer_reads = 0 if layered is None else sizeof * layered.read
  achieved = kernel_bytes / (kern_ms * 1e-3) / 1e9
  traffic, traffic_sha = measured_traffic(args.n, args.p, args.dtype,
                                          args.geometry, args.jitter)
  passes = run.vector_passes if hasattr(run, 'vector_passes') else 8
  # the same GLOBAL mesh on one GPU (committed measurement), so that an N > 1
  # line reads directly as north_star's strong-scaling speed-up
  strong_ref = None
  if world > 1 and not periodic_dims and not args.jitter:
    gdim = (args.n if args.scaling == 'strong'
            else (args.n * grid_b[0] if len(set(grid_b)) == 1 else None))
    ref = STRONG_REF.get((gdim, args.p, args.dtype))
    if ref is not None:
      strong_ref = {'global_elements': gdim ** 3, 'n_gpus': 1,
                    'ms_per_step': ref[0], 'source': ref[1],
                    'speedup_vs_strong_ref': ref[0] / ms_per_step}

  verify = None
  if args.verify:
    # A x = A x* through the operator and solver that were just timed
    xs = manufactured_solution(mesh, bool(periodic_dims)).to(tdtype)
    if 'boundary' in mesh.physical_masks:
      xs = xs * (~mesh.physical_masks['boundary']).to(tdtype)
    vtol = 1e-10 if args.dtype == 'f64' else 1e-5
    if world > 1 and args.partitioned == 'consistent':
      from swirl_fem_amd.distributed import solver
      xv, vinfo = solver.cg(A, A(xs).clone(), part.plan, tol=vtol,
                            maxiter=20000, assembled_rhs=True)
    elif world > 1:
      from swirl_fem_amd.linalg.cg import cg as cg_solve
      xv, vinfo = cg_solve(A, A(xs).clone(), tol=vtol, maxiter=20000,
                           M=mesh.exchange, reduce_fn=part.reduce_sum_)
    else:
      from swirl_fem_amd.linalg.cg import cg as cg_solve
      xv, vinfo = cg_solve(A, A(xs).clone(), tol=vtol, maxiter=20000)
    err = torch.stack([(xv - xs).abs().max(), xs.abs().max()]).double()
    if world > 1:
      err = comm.all_reduce_max_(err)
    verify = {'rel_err_vs_manufactured': float(err[0] / err[1]),
              'iterations': int(vinfo['num_iterations']),
              'status': str(vinfo.get('status')), 'tol': vtol}

  if rank == 0:
    res = {
        'metric': 'GDOF/s per CG iteration, 3D p=%d %s' % (
            args.p, 'Helmholtz' if args.mass_coeff else 'Laplacian'),
        'value': value, 'unit': 'GDOF/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {
            'workload': '3D %s CG iteration, %d^3 hex elements %s, '
                        'p=%d GLL collocated, %s, %s' % (
                            'Helmholtz' if args.mass_coeff else 'Laplacian',
                            args.n, 'per GPU' if args.scaling == 'weak'
                            else 'in total', args.p, args.dtype,
                            'triply periodic' if periodic_dims else 'Dirichlet'),
            'elements_per_gpu': E, 'dofs_global': N_global,
            'blocks': 'x'.join(map(str, block_grid(world))),
            'periodic_dims': list(periodic_dims),
            'partitioned_cg': (args.partitioned + (
                '' if args.no_overlap or args.partitioned != 'consistent'
                else ', exchange overlapped with interior elements'))
            if world > 1 else None,
            'backend': {'nccl': 'rccl', 'gloo': 'gloo (rehearsal)',
                        'threads': 'threads of one process on one GPU '
                                   '(rehearsal)'}[args.backend]
            if world > 1 else None,
            'world_size_seen': comm.get_world_size() if world > 1 else 1,
            'verify': verify,
            'per_rank': per_rank,
            'strong_ref': strong_ref,
            'apply_only_gdofs': N_local * world / (apply_ms * 1e-3) / 1e9,
            'apply_ms': apply_ms, 'setup_s': setup_s,
            'repeat_ms_per_step': repeat_ms,
            'median_ms_per_step': float(np.median([ms_per_step] + repeat_ms)),
            'assembly': (
                'atomic (shared nodes: global_atomic_add after clearing their '
                'range)' if layered is None else
                'layered: %d layers behind the nodal vector, %d of %d values '
                'per apply stored to a further layer, no atomics, nothing '
                'cleared; r -= alpha Ap adds the layers up' % (
                    len(layered.layers), layered.written - N_local, N_local)),
            'switches': switches.active(),
            'solver': {
                'lazy_x': (None if getattr(run, 'lazy', None) is None
                           else 'x takes its terms every %d-th iteration '
                                '(bitwise the same x; any %d consecutive '
                                'iterations hold exactly one such update per '
                                '%d)' % ((run.lazy[0].shape[0],) * 3)),
                'inner_products': ('stored partial sums added in a fixed '
                                   'order (bitwise reproducible solve)'
                                   if getattr(run, 'det', None) is not None
                                   else 'atomically accumulated partial sums'),
            },
            'geometry': ('%s: %d affine + %d multilinear elements (factors '
                         'evaluated in registers), %d with 6 stored factors '
                         'per point' % (args.geometry, op.num_affine,
                                        op.num_multilinear, op.num_curved)),
            'mesh_jitter': args.jitter,
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic,
            'traffic_profiled_at': traffic_sha,
            'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                              'of this command (%s; FETCH_SIZE doubled as '
                              'MI355X_MICROARCH.md prescribes); null when the '
                              'workload was not profiled or the kernel sources '
                              '(sha256 in traffic_profiled_at) have changed '
                              'since' % TRAFFIC_FILE,
            'kernel': op.kernel_name(args.mass_coeff, 1.0,
                                     layered=layered is not None),
            'kernel_ms': kern_ms,
            'kernel_ms_covers': 'one apply as the solver issues it: the fused '
                                'kernel plus the clearing of the atomically '
                                'accumulated node range it needs, if any '
                                '(layered assembly: none)',
            'bytes_per_launch': kernel_bytes,
            'bytes_model': ('layered assembly: out is s x (N + values stored '
                            'to further layers) instead of s N; ' if layered
                            is not None else '') +
                           'u s N + out s N + connectivity (432 B per element '
                           'from a facet table, + 4 B on a chain list; 4 n + '
                           '2 S per element on index rows) + geometry (64 B '
                           'per affine / box element, 24 reals per multilinear '
                           'one, %d reals per point where factors are stored): '
                           'HelmholtzOperator.bytes_per_apply' % ngeo,
            'measured_stream_peak': stream_gbs,
            'frac_of_measured_stream': achieved / stream_gbs,
            # NOT an HBM fraction: the SURVEY 8(d) stored-6-factor model
            # charges 6 factors per point to every element; on affine /
            # multilinear elements this kernel never reads them, so the ratio
            # may exceed 1.  It compares GDOF/s with the model's 85.9 GDOF/s.
            'model_gdofs_ratio': model_bytes / (kern_ms * 1e-3) / 1e9 /
                                 HBM_PEAK_GBS,
            'model_bytes_per_launch': model_bytes,
        },
    }
    cg_bytes = kernel_bytes + passes * sizeof * N_local + layer_reads
    res['roofline_cg_iteration'] = {
        'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
        'achieved': cg_bytes / (ms_per_step * 1e-3) / 1e9,
        'frac': cg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'bytes_per_iteration': cg_bytes,
        'vector_passes': passes,
        'model_gdofs_ratio': (model_bytes + 11 * sizeof * N_local) /
                             (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'note': 'bytes this build moves per iteration: the apply above + %d '
                'N-vector passes (p.Ap and r.r fused into kernels that stream '
                'the vectors anyway); model_gdofs_ratio is against SURVEY '
                "8(d)'s stored-factor apply + 11 passes" % passes}
    if general is not None:
      res['roofline_stored_factors'] = {
          'bound': 'hbm', 'achieved': stored_bytes / (general * 1e-3) / 1e9,
          'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
          'frac': stored_bytes / (general * 1e-3) / 1e9 / HBM_PEAK_GBS,
          'kernel_ms': general, 'bytes_per_launch': stored_bytes,
          'kernel': stored_kernel,
          'note': 'the same mesh with geometry=stored: every element reads '
                  'its %d factors per point, as curved elements do (the '
                  'SURVEY 8(d) byte model, with the connectivity this build '
                  'reads instead of 4 E n)' % ngeo}
    if world == 1 and not args.no_cpu_baseline:
      res['cpu_baseline'] = cpu_baseline(P, ne=min(32, args.n) if args.p <= 7 else min(12, args.n))
    else:
      res['cpu_baseline'] = None
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
  main()
