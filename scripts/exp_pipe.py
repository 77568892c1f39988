"""A/B of the pipelined persistent apply kernel (run on the GPU box).

Each configuration runs in its own process (the switches are read once):
  python scripts/exp_pipe.py            # driver: spawns the variants
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
  import numpy as np, torch
  from swirl_fem_amd.distributed import blocks
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  n = int(os.environ.get('N', '64')); P = 8
  dev = torch.device('cuda:0')
  jit = float(os.environ.get('JITTER', '0'))
  part = blocks.build_block_partition(n, P, (1, 1, 1), 0, device=dev, jitter=jit)
  mesh = part.mesh
  grid = Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE)
  fes = FiniteElementSpace.create(mesh, Quadrature1D.create_from_nodes_1d(grid))
  bm = mesh.physical_masks.get('boundary')
  g = torch.Generator(device=dev).manual_seed(5)
  u = torch.randn(mesh.num_nodes, dtype=torch.float64, device=dev, generator=g)
  out = torch.empty_like(u)
  res = {}
  for geo in ('auto', 'stored'):
    op = fes.helmholtz_operator(bm, geo)
    for l0 in (0.0, 0.7):
      for _ in range(3): op.apply(u, l0, 1.0, out=out)
      torch.cuda.synchronize()
      lo, hi = op.zero_range
      evs = []
      for _ in range(20):
        out[lo:hi].zero_()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(); op.apply(u, l0, 1.0, out=out, zero=False); s1.record()
        evs.append((s0, s1))
      torch.cuda.synchronize()
      ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
      chk = float(out.double().abs().sum()); chk2 = float((out.double() * u).sum())
      print(f"RESULT {os.environ.get('TAG')} geo={geo} jitter={jit} l0={l0} kernel_ms={ms:.4f} abs_sum={chk:.12e} dot={chk2:.12e}", flush=True)
    del op

if os.environ.get('CHILD'):
  child()
else:
  variants = [('nopipe', {'SFEM_PIPE': '0'}),
              ('pipe3', {'SFEM_PIPE': '1', 'SFEM_PIPE_MINW': '3'}),
              ('pipe2', {'SFEM_PIPE': '1', 'SFEM_PIPE_MINW': '2'}),
              ('pipe3x16', {'SFEM_PIPE': '1', 'SFEM_PIPE_MINW': '3', 'SFEM_PIPE_WAVES_PER_CU': '24'}),
              ('pipe2x16', {'SFEM_PIPE': '1', 'SFEM_PIPE_MINW': '2', 'SFEM_PIPE_WAVES_PER_CU': '16'})]
  for jit in ('0', '0.2'):
    for tag, env in variants:
      e = dict(os.environ, CHILD='1', TAG=tag, JITTER=jit, **env)
      r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e,
                         capture_output=True, text=True, timeout=600)
      out = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
      print('\n'.join(out) if out else (r.stdout[-2000:] + r.stderr[-3000:]), flush=True)
