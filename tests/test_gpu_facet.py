"""GPU parity of the compact-connectivity (facet table) Helmholtz kernels
(`csrc/sfem_helmholtz_facet.h`): table builder vs the index rows it replaces,
operator vs the CPU oracle and vs the index-row kernels, 3D, P = 6..12.

fp64 tolerance 1e-10 relative, fp32 1e-5 (BASELINE.json north_star) on
float32-representable inputs (tests/fp32util.py; observed errors:
profiles/r04_fp32_errors.md).
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import sfem_oracle as O
from swirl_fem_amd import _lib, _ops
from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
from swirl_fem_amd.core.fespace import FiniteElementSpace
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
from swirl_fem_amd.core import operators
from swirl_fem_amd.core.mesh_refiner import refine_premesh
from tests.fp32util import F32Rng, f32_mesh, tolerance

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GLL = NodeType.GAUSS_LOBATTO_LEGENDRE
TOL = {torch.float64: 1e-10, torch.float32: 1e-5}


def dev(x, dtype=None):
  t = torch.as_tensor(np.ascontiguousarray(x), device=DEV)
  return t if dtype is None else t.to(dtype)


def relerr(a, b):
  a = a.detach().cpu().numpy().astype(np.float64)
  return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def random_orientations(pm, rng):
  """Random element order and one of the 48 vertex orderings per element."""
  orients = [(perm, axes) for perm in itertools.permutations(range(3))
             for r in range(4) for axes in itertools.combinations(range(3), r)]
  el = []
  for e in pm.elements[rng.permutation(pm.num_elements)]:
    perm, axes = orients[rng.integers(len(orients))]
    el.append(np.flip(e.reshape(2, 2, 2).transpose(perm), axes).reshape(-1))
  return pm.replace(elements=np.array(el, dtype=np.int32))


def make_mesh(n, P, mode, rng, rotate=False):
  pm = unit_cube_mesh(n, ndim=3)
  x = pm.node_coords.copy()
  if mode == 'sheared':
    x = x @ (np.eye(3) + 0.3 * rng.uniform(-1, 1, (3, 3))).T + 0.1
  elif mode == 'jittered':
    x = x + 0.1 / n * rng.uniform(-1, 1, x.shape)
  elif mode == 'stretched':       # boxes of different sizes: still diagonal
    x = x ** np.array([1.0, 1.5, 2.0])
  pm = pm.replace(node_coords=x)
  if rotate:
    pm = random_orientations(pm, rng)
  return refine_premesh(pm, Nodes1D.create(P, GLL))


def reference(ofes, u, l0, l1, mask):
  ul = ofes.gather(u)
  loc = 0.0
  if l0:
    loc = loc + l0 * ofes.mass_local(ul)
  if l1:
    loc = loc + l1 * ofes.stiffness_local(ul)
  out = ofes.scatter(loc)
  if mask is not None:
    keep = 1.0 - mask.astype(np.float64)
    out = out * (keep if out.ndim == 1 else keep[:, None])
  return out


def table_ids(tab, P):
  """The (E, P^3) index rows a table stands for, and its flag bits."""
  tab = tab.cpu().numpy().astype(np.int64)
  cls = lambda a: 0 if a == 0 else (2 if a == P - 1 else 1)
  E = tab.shape[0]
  ids = np.zeros((E, P, P, P), dtype=np.int64)
  flags = np.zeros((E, P, P, P), dtype=np.int64)
  for a, i, j in itertools.product(range(P), repeat=3):
    ent = tab[:, cls(a) * 9 + cls(i) * 3 + cls(j)]
    code = ent[:, 0] & 0xFFFFFFFF
    ids[:, a, i, j] = ((code & 0x3FFFFFFF) + ent[:, 1] * (a - 1) +
                       ent[:, 2] * (i - 1) + ent[:, 3] * (j - 1))
    flags[:, a, i, j] = code >> 30
  return ids.reshape(E, -1), flags.reshape(E, -1)


@pytest.mark.parametrize('P', [3, 4, 6, 7, 8, 9, 12])
def test_facet_table_reproduces_the_index_rows(P):
  rng = np.random.default_rng(P)
  n = 2 if P > 8 else 3
  rp = make_mesh(n, P, 'jittered', rng, rotate=True)
  mesh = rp.finalize(device=DEV)
  mult = mesh.assembly_plan().multiplicity
  mask = mesh.physical_masks['boundary'].to(torch.uint8).contiguous()
  tab, ok = _ops.facet_table(mesh.elements, mask, mult, P)
  assert bool(ok.all())
  ids, flags = table_ids(tab, P)
  assert np.array_equal(ids, rp.elements)
  enc = _ops.encode_elements(mesh.elements, mask, mult).cpu().numpy()
  assert np.array_equal(flags, (enc.astype(np.int64) & 0xFFFFFFFF) >> 30)
  # elements that are NOT 27 affine maps, or whose flags cut through a facet,
  # must be refused (and only those)
  el = rp.elements.copy()
  el[1, [5, 6]] = el[1, [6, 5]]   # two nodes of one facet swapped (P = 3:
                                  # of two one-node facets, still 27 maps)
  el[2, 7] = -1                                 # padding slot
  m2 = mask.clone()
  inner = int(rp.elements[0].reshape(P, P, P)[0, 1, 1])   # a face-interior node
  face = rp.elements[0].reshape(P, P, P)[0, 1:-1, 1:-1].reshape(-1)
  only_one = P > 3 and not bool(mask[inner])
  if only_one:
    m2[inner] = 1
  tab2, ok2 = _ops.facet_table(dev(el), m2, mult, P)
  ok2 = ok2.cpu().numpy()
  assert ok2[1] == (P == 3) and not ok2[2]
  if only_one:      # every element that holds the re-flagged face is refused
    holds = np.isin(rp.elements, face).any(axis=1)
    assert not ok2[holds].any()
    rest = ~holds
    rest[[1, 2]] = False
    assert ok2[rest].all()


@pytest.mark.parametrize('P,chain', [
    (6, '16'), (6, '3'), (6, 'off'), (7, '16'), (7, '3'), (7, 'off'),
    (8, '16'), (8, '3'), (8, 'off'),
    # several waves per element
    (9, '16'), (10, 'off'), (11, '3'), (12, '16'), (12, 'off')])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_facet_helmholtz_matches_oracle(P, dtype, chain, monkeypatch):
  """`chain`: scalar fields walk chains of elements (segments of <= 16 / 3
  elements, the shared face carried in registers) or one element per wave
  (P <= 8) / workgroup (P >= 9)."""
  tol = tolerance(dtype, P)
  if chain == 'off':
    monkeypatch.setenv('SFEM_CHAIN', '0')
    monkeypatch.setenv('SFEM_CHAIN_LEN', '8')    # built, not used
  else:
    monkeypatch.setenv('SFEM_CHAIN_LEN', chain)
  modes = (('structured', False), ('stretched', True), ('sheared', True),
           ('jittered', True))
  if P >= 9:          # the oracle's dense element matrices grow as P^6
    modes = (('structured', False), ('sheared', True), ('jittered', True))
  for mode, rotate in modes:
    rng = F32Rng(100 * P + len(mode))
    rp = f32_mesh(make_mesh(3 if chain == '3' and P == 6 else 2, P, mode, rng,
                            rotate), dtype)
    mesh = rp.finalize(device=DEV, dtype=dtype)
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
    ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
    bmask = mesh.physical_masks['boundary']
    ops = {g: fes.helmholtz_operator(bmask, g)
           for g in ('auto', 'multilinear', 'stored')}
    monkeypatch.setenv('SFEM_FACET', '0')
    from swirl_fem_amd.core import operators
    plain = operators.HelmholtzOperator.create(fes, bmask, 'auto')
    monkeypatch.delenv('SFEM_FACET')
    assert plain.facet_parts is None
    for g, op in ops.items():
      assert op.facet_parts is not None, (mode, g)
      # (multilinear elements of P >= 9 keep their index rows)
      rows_ok = P >= 9 and (g == 'multilinear' or mode == 'jittered') and \
          g != 'stored'
      assert all(('facet_table' in p) != rows_ok
                 for p in op.facet_parts), (mode, g)
      chained = [p for p in op.facet_parts if 'chains' in p]
      assert len(chained) == (len(op.facet_parts) if P <= 8 else 0), (mode, g)
      if rows_ok:
        assert 'helmholtz_kernel' in op.kernel_name()
      elif chain == 'off' or not chained:
        assert 'helmholtz_facet_kernel' in op.kernel_name()
      else:
        assert 'helmholtz_chain_kernel' in op.kernel_name()
        if not rotate:      # structured n^3 mesh: n^2 chains of n
          off = op.facet_parts[0]['chains'][0]
          n = round(mesh.num_elements ** (1 / 3))
          assert off.tolist() == list(range(0, n ** 3 + 1, n))
    kinds = {p['geo_mode'] for p in ops['auto'].facet_parts}
    if dtype == torch.float64:
      want = {'structured': {_lib.GEO_BOX}, 'stretched': {_lib.GEO_BOX},
              'sheared': {_lib.GEO_AFFINE},
              'jittered': {_lib.GEO_MULTILINEAR}}[mode]
      assert kinds == want, (mode, kinds)
    mk = bmask.cpu().numpy()
    for nc in (1, 3):
      u = rng.standard_normal((mesh.num_nodes, nc))
      uu = u[:, 0] if nc == 1 else u
      ud = dev(uu, dtype) if nc == 1 else dev(u.T.copy(), dtype).t()
      for l0, l1 in ((0.0, 1.0), (0.6, 1.4), (1.0, 0.0)):
        ref = reference(ofes, uu, l0, l1, mk)
        for g, op in ops.items():
          got = op.apply(ud, l0, l1)
          assert relerr(got, ref) < tol, (mode, g, nc, l0, l1)
        assert relerr(plain.apply(ud, l0, l1), ref) < tol
      # fused u . A u
      ref = reference(ofes, uu, 0.0, 1.0, mk)
      want, scale = float((uu * ref).sum()), float(np.abs(uu * ref).sum())
      for g, op in ops.items():
        parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                            device=DEV)
        got = op.apply(ud, 0.0, 1.0, dot_out=parts)
        assert relerr(got, ref) < tol
        assert abs(float(parts.sum()) - want) <= 10 * tol * scale, (mode, g)
    # interleaved (N, nc) fields keep the index-row kernels
    u = rng.standard_normal((mesh.num_nodes, 2))
    got = ops['auto'].apply(dev(u, dtype), 0.3, 1.0)
    assert relerr(got, reference(ofes, u, 0.3, 1.0, mk)) < tol


def test_facet_and_index_row_elements_in_one_operator():
  """Elements the builder refuses (here: a Dirichlet mask that holds single
  nodes of a face, and nothing else of it) fall back to the index rows; the
  operator applies both groups and matches the oracle."""
  P = 8
  rng = np.random.default_rng(5)
  rp = make_mesh(3, P, 'jittered', rng, rotate=True)
  mesh = rp.finalize(device=DEV)
  fes = FiniteElementSpace.create(
      mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
  ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
  mask = mesh.physical_masks['boundary'].cpu().numpy().copy()
  pick = rng.choice(np.nonzero(~mask)[0], 5, replace=False)
  mask[pick] = True
  op = fes.helmholtz_operator(dev(mask))
  n_facet = sum(p['elem_list'].numel() if 'elem_list' in p else
                mesh.num_elements for p in op.facet_parts
                if 'facet_table' in p)
  assert 0 < n_facet < mesh.num_elements
  assert any('facet_table' not in p for p in op.facet_parts)
  u = rng.standard_normal(mesh.num_nodes)
  for l0, l1 in ((0.0, 1.0), (0.5, 2.0)):
    ref = reference(ofes, u, l0, l1, mask)
    assert relerr(op.apply(dev(u), l0, l1), ref) < 1e-10
  # split(): both halves keep their share of either group
  half = torch.arange(mesh.num_elements, device=DEV) % 2 == 0
  a, b = op.split(half)
  out = a.apply(dev(u), 0.5, 2.0)
  out = b.apply(dev(u), 0.5, 2.0, out=out, zero=False)
  # owned nodes of `a` are plain stores in both halves: compare shared-safe sum
  ra = reference(O.FESpace(rp.node_coords, rp.elements[::2], (P, 'gll'),
                           (P, 'gll')), u, 0.5, 2.0, mask)
  rb = reference(O.FESpace(rp.node_coords, rp.elements[1::2], (P, 'gll'),
                           (P, 'gll')), u, 0.5, 2.0, mask)
  assert relerr(out, ra + rb) < 1e-10


@pytest.mark.parametrize('P', [8, 12])
def test_facet_kernels_with_64_bit_addressing(P, monkeypatch):
  """Fields of 4 GiB and more (the 128^3 mesh on one GPU: 5.8 GB per vector)
  take instantiations that form 64-bit addresses; forced here on a small
  mesh (`SFEM_FACET_OFF64=1`), chains and single elements, scalar and
  component-major."""
  rng = F32Rng(77)
  monkeypatch.setenv('SFEM_FACET_OFF64', '1')
  for mode, dtype in (('structured', torch.float64), ('sheared', torch.float64),
                      ('jittered', torch.float32)):
    rp = f32_mesh(make_mesh(2, P, mode, rng, rotate=mode != 'structured'),
                  dtype)
    mesh = rp.finalize(device=DEV, dtype=dtype)
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
    ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
    bmask = mesh.physical_masks['boundary']
    mk = bmask.cpu().numpy()
    for g in ('auto', 'stored'):
      op = fes.helmholtz_operator(bmask, g)
      for nc in (1, 2):
        u = rng.standard_normal((mesh.num_nodes, nc))
        uu = u[:, 0] if nc == 1 else u
        ud = dev(uu, dtype) if nc == 1 else dev(u.T.copy(), dtype).t()
        got = op.apply(ud, 0.4, 1.2)
        assert relerr(got, reference(ofes, uu, 0.4, 1.2, mk)) < tolerance(
            dtype, P)


def test_chain_segments_follow_the_mesh_size(monkeypatch):
  """Small meshes get short (or no) chains, so that a launch still fills the
  chip: 8 elements per workgroup only from 131 072 elements on."""
  from swirl_fem_amd.core import operators
  monkeypatch.delenv('SFEM_CHAIN_LEN', raising=False)
  assert operators.chain_segment_length(16 ** 3) == 1
  assert operators.chain_segment_length(32 ** 3) == 2
  assert operators.chain_segment_length(64 ** 3) == 8
  monkeypatch.setenv('SFEM_CHAIN_LEN', '5')
  assert operators.chain_segment_length(16 ** 3) == 5


@pytest.mark.parametrize('order,periodic,dtype', [
    (7, (), torch.float64), (7, (0, 1, 2), torch.float64),
    (5, (1,), torch.float64), (6, (), torch.float64),
    (7, (0, 1, 2), torch.float32), (5, (), torch.float32)])
def test_stokes_box_kernels_match_index_rows(order, periodic, dtype,
                                             monkeypatch):
  """div / grad_t on an anisotropic Cartesian box (SFEM_GEO_BOX chain kernels:
  one derivative per component) against the index-row kernels and the affine
  chain kernels, with and without the scale factors of E."""
  from swirl_fem_amd.common.premesh_commons import box_mesh
  from swirl_fem_amd.core import layout
  from swirl_fem_amd.navier_stokes.navier_stokes import StokesSEM
  pm = box_mesh((5, 4, 3), (0.0, -1.0, 0.5), (2.0, 0.2, 3.5),
                periodic_dims=periodic)
  bcs = {} if len(periodic) == 3 else {'boundary': (1, 0.0)}
  monkeypatch.setenv('SFEM_CHAIN_LEN', '3')
  # (by default the divergence walks chains on fp64 box elements only)
  monkeypatch.setenv('SFEM_STOKES_FACET_DIV', 'all')

  def build(**env):
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    sem = StokesSEM.create(pm, bcs, order=order, device=DEV, dtype=dtype)
    op = sem._divgrad()
    for k in env:
      monkeypatch.delenv(k)
    return sem, op

  sem, op = build()
  modes = [q['geo_mode'] for q in op.facet_parts]
  assert modes == [operators._GEO_BOX], modes
  _, op_aff = build(SFEM_BOX='0')
  assert [q['geo_mode'] for q in op_aff.facet_parts] == [operators._GEO_AFFINE]
  _, op_rows = build(SFEM_STOKES_FACET='0')
  assert op_rows.facet_parts is None
  nv, npr = sem.velocity.mesh.num_nodes, op.num_pressure_nodes
  g = torch.Generator(device='cpu').manual_seed(order)
  p = torch.randn(npr, dtype=dtype, generator=g).to(DEV)
  u = layout.empty_component_major((nv, 3), dtype, torch.device(DEV))
  u.copy_(torch.randn(nv, 3, dtype=dtype, generator=g).to(DEV))
  s1 = (torch.rand(nv, dtype=dtype, generator=g) + 0.5).to(DEV)
  s3 = layout.empty_component_major((nv, 3), dtype, torch.device(DEV))
  s3.copy_((torch.rand(nv, 3, dtype=dtype, generator=g) + 0.5).to(DEV))
  tol = 1e-13 if dtype == torch.float64 else 1e-5
  for scale in (None, s1, s3):
    want = op_rows.grad_t(p, component_major=True, scale=scale)
    for other in (op, op_aff):
      got = other.grad_t(p, component_major=True, scale=scale)
      err = (got - want).abs().max() / want.abs().max()
      assert err < tol, ('grad_t', scale is not None, float(err))
    want = op_rows.div(u, scale=scale)
    for other in (op, op_aff):
      got = other.div(u, scale=scale)
      err = (got - want).abs().max() / want.abs().max()
      assert err < tol, ('div', scale is not None, float(err))
  # the fused p . (D u) of the pressure CG
  dots = torch.zeros(1024, dtype=torch.float64, device=DEV)
  got = op.div(u, dot_with=p, dot_out=dots)
  assert abs(float(dots.sum()) - float(p.double() @ got.double())) < (
      1e-10 if dtype == torch.float64 else 1e-5) * float(p.norm() * got.norm())


HEADLINE_KERNEL = ('sfem::helmholtz_chain_kernel<double, 8, '
                   'sfem::BoxElem<double, 8, false>, ')


def _bench_record_kernel():
  """`roofline.kernel` of the newest driver record in the tree, if any."""
  import glob
  import json
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for path in sorted(glob.glob(os.path.join(root, 'BENCH_r*.json')))[::-1]:
    try:
      rec = json.load(open(path))
    except (OSError, ValueError):
      continue
    kern = (rec.get('parsed') or rec).get('roofline', {}).get('kernel')
    if kern:
      return kern
  return None


@pytest.mark.parametrize('chain_len', [8, 4])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_headline_chain_instantiation_matches_oracle(chain_len, dtype,
                                                     monkeypatch):
  """The instantiation bench.py times -- `helmholtz_chain_kernel<double, 8,
  BoxElem<double, 8, false>>` walking segments of 8 -- against the oracle at
  the chain length it runs: an 8 x 2 x 2 box of P = 8 elements gives four
  chains of 8 along the local axis 0, so every element class of a segment
  (first: carry out only; middle: carried in AND out, both flag edits of
  sfem_helmholtz_facet.h:936-938 at once; last: carry in only) occurs; with
  segments of 4 the face between two segments takes the atomics of both.
  Structured / stretched boxes run the box kernel, the sheared mesh the
  affine chain kernel.  Dirichlet mask, pure stiffness and Helmholtz, fused
  u . A u, scalar and component-major fields (reference semantics:
  core/gather_scatter.py:121-133, core/fespace.py:405-471)."""
  from swirl_fem_amd.common.premesh_commons import box_mesh
  P = 8
  tol = TOL[dtype]
  monkeypatch.setenv('SFEM_CHAIN_LEN', str(chain_len))
  real = 'double' if dtype == torch.float64 else 'float'
  for mode in ('structured', 'stretched', 'sheared'):
    rng = F32Rng(8 * chain_len + len(mode))
    pm = box_mesh((8, 2, 2), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    x = pm.node_coords.copy()
    if mode == 'stretched':
      x = x ** np.array([1.0, 1.5, 2.0])
    elif mode == 'sheared':
      x = x @ (np.eye(3) + 0.3 * rng.uniform(-1, 1, (3, 3))).T + 0.1
    rp = f32_mesh(refine_premesh(pm.replace(node_coords=x),
                                 Nodes1D.create(P, GLL)), dtype)
    mesh = rp.finalize(device=DEV, dtype=dtype)
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
    ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
    bmask = mesh.physical_masks['boundary']
    mk = bmask.cpu().numpy()
    o32 = None
    for mask_t, mask_np in ((bmask, mk), (None, None)):
      op = fes.helmholtz_operator(mask_t)
      assert len(op.facet_parts) == 1
      part = op.facet_parts[0]
      off = part['chains'][0].tolist()
      assert off == list(range(0, 33, chain_len)), off
      elem = ('sfem::FacetElem<%s, 8, 1, false>' % real if mode == 'sheared'
              else 'sfem::BoxElem<%s, 8, false>' % real)
      want_name = 'sfem::helmholtz_chain_kernel<%s, 8, %s, ' % (real, elem)
      assert op.kernel_name(0.0, 1.0) == want_name
      if mode != 'sheared' and dtype == torch.float64:
        assert op.kernel_name(0.0, 1.0) == HEADLINE_KERNEL
        rec = _bench_record_kernel()
        assert rec is None or rec == HEADLINE_KERNEL, rec
      for nc in (1, 3):
        u = rng.standard_normal((mesh.num_nodes, nc))
        uu = u[:, 0] if nc == 1 else u
        ud = dev(uu, dtype) if nc == 1 else dev(u.T.copy(), dtype).t()
        for l0, l1 in ((0.0, 1.0), (0.6, 1.4)):
          ref = reference(ofes, uu, l0, l1, mask_np)
          got = op.apply(ud, l0, l1)
          bar = tol
          if dtype == torch.float32 and mode == 'sheared':
            # elements of aspect ratio 4, sheared: the REFERENCE ALGORITHM in
            # float32 is 1.2e-5 .. 2.7e-5 from the fp64 oracle on this mesh
            # (profiles/r04_fp32_errors.md, `sheared 8x2x2`: the kernel's
            # error is 0.6 .. 1.05 of it); the bar is 1e-5 or 1.5 x that
            # reference error, whichever is larger
            if o32 is None:
              o32 = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'),
                              (P, 'gll'), dtype=np.float32)
            bar = max(tol, 1.5 * relerr(torch.as_tensor(reference(
                o32, uu.astype(np.float32), np.float32(l0), np.float32(l1),
                mask_np)), ref))
          assert relerr(got, ref) < bar, (mode, nc, l0, l1, mask_t is None)
          if (l0, l1) == (0.0, 1.0):
            bar_stiffness = bar
        ref = reference(ofes, uu, 0.0, 1.0, mask_np)
        want = float((uu * ref).sum())
        scale = float(np.abs(uu * ref).sum())
        parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                            device=DEV)
        got = op.apply(ud, 0.0, 1.0, dot_out=parts)
        assert relerr(got, ref) < bar_stiffness, (mode, nc, 'dot')
        assert abs(float(parts.sum()) - want) <= 10 * tol * scale, (mode, nc)


def test_chain_launch_without_an_instantiation_is_an_error(monkeypatch):
  """P >= 9 elements with stored / multilinear geometry have no chain kernel
  (`FacetElem::CHAINS`): a launch that asks for one must fail with
  SFEM_EUNSUPPORTED instead of returning SFEM_OK with nothing written, and
  `SFEM_CHAIN_HI=1` must not attach chains to such parts."""
  P = 9
  rng = np.random.default_rng(3)
  rp = make_mesh(2, P, 'jittered', rng, rotate=False)
  mesh = rp.finalize(device=DEV)
  fes = FiniteElementSpace.create(
      mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
  monkeypatch.setenv('SFEM_CHAIN_HI', '1')
  monkeypatch.setenv('SFEM_CHAIN_LEN', '2')
  op = fes.helmholtz_operator(None, 'stored')
  assert all('chains' not in q for q in op.facet_parts)
  u = dev(rng.standard_normal(mesh.num_nodes))
  good = op.apply(u)
  ids = torch.arange(mesh.num_elements, device=DEV)
  bad = dict(op.facet_parts[0],
             chains=operators.facet_chains(mesh.elements, ids, P, 2),
             chain_len=2)
  op.facet_parts = [bad]
  with pytest.raises(RuntimeError, match='no chain kernel'):
    op.apply(u)
  # box / affine elements of P >= 9 do have one (opt-in)
  rp = make_mesh(2, P, 'structured', rng)
  mesh = rp.finalize(device=DEV)
  fes = FiniteElementSpace.create(
      mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
  opc = fes.helmholtz_operator(None)
  assert all('chains' in q for q in opc.facet_parts)
  monkeypatch.delenv('SFEM_CHAIN_HI')
  # (`fes.helmholtz_operator` keeps one operator per mask and geometry)
  ref = operators.HelmholtzOperator.create(fes, None)
  assert all('chains' not in q for q in ref.facet_parts)
  u = dev(rng.standard_normal(mesh.num_nodes))
  assert relerr(opc.apply(u), ref.apply(u).cpu().numpy()) < 1e-12
  del good


@pytest.mark.parametrize('P,chain', [(6, '3'), (7, '16'), (8, '3'), (8, '16'),
                                     (8, 'off'), (9, 'off'), (12, 'off')])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_layered_assembly_matches_oracle(P, chain, dtype, monkeypatch):
  """Layered assembly (`build_layer_plan`, `apply_layered`, `fold_layers`):
  every (element, facet) writes its contribution to a layer of its own with a
  plain store, the layers are added up afterwards -- the direct-stiffness sum
  of core/gather_scatter.py:130-133 without atomics.  Against the oracle and
  against the atomic assembly of the same operator, on every geometry kind,
  with chains (carried faces are written by the receiver only) and without,
  rotated elements, Dirichlet rows, fused u . A u; and bitwise reproducible."""
  tol = tolerance(dtype, P)
  if chain == 'off':
    monkeypatch.setenv('SFEM_CHAIN', '0')
  else:
    monkeypatch.setenv('SFEM_CHAIN_LEN', chain)
  modes = (('structured', False), ('stretched', True), ('sheared', True),
           ('jittered', True))
  if P >= 9:
    modes = (('structured', False), ('sheared', True))
  for mode, rotate in modes:
    rng = F32Rng(7 * P + len(mode))
    n = 3 if P <= 7 else 2
    rp = f32_mesh(make_mesh(n, P, mode, rng, rotate), dtype)
    mesh = rp.finalize(device=DEV, dtype=dtype)
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
    ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
    bmask = mesh.physical_masks['boundary']
    N = mesh.num_nodes
    for geometry in ('auto', 'stored'):
      for mask_t in (bmask, None):
        op = operators.HelmholtzOperator.create(fes, mask_t, geometry)
        plan = op.layer_plan()
        assert plan is not None, (mode, geometry)
        assert plan.extent >= N and len(plan.layers) >= 1
        # interior vertices of an n^3 mesh have 8 holders (4 writers when
        # the chains carry one direction)
        assert len(plan.layers) + 1 >= (4 if chain != 'off' and P <= 8 else 8) \
            or n < 3
        lens = [l for l, _ in plan.layers]
        assert lens == sorted(lens, reverse=True) and lens[0] <= N + 3
        if P <= 8 and chain != 'off':
          assert 'helmholtz_chain_kernel' in op.kernel_name(layered=True)
        assert op.kernel_name(layered=True).endswith('*, true>')
        mk = None if mask_t is None else mask_t.cpu().numpy()
        u = rng.standard_normal(N)
        ud = dev(u, dtype)
        for l0, l1 in ((0.0, 1.0), (0.6, 1.4), (1.0, 0.0)):
          ref = reference(ofes, u, l0, l1, mk)
          ext = op.new_extended()
          parts = torch.zeros(_lib.SFEM_DOT_SLOTS, dtype=torch.float64,
                              device=DEV)
          op.apply_layered(ud, ext, l0, l1, dot_out=parts)
          raw = ext.clone()
          got = _ops.fold_layers(ext, N, plan.layers)
          assert relerr(got, ref) < tol, (mode, geometry, l0, l1)
          atomic = op.apply(ud, l0, l1)
          assert relerr(got, atomic.double().cpu().numpy()) < (
              1e-13 if dtype == torch.float64 else 2e-6)
          want = float((u * ref).sum())
          scale = float(np.abs(u * ref).sum())
          assert abs(float(parts.sum()) - want) <= 10 * tol * scale
          # r -= alpha (Ap from its layers): skipping the chunks nobody writes
          # (`plan.masks`) changes nothing, bit for bit
          scal = torch.zeros(_lib.SFEM_CG_NSCALARS, dtype=torch.float64,
                             device=DEV)
          scal[0], scal[1] = 1.0, 3.0
          r_m, r_u = ud.clone(), ud.clone()
          _ops.cg_update_r_layered(r_m, raw, plan.layers, scal, 0,
                                   masks=plan.masks)
          _ops.cg_update_r_layered(r_u, raw, plan.layers, scal, 0)
          assert torch.equal(r_m, r_u)
          want_r = ud.double() - (1.0 / 3.0) * got.double()
          assert float((r_m.double() - want_r).abs().max()) <= 4 * tol * float(
              want_r.abs().max())
          assert 0 < plan.read <= sum(l for l, _ in plan.layers)
          # folding at SOME nodes first (what a partitioned operator does at
          # its interface) moves their layer values into the nodal vector and
          # clears the slots: the complete fold is unchanged
          pick = torch.unique(torch.randint(
              0, plan.layers[0][0], (200,), device=DEV,
              generator=torch.Generator(device=DEV).manual_seed(1)))
          pick = pick[pick < N]
          part = raw.clone()
          _ops.fold_layers_at(part, pick, N, plan.layers)
          for ln, off in plan.layers:
            inside = pick[pick < ln]
            if inside.numel():
              assert float(part[off + inside].abs().max()) == 0.0
          assert float((part[pick] - got[pick]).abs().max()) <= 1e-14 * float(
              got.abs().max()) * (1 if dtype == torch.float64 else 1e8)
          assert float((_ops.fold_layers(part, N, plan.layers) - got).abs()
                       .max()) <= (1e-14 if dtype == torch.float64 else
                                   1e-6) * float(got.abs().max())
          # the same buffer again (slots nobody writes still zero), and a
          # fresh one: bit for bit the same
          op.apply_layered(ud, ext, l0, l1)
          assert torch.equal(ext[N:], raw[N:])
          again = _ops.fold_layers(ext, N, plan.layers).clone()
          ext2 = op.new_extended()
          op.apply_layered(ud, ext2, l0, l1)
          assert torch.equal(ext2, raw)
          assert torch.equal(again, _ops.fold_layers(ext2, N, plan.layers))
    # a split operator covers part of the mesh: no plan; so does one whose
    # elements partly keep their index rows
    half = torch.arange(mesh.num_elements, device=DEV) % 2 == 0
    a, b = op.split(half)
    assert a.layer_plan() is None and b.layer_plan() is None
    monkeypatch.setenv('SFEM_LAYERED', '0')
    assert operators.HelmholtzOperator.create(fes, None).layer_plan() is None
    monkeypatch.delenv('SFEM_LAYERED')


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_cg_with_layered_assembly(dtype, monkeypatch):
  """`linalg.cg` adds the layers of Ap up inside `r -= alpha Ap`
  (`sfem_cg_update_r_layered`): same iterates and iteration count as the
  atomic assembly and as the oracle's CG (linalg/cg.py:54-97), also under
  graph replay and with a preconditioner slot.  With the inner products
  summed from stored per-wave / per-workgroup partial sums in a fixed order
  (`sfem_cg_scalars_n`, default) no atomic is left in the iteration: whole
  solves are equal BIT FOR BIT from run to run and under graph replay;
  `SFEM_DETERMINISTIC=0` (atomically accumulated sums) agrees to rounding."""
  from swirl_fem_amd.linalg.cg import CGRunner, cg
  P = 8
  monkeypatch.setenv('SFEM_CHAIN_LEN', '2')
  rng = F32Rng(11)
  rp = f32_mesh(make_mesh(2, P, 'stretched', rng), dtype)
  mesh = rp.finalize(device=DEV, dtype=dtype)
  fes = FiniteElementSpace.create(
      mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
  ofes = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
  bmask = mesh.physical_masks['boundary']
  mk = bmask.cpu().numpy()
  op = operators.HelmholtzOperator.create(fes, bmask)
  A = op.linear_operator(0.0, 1.0)
  b = rng.standard_normal(mesh.num_nodes) * ~mk
  bd = dev(b, dtype)
  f64 = dtype == torch.float64
  tol = 1e-10 if f64 else 1e-5

  run = CGRunner(A, bd, tol=tol)
  assert run.layered is not None and len(run.layered.layers) >= 3
  monkeypatch.setenv('SFEM_LAYERED', '0')
  op_a = operators.HelmholtzOperator.create(fes, bmask)
  plain = CGRunner(op_a.linear_operator(0.0, 1.0), bd, tol=tol)
  monkeypatch.delenv('SFEM_LAYERED')
  assert plain.layered is None
  for it in range(12):
    run.step()
    plain.step()
    err = float((run.x - plain.x).abs().max() / plain.x.abs().max())
    assert err < (1e-11 if f64 else 1e-4), (it, err)
  # whole solves: oracle, atomic, layered (twice: bitwise), graph replay
  xo, info_o = O.cg(lambda v: reference(ofes, v, 0.0, 1.0, mk), b, tol=tol)
  x1, i1 = cg(A, bd, tol=tol)
  x2, i2 = cg(A, bd, tol=tol)
  xa, ia = cg(op_a.linear_operator(0.0, 1.0), bd, tol=tol)
  xg, ig = cg(A, bd, tol=tol, graph=True)
  assert torch.equal(x1, x2) and i1['num_iterations'] == i2['num_iterations']
  assert torch.equal(x1, xg) and ig['num_iterations'] == i1['num_iterations']
  assert CGRunner(A, bd, tol=tol).det is not None
  monkeypatch.setenv('SFEM_DETERMINISTIC', '0')
  assert CGRunner(A, bd, tol=tol).det is None
  xn, i_n = cg(A, bd, tol=tol)
  monkeypatch.delenv('SFEM_DETERMINISTIC')
  # (two solves to tol = 1e-10 that may stop one iteration apart)
  close = lambda a, c: float((a - c).abs().max()) <= (
      1e-8 if f64 else 2e-3) * float(c.abs().max())
  assert close(xn, x1) and abs(i_n['num_iterations'] -
                               i1['num_iterations']) <= 1
  slack = 1 if f64 else 3      # 268 iterations: the stop test can flip by one
  assert abs(i1['num_iterations'] - info_o['num_iterations']) <= slack
  assert abs(i1['num_iterations'] - ia['num_iterations']) <= slack
  assert relerr(x1, xo) < (1e-8 if f64 else 2e-3)
  assert relerr(x1, xa.double().cpu().numpy()) < (1e-9 if f64 else 2e-3)
  # a preconditioner in the M slot (Jacobi): the update runs with fuse_rr = 0
  diag = op.apply(torch.ones_like(bd), 1.0, 0.0) + bmask.to(bd.dtype)
  M = lambda r: r / diag
  xm, im = cg(A, bd, tol=tol, M=M)
  xm_a, im_a = cg(op_a.linear_operator(0.0, 1.0), bd, tol=tol, M=M)
  assert abs(im['num_iterations'] - im_a['num_iterations']) <= slack + 2
  assert relerr(xm, xm_a.double().cpu().numpy()) < (1e-9 if f64 else 2e-3)


def test_kernarg_layout_selftest():
  """The facet / chain kernels read their matrix through the kernarg segment
  at a hand-computed offset (`FacetKernarg::MAT_OFF`); the library can check
  that assumption on the device and the Python layer does so before the first
  facet launch."""
  from swirl_fem_amd import _ops
  _ops._KERNARG_CHECKED.clear()
  _ops.kernarg_selftest(torch.device(DEV))
  assert torch.device(DEV) in _ops._KERNARG_CHECKED
  bad = torch.zeros(1, dtype=torch.int32, device=DEV)
  rc = _lib.load().sfem_kernarg_selftest(_ops._ptr(bad), _ops._stream(
      torch.device(DEV)))
  assert rc == 0 and int(bad.item()) == 0
