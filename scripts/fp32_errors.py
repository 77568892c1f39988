"""Observed fp32 errors of every fused kernel against the fp64 oracle, next to
the error the REFERENCE ALGORITHM itself carries when it is evaluated in
float32 (dense Kronecker einsum, stored J^-1 / det J, `np.add.at` scatter in
np.float32: `oracle.sfem_oracle.FESpace(dtype=np.float32)` -- what the JAX
reference gives without x64, core/interpolation.py:262-263, :291-292).

Error measure (the one the parity tests use): max |got - ref64| / max |ref64|.
Inputs are float32-representable, so input rounding is not counted.

  python scripts/fp32_errors.py [out.jsonl]     (needs the GPU)
Writes one JSON line per (case, form, kernel route); `--md` renders
profiles/r04_fp32_errors.md from such a file.
"""
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def f32(x):
  return np.asarray(x, dtype=np.float32).astype(np.float64)


def relerr(a, b):
  a = np.asarray(a, dtype=np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def random_orientations(pm, rng):
  d = pm.ndim
  orients = [(perm, axes) for perm in itertools.permutations(range(d))
             for r in range(d + 1) for axes in itertools.combinations(range(d), r)]
  el = []
  for e in pm.elements[rng.permutation(pm.num_elements)]:
    perm, axes = orients[rng.integers(len(orients))]
    el.append(np.flip(e.reshape([2] * d).transpose(perm), axes).reshape(-1))
  return pm.replace(elements=np.array(el, dtype=np.int32))


def make_mesh(ndim, n, P, mode, rng):
  from swirl_fem_amd.common.premesh_commons import box_mesh, unit_cube_mesh
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  pm = unit_cube_mesh(n, ndim=ndim)
  if mode == 'sheared 8x2x2':      # elements of aspect ratio 4, then sheared
    pm = box_mesh((8, 2, 2), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
  x = pm.node_coords.copy()
  if mode in ('sheared', 'sheared 8x2x2'):
    x = x @ (np.eye(ndim) + 0.3 * rng.uniform(-1, 1, (ndim, ndim))).T + 0.1
  elif mode in ('jittered', 'curved'):
    x = x + 0.1 / n * rng.uniform(-1, 1, x.shape)
  pm = pm.replace(node_coords=x)
  if mode in ('jittered', 'curved'):
    pm = random_orientations(pm, rng)
  rp = refine_premesh(pm, Nodes1D.create(P, NodeType.GAUSS_LOBATTO_LEGENDRE))
  if mode == 'curved':
    xc = rp.node_coords.copy()
    bump = 0.03 * np.sin(np.pi * xc[:, 0]) * np.sin(2 * np.pi * xc[:, 1])
    xc[:, -1] += bump
    rp = rp.replace(node_coords=xc)
  # float32-representable coordinates: both sides see the same mesh
  return rp.replace(node_coords=f32(rp.node_coords))


def helmholtz_rows(out):
  import torch
  from oracle import sfem_oracle as O
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  GLL = NodeType.GAUSS_LOBATTO_LEGENDRE
  dev = 'cuda:0'
  cases = [(3, 3, 4), (3, 2, 6), (3, 2, 8), (3, 2, 10), (3, 2, 12),
           (2, 4, 4), (2, 3, 6), (2, 3, 8), (2, 3, 12)]
  forms = {'mass': (1.0, 0.0), 'stiffness': (0.0, 1.0),
           'helmholtz': (0.6, 1.4)}
  for (ndim, n, P), mode in list(itertools.product(
      cases, ('structured', 'sheared', 'jittered', 'curved'))) + [
          ((3, 2, 8), 'sheared 8x2x2')]:
    rng = np.random.default_rng(1000 * ndim + 10 * P + len(mode))
    rp = make_mesh(ndim, n, P, mode, rng)
    mesh = rp.finalize(device=dev, dtype=torch.float32)
    fes = FiniteElementSpace.create(
        mesh, Quadrature1D.create_from_nodes_1d(Nodes1D.create(P, GLL)))
    o64 = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'))
    o32 = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (P, 'gll'),
                    dtype=np.float32)
    bmask = mesh.physical_masks['boundary']
    keep = 1.0 - bmask.cpu().numpy().astype(np.float64)
    routes = {g: fes.helmholtz_operator(bmask, g)
              for g in ('auto', 'multilinear', 'stored')}
    if mode == 'curved':
      routes = {g: o for g, o in routes.items() if g != 'multilinear'}
    os.environ['SFEM_FACET'] = '0'
    routes['index rows'] = fes.helmholtz_operator(bmask, 'auto')
    del os.environ['SFEM_FACET']
    for nc in (1, ndim):
      u = f32(rng.standard_normal((mesh.num_nodes, nc)))
      uu = u[:, 0] if nc == 1 else u
      ud = (torch.as_tensor(uu, device=dev).float() if nc == 1 else
            torch.as_tensor(u.T.copy(), device=dev).float().t())
      for form, (l0, l1) in forms.items():
        if nc > 1 and form != 'helmholtz':
          continue

        def ref(o, v):
          ul = o.gather(v)
          loc = 0
          if l0:
            loc = loc + np.asarray(l0, o.dtype) * o.mass_local(ul)
          if l1:
            loc = loc + np.asarray(l1, o.dtype) * o.stiffness_local(ul)
          r = o.scatter(loc)
          k = keep.astype(o.dtype)
          return r * (k if r.ndim == 1 else k[:, None])

        r64 = ref(o64, uu)
        e32 = relerr(ref(o32, uu.astype(np.float32)), r64)
        for g, op in routes.items():
          got = op.apply(ud, l0, l1).cpu().numpy()
          row = {'op': 'helmholtz', 'ndim': ndim, 'P': P, 'mesh': mode,
                 'form': form, 'ncomp': nc, 'route': g,
                 'kernel': op.kernel_name(l0, l1, nc).split('<')[0].replace(
                     'sfem::', ''),
                 'hip_f32': relerr(got, r64), 'reference_f32': e32}
          out.write(json.dumps(row) + '\n')
          out.flush()


def stokes_rows(out):
  import torch
  from oracle import sfem_oracle as O
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core import layout, operators
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  GLL, GL = NodeType.GAUSS_LOBATTO_LEGENDRE, NodeType.GAUSS_LEGENDRE
  dev = 'cuda:0'
  os.environ['SFEM_STOKES_FACET_DIV'] = 'all'
  for (ndim, n, P), mode in itertools.product(
      [(2, 3, 4), (2, 4, 6), (2, 2, 12), (3, 2, 4), (3, 2, 6), (3, 2, 8),
       (3, 1, 12)], ('structured', 'sheared', 'jittered')):
    rng = np.random.default_rng(77 * ndim + P + len(mode))
    pm = unit_cube_mesh(n, ndim=ndim)
    x = pm.node_coords.copy()
    if mode == 'sheared':
      x = x @ (np.eye(ndim) + 0.3 * rng.uniform(-1, 1, (ndim, ndim))).T
    if mode == 'jittered':
      x = x + 0.15 / n * rng.uniform(-1, 1, x.shape)
    pm = pm.replace(node_coords=f32(x))
    rv = refine_premesh(pm, Nodes1D.create(P, GLL))
    rq = refine_premesh(pm, Nodes1D.create(P - 2, GL))
    rv = rv.replace(node_coords=f32(rv.node_coords))
    rq = rq.replace(node_coords=f32(rq.node_coords))
    quad = Quadrature1D.create(P, GLL)
    vsp = FiniteElementSpace.create(rv.finalize(device=dev,
                                                dtype=torch.float32), quad)
    psp = FiniteElementSpace.create(rq.finalize(device=dev,
                                                dtype=torch.float32), quad)
    sides = {}
    for name, dt in (('f64', np.float64), ('f32', np.float32)):
      sides[name] = (
          O.FESpace(rv.node_coords, rv.elements, (P, 'gll'), (P, 'gll'), dt),
          O.FESpace(rq.node_coords, rq.elements, (P - 2, 'gl'), (P, 'gll'),
                    dt))
    bmask = vsp.mesh.physical_masks['boundary']
    keep = ~bmask.cpu().numpy()
    u = f32(rng.standard_normal((vsp.mesh.num_nodes, ndim)))
    p = f32(rng.standard_normal(psp.mesh.num_nodes))

    def refs(ov, op):
      d = op.scatter(O.div_local(ov, op, ov.gather(u.astype(ov.dtype))))
      g = keep[:, None] * ov.scatter(
          O.div_t_local(ov, op, op.gather(p.astype(ov.dtype))))
      return d, g

    d64, g64 = refs(*sides['f64'])
    d32, g32 = refs(*sides['f32'])
    sc = f32(rng.uniform(0.5, 2.0, vsp.mesh.num_nodes))

    def e_ref(ov, op, g):      # E = D (s * masked D^T p), navier_stokes.py:345-348
      w = (sc[:, None] * g).astype(ov.dtype)
      return op.scatter(O.div_local(ov, op, ov.gather(w)))

    e64, e32 = e_ref(*sides['f64'], g64), e_ref(*sides['f32'], g32)
    for geometry in ('auto', 'multilinear', 'stored'):
      try:
        fused = operators.StokesDivGrad.create(vsp, psp, bmask, geometry)
      except NotImplementedError as e:   # eligibility check in fp32
        out.write(json.dumps({'op': 'stokes', 'ndim': ndim, 'P': P,
                              'mesh': mode, 'route': geometry,
                              'skipped': str(e)}) + '\n')
        continue
      ud = layout.empty_component_major((vsp.mesh.num_nodes, ndim),
                                        torch.float32, torch.device(dev))
      ud.copy_(torch.as_tensor(u, device=dev).float())
      pd = torch.as_tensor(p, device=dev).float()
      kinds = sorted({q['geo_mode'] for q in
                      (fused.facet_parts or fused.parts)})
      sd = torch.as_tensor(sc, device=dev).float()
      forms = [('div', fused.div(ud), d64, d32),
               ('grad_t', fused.grad_t(pd, component_major=True), g64, g32),
               ('E two kernels', fused.div(
                   fused.grad_t(pd, component_major=True), scale=sd), e64,
                e32)]
      if fused.penc is None:
        forms.append(('E split', fused.e_apply(pd, scale=sd), e64, e32))
      for form, got, r64, r32 in forms:
        out.write(json.dumps({
            'op': 'stokes', 'ndim': ndim, 'P': P, 'mesh': mode, 'form': form,
            'route': geometry, 'geo_modes': kinds,
            'facet': fused.facet_parts is not None,
            'hip_f32': relerr(got.cpu().numpy(), r64),
            'reference_f32': relerr(r32, r64)}) + '\n')
        out.flush()
  del os.environ['SFEM_STOKES_FACET_DIV']


def convection_rows(out):
  import torch
  from oracle import sfem_oracle as O
  from swirl_fem_amd.common.premesh_commons import unit_cube_mesh
  from swirl_fem_amd.core import operators
  from swirl_fem_amd.core.fespace import FiniteElementSpace
  from swirl_fem_amd.core.interpolation import Nodes1D, NodeType, Quadrature1D
  from swirl_fem_amd.core.mesh_refiner import refine_premesh
  GLL = NodeType.GAUSS_LOBATTO_LEGENDRE
  dev = 'cuda:0'
  for (ndim, n, P, extra), mode in itertools.product(
      [(2, 4, 6, 2), (3, 2, 4, 2), (3, 2, 8, 2)], ('structured', 'jittered')):
    rng = np.random.default_rng(5 * P + ndim)
    pm = unit_cube_mesh(n, ndim=ndim)
    x = pm.node_coords.copy()
    if mode == 'jittered':
      x = x + 0.15 / n * rng.uniform(-1, 1, x.shape)
    rp = refine_premesh(pm.replace(node_coords=f32(x)), Nodes1D.create(P, GLL))
    rp = rp.replace(node_coords=f32(rp.node_coords))
    q = P + extra
    fes = FiniteElementSpace.create(
        rp.finalize(device=dev, dtype=torch.float32),
        Quadrature1D.create(q, GLL))
    try:
      conv = operators.ConvectionOperator.create(fes)
    except NotImplementedError as e:
      out.write(json.dumps({'op': 'convection', 'ndim': ndim, 'P': P,
                            'mesh': mode, 'skipped': str(e)}) + '\n')
      continue
    ul = f32(rng.standard_normal(rp.elements.shape + (ndim,)))
    res = {}
    for name, dt in (('f64', np.float64), ('f32', np.float32)):
      o = O.FESpace(rp.node_coords, rp.elements, (P, 'gll'), (q, 'gll'), dt)
      res[name] = o.convection_local(ul.astype(dt), ul.astype(dt))
    got = conv.apply_local(torch.as_tensor(ul, device=dev).float())
    out.write(json.dumps({
        'op': 'convection', 'ndim': ndim, 'P': P, 'q': q, 'mesh': mode,
        'form': 'u.grad(u) v', 'route': 'two-grid',
        'hip_f32': relerr(got.cpu().numpy(), res['f64']),
        'reference_f32': relerr(res['f32'], res['f64'])}) + '\n')
    out.flush()


def render(path, md):
  rows = [json.loads(l) for l in open(path)]
  lines = [
      '# fp32: observed error of the fused kernels vs the reference algorithm in float32',
      '',
      'Measure: max |result - oracle(fp64)| / max |oracle(fp64)| on the same '
      'float32-representable inputs (`scripts/fp32_errors.py`, one MI355X).',
      '`hip` = this build\'s kernel in fp32; `reference` = the oracle\'s '
      'restatement of the reference algorithm evaluated in np.float32 (dense '
      'Kronecker matrices, stored J^-1 and det J, einsum contractions: '
      '`oracle.sfem_oracle.FESpace(dtype=np.float32)`), i.e. the rounding the '
      'JAX reference itself carries at fp32. north_star tolerance: 1e-5.',
      '']
  worst = {}
  for r in rows:
    if 'skipped' in r:
      continue
    key = (r['op'], r['ndim'], r['P'], r['mesh'])
    w = worst.setdefault(key, {'hip': 0.0, 'ref': 0.0, 'where': ''})
    if r['hip_f32'] > w['hip']:
      w['hip'] = r['hip_f32']
      w['where'] = '%s / %s' % (r.get('form', ''), r.get('route', ''))
    w['ref'] = max(w['ref'], r['reference_f32'])
  lines += ['| operator | ndim | P | mesh | worst hip (form / route) | hip | reference in f32 | hip <= 1e-5 |',
            '|---|---|---|---|---|---|---|---|']
  for (op, ndim, P, mesh), w in sorted(worst.items()):
    lines.append('| %s | %d | %d | %s | %s | %.2e | %.2e | %s |' % (
        op, ndim, P, mesh, w['where'], w['hip'], w['ref'],
        'yes' if w['hip'] <= 1e-5 else '**no**'))
  over = [r for r in rows if 'hip_f32' in r and r['hip_f32'] > 1e-5]
  lines += ['', '## Rows above 1e-5 (%d of %d)' % (
      len(over), sum('hip_f32' in r for r in rows)), '']
  if over:
    lines += ['| operator | ndim | P | mesh | form | ncomp | route | hip | reference in f32 |',
              '|---|---|---|---|---|---|---|---|---|']
    for r in over:
      lines.append('| %s | %d | %d | %s | %s | %s | %s | %.2e | %.2e |' % (
          r['op'], r['ndim'], r['P'], r['mesh'], r.get('form', ''),
          r.get('ncomp', ''), r.get('route', ''), r['hip_f32'],
          r['reference_f32']))
  else:
    lines.append('none')
  skipped = [r for r in rows if 'skipped' in r]
  if skipped:
    lines += ['', '## Not on a fused fp32 route', '']
    for r in skipped:
      lines.append('* %s ndim=%d P=%d %s %s: %s' % (
          r['op'], r['ndim'], r['P'], r['mesh'], r.get('route', ''),
          r['skipped']))
  open(md, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--md':
    render(sys.argv[2], sys.argv[3])
    sys.exit(0)
  path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/fp32_errors.jsonl'
  os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
  with open(path, 'w') as out:
    helmholtz_rows(out)
    stokes_rows(out)
    convection_rows(out)
  print('wrote', path)
