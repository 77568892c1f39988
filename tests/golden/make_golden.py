"""Generates the golden fixtures under tests/golden/ from the reference.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

What runs: the reference's own NumPy/SciPy code, unmodified, imported from
/root/reference -- `Nodes1D`, `Quadrature1D`, `BarycentricInterpolator`
(core/interpolation.py), `unit_cube_mesh` (common/premesh_commons.py),
`refine_premesh` (core/mesh_refiner.py) and the index builders of
core/gather_scatter.py.  Those modules `import jax` / `flax` /
`more_itertools` at the top although the functions used here never touch
them; the packages are not installed in this image, so inert placeholder
modules are registered first purely to let the import statements succeed.  No
JAX behaviour is emulated and nothing that needs JAX (fespace, Mesh ops, cg,
navier_stokes) is called -- those parts of the oracle are pinned by the
reference's analytic known-answer tests instead (tests/test_oracle_*.py).

The fixtures are plain data (inputs + outputs); neither the reference source
nor this shim travels to the GPU box.
"""

import dataclasses
import itertools
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _install_placeholders():
  def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m

  jax = mod('jax')
  jax.numpy = mod('jax.numpy')
  mod('jax.typing', ArrayLike=object)
  jax.tree_util = mod('jax.tree_util', tree_map=None, tree_leaves=None)
  jax.lax = mod('jax.lax')
  jax.Array = object
  jax.vmap = None
  # jax.tree.map over python lists of arrays == a list comprehension
  jax.tree = mod('jax.tree', map=lambda f, *xs: [f(*a) for a in zip(*xs)])
  flax = mod('flax')
  flax.struct = mod(
      'flax.struct', dataclass=dataclasses.dataclass,
      field=lambda pytree_node=True, **kw: dataclasses.field(**kw))

  def pairwise(it):
    a, b = itertools.tee(it)
    next(b, None)
    return zip(a, b)

  def powerset(it):
    s = list(it)
    return itertools.chain.from_iterable(
        itertools.combinations(s, r) for r in range(len(s) + 1))

  mod('more_itertools', pairwise=pairwise, powerset=powerset)


def main():
  _install_placeholders()
  sys.path.insert(0, '/root/reference')
  warnings.simplefilter('ignore', RuntimeWarning)
  from swirl_fem.core import interpolation as I
  from swirl_fem.core import gather_scatter as GS
  from swirl_fem.core import mesh_refiner
  from swirl_fem.core.premesh import Premesh
  from swirl_fem.common import premesh_commons
  from swirl_fem.common import facet_util

  NT = I.NodeType
  types_ = {'nc': NT.NEWTON_COTES, 'gl': NT.GAUSS_LEGENDRE,
            'gll': NT.GAUSS_LOBATTO_LEGENDRE}

  # ------------------------------------------------------------------ 1D data
  out = {}
  for name, nt in types_.items():
    for p in range(2, 13):
      nodes = I.Nodes1D.create(p, nt)
      quad = I.Quadrature1D.create_from_nodes_1d(nodes)
      bi = I.BarycentricInterpolator(1, nodes, nodes)
      out[f'{name}{p}_nodes'] = nodes.node_values
      out[f'{name}{p}_weights'] = quad.weights
      out[f'{name}{p}_bary'] = bi._barycentric_weights()
      out[f'{name}{p}_D'] = bi._differentiation_matrix_1d()
  pairs = []
  for p in range(2, 13):
    pairs += [('gll', p, 'gl', q) for q in (p, p + 1, p + 2)]
    pairs += [('gll', p, 'gll', p + 2), ('gll', p, 'gll', p + 1),
              ('nc', 2, 'gll', p), ('nc', 2, 'gl', p), ('nc', p, 'gl', p)]
    if p >= 3:
      pairs += [('gll', p, 'gll', p - 1), ('gll', p - 1, 'gll', p)]
    if p >= 4:
      pairs += [('gl', p - 2, 'gll', p)]
  for (gn, gp, en, ep) in sorted(set(pairs)):
    bi = I.BarycentricInterpolator(
        1, I.Nodes1D.create(gp, types_[gn]), I.Nodes1D.create(ep, types_[en]))
    out[f'I_{gn}{gp}_{en}{ep}'] = bi._interpolation_matrix_1d()
  # a few dense Kronecker forms (axis ordering check)
  for (d, gp, ep) in [(2, 3, 4), (3, 3, 3), (3, 2, 3)]:
    bi = I.BarycentricInterpolator(
        d, I.Nodes1D.create(gp, NT.GAUSS_LOBATTO_LEGENDRE),
        I.Nodes1D.create(ep, NT.GAUSS_LEGENDRE))
    out[f'kron_M_d{d}_gll{gp}_gl{ep}'] = bi.interpolation_matrix()
    out[f'kron_G_d{d}_gll{gp}_gl{ep}'] = bi.interpolation_matrix_grad()
  # single evaluation point (BDF/EXT style)
  for k in range(1, 5):
    grid = I.Nodes1D.create(k + 1, NT.NEWTON_COTES)
    h = 2 / k
    ext = I.BarycentricInterpolator(
        1, grid, I.Nodes1D.create_single_point(np.array(1 + h)))
    bdf = I.BarycentricInterpolator(
        1, grid, I.Nodes1D.create_single_point(np.array(1.0)))
    out[f'ext{k}_M'] = ext.interpolation_matrix().reshape(-1)
    out[f'bdf{k}_G'] = bdf.interpolation_matrix_grad().reshape(-1) * h
  out['weights_nd_gll4_d3'] = I.Quadrature1D.create(
      4, NT.GAUSS_LOBATTO_LEGENDRE).weights_nd(3)
  np.savez_compressed(os.path.join(HERE, 'interp1d.npz'), **out)
  print('interp1d.npz', len(out), 'arrays')

  # -------------------------------------------------------------- facet util
  fo = {}
  for d, npts in [(1, 3), (2, 3), (2, 4), (1, 6)]:
    mp = facet_util.get_orderings_mapping(d, npts)
    fo[f'ord_d{d}_n{npts}_keys'] = np.array(list(mp.keys()))
    fo[f'ord_d{d}_n{npts}_vals'] = np.stack(list(mp.values()))
  np.savez_compressed(os.path.join(HERE, 'facet_util.npz'), **fo)

  # ------------------------------------------------------------------ meshes
  def premesh_dict(pm, prefix, d):
    d[prefix + 'node_coords'] = pm.node_coords
    d[prefix + 'elements'] = pm.elements
    for k, v in pm.physical_groups.items():
      d[prefix + 'group_' + k] = v
    if pm.periodic_links is not None:
      d[prefix + 'periodic_links'] = pm.periodic_links
    if pm.partitions is not None:
      d[prefix + 'partitions'] = pm.partitions

  def finalize_arrays(pm, prefix, d):
    """Index-builder outputs the reference's Premesh.finalize computes."""
    if pm.partitions is None:
      ni = GS.get_unique_node_indices(
          np.arange(pm.num_nodes, dtype=np.int32), pm.periodic_links)
      gi, ui = GS.get_exchange_indices(ni)
      d[prefix + 'node_indices'] = ni
      d[prefix + 'gather_indices'] = gi
      d[prefix + 'unique_indices'] = ui
    else:
      eidx = GS.group_by_partitions(pm.partitions)
      pad = eidx == -1
      elements = np.where(pad[..., None], -1, pm.elements[np.where(pad, 0, eidx)])
      local_nodes, local_elements = GS.get_local_elements(elements)
      ni = GS.get_unique_node_indices(local_nodes, pm.periodic_links)
      gi, ui = GS.get_exchange_indices(ni)
      assert ui is None
      d[prefix + 'element_indices'] = eidx
      d[prefix + 'local_nodes'] = local_nodes
      d[prefix + 'local_elements'] = local_elements
      d[prefix + 'node_indices'] = ni
      d[prefix + 'gather_indices'] = gi

  def scramble(pm, seed):
    """Random element order + random cube orientation per element."""
    rng = np.random.default_rng(seed)
    d = pm.ndim
    orients = []
    for perm in itertools.permutations(range(d)):
      for r in range(d + 1):
        for axes in itertools.combinations(range(d), r):
          orients.append((perm, axes))
    el = pm.elements[rng.permutation(pm.num_elements)]
    new = []
    for e in el:
      perm, axes = orients[rng.integers(len(orients))]
      new.append(np.flip(e.reshape([2] * d).transpose(perm), axes).reshape(-1))
    return dataclasses.replace(pm, elements=np.array(new, dtype=np.int32))

  cases = {
      'q2d_n3_p4': dict(n=3, ndim=2, p=4),
      'q2d_n4_p3_per0': dict(n=4, ndim=2, p=3, periodic_dims=(0,)),
      'q2d_n4_p3_per01': dict(n=4, ndim=2, p=3, periodic_dims=(0, 1)),
      'q2d_n16_p4': dict(n=16, ndim=2, p=4),
      'h3d_n2_p3': dict(n=2, ndim=3, p=3),
      'h3d_n2_p4': dict(n=2, ndim=3, p=4, a=-1.0, b=2.0),
      'h3d_n3_p3_per012': dict(n=3, ndim=3, p=3, periodic_dims=(0, 1, 2)),
      'h3d_n2_p2': dict(n=2, ndim=3, p=2),
      'l1d_n5_p5': dict(n=5, ndim=1, p=5),
      'q2d_n3_gl3': dict(n=3, ndim=2, p=3, nt='gl'),
      'h3d_n2_gl2': dict(n=2, ndim=3, p=2, nt='gl'),
      'q2d_n4_p3_part22': dict(n=4, ndim=2, p=3,
                               partitions=np.arange(4).reshape(2, 2)),
      'q2d_n4_p3_part22_per0': dict(n=4, ndim=2, p=3, periodic_dims=(0,),
                                    partitions=np.arange(4).reshape(2, 2)),
      'h3d_n4_p3_part222_per012': dict(
          n=4, ndim=3, p=3, periodic_dims=(0, 1, 2),
          partitions=np.arange(8).reshape(2, 2, 2)),
      'h3d_n2_p3_part211': dict(n=2, ndim=3, p=3,
                                partitions=np.arange(2).reshape(2, 1, 1)),
      'q2d_n3_p5_scr': dict(n=3, ndim=2, p=5, scramble=7),
      'h3d_n2_p4_scr': dict(n=2, ndim=3, p=4, scramble=11),
      'h3d_n3_p3_scr': dict(n=3, ndim=3, p=3, scramble=13),
      'h3d_n2_p5_scr_per2': dict(n=2, ndim=3, p=5, scramble=5,
                                 periodic_dims=(2,)),
  }
  md = {}
  for name, c in cases.items():
    pm = premesh_commons.unit_cube_mesh(
        c['n'], ndim=c['ndim'], a=c.get('a', 0.0), b=c.get('b', 1.0),
        periodic_dims=c.get('periodic_dims', ()),
        partitions=c.get('partitions'))
    premesh_dict(pm, name + '/cube/', md)
    if 'scramble' in c:
      pm = scramble(pm, c['scramble'])
      premesh_dict(pm, name + '/scrambled/', md)
    grid = I.Nodes1D.create(c['p'], types_[c.get('nt', 'gll')])
    rp = mesh_refiner.refine_premesh(pm, grid)
    premesh_dict(rp, name + '/refined/', md)
    try:
      finalize_arrays(rp, name + '/final/', md)
    except NotImplementedError as e:
      md[name + '/final/not_implemented'] = np.array(str(e))
  # uneven partition sizes (padding) on a 1D mesh, refined to p=3
  pm = premesh_commons.unit_cube_mesh(6, ndim=1)
  pm = dataclasses.replace(pm, partitions=np.array([0, 0, 1, 1, 2, 3],
                                                   dtype=np.int32))
  rp = mesh_refiner.refine_premesh(
      pm, I.Nodes1D.create(3, NT.GAUSS_LOBATTO_LEGENDRE))
  premesh_dict(pm, 'l1d_n6_p3_uneven/cube/', md)
  premesh_dict(rp, 'l1d_n6_p3_uneven/refined/', md)
  finalize_arrays(rp, 'l1d_n6_p3_uneven/final/', md)
  np.savez_compressed(os.path.join(HERE, 'meshes.npz'), **md)
  print('meshes.npz', len(md), 'arrays')


if __name__ == '__main__':
  main()
