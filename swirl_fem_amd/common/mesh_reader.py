"""Parses a Gmsh mesh file into a `Premesh`.

Same entry point and semantics as the reference's `common/mesh_reader.py`
(`read(path, ndim)` :78-114, Gmsh -> lexicographic vertex permutation :40-44,
periodic links from the `$Periodic` section :47-75).  The reference delegates
the file parsing to the third-party `meshio` package (absent here and not
needed): this module reads the ASCII MSH 4.1 and MSH 2.2 formats itself and
exposes exactly what the reference consumes of a `meshio.Mesh`:

  * `points`: node coordinates in file order, nodes renumbered 0-based in that
    order (meshio's gmsh reader does the same tag -> index mapping),
  * `cells_dict[type]`: the cells of one type, blocks concatenated in file
    order, Gmsh vertex order, 0-based node indices,
  * `gmsh_periodic`: `(entity_dim, (slave, master), affine, node_pairs)` with
    `node_pairs[:, 0]` the slave and `[:, 1]` the master node index.
"""

from __future__ import annotations

import dataclasses
import os

import numpy as np

from swirl_fem_amd.core.premesh import Premesh

# Gmsh -> tensor-product (lexicographic, axis 0 slowest) vertex ordering
# (reference common/mesh_reader.py:24-44):
#   quad  3--2     1--3        hexahedron  Gmsh 0..7 -> (x, y, z) corner
#         |  |     |  |                    order with x slowest
#         0--1     0--2
_NODE_ORDERING_PERMUTATIONS = {
    1: [0, 1],
    2: [0, 3, 1, 2],
    3: [0, 4, 3, 7, 1, 5, 2, 6],
}

# gmsh element type -> (meshio-style name, number of nodes)
_GMSH_TYPES = {
    1: ('line', 2), 2: ('triangle', 3), 3: ('quad', 4), 4: ('tetra', 4),
    5: ('hexahedron', 8), 6: ('wedge', 6), 7: ('pyramid', 5), 8: ('line3', 3),
    9: ('triangle6', 6), 10: ('quad9', 9), 11: ('tetra10', 10),
    12: ('hexahedron27', 27), 15: ('vertex', 1), 16: ('quad8', 8),
    17: ('hexahedron20', 20),
}


@dataclasses.dataclass
class GmshMesh:
  """The parts of a Gmsh file that `read` needs (see module docstring)."""
  points: np.ndarray
  cells_dict: dict
  gmsh_periodic: list


def _sections(text: str) -> dict:
  out, name, buf = {}, None, []
  for line in text.splitlines():
    s = line.strip()
    if not s:
      continue
    if s.startswith('$End'):
      if name is not None:
        out.setdefault(name, []).append(buf)
      name, buf = None, []
    elif s.startswith('$'):
      name, buf = s[1:], []
    elif name is not None:
      buf.append(s)
  return out


class _Tokens:
  def __init__(self, lines):
    self.t = ' '.join(lines).split()
    self.i = 0

  def ints(self, n):
    v = [int(x) for x in self.t[self.i:self.i + n]]
    if len(v) != n:
      raise ValueError('unexpected end of a Gmsh section')
    self.i += n
    return v

  def floats(self, n):
    v = [float(x) for x in self.t[self.i:self.i + n]]
    if len(v) != n:
      raise ValueError('unexpected end of a Gmsh section')
    self.i += n
    return v

  def int(self):
    return self.ints(1)[0]


def _parse_v4(sec) -> GmshMesh:
  tk = _Tokens(sec['Nodes'][0])
  nblocks, nnodes, _, _ = tk.ints(4)
  tags = np.empty(nnodes, dtype=np.int64)
  points = np.empty((nnodes, 3), dtype=np.float64)
  at = 0
  for _ in range(nblocks):
    edim, _, parametric, nb = tk.ints(4)
    tags[at:at + nb] = tk.ints(nb)
    width = 3 + (edim if parametric else 0)   # x y z [u [v [w]]]
    points[at:at + nb] = np.asarray(tk.floats(width * nb)).reshape(
        nb, width)[:, :3]
    at += nb
  if at != nnodes:
    raise ValueError(f'$Nodes announces {nnodes} nodes, found {at}')
  index = -np.ones(int(tags.max()) + 1, dtype=np.int64)
  index[tags] = np.arange(nnodes)

  cells = {}
  tk = _Tokens(sec['Elements'][0])
  nblocks, _, _, _ = tk.ints(4)
  for _ in range(nblocks):
    _, _, etype, nb = tk.ints(4)
    if etype not in _GMSH_TYPES:
      raise ValueError(f'unsupported Gmsh element type {etype}')
    name, nn = _GMSH_TYPES[etype]
    rows = np.asarray(tk.ints((nn + 1) * nb), dtype=np.int64).reshape(nb, nn + 1)
    cells.setdefault(name, []).append(index[rows[:, 1:]])
  cells_dict = {k: np.concatenate(v) for k, v in cells.items()}

  periodic = []
  for lines in sec.get('Periodic', []):
    tk = _Tokens(lines)
    for _ in range(tk.int()):
      edim, stag, mtag = tk.ints(3)
      affine = tk.floats(tk.int())
      npairs = tk.int()
      pairs = np.asarray(tk.ints(2 * npairs), dtype=np.int64).reshape(npairs, 2)
      periodic.append((edim, (stag, mtag), affine, index[pairs]))
  return GmshMesh(points=points, cells_dict=cells_dict, gmsh_periodic=periodic)


def _parse_v2(sec) -> GmshMesh:
  tk = _Tokens(sec['Nodes'][0])
  nnodes = tk.int()
  tags = np.empty(nnodes, dtype=np.int64)
  points = np.empty((nnodes, 3), dtype=np.float64)
  for k in range(nnodes):
    tags[k] = tk.int()
    points[k] = tk.floats(3)
  index = -np.ones(int(tags.max()) + 1, dtype=np.int64)
  index[tags] = np.arange(nnodes)

  cells = {}
  tk = _Tokens(sec['Elements'][0])
  for _ in range(tk.int()):
    _, etype, ntags = tk.ints(3)
    tk.ints(ntags)
    if etype not in _GMSH_TYPES:
      raise ValueError(f'unsupported Gmsh element type {etype}')
    name, nn = _GMSH_TYPES[etype]
    cells.setdefault(name, []).append(index[np.asarray(tk.ints(nn))])
  cells_dict = {k: np.stack(v) for k, v in cells.items()}

  periodic = []
  for lines in sec.get('Periodic', []):
    # "dim slave master" [, "Affine" 16 values], count, pairs
    toks = ' '.join(lines).split()
    i = 0
    count = int(toks[i]); i += 1
    for _ in range(count):
      edim, stag, mtag = (int(x) for x in toks[i:i + 3]); i += 3
      affine = []
      if toks[i] == 'Affine':
        affine = [float(x) for x in toks[i + 1:i + 17]]; i += 17
      npairs = int(toks[i]); i += 1
      pairs = np.asarray([int(x) for x in toks[i:i + 2 * npairs]],
                         dtype=np.int64).reshape(npairs, 2)
      i += 2 * npairs
      periodic.append((edim, (stag, mtag), affine, index[pairs]))
  return GmshMesh(points=points, cells_dict=cells_dict, gmsh_periodic=periodic)


def read_gmsh(path) -> GmshMesh:
  """Parses an ASCII `.msh` file (format 4.1 or 2.2)."""
  with open(os.fspath(path), 'r') as f:
    sec = _sections(f.read())
  if 'MeshFormat' not in sec:
    raise ValueError(f'{path}: not a Gmsh file ($MeshFormat missing)')
  fmt = sec['MeshFormat'][0][0].split()
  version, is_binary = float(fmt[0]), int(fmt[1])
  if is_binary:
    raise ValueError(f'{path}: binary Gmsh files are not supported')
  if 'Nodes' not in sec or 'Elements' not in sec:
    raise ValueError(f'{path}: $Nodes / $Elements missing')
  if 4.0 < version < 5.0:
    return _parse_v4(sec)
  if 2.0 <= version < 3.0:
    return _parse_v2(sec)
  raise ValueError(f'{path}: unsupported Gmsh format version {fmt[0]}')


def _get_periodic_links(mesh: GmshMesh, ndim: int) -> np.ndarray:
  """Pairs of `(ndim-1)`-dimensional facets joined by a periodic connection
  (reference :47-75): every facet cell whose nodes are all slaves of one
  periodic entity of that dimension, with the facet of their masters."""
  # One slave -> master node map per periodic entity pair.  (The reference
  # merges all pairs into a single dict, :57-62, so on a mesh that is periodic
  # in several directions the edge / corner nodes keep only the last
  # direction's master; a facet is matched here against the map of the one
  # entity it lies on.)
  maps = [dict(node_pairs.tolist())
          for entity_dim, _, _, node_pairs in mesh.gmsh_periodic
          if entity_dim == ndim - 1]
  facet_type = {1: 'line', 2: 'quad'}[ndim - 1]
  links = []
  for facet in mesh.cells_dict[facet_type]:
    for src_tgt in maps:
      if all(int(x) in src_tgt for x in facet):
        links.append(np.stack([facet, [src_tgt[int(x)] for x in facet]]))
        break
  links = np.stack(links).astype(np.int32)
  # Deviation from the reference, which keeps Gmsh's cyclic vertex order for
  # the quadrilateral facets of 3D links: `refine_premesh` reads facets as
  # lexicographic 2 x 2 tensors (reference core/mesh_refiner.py:155), so a
  # cyclic quad would pair diagonals instead of edges.  Both sides of a link
  # are permuted alike, which keeps the vertex-to-vertex correspondence.
  return links[..., _NODE_ORDERING_PERMUTATIONS[ndim - 1]]


def read(path, ndim: int) -> Premesh:
  """Reads the Gmsh mesh at `path` into a `Premesh` (reference :78-114)."""
  if ndim not in [1, 2, 3]:
    raise ValueError(f'Invalid ndim: {ndim=}. Valid spatial dimensions are '
                     '1, 2 and 3.')
  mesh = read_gmsh(path)
  node_coords = mesh.points[:, :ndim]
  elem_type = {1: 'line', 2: 'quad', 3: 'hexahedron'}[ndim]
  if elem_type not in mesh.cells_dict:
    raise ValueError(
        f'Reading mesh of {ndim=} but cells of type {elem_type=} not found '
        f'in {mesh.cells_dict.keys()=}')
  # reorder the vertices of each element lexicographically
  elements = mesh.cells_dict[elem_type][:, _NODE_ORDERING_PERMUTATIONS[ndim]]
  periodic_links = (_get_periodic_links(mesh, ndim=ndim)
                    if mesh.gmsh_periodic else None)
  return Premesh.create(node_coords=node_coords, elements=elements,
                        periodic_links=periodic_links)
