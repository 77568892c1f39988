"""Gmsh reader and METIS-free partitioner: the reference's own test cases
(common/mesh_reader_test.py:25-80, common/mesh_partitioner_test.py:38-82) on
the reference's data files (tests/golden/msh/*.msh), plus format coverage."""
import collections
import math
import os

import numpy as np
import pytest

from swirl_fem_amd.common import mesh_partitioner, mesh_reader
from swirl_fem_amd.core.interpolation import Nodes1D, NodeType
from swirl_fem_amd.core.mesh_refiner import refine_premesh
from swirl_fem_amd.core.premesh import Premesh

MSH = os.path.join(os.path.dirname(__file__), 'golden', 'msh')


def test_read_1d_mesh():
  pm = mesh_reader.read(os.path.join(MSH, 'line1d.msh'), ndim=1)
  assert pm.node_coords.shape == (17, 1)
  assert pm.elements.shape == (16, 2)
  assert pm.periodic_links is None
  np.testing.assert_array_almost_equal(np.sort(pm.node_coords.flatten()),
                                       np.linspace(0, 1, num=17))


MSH41_QUAD = """$MeshFormat
4.1 0 8
$EndMeshFormat
$Nodes
1 4 1 4
2 1 0 4
1
2
3
4
0 0 0
1 0 0
1 1 0
0 1 0
$EndNodes
$Elements
1 1 1 1
2 1 3 1
1 1 2 3 4
$EndElements
"""

MSH22_QUAD = """$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
4
1 0 0 0
2 1 0 0
3 1 1 0
4 0 1 0
$EndNodes
$Elements
1
1 3 2 0 1 1 2 3 4
$EndElements
"""


@pytest.mark.parametrize('text', [MSH41_QUAD, MSH22_QUAD])
def test_read_single_element_mesh_2d(tmp_path, text):
  path = tmp_path / 'test.msh'
  path.write_text(text)
  pm = mesh_reader.read(path, ndim=2)
  assert pm.node_coords.shape == (4, 2)
  np.testing.assert_array_equal(pm.node_coords,
                                [[0, 0], [1, 0], [1, 1], [0, 1]])
  # vertex order changes to lexicographic
  np.testing.assert_array_equal(pm.elements, [[0, 3, 1, 2]])
  assert pm.periodic_links is None


def test_read_2d_periodic_mesh():
  pm = mesh_reader.read(os.path.join(MSH, 'kovasznay.msh'), ndim=2)
  assert pm.node_coords.shape == (65, 2)
  assert pm.elements.shape == (48, 4)
  assert pm.periodic_links.shape == (4, 2, 2)
  # links join the bottom and top edges: same x, y differing by the period
  x = pm.node_coords
  a, b = x[pm.periodic_links[:, 0]], x[pm.periodic_links[:, 1]]
  np.testing.assert_allclose(a[..., 0], b[..., 0], atol=1e-9)
  assert np.allclose(np.abs(a[..., 1] - b[..., 1]), 2 * np.pi)


def test_read_3d_meshes():
  pm = mesh_reader.read(os.path.join(MSH, 'cube.msh'), ndim=3)
  assert pm.node_coords.shape == (125, 3)
  assert pm.elements.shape == (64, 8)
  assert pm.periodic_links is None
  # lexicographic vertex order: positive Jacobian, axis 0 slowest
  xe = pm.node_coords[pm.elements].reshape(64, 2, 2, 2, 3)
  jac = np.stack([xe[:, 1, 0, 0] - xe[:, 0, 0, 0], xe[:, 0, 1, 0] - xe[:, 0, 0, 0],
                  xe[:, 0, 0, 1] - xe[:, 0, 0, 0]], axis=1)
  assert (np.linalg.det(jac) > 0).all()
  pp = mesh_reader.read(os.path.join(MSH, 'periodic_cube.msh'), ndim=3)
  assert pp.node_coords.shape == (125, 3)
  assert pp.elements.shape == (64, 8)
  assert pp.periodic_links.shape == (48, 2, 4)
  # a triply periodic 4^3 cube refined to order 3 has (4*3)^3 unique nodes
  mesh = refine_premesh(pp, Nodes1D.create(
      4, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize_all()
  gi = mesh['exchange_gather_indices']
  assert mesh['node_coords'].shape[0] == 13 ** 3
  unique = 13 ** 3 - (len(gi) - len(np.unique(mesh['exchange_unique_indices'])))
  assert unique == 12 ** 3


def test_reader_errors(tmp_path):
  with pytest.raises(ValueError, match='Invalid ndim'):
    mesh_reader.read(os.path.join(MSH, 'cube.msh'), ndim=4)
  with pytest.raises(ValueError, match='not found'):
    mesh_reader.read(os.path.join(MSH, 'kovasznay.msh'), ndim=3)
  bad = tmp_path / 'bad.msh'
  bad.write_text('$MeshFormat\n4.1 1 8\n$EndMeshFormat\n')
  with pytest.raises(ValueError, match='binary'):
    mesh_reader.read(bad, ndim=2)


def _unit_interval_mesh(num_elements):
  num_nodes = 1 + num_elements
  return Premesh.create(
      node_coords=np.linspace(0, 1, num_nodes).reshape((num_nodes, 1)),
      elements=np.array([[i, i + 1] for i in range(num_elements)]))


def _check_balance(parts, num_elements, num_partitions):
  counts = collections.Counter(int(p) for p in parts)
  assert len(counts) == min(num_partitions, num_elements)
  for pid, count in counts.items():
    assert 0 <= pid <= num_partitions - 1
    assert (math.floor(num_elements / num_partitions) <= count <=
            math.ceil(num_elements / num_partitions))


@pytest.mark.parametrize('num_elements,num_partitions',
                         [(2, 2), (8, 2), (16, 4), (15, 4), (35, 8), (7, 3)])
def test_partition_1d_mesh(num_elements, num_partitions):
  part = mesh_partitioner.partition(_unit_interval_mesh(num_elements),
                                    num_partitions=num_partitions)
  assert part.num_nodes == num_elements + 1
  assert part.num_elements == num_elements
  assert len(part.partitions) == num_elements
  _check_balance(part.partitions, num_elements, num_partitions)
  for p in range(num_partitions):
    elems = set(i for i, k in enumerate(part.partitions) if p == k)
    assert elems == set(range(min(elems), 1 + max(elems)))


@pytest.mark.parametrize('num_partitions', [2, 3, 4, 8])
def test_partition_unit_cube(num_partitions):
  pm = mesh_reader.read(os.path.join(MSH, 'cube.msh'), ndim=3)
  part = mesh_partitioner.partition(pm, num_partitions=num_partitions)
  assert part.node_coords.shape == (125, 3)
  assert part.elements.shape == (64, 8)
  _check_balance(part.partitions, 64, num_partitions)
  if num_partitions == 8:
    # octants of the cube: 98 of the 125 vertices lie inside one partition
    owners = collections.defaultdict(set)
    for e, p in zip(part.elements, part.partitions):
      for v in e:
        owners[int(v)].add(int(p))
    assert sum(len(s) == 1 for s in owners.values()) == 8 * 8
  # the partitioned premesh finalises into the reference-style (P, S) tables
  arrays = refine_premesh(part, Nodes1D.create(
      3, NodeType.GAUSS_LOBATTO_LEGENDRE)).finalize_all(axis_name='parts')
  assert arrays['elements'].shape[0] == num_partitions


def test_hdf5_snapshot_container_round_trip(tmp_path):
  """`write_snapshots` produces HDF5 (reference niles/datagen/datagen.py:
  127-165: datasets 't', 'u', 'p' at the root) without h5py, through the HDF5
  C library; files read back bit for bit, and the superblock signature is the
  one every HDF5 reader checks."""
  import numpy as np
  from swirl_fem_amd.niles.datagen import datagen, h5lite
  assert h5lite.available()
  rng = np.random.default_rng(0)
  data = {'t': np.linspace(0.0, 1e-3, 6), 'u': rng.standard_normal((6, 50, 2)),
          'p': rng.standard_normal((6, 17)).astype(np.float32)}
  path = datagen.write_snapshots(str(tmp_path / 'cycle_0_5'), data)
  assert path.endswith('.hdf5')
  with open(path, 'rb') as f:
    assert f.read(8) == b'\x89HDF\r\n\x1a\n'
  back = datagen.read_snapshots(path)
  assert sorted(back) == ['p', 't', 'u']
  for k, v in data.items():
    assert back[k].dtype == v.dtype and back[k].shape == v.shape
    np.testing.assert_array_equal(back[k], v)
  # integer / empty / scalar-shaped datasets, and overwriting a file
  more = {'ids': np.arange(5, dtype=np.int32), 'empty': np.zeros((0, 3)),
          'flag': np.array([True, False])}
  h5lite.write(path, more)
  back = h5lite.read(path)
  assert back['ids'].tolist() == [0, 1, 2, 3, 4] and back['empty'].shape == (0, 3)
  assert back['flag'].tolist() == [1, 0] and 'u' not in back
  npz = datagen.write_snapshots(str(tmp_path / 'c'), data, format='npz')
  assert npz.endswith('.npz')
  np.testing.assert_array_equal(datagen.read_snapshots(npz)['u'], data['u'])
  with pytest.raises(ValueError):
    datagen.write_snapshots(str(tmp_path / 'd'), data, format='netcdf')
