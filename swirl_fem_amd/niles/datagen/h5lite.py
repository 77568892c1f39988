"""Minimal HDF5 reader / writer over the HDF5 C library (ctypes).

The reference writes its snapshots with `h5py` (`niles/datagen/datagen.py:
127-165`: datasets 't', 'u', 'p' at the root of one file per cycle).  `h5py`
is preferred when it is importable; this module binds the few `libhdf5` calls
needed for the same files (contiguous numeric datasets at the root group) so
that the container format does not depend on a Python package being present.
Files written here open in h5py / HDF5 tools and vice versa.

    write(path, {'t': t, 'u': u, 'p': p})        read(path) -> dict of arrays
"""

from __future__ import annotations

import ctypes
import ctypes.util
import glob
import os

from swirl_fem_amd import switches

import numpy as np

_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5T_INTEGER, _H5T_FLOAT = 0, 1
_H5T_SGN_NONE = 0
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0

_lib = None


class H5Error(RuntimeError):
  pass


def _candidates():
  env = switches.get('SFEM_HDF5_LIB')
  if env:
    yield env
  found = ctypes.util.find_library('hdf5')
  if found:
    yield found
  for pattern in ('/opt/conda/lib/libhdf5.so*', '/usr/lib/*/libhdf5*.so*',
                  '/usr/lib/*/hdf5/serial/libhdf5.so*',
                  '/usr/local/lib/libhdf5.so*'):
    for path in sorted(glob.glob(pattern)):
      if '_hl' not in path and '_cpp' not in path and 'fortran' not in path:
        yield path


def available() -> bool:
  try:
    _load()
    return True
  except H5Error:
    return False


def _load():
  global _lib
  if _lib is not None:
    return _lib
  lib = None
  for path in _candidates():
    try:
      lib = ctypes.CDLL(path)
      break
    except OSError:
      continue
  if lib is None:
    raise H5Error('no HDF5 C library found (set SFEM_HDF5_LIB or install '
                  'h5py)')
  hid = ctypes.c_int64                      # hid_t is 64-bit since HDF5 1.10
  major, minor, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
  lib.H5open()
  lib.H5get_libversion(ctypes.byref(major), ctypes.byref(minor),
                       ctypes.byref(rel))
  if (major.value, minor.value) < (1, 10):
    raise H5Error(f'HDF5 {major.value}.{minor.value} is too old (need 1.10)')
  sig = {
      'H5Fcreate': (hid, [ctypes.c_char_p, ctypes.c_uint, hid, hid]),
      'H5Fopen': (hid, [ctypes.c_char_p, ctypes.c_uint, hid]),
      'H5Fclose': (ctypes.c_int, [hid]),
      'H5Screate_simple': (hid, [ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p]),
      'H5Sclose': (ctypes.c_int, [hid]),
      'H5Dcreate2': (hid, [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]),
      'H5Dopen2': (hid, [hid, ctypes.c_char_p, hid]),
      'H5Dwrite': (ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
      'H5Dread': (ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
      'H5Dget_space': (hid, [hid]),
      'H5Dget_type': (hid, [hid]),
      'H5Dclose': (ctypes.c_int, [hid]),
      'H5Sget_simple_extent_ndims': (ctypes.c_int, [hid]),
      'H5Sget_simple_extent_dims': (ctypes.c_int, [hid, ctypes.c_void_p,
                                                   ctypes.c_void_p]),
      'H5Tget_class': (ctypes.c_int, [hid]),
      'H5Tget_size': (ctypes.c_size_t, [hid]),
      'H5Tget_sign': (ctypes.c_int, [hid]),
      'H5Tclose': (ctypes.c_int, [hid]),
      'H5Gget_num_objs': (ctypes.c_int, [hid, ctypes.c_void_p]),
      'H5Gget_objname_by_idx': (ctypes.c_ssize_t, [hid, ctypes.c_uint64,
                                                  ctypes.c_char_p,
                                                  ctypes.c_size_t]),
      'H5Gopen2': (hid, [hid, ctypes.c_char_p, hid]),
      'H5Gclose': (ctypes.c_int, [hid]),
      'H5Eset_auto2': (ctypes.c_int, [hid, ctypes.c_void_p, ctypes.c_void_p]),
  }
  for name, (res, args) in sig.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as exc:   # e.g. a build without the 1.6 API
      raise H5Error(f'libhdf5 lacks {name}: {exc}') from exc
    fn.restype, fn.argtypes = res, args
  lib.H5Eset_auto2(0, None, None)           # errors come back as status codes
  lib._hid = hid
  _lib = lib
  return lib


_NATIVE = {
    np.dtype('float64'): 'H5T_NATIVE_DOUBLE_g',
    np.dtype('float32'): 'H5T_NATIVE_FLOAT_g',
    np.dtype('int64'): 'H5T_NATIVE_INT64_g',
    np.dtype('int32'): 'H5T_NATIVE_INT32_g',
    np.dtype('uint8'): 'H5T_NATIVE_UINT8_g',
    np.dtype('int8'): 'H5T_NATIVE_INT8_g',
    np.dtype('uint32'): 'H5T_NATIVE_UINT32_g',
    np.dtype('uint64'): 'H5T_NATIVE_UINT64_g',
}


def _native(lib, dtype):
  name = _NATIVE.get(np.dtype(dtype))
  if name is None:
    raise H5Error(f'dtype {dtype} is not supported by h5lite')
  return ctypes.c_int64.in_dll(lib, name).value


def write(path: str, datasets: dict) -> None:
  """Creates `path` (truncating) with one contiguous dataset per entry."""
  lib = _load()
  f = lib.H5Fcreate(os.fsencode(path), _H5F_ACC_TRUNC, _H5P_DEFAULT,
                    _H5P_DEFAULT)
  if f < 0:
    raise H5Error(f'cannot create {path}')
  try:
    for name, value in datasets.items():
      a = np.ascontiguousarray(value)
      if a.dtype == np.bool_:
        a = a.astype(np.uint8)
      mem = _native(lib, a.dtype)
      dims = (ctypes.c_uint64 * max(a.ndim, 1))(*a.shape)
      space = lib.H5Screate_simple(a.ndim, dims, None)
      if space < 0:
        raise H5Error(f'dataspace of {name!r}')
      dset = lib.H5Dcreate2(f, name.encode(), mem, space, _H5P_DEFAULT,
                            _H5P_DEFAULT, _H5P_DEFAULT)
      if dset < 0:
        lib.H5Sclose(space)
        raise H5Error(f'cannot create dataset {name!r}')
      rc = 0
      if a.size:
        rc = lib.H5Dwrite(dset, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                          a.ctypes.data_as(ctypes.c_void_p))
      lib.H5Dclose(dset)
      lib.H5Sclose(space)
      if rc < 0:
        raise H5Error(f'writing dataset {name!r} failed')
  finally:
    if lib.H5Fclose(f) < 0:
      raise H5Error(f'closing {path} failed')


def read(path: str) -> dict:
  """All numeric datasets at the root of `path` as NumPy arrays."""
  lib = _load()
  f = lib.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, _H5P_DEFAULT)
  if f < 0:
    raise H5Error(f'cannot open {path}')
  out = {}
  try:
    root = lib.H5Gopen2(f, b'/', _H5P_DEFAULT)
    count = ctypes.c_uint64(0)
    if root < 0 or lib.H5Gget_num_objs(root, ctypes.byref(count)) < 0:
      raise H5Error(f'cannot list the root group of {path}')
    names = []
    for k in range(count.value):
      buf = ctypes.create_string_buffer(1024)
      if lib.H5Gget_objname_by_idx(root, k, buf, 1024) < 0:
        raise H5Error(f'cannot read the name of object {k} of {path}')
      names.append(buf.value)
    lib.H5Gclose(root)
    for name in names:
      dset = lib.H5Dopen2(f, name, _H5P_DEFAULT)
      if dset < 0:
        continue                                      # a group: not ours
      space, ftype = lib.H5Dget_space(dset), lib.H5Dget_type(dset)
      nd = lib.H5Sget_simple_extent_ndims(space)
      dims = (ctypes.c_uint64 * max(nd, 1))()
      if nd > 0:
        lib.H5Sget_simple_extent_dims(space, dims, None)
      cls, size = lib.H5Tget_class(ftype), lib.H5Tget_size(ftype)
      if cls == _H5T_FLOAT:
        dtype = np.dtype(f'f{size}')
      elif cls == _H5T_INTEGER:
        kind = 'u' if lib.H5Tget_sign(ftype) == _H5T_SGN_NONE else 'i'
        dtype = np.dtype(f'{kind}{size}')
      else:
        dtype = None
      if dtype is not None and dtype in _NATIVE:
        a = np.empty(tuple(dims[:nd]), dtype=dtype)
        if a.size and lib.H5Dread(dset, _native(lib, dtype), _H5S_ALL,
                                  _H5S_ALL, _H5P_DEFAULT,
                                  a.ctypes.data_as(ctypes.c_void_p)) < 0:
          raise H5Error(f'reading dataset {name!r} failed')
        out[name.decode()] = a
      lib.H5Tclose(ftype)
      lib.H5Sclose(space)
      lib.H5Dclose(dset)
  finally:
    lib.H5Fclose(f)
  return out
