"""ctypes loader for the C-ABI library `libsfem_hip.so` (include/sfem.h).

There is no CPU fallback: if the library is missing or a tensor is not on the
GPU the call raises.  Build the library with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C swirl_fem_amd/csrc -j8`.
"""

from __future__ import annotations

import ctypes
import os

from swirl_fem_amd import switches

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFEM_LIB: another build of the same library (kernel A/B experiments)
LIB_PATH = switches.get('SFEM_LIB') or os.path.join(_HERE, 'libsfem_hip.so')
ABI_VERSION = 7

SFEM_F32, SFEM_F64 = 0, 1
SFEM_CG_NSCALARS_NAMED = 16
SFEM_CG_NSCALARS = 80      # 16 named scalars + 64 partial sums of gamma_new
CG_STATUS = {0: 'running', 1: 'converged', 2: 'maxiter', 3: 'breakdown_pAp',
             4: 'breakdown_gamma'}
SFEM_DOT_SLOTS = 1024
SFEM_CG_MEAN_SUMS = 256    # 2 parities x (64 sums of 1.r + 64 sums of w.r)
SFEM_MAX_LAYERS = 15
SFEM_FOLD_GROUPS = 256
SFEM_LAYER_CHUNK = 512
SFEM_CG_LAZY_MAX = 8
SFEM_IDX_MASK = 0x3FFFFFFF   # node id field of an encoded index (and its pad value)
SFEM_ENS_NSCALARS = 16   # per member: the named scalars of the single solve
SFEM_ENS_GROUPS = 32      # stored partial sums per member and inner product
SFEM_ENS_MAX_MEMBERS = 4096

c_i32, c_i64, c_dbl, c_ptr = (ctypes.c_int32, ctypes.c_int64, ctypes.c_double,
                              ctypes.c_void_p)


class HelmholtzArgs(ctypes.Structure):
  """Mirror of `struct sfem_helmholtz_args`."""
  _fields_ = [
      ('u', c_ptr), ('out', c_ptr), ('enc', c_ptr), ('geo', c_ptr),
      ('geo_elem', c_ptr), ('geo_index', c_ptr), ('elem_list', c_ptr),
      ('dmat', c_ptr), ('weights', c_ptr), ('nodes', c_ptr),
      ('num_elements', c_i64), ('num_listed', c_i64), ('num_nodes', c_i64),
      ('zero_begin', c_i64), ('zero_end', c_i64), ('ndim', c_i32),
      ('P', c_i32), ('ncomp', c_i32), ('dtype', c_i32), ('geo_mode', c_i32),
      ('colored', c_i32), ('lambda0', c_dbl), ('lambda1', c_dbl),
      ('node_stride', c_i64), ('comp_stride', c_i64), ('dot_out', c_ptr),
      ('shared_order', c_ptr), ('shared_stride', c_i32),
      ('cluster_elems', c_ptr), ('cluster_offsets', c_ptr),
      ('cluster_nodes', c_ptr), ('num_clusters', c_i64),
      ('facet_table', c_ptr), ('geo_const', c_ptr),
      ('chain_offsets', c_ptr), ('chain_elems', c_ptr), ('num_chains', c_i64),
      ('layered_extent', c_i64), ('dot_slots', c_i64),
  ]


class StokesArgs(ctypes.Structure):
  """Mirror of `struct sfem_stokes_args`."""
  _fields_ = [
      ('u', c_ptr), ('out', c_ptr), ('p_in', c_ptr), ('p_out', c_ptr),
      ('scale', c_ptr), ('enc', c_ptr), ('penc', c_ptr), ('kfac', c_ptr),
      ('geo_elem', c_ptr), ('geo_index', c_ptr), ('elem_list', c_ptr),
      ('dmat', c_ptr), ('weights', c_ptr), ('nodes', c_ptr), ('interp', c_ptr),
      ('num_elements', c_i64), ('num_listed', c_i64), ('num_nodes', c_i64),
      ('zero_begin', c_i64), ('zero_end', c_i64), ('ndim', c_i32),
      ('P', c_i32), ('dtype', c_i32), ('geo_mode', c_i32),
      ('node_stride', c_i64), ('comp_stride', c_i64),
      ('scale_per_node', c_i32), ('shared_order', c_ptr),
      ('shared_stride', c_i32), ('dot_out', c_ptr),
      ('facet_table', c_ptr), ('chain_offsets', c_ptr), ('chain_elems', c_ptr),
      ('num_chains', c_i64),
  ]


GEO_POINT, GEO_AFFINE, GEO_MULTILINEAR, GEO_BOX = 0, 1, 3, 5

# name -> argument types (all functions return int unless noted)
SIGNATURES = {
    'sfem_gather': [c_ptr, c_ptr, c_ptr, c_i64, c_dbl, c_i32, c_ptr],
    'sfem_gather_rows': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    'sfem_scatter_add': [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_i32,
                         c_ptr],
    'sfem_scatter_csr': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                         c_ptr],
    'sfem_exchange_local': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr,
                            c_i64, c_i32, c_i32, c_ptr],
    'sfem_exchange_classes': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64,
                              c_i32, c_ptr],
    'sfem_zero_strips': [c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr],
    'sfem_subtract_weighted_mean': [c_ptr, c_ptr, c_dbl, c_ptr, c_ptr, c_i64,
                                    c_ptr, c_i32, c_ptr],
    'sfem_pack': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    'sfem_unpack_add': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    'sfem_pack_strided': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64, c_i64,
                          c_i32, c_ptr],
    'sfem_unpack_add_atomic': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64,
                               c_i64, c_i32, c_ptr],
    'sfem_geom_factors': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i32,
                          c_ptr, c_ptr, c_ptr, c_i32, c_ptr],
    'sfem_basis_eval': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                        c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    'sfem_basis_eval_t': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                          c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                          c_ptr],
    'sfem_helmholtz_setup': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                             c_i32, c_ptr],
    'sfem_encode_elements': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr],
    'sfem_helmholtz_apply': [ctypes.POINTER(HelmholtzArgs), c_ptr],
    'sfem_helmholtz_setup_multilinear': [c_ptr, c_ptr, c_i64, c_i32, c_i32,
                                         c_i32, c_ptr],
    'sfem_helmholtz_local': [ctypes.POINTER(HelmholtzArgs), c_ptr],
    'sfem_facet_table_build': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                               c_i32, c_ptr],
    'sfem_helmholtz_setup_affine': [c_ptr, c_ptr, c_i64, c_dbl, c_i32, c_ptr],
    'sfem_helmholtz_cluster_limits': [c_i32, c_i32, ctypes.POINTER(c_i32),
                                      ctypes.POINTER(c_i32)],
    'sfem_dot': [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr],
    'sfem_dot_accumulate': [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr],
    'sfem_dot_indexed': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64,
                         c_i64, c_dbl, c_ptr, c_i32, c_ptr],
    'sfem_cg_scalars': [c_ptr, c_i32, c_dbl, c_dbl, c_dbl, c_ptr, c_ptr],
    'sfem_cg_update_xr': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i32,
                          c_i32, c_ptr],
    'sfem_cg_update_p': [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr],
    'sfem_cg_update_r': [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_i32, c_ptr],
    'sfem_cg_update_xp': [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr],
    'sfem_cg_update_r_layered': [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i32,
                                 c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    'sfem_kernarg_selftest': [c_ptr, c_ptr],
    'sfem_fdm_solve': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                       c_i32, c_i32, c_ptr],
    'sfem_fdm_solve_sums': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                            c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr],
    'sfem_add_element_constants': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i64,
                                   c_i32, c_ptr],
    'sfem_ell_chebyshev': [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                           c_i32, c_i32, c_dbl, c_dbl, c_i32, c_ptr],
    'sfem_ens_dot': [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i32, c_i32, c_ptr],
    'sfem_ens_init': [c_ptr, c_ptr, c_i32, c_dbl, c_dbl, c_dbl, c_ptr],
    'sfem_ens_update_r': [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_i32,
                          c_ptr],
    'sfem_ens_close': [c_ptr, c_ptr, c_i32, c_dbl, c_ptr],
    'sfem_ens_update_xp': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i32,
                           c_ptr],
    'sfem_ens_update_r_mean': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr,
                               c_ptr, c_i32, c_ptr],
    'sfem_ens_close_mean': [c_ptr, c_ptr, c_ptr, c_dbl, c_i32, c_dbl, c_ptr],
    'sfem_ens_update_xp_mean': [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr,
                                c_i32, c_ptr],
    'sfem_ens_subtract_weighted_mean': [c_ptr, c_ptr, c_dbl, c_ptr, c_ptr,
                                        c_i64, c_i32, c_i32, c_ptr],
    'sfem_fold_layers': [c_ptr, c_i64, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    'sfem_fold_layers_at': [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32,
                            c_i32, c_ptr],
    'sfem_cg_scalars_n': [c_ptr, c_i32, c_dbl, c_dbl, c_dbl, c_ptr, c_i64,
                          c_ptr],
    'sfem_cg_update_r_layered_det': [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i32,
                                     c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                     c_i32, c_ptr],
    'sfem_cg_update_xp_lazy': [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr,
                               c_i32, c_i32, c_ptr],
    'sfem_cg_flush_x': [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_i32,
                        c_ptr],
    'sfem_cg_update_r_mean': [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i32,
                              c_ptr],
    'sfem_cg_update_xp_mean': [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_dbl,
                               c_i32, c_ptr],
    'sfem_axpby': [c_dbl, c_ptr, c_dbl, c_ptr, c_i64, c_i32, c_ptr],
    'sfem_stokes_setup': [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                          c_i32, c_ptr],
    'sfem_stokes_div': [c_ptr, c_ptr],
    'sfem_stokes_grad_t': [c_ptr, c_ptr],
    'sfem_stokes_e_first': [c_ptr, c_ptr],
    'sfem_stokes_e_second': [c_ptr, c_ptr],
    'sfem_stokes_convect_local': [c_ptr, c_ptr],
    'sfem_abi_version': [],
}

_lib = None


class SfemError(RuntimeError):
  """A C-ABI call returned a non-zero status."""


def load() -> ctypes.CDLL:
  """Loads (once) and returns the library; raises if it is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise SfemError(
        f'{LIB_PATH} is missing: the HIP extension has not been built. '
        'Run `make -C swirl_fem_amd/csrc -j8` (needs hipcc, gfx950). There is '
        'no CPU fallback.')
  lib = ctypes.CDLL(LIB_PATH)
  for name, argtypes in SIGNATURES.items():
    fn = getattr(lib, name)           # AttributeError if a symbol is missing
    fn.argtypes = argtypes
    fn.restype = ctypes.c_int
  lib.sfem_last_error.argtypes = []
  lib.sfem_last_error.restype = ctypes.c_char_p
  if lib.sfem_abi_version() != ABI_VERSION:
    raise SfemError(f'ABI version mismatch: library '
                    f'{lib.sfem_abi_version()} != binding {ABI_VERSION}')
  _lib = lib
  return lib


def check(status: int, who: str):
  if status != 0:
    msg = load().sfem_last_error().decode(errors='replace')
    raise SfemError(f'{who} failed with status {status}: {msg}')
