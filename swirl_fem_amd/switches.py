"""Every `SFEM_*` environment switch of the package, in one table.

All of them select between code paths that give the same results (kernel
variants kept for A/B measurements and for the parity tests that pit one
variant against the other); none is needed in normal use.  The Python side
reads a switch ONLY through `get` / `enabled`, at the moment the decision is
taken (the tests flip them inside one process); a name that is not in the
table raises, and `SFEM_*` variables in the environment that the table does
not know are reported once (`check_environment`) -- a stray or misspelt
variable must not silently change the launched kernel.  `active()` is what
`bench.py` prints into `config.switches`, so that a result line says which
variants produced it.  INTEGRATION.md carries the same table
(`python -m swirl_fem_amd.switches` prints it).
"""

from __future__ import annotations

import os
import warnings

# name -> (default, read by, meaning)
SWITCHES = {
    'SFEM_LIB': (None, '_lib.py',
                 'path of another build of libsfem_hip.so (kernel A/B runs, '
                 '`scripts/build_variant.sh`)'),
    'SFEM_HDF5_LIB': (None, 'niles/datagen/h5lite.py',
                      'path of the HDF5 C library for snapshot files when '
                      'h5py is not installed'),
    'SFEM_FACET': ('1', 'core/operators.py',
                   '0: 3D P = 6..12 operators keep their index rows instead '
                   'of the compact facet tables'),
    'SFEM_BOX': ('1', 'core/operators.py',
                 '0: Cartesian elements run the general affine kernels'),
    'SFEM_CHAIN': ('1', 'core/operators.py, _ops.py',
                   '0: one element per wave instead of chain segments'),
    'SFEM_CHAIN_LEN': (None, 'core/operators.py',
                       'elements per chain segment (default: 8 on large '
                       'meshes, shorter on small ones, '
                       '`operators.chain_segment_length`)'),
    'SFEM_CHAIN_HI': ('0', 'core/operators.py',
                      '1: chains also for P >= 9 box / affine elements '
                      '(slower: measured 1.73 vs 1.47 ms at p = 11)'),
    'SFEM_CHAIN_VECTOR': ('1', '_ops.py',
                          '0: component-major vector fields are not walked '
                          'as chains'),
    'SFEM_LAYERED': ('1', 'core/operators.py',
                     '0: CG keeps the atomic assembly; force: layered '
                     'assembly also for elements with stored factors'),
    'SFEM_DETERMINISTIC': ('1', 'linalg/cg.py',
                           '0: with layered assembly the two inner products '
                           'of a CG iteration still accumulate their partial '
                           'sums with atomics (not bitwise reproducible; one '
                           'scalar launch less per iteration)'),
    'SFEM_LAZY_X': ('4', 'linalg/cg.py',
                    'directions the CG keeps before it adds them to x '
                    '(vectors of 256 MB and more; 0 or 1: x += alpha p every '
                    'iteration)'),
    'SFEM_LAZY_X_MIN_MB': (None, 'linalg/cg.py',
                           'smallest vector (MiB) that gets the lazy x update '
                           '(default 256; the tests set 0)'),
    'SFEM_SORTED_SCATTER': ('1', 'core/operators.py',
                            '0: index-row kernels issue their atomics in slot '
                            'order instead of node order'),
    'SFEM_CLUSTER': ('0', 'core/operators.py',
                     "1: assembly='auto' means cluster assembly (slower)"),
    'SFEM_INTERP': ('1', 'the library (getenv per launch)',
                    '0: values-only sfem_basis_eval / sfem_basis_eval_t run '
                    'the generic kernels instead of the compile-time-sized '
                    'interpolation (csrc/sfem_interp.h)'),
    'SFEM_MFMA': ('0', '_ops.py and the library (getenv per launch)',
                  '1: p = 11 fp32 index-row elements run the matrix-core '
                  'kernel (slower: 1.25 vs 1.0 ms)'),
    'SFEM_FACET_OFF64': ('0', 'the library (getenv per launch)',
                         '1: facet kernels form 64-bit addresses even for '
                         'fields below 4 GiB (tests)'),
    'SFEM_STOKES_FACET': ('1', 'core/operators.py',
                          '0: Stokes div / grad_t keep their index rows'),
    'SFEM_STOKES_FACET_DIV': ('box', 'core/operators.py',
                              'all: the divergence walks chains on every '
                              'geometry, not only on box elements (tests)'),
    'SFEM_SPLIT_E': ('0', 'navier_stokes/navier_stokes.py',
                     '1: pressure operator E in two fused halves '
                     '(`sfem_stokes_e_first/second`)'),
    'SFEM_STOKES_LAYERED': ('auto', 'navier_stokes/navier_stokes.py',
                            'pressure operator E on index-row kernels: D^T '
                            'with one position per writer instead of atomics '
                            '(`StokesDivGrad.e_layered`); auto = 2D meshes, '
                            '1 = 3D index rows too, 0 = never'),
    'SFEM_FUSED_DOTS': ('1', 'navier_stokes/navier_stokes.py',
                        '0: the pressure CG computes p.Ep and r.Mr with '
                        'separate dot kernels'),
    'SFEM_FUSED_MEAN': ('1', 'linalg/cg.py',
                        '0: the mean projection of the pressure '
                        'preconditioner is applied as a separate kernel'),
    'SFEM_GRAPHS': ('1', 'navier_stokes/navier_stokes.py',
                    '0: solver iterations are never replayed as HIP graphs'),
    'SFEM_GRAPH_REUSE': ('1', 'navier_stokes/navier_stokes.py',
                         '0: a recorded iteration is not kept across solves'),
    'SFEM_GRAPH_MAX_NUMEL': (str(1 << 25), 'navier_stokes/navier_stokes.py',
                             'largest vector (entries) whose solves are '
                             'replayed as graphs'),
    'SFEM_PRESSURE_PC': ('projection', 'examples/navier_stokes_driver.py',
                         "pressure preconditioner of the drivers: 'projection' "
                         "(the reference's nullspace projection) or 'schwarz' "
                         '(element-wise fast diagonalisation + piecewise-'
                         'constant coarse level, '
                         'navier_stokes/pressure_preconditioner.py)'),
    'SFEM_VELOCITY_PC': ('exchange', 'navier_stokes/navier_stokes.py',
                         "preconditioner of the stepper's Helmholtz solve: "
                         "'exchange' (the reference's M = QQ^T) or 'mass' "
                         '(inverse assembled lumped mass, scaled so that the '
                         "reference's stopping rule still holds)"),
    'SFEM_PC_FUSED': ('1', 'navier_stokes/pressure_preconditioner.py',
                      "0: the 'schwarz' preconditioner forms the element sums, "
                      'the coarse correction and the mean removal in separate '
                      'passes'),
    'SFEM_PC_COARSE_ITERS': (None, 'navier_stokes/pressure_preconditioner.py',
                             'Chebyshev steps of the coarse solve inside the '
                             "'schwarz' preconditioner (default: from the "
                             'spectrum bounds, ~ sqrt(kappa) ln(200) / 2)'),
    'SFEM_PRESSURE_PROJECTION': ('0', 'navier_stokes/navier_stokes.py',
                                 'number of earlier pressure increments the '
                                 'steppers project the next pressure solve '
                                 'onto (successive right-hand sides; 0 = off: '
                                 "every solve starts from zero as in the "
                                 'reference)'),
    # test-size knobs (tests/ only)
    'SFEM_TEST_FULL_N': ('64', 'tests/', 'elements per direction of the '
                         'full-size property tests'),
    'SFEM_TEST_P11_N': ('64', 'tests/', 'the same for the p = 11 block'),
}

_checked = False


def check_environment() -> list:
  """Names of `SFEM_*` variables in the environment that no code reads
  (warned about once per process)."""
  global _checked
  stray = sorted(k for k in os.environ
                 if k.startswith('SFEM_') and k not in SWITCHES)
  if stray and not _checked:
    warnings.warn('unknown SFEM_* environment variables (ignored): ' +
                  ', '.join(stray) + '; known switches: '
                  'python -m swirl_fem_amd.switches', RuntimeWarning,
                  stacklevel=2)
  _checked = True
  return stray


def get(name: str) -> str | None:
  """Current value of a switch (its default when unset)."""
  if name not in SWITCHES:
    raise KeyError(f'{name} is not a registered switch (swirl_fem_amd/'
                   'switches.py)')
  if not _checked:
    check_environment()
  return os.environ.get(name, SWITCHES[name][0])


def enabled(name: str) -> bool:
  """True unless the switch is '0' (for switches whose default is on) /
  only if it is '1' (default off)."""
  default = SWITCHES[name][0]
  value = get(name)
  return value != '0' if default != '0' else value == '1'


def active() -> dict:
  """Switches set in the environment to something other than their default."""
  return {k: os.environ[k] for k in sorted(SWITCHES)
          if k in os.environ and os.environ[k] != SWITCHES[k][0]}


def markdown_table() -> str:
  rows = ['| switch | default | read by | meaning |', '|---|---|---|---|']
  for k, (d, where, doc) in SWITCHES.items():
    rows.append('| `%s` | %s | %s | %s |' % (
        k, '(unset)' if d is None else '`%s`' % d, where, doc))
  return '\n'.join(rows)


if __name__ == '__main__':
  print(markdown_table())
