"""fp32 parity at north_star's 1e-5: inputs the float32 kernels can hold.

The fp32 tests compare a float32 kernel with the float64 oracle.  If the
oracle is fed float64 inputs that the kernel only sees rounded, the rounding
of the INPUTS (6e-8 per entry, amplified by the derivative matrices and by
h^-1 in the Jacobians) is charged to the kernel -- that, not the kernels'
arithmetic, is what had pushed the tolerances of rounds 1-3 to 2e-5..5e-5
(`profiles/r04_fp32_errors.md`: with representable inputs every fused kernel
is within 1e-5 wherever the reference algorithm in float32 is).  `F32Rng`
draws float32-representable numbers, `f32_mesh` rounds the refined node
coordinates once, so both sides see the same mesh and the same field.
"""
import numpy as np


def f32r(x):
  """`x` rounded to float32, as float64."""
  return np.asarray(x, dtype=np.float32).astype(np.float64)


def f32_mesh(rp, dtype=None):
  """The refined premesh with float32-representable node coordinates (only
  when `dtype` is torch.float32 or None: the fp64 runs keep exact meshes,
  whose affine / box classification works at 1e-11)."""
  import torch
  if dtype is not None and dtype != torch.float32:
    return rp
  return rp.replace(node_coords=f32r(rp.node_coords))


class F32Rng:
  """`np.random.Generator` whose real-valued draws are float32-representable."""

  def __init__(self, seed):
    self._rng = np.random.default_rng(seed)

  def standard_normal(self, *a, **k):
    return f32r(self._rng.standard_normal(*a, **k))

  def uniform(self, *a, **k):
    return f32r(self._rng.uniform(*a, **k))

  def random(self, *a, **k):
    return f32r(self._rng.random(*a, **k))

  def __getattr__(self, name):       # permutation, integers, choice, ...
    return getattr(self._rng, name)


def tolerance(dtype, P):
  """Relative tolerance of a fused kernel against the fp64 oracle: north_star's
  1e-10 (fp64) / 1e-5 (fp32).  One documented exception, P >= 11 in fp32: 2e-5.
  There the REFERENCE ALGORITHM evaluated in float32 is itself 1.0e-5..1.8e-5
  from the fp64 oracle (profiles/r04_fp32_errors.md, rows `helmholtz | 2 | 12`,
  `helmholtz | 3 | 12`, `stokes | 3 | 12`), the kernels 0.3e-5..1.0e-5 on most
  meshes and 1.8e-5 on one (2D, jittered, stored factors) -- a 12-point
  derivative matrix has entries of 1e2 and sums 12 products per line, 36 per
  point, in single precision."""
  import torch
  if dtype == torch.float64:
    return 1e-10
  return 1e-5 if P <= 10 else 2e-5
