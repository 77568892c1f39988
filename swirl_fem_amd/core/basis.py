"""Single-interpolator helpers used by `BarycentricInterpolator.interpolate*`."""

from __future__ import annotations

import numpy as np
import torch

from swirl_fem_amd import _ops
from swirl_fem_amd.core import interpolation


def _mats(interp, like):
  i1, _ = interpolation.matrices_1d(interp.gridpoints_1d, interp.evalpoints_1d)
  g1 = interp._interp_grad_matrix_1d()
  return tuple(torch.as_tensor(np.array(m), dtype=like.dtype,
                               device=like.device) for m in (i1, g1))


def interp(interpolator, u3):
  """u3 (E, n, nc) -> (E, Q, nc)."""
  i1, g1 = _mats(interpolator, u3)
  from swirl_fem_amd.core import autodiff
  ev = autodiff.basis_eval if autodiff.needs_grad(u3) else _ops.basis_eval
  val, _ = ev(
      u3, i1, g1, None, interpolator.ndim,
      interpolator.gridpoints_1d.num_points,
      interpolator.evalpoints_1d.num_points, False, True, False)
  return val


def ref_grad(interpolator, u3):
  """u3 (E, n, nc) -> reference-space gradient (E, Q, d, nc)."""
  i1, g1 = _mats(interpolator, u3)
  _, g = _ops.basis_eval(
      u3, i1, g1, None, interpolator.ndim,
      interpolator.gridpoints_1d.num_points,
      interpolator.evalpoints_1d.num_points, False, False, True)
  return g
