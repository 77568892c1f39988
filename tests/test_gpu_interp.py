"""The compile-time-sized tensor interpolation (csrc/sfem_interp.h) behind the
values-only cases of `sfem_basis_eval` / `sfem_basis_eval_t` (reference:
core/interpolation.py:260-263 and its transpose in `local_covector`,
core/fespace.py:405-471) against the generic kernels and a float64 einsum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _einsum(mat, x, d):
  """(M (x) .. (x) M) x over the element axis: x (E, n^d, nc) -> (E, m^d, nc)."""
  E, _, nc = x.shape
  n = mat.shape[1]
  t = x.reshape((E,) + (n,) * d + (nc,)).double()
  m = mat.double()
  if d == 2:
    return torch.einsum('ai,bj,eijk->eabk', m, m, t).reshape(E, -1, nc)
  return torch.einsum('ai,bj,cl,eijlk->eabck', m, m, m, t).reshape(E, -1, nc)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('d,P,q,nc', [(3, 8, 10, 3), (3, 8, 7, 3), (3, 7, 8, 3),
                                      (2, 9, 11, 2), (2, 6, 5, 2), (3, 4, 6, 1),
                                      (3, 12, 14, 3), (2, 14, 12, 8),
                                      (3, 5, 5, 2), (3, 6, 9, 1)])
def test_values_only_basis_kernels(d, P, q, nc, dtype, monkeypatch):
  from swirl_fem_amd import _ops
  g = torch.Generator(device=DEV).manual_seed(d * 1000 + P * 10 + q)
  E = 37
  rnd = lambda *s: torch.randn(*s, dtype=dtype, device=DEV, generator=g)
  i1, g1 = rnd(q, P), rnd(q, P)
  u = rnd(E, P ** d, nc)
  c0 = rnd(E, q ** d, nc)
  wdet = rnd(E, q ** d).abs() + 0.5
  fwd = lambda: _ops.basis_eval(u, i1, g1, None, d, P, q, False, True, False)[0]
  bwd = lambda: _ops.basis_eval_t(c0, None, i1, g1, None, wdet, d, P, q, nc,
                                  False)
  monkeypatch.setenv('SFEM_INTERP', '0')
  fwd_generic, bwd_generic = fwd(), bwd()
  monkeypatch.setenv('SFEM_INTERP', '1')
  fwd_fast, bwd_fast = fwd(), bwd()
  # same contraction order, same summation order: the same bits (sizes outside
  # the instantiated range, here q - P = 3, run the generic kernel either way)
  assert torch.equal(fwd_fast, fwd_generic)
  assert torch.equal(bwd_fast, bwd_generic)
  tol = 1e-12 if dtype == torch.float64 else 2e-5
  want = _einsum(i1, u, d)
  assert float((fwd_fast.double() - want).abs().max()) <= tol * float(
      want.abs().max())
  want_t = _einsum(i1.t(), wdet[:, :, None] * c0, d)
  assert float((bwd_fast.double() - want_t).abs().max()) <= tol * float(
      want_t.abs().max())
