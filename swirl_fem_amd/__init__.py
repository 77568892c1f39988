"""swirl_fem_amd: MI355X-native spectral-element operator engine.

Keeps the `Mesh` / `FiniteElementSpace` / `cg` / `StokesSEM` API of
google-research/swirl-fem; the numerics run in hand-written HIP kernels for
gfx950 behind the C-ABI declared in `include/sfem.h`.
"""

__version__ = '0.1.0'
