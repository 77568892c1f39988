"""Setup products for direct-stiffness summation on the device.

`AssemblyPlan` classifies the nodes of a mesh by multiplicity (how many element
slots reference them) and, on demand, builds the inverse map (CSR by node) used
by the deterministic assembly `sfem_scatter_csr`.  These are one-off setup
computations (torch index ops on the device); the reference has no counterpart
because XLA's scatter-add hides them (core/gather_scatter.py:130-133).
"""

from __future__ import annotations

import dataclasses

import torch


@dataclasses.dataclass(eq=False)
class AssemblyPlan:
  multiplicity: torch.Tensor        # (N,) int32
  zero_range: tuple                 # [lo, hi) covering every node with mult != 1
  num_shared: int
  _mesh: object = None
  _csr: tuple | None = None

  @classmethod
  def build(cls, mesh) -> 'AssemblyPlan':
    el = mesh.elements.reshape(-1)
    valid = el[el >= 0].to(torch.int64)
    mult = torch.bincount(valid, minlength=mesh.num_nodes).to(torch.int32)
    other = torch.nonzero(mult != 1).reshape(-1)
    if other.numel():
      rng = (int(other.min()), int(other.max()) + 1)
    else:
      rng = (0, 0)
    num_shared = int((mult > 1).sum())
    return cls(multiplicity=mult, zero_range=rng, num_shared=num_shared,
               _mesh=mesh)

  def coloring(self, seed: int = 0):
    """Conflict-free colour classes of the elements (no two elements of one
    class share a node) and, per element slot, whether it is the FIRST toucher
    of its node in colour order.

    Greedy maximal-independent-set colouring with random priorities
    (Luby / Jones-Plassmann): a class is grown by repeatedly adding the
    remaining elements that hold the largest priority at every one of their
    corner nodes and are not adjacent to the class so far.  In a conforming
    tensor-product mesh two elements that share any node share a corner node,
    so corners decide adjacency; the result is verified on all shared slots.

    Returns (colors (E,) int32, num_colors, first (E, n) bool).
    """
    if '_coloring_cache' in self.__dict__:
      return self.__dict__['_coloring_cache']
    mesh = self._mesh
    el = mesh.elements.to(torch.int64)
    E, n = el.shape
    d = mesh.ndim
    P = mesh.gridpoints_1d.num_points
    N = mesh.num_nodes
    # corner slots of the lexicographic node lattice
    idx = torch.arange(n, device=el.device).reshape([P] * d)
    corner = idx[tuple([[0, P - 1]] * 1)] if d == 1 else idx
    for ax in range(d):
      corner = corner.index_select(ax, torch.tensor([0, P - 1],
                                                    device=el.device))
    cn = el[:, corner.reshape(-1)]                       # (E, 2^d)
    valid = cn >= 0
    cn = torch.where(valid, cn, torch.zeros_like(cn))
    gen = torch.Generator(device=el.device).manual_seed(seed)
    prio = (torch.randperm(E, device=el.device, generator=gen) + 1).to(
        torch.int64)
    colors = torch.full((E,), -1, dtype=torch.int32, device=el.device)
    num_colors = 0
    real = valid.any(dim=1)                              # not a padding element
    colors[~real] = 0
    while bool((colors < 0).any()):
      cand = colors < 0
      used = torch.zeros(N, dtype=torch.bool, device=el.device)
      while bool(cand.any()):
        p = torch.where(cand, prio, torch.zeros_like(prio))
        node_max = torch.zeros(N, dtype=torch.int64, device=el.device)
        node_max.scatter_reduce_(0, cn.reshape(-1),
                                 p[:, None].expand_as(cn).reshape(-1), 'amax')
        sel = cand & ((node_max[cn] == p[:, None]) | ~valid).all(dim=1)
        colors[sel] = num_colors
        used[cn[sel].reshape(-1)] = True
        cand = cand & ~sel & ~(used[cn] & valid).any(dim=1)
      num_colors += 1
    num_colors = max(num_colors, 1)
    # first toucher of every node = the slot whose element has the least colour
    slot_color = colors.to(torch.int64)[:, None].expand(E, n)
    ok = el >= 0
    node_min = torch.full((N,), num_colors, dtype=torch.int64,
                          device=el.device)
    node_min.scatter_reduce_(0, el[ok], slot_color[ok], 'amin')
    first = ok & (node_min[torch.where(ok, el, torch.zeros_like(el))] ==
                  slot_color)
    # verification on every slot: a node is touched at most once per colour
    key = el[ok] * num_colors + slot_color[ok]
    if key.numel() and int(torch.bincount(key).max()) > 1:
      raise RuntimeError('element colouring has a node conflict '
                         '(non-conforming mesh?)')
    self.__dict__['_coloring_cache'] = (colors, num_colors, first)
    return self.__dict__['_coloring_cache']

  def csr(self):
    """(offsets (N+1,) int64, slots (nnz,) int32), slots ascending per node."""
    if self._csr is None:
      el = self._mesh.elements.reshape(-1).to(torch.int64)
      n = self._mesh.num_nodes
      key = torch.where(el >= 0, el, torch.full_like(el, n))
      order = torch.argsort(key, stable=True)
      counts = torch.bincount(key, minlength=n + 1)[:n]
      offsets = torch.zeros(n + 1, dtype=torch.int64, device=el.device)
      offsets[1:] = torch.cumsum(counts, 0)
      slots = order[: int(offsets[-1])].to(torch.int32).contiguous()
      self._csr = (offsets.contiguous(), slots)
    return self._csr
