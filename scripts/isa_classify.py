"""Instruction classes of the headline kernels (static ISA of gfx950).

Compiles the p = 7 fp64 instantiations of the index-row kernel (round 2) and
of the facet / chain kernels (round 3) with -save-temps and sorts every
instruction of each kernel body into: fp64 arithmetic, integer / address
arithmetic, flag tests and selects, register moves, cross-lane, LDS, vector
memory, scalar ALU, scalar control flow, scalar memory, waits.  Writes
profiles/r03_isa_classes.md.  (Static counts: the index-row kernel repeats its
sorted shared scatter about five times per element, the chain kernels run
their loop body once per element; the per-element DYNAMIC counts are the
SQ_INSTS_* counters in profiles/traffic_r03.json.)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "sfem_helmholtz_facet.h"
namespace sfem {
template __global__ void helmholtz_kernel<double, 8, 3, true, true, GEO_AFFINE, true, false>(HelmholtzParams<double>, DMat<double, 8>);
template __global__ void helmholtz_facet_kernel<double, 8, FacetElem<double, 8, GEO_AFFINE, false>, true, true>(FacetParams<double>, DMat<double, 8>, FacetSlots);
template __global__ void helmholtz_facet_kernel<double, 8, BoxElem<double, 8, false>, true, true>(FacetParams<double>, SMat<double, 8>, FacetSlots);
template __global__ void helmholtz_chain_kernel<double, 8, FacetElem<double, 8, GEO_AFFINE, false>, true>(FacetParams<double>, DMat<double, 8>, FacetSlots);
template __global__ void helmholtz_chain_kernel<double, 8, BoxElem<double, 8, false>, true>(FacetParams<double>, SMat<double, 8>, FacetSlots);
}
'''
NAMES = ['index rows, affine (round 2 headline)', 'facet table, affine',
         'facet table, box', 'chain, affine', 'chain, box (round 3 headline)']


def classify(op):
  if op.startswith('s_waitcnt') or op == 's_nop':
    return 'waits / nops'
  if op.startswith('s_load') or op.startswith('s_buffer_load'):
    return 'scalar memory'
  if op.startswith('s_cbranch') or op.startswith('s_branch') or 'saveexec' in op or op in (
      's_endpgm', 's_barrier', 's_setpc_b64'):
    return 'scalar control flow'
  if op.startswith('s_'):
    return 'scalar ALU'
  if op.startswith('ds_bpermute') or op.startswith('ds_swizzle') or op.startswith('ds_permute'):
    return 'cross-lane'
  if op.startswith('ds_'):
    return 'LDS'
  if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_'):
    return 'vector memory'
  if op.startswith('scratch_'):
    return 'scratch (spills)'
  if 'f64' in op or 'f32' in op:
    return 'floating-point arithmetic'
  if op.startswith('v_cmp') or op.startswith('v_cndmask') or op.startswith('v_and') or op.startswith(
      'v_or') or op.startswith('v_not') or op.startswith('v_xor') or op.startswith('v_bfe') or op.startswith(
      'v_bitop') or op.startswith('v_bfi'):
    return 'flag tests / selects'
  if op.startswith('v_mov') or op.startswith('v_accvgpr'):
    return 'register moves'
  if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane') or \
      op.startswith('v_mbcnt') or 'dpp' in op:
    return 'cross-lane'
  if op.startswith('v_'):
    return 'integer / address arithmetic'
  return 'other'


def main():
  with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, 'k.hip'), 'w').write(SRC)
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-ffp-contract=fast', '-Wno-array-bounds', '-I' + os.path.join(ROOT, 'include'),
                    '-I' + os.path.join(ROOT, 'swirl_fem_amd', 'csrc'), '-c', 'k.hip', '-o', 'k.o',
                    '-save-temps'], cwd=tmp, check=True, capture_output=True)
    asm = open(os.path.join(tmp, 'k-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
  funcs = re.split(r'\n\t\.globl\t', asm)[1:1 + len(NAMES)]
  table, meta = [], []
  for f in funcs:
    body = f.split('s_endpgm')[0]
    c = collections.Counter()
    for line in body.split('\n'):
      m = re.match(r'\t([a-z_0-9]+)', line)
      if m:
        c[classify(m.group(1))] += 1
    table.append(c)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', f) or re.search(r'; NumVgprs: (\d+)', f)
    sc = re.search(r'; ScratchSize: (\d+)', f)
    meta.append((vg.group(1) if vg else '?', sc.group(1) if sc else '?'))
  classes = ['floating-point arithmetic', 'integer / address arithmetic', 'flag tests / selects',
             'register moves', 'cross-lane', 'LDS', 'vector memory', 'scratch (spills)', 'scalar ALU',
             'scalar control flow', 'scalar memory', 'waits / nops']
  out = ['# Instruction classes of the p = 7 fp64 apply kernels (static gfx950 ISA)', '',
         '`python scripts/isa_classify.py` (hipcc -O3 -save-temps of the headline instantiations, '
         'one column per kernel).', '',
         '| class | ' + ' | '.join(NAMES) + ' |', '|---|' + '---:|' * len(NAMES)]
  for cl in classes:
    out.append(f'| {cl} | ' + ' | '.join(str(t[cl]) for t in table) + ' |')
  valu = ['floating-point arithmetic', 'integer / address arithmetic', 'flag tests / selects', 'register moves']
  out.append('| **vector ALU (first four rows)** | ' + ' | '.join(
      '**%d**' % sum(t[c] for c in valu) for t in table) + ' |')
  out.append('| VGPRs / scratch bytes | ' + ' | '.join(f'{v} / {s}' for v, s in meta) + ' |')
  out += ['', 'Static counts: the index-row kernel runs its sorted shared scatter (a loop of about 40 '
          'instructions) five times per element and the chain kernels execute their loop body once per '
          'element after a prologue of about 120 instructions; the dynamic per-element counts are the '
          '`SQ_INSTS_*` counters in `profiles/traffic_r03.json` (index rows: 836 VALU / 134 SALU per '
          'element in `gpurun_out/pmc_facet`, round 2: 946 / 143).', '']
  open(os.path.join(ROOT, 'profiles', 'r03_isa_classes.md'), 'w').write('\n'.join(out))
  print('\n'.join(out))


if __name__ == '__main__':
  main()
