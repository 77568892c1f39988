"""The drop-in boundary is a C ABI: a plain-C program (gcc, no Python, no
torch) includes include/sfem.h, links libsfem_hip.so and drives the kernels
with raw device pointers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'c_abi', 'abi_check.c')
LIBDIR = os.path.join(ROOT, 'swirl_fem_amd')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(out):
  if shutil.which('gcc') is None:
    pytest.skip('gcc not available')
  if not os.path.exists(os.path.join(LIBDIR, 'libsfem_hip.so')):
    pytest.fail('libsfem_hip.so missing: run __graft_entry__.build()')
  cmd = ['gcc', '-std=c11', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__',
         f'-I{ROCM}/include', f'-I{ROOT}/include', SRC, '-o', out,
         f'-L{LIBDIR}', '-lsfem_hip', f'-L{ROCM}/lib', '-lamdhip64', '-lm',
         f'-Wl,-rpath,{LIBDIR}', f'-Wl,-rpath,{ROCM}/lib']
  r = subprocess.run(cmd, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-3000:]
  return out


def test_header_is_plain_c_and_client_links(tmp_path):
  """include/sfem.h compiles as C11 with -Wall -Werror and every symbol the
  client uses resolves against the shared library (no GPU needed)."""
  exe = _build(str(tmp_path / 'abi_check'))
  assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_plain_c_client_runs(tmp_path):
  exe = _build(str(tmp_path / 'abi_check'))
  r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
  assert 'c-abi OK' in r.stdout
