"""Prints VGPR / scratch / occupancy of the kernels of one .hip file.
usage: python scripts/kernel_resources.py swirl_fem_amd/csrc/sfem_stokes_f64_3d.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
d = os.path.dirname(os.path.abspath(src))
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                    '-ffp-contract=fast', f'-I{d}/../../include', f'-I{d}', '-c', src, '-o', '/tmp/_kr.o',
                    '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
cur = None; rows = {}
for line in r.stderr.splitlines():
  m = re.search(r'remark: +(.*?) \[-Rpass', line)
  if not m: continue
  t = m.group(1)
  if t.startswith('Function Name:'):
    name = t.split(':', 1)[1].strip()
    dem = subprocess.run(['/usr/bin/c++filt', name], capture_output=True, text=True).stdout.strip()
    cur = re.sub(r'\(.*', '', dem).replace('void sfem::', ''); rows[cur] = {}
  elif cur and ':' in t:
    k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
  if flt in k:
    print(f"{k:60s} VGPR {v.get('VGPRs'):>4s} AGPR {v.get('AGPRs','0'):>3s} SGPR {v.get('TotalSGPRs'):>4s} scratch {v.get('ScratchSize [bytes/lane]'):>5s} occ {v.get('Occupancy [waves/SIMD]')} LDS {v.get('LDS Size [bytes/block]')}")
