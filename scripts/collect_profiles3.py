"""Copies the judged summaries of scripts/gpu_round3.sh from gpurun_out/ to
profiles/ and derives HBM traffic per launch of the dominant kernels.

read bytes = 2 * FETCH_SIZE KiB (gfx950 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"), write + atomic bytes = WRITE_SIZE KiB.  Every
entry carries the sha256 of the kernel sources it was measured on
(bench.kernel_source_hash): bench.py reports `traffic` only while it matches."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
rnd = tag[:3]
out = os.path.join(ROOT, 'profiles'); src = os.path.join(ROOT, 'gpurun_out')
try:      # written on the GPU box by gpu_round3.sh: the sources that were measured
  src_hash = open(os.path.join(src, f'src_hash_{tag}.txt')).read().strip()
except OSError:
  src_hash = bench.kernel_source_hash()


def one(pattern):
  hits = glob.glob(os.path.join(src, pattern), recursive=True)
  return hits[0] if hits else None


def stats(name, match):
  """{kernel: (avg ns, calls)} of the kernels whose name contains `match`."""
  f = one(f'prof_{tag}_{name}/**/*_kernel_stats.csv')
  if f is None:
    return {}
  shutil.copy(f, os.path.join(out, f'{rnd}_bench_{name}_kernel_stats.csv'))
  rows = [r for r in csv.DictReader(open(f)) if match(r['Name'])]
  rows.sort(key=lambda r: -float(r['TotalDurationNs']))
  return {r['Name']: (float(r['AverageNs']), int(r['Calls'])) for r in rows}


def counter(name, key, kernel):
  f = one(f'pmc_{tag}_{name}_{key}/**/*_counter_collection.csv')
  if f is None:
    return {}
  acc = {}
  for r in csv.DictReader(open(f)):
    if r['Kernel_Name'] == kernel:
      acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
  return {k: sum(v) / len(v) for k, v in acc.items()}


def last_json(path):
  try:
    lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
    return json.loads(lines[-1]) if lines else None
  except OSError:
    return None


helm = lambda k: 'helmholtz' in k and 'setup' not in k
keys = {'auto': 'n64_p7_f64_auto', 'affine': 'n64_p7_f64_auto_affine_kernel',
        'stored': 'n64_p7_f64_stored', 'jitter': 'n64_p7_f64_auto_jitter',
        'p11': 'n64_p11_f32_auto'}
tpath = os.path.join(out, f'traffic_{rnd}.json')
try:
  traffic = json.load(open(tpath))
except (OSError, ValueError):
  traffic = {}
traffic['_comment'] = (
    'HBM bytes per launch of the dominant kernels: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate '
    'passes (read = 2 * FETCH_SIZE KiB, write + atomic = WRITE_SIZE KiB); scripts/gpu_round3.sh ' + tag +
    '; src_hash = sha256 of the kernel sources (bench.kernel_source_hash)')
for name, key in keys.items():
  st = stats(name, helm)
  if not st:
    continue
  kernel = next(iter(st))
  avg_ns, calls = st[kernel]
  c = {}
  for k in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_EA0_ATOMIC_sum', 'sq1', 'sq2'):
    c.update(counter(name, k, kernel))
  line = last_json(os.path.join(src, f'prof_{tag}_{name}.log'))
  entry = {'kernel': kernel, 'kernel_avg_ms_rocprof_stats': avg_ns / 1e6, 'calls': calls, 'src_hash': src_hash}
  if line:
    entry['bytes_must_move'] = line['roofline']['bytes_per_launch']
    entry['bench_kernel_ms_incl_zero_fill'] = line['roofline']['kernel_ms']
  if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    rd, wr = 2 * c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024
    entry.update({'bytes': round(rd + wr), 'read_bytes': round(rd), 'write_bytes': round(wr)})
  entry.update({k + ('_KiB' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''): v for k, v in c.items()})
  traffic[key] = entry
  print(key, kernel[:60], f'{avg_ns/1e6:.4f} ms', {k: f'{v:.4g}' for k, v in c.items()})
# the fused Stokes kernels
st = stats('stokes', lambda k: 'stokes' in k and 'setup' not in k)
line = last_json(os.path.join(src, f'prof_{tag}_stokes.log'))
for kernel, (avg_ns, calls) in st.items():
  which = 'stokes_grad_t' if 'grad_t' in kernel else ('stokes_div' if 'div' in kernel else None)
  if which is None:
    continue
  c = {}
  for k in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_EA0_ATOMIC_sum'):
    c.update(counter('stokes', k, kernel))
  entry = {'kernel': kernel, 'kernel_avg_ms_rocprof_stats': avg_ns / 1e6, 'calls': calls, 'src_hash': src_hash}
  if line:
    must = line[which]['bytes_must_move']
    if 'chain' in kernel and 'connectivity' not in line:
      # (a log from before prof_stokes.py charged the facet kernels their own
      # connectivity: 436 B per element instead of the index rows)
      E_, n_ = line['n'] ** 3, line['P'] ** 3
      must -= (4 * n_ - 436) * E_ + (2 * 296 * E_ if which == 'stokes_grad_t' else 0)
    entry['bytes_must_move'] = must
    entry['frac_of_8TBs_on_must_move'] = must / (avg_ns * 1e-9) / 8e12
  if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    rd, wr = 2 * c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024
    entry.update({'bytes': round(rd + wr), 'read_bytes': round(rd), 'write_bytes': round(wr)})
  entry.update({k + ('_KiB' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''): v for k, v in c.items()})
  traffic[f'{which}_n64_p7_f64'] = entry
  print(which, kernel[:60], f'{avg_ns/1e6:.4f} ms', {k: f'{v:.4g}' for k, v in c.items()})
json.dump(traffic, open(tpath, 'w'), indent=1)
for name in ('auto', 'affine', 'jitter', 'stored', 'p11', 'stokes'):
  line = last_json(os.path.join(src, f'prof_{tag}_{name}.log'))
  if line:
    json.dump(line, open(os.path.join(out, f'{rnd}_bench_{name}_under_rocprof.json'), 'w'), indent=1)
