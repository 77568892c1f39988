"""Copies the judged summaries of scripts/gpu_round2.sh from gpurun_out/ to
profiles/ and derives HBM traffic per launch of the dominant kernel.

read bytes = 2 * FETCH_SIZE KiB (gfx950 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"), write + atomic bytes = WRITE_SIZE KiB."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = tag[:3]
out = os.path.join(ROOT, 'profiles'); src = os.path.join(ROOT, 'gpurun_out')
sha = os.environ.get('SFEM_GIT_SHA', '')
if not sha:
  try:
    sha = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'],
                         capture_output=True, text=True).stdout.strip()
  except OSError:
    sha = ''


def one(pattern):
  hits = glob.glob(os.path.join(src, pattern), recursive=True)
  return hits[0] if hits else None


def dominant(name):
  """(kernel name, avg ns, calls) of the most expensive helmholtz kernel."""
  f = one(f'prof_{tag}_{name}/**/*_kernel_stats.csv')
  if f is None:
    return None
  shutil.copy(f, os.path.join(out, f'{rnd}_bench_{name}_kernel_stats.csv'))
  rows = [r for r in csv.DictReader(open(f)) if 'helmholtz' in r['Name'] and 'setup' not in r['Name']]
  rows.sort(key=lambda r: -float(r['TotalDurationNs']))
  r = rows[0]
  return r['Name'], float(r['AverageNs']), int(r['Calls'])


def counter(name, key, kernel):
  f = one(f'pmc_{tag}_{name}_{key}/**/*_counter_collection.csv')
  if f is None:
    return {}
  acc = {}
  for r in csv.DictReader(open(f)):
    if r['Kernel_Name'] == kernel:
      acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
  return {k: sum(v) / len(v) for k, v in acc.items()}


keys = {'auto': 'n64_p7_f64_auto', 'stored': 'n64_p7_f64_stored',
        'jitter': 'n64_p7_f64_auto_jitter', 'p11': 'n64_p11_f32_auto'}
traffic = {'_comment': 'HBM bytes per launch of the dominant fused-apply kernel: rocprofv3 --pmc FETCH_SIZE / '
                       'WRITE_SIZE in separate passes of bench.py (read = 2 * FETCH_SIZE KiB, write + atomic = '
                       'WRITE_SIZE KiB); scripts/gpu_round2.sh ' + tag}
for name, key in keys.items():
  dom = dominant(name)
  if dom is None:
    continue
  kernel, avg_ns, calls = dom
  c = {}
  for k in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_EA0_ATOMIC_sum'):
    c.update(counter(name, k, kernel))
  if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
    continue
  rd, wr = 2 * c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024
  traffic[key] = {'bytes': round(rd + wr), 'read_bytes': round(rd), 'write_bytes': round(wr),
                  'FETCH_SIZE_KiB': c['FETCH_SIZE'], 'WRITE_SIZE_KiB': c['WRITE_SIZE'],
                  'kernel': kernel, 'kernel_avg_ms_rocprof_stats': avg_ns / 1e6, 'calls': calls,
                  'git_sha': sha,
                  **{k: v for k, v in c.items() if k not in ('FETCH_SIZE', 'WRITE_SIZE')}}
  print(key, kernel[:70], f'{avg_ns/1e6:.4f} ms  read {rd/1e9:.3f} GB  write {wr/1e9:.3f} GB', {k: f'{v:.4g}' for k, v in c.items()})
json.dump(traffic, open(os.path.join(out, f'traffic_{rnd}.json'), 'w'), indent=1)
# issue-side counters of the p = 11 kernel
sq = {}
dom = dominant('p11')
if dom:
  for i in (1, 2, 3):
    sq.update(counter('p11', f'sq{i}', dom[0]))
  json.dump({'kernel': dom[0], 'avg_ms': dom[1] / 1e6, 'counters_per_launch': sq, 'git_sha': sha},
            open(os.path.join(out, f'{rnd}_p11_f32_counters.json'), 'w'), indent=1)
  for k in sorted(sq):
    print('p11 %-32s %.4g' % (k, sq[k]))
