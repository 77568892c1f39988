"""Copies the judged summaries of one GPU round from gpurun_out/ to profiles/.

  python scripts/collect_profiles.py r01e

* kernel-trace --stats summaries of the default and stored-factor bench runs,
* bench JSON lines,
* HBM traffic per launch of the dominant kernel from the FETCH_SIZE /
  WRITE_SIZE passes: read bytes = 2 * FETCH_SIZE * 1024 (gfx950 tallies 128-B
  requests at 64 B, MI355X_MICROARCH.md), write bytes = WRITE_SIZE * 1024.
"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, 'profiles'); src = os.path.join(ROOT, 'gpurun_out')
rnd = tag[:3]


def one(pattern):
  hits = glob.glob(os.path.join(src, pattern), recursive=True)
  if not hits:
    raise SystemExit(f'missing {pattern}')
  return hits[0]


shutil.copy(one(f'prof_{tag}/**/*_kernel_stats.csv'), os.path.join(out, f'{rnd}_bench_auto_kernel_stats.csv'))
shutil.copy(one(f'prof_{tag}_stored/**/*_kernel_stats.csv'), os.path.join(out, f'{rnd}_bench_stored_factors_kernel_stats.csv'))
shutil.copy(os.path.join(src, f'bench_{tag}.json'), os.path.join(out, f'{rnd}_bench_auto.json'))
shutil.copy(os.path.join(src, f'bench_{tag}_stored.json'), os.path.join(out, f'{rnd}_bench_stored_factors.json'))


def counter(mode, which):
  # GM template argument: 1 = affine (default mesh), 0 = stored factors
  # helmholtz_kernel<T, P, DIM, GS, SCALAR, GM, SORTED, MASS>
  suffix = 'true, true, 1,' if mode == 'affine' else 'true, true, 0,'
  f = one(f'pmc_{tag}_{mode}_{which}/**/*_counter_collection.csv')
  vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
          if 'helmholtz_kernel' in r['Kernel_Name'] and suffix in r['Kernel_Name']]
  return sum(vals) / len(vals), len(vals)


traffic = {'_comment': 'HBM bytes per launch of sfem::helmholtz_kernel<double, 8, 3, true, true, GM>, '
                       'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py '
                       '(read = 2 * FETCH_SIZE KiB, write = WRITE_SIZE KiB); round ' + tag}
for mode, key in (('affine', 'n64_p7_f64_auto'), ('stored', 'n64_p7_f64_stored')):
  fetch, n1 = counter(mode, 'fetch'); write, n2 = counter(mode, 'write')
  rd, wr = 2 * fetch * 1024, write * 1024
  traffic[key] = {'bytes': round(rd + wr), 'FETCH_SIZE_KiB': fetch, 'WRITE_SIZE_KiB': write,
                  'read_bytes': round(rd), 'write_bytes': round(wr), 'launches_averaged': [n1, n2]}
  print(key, f'read {rd/1e9:.3f} GB write {wr/1e9:.3f} GB total {(rd+wr)/1e9:.3f} GB')
json.dump(traffic, open(os.path.join(out, f'traffic_{rnd}.json'), 'w'), indent=1)
for name in ('auto', 'stored_factors'):
  rows = list(csv.DictReader(open(os.path.join(out, f'{rnd}_bench_{name}_kernel_stats.csv'))))
  for r in rows[:6]:
    print(name, r['Name'][:60].ljust(60), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
