"""Per-kernel totals from a rocprofv3 results .db (rocpd output): name, launches, total ms, mean us.
  python scripts/kernel_stats_db.py gpurun_out/prof/x_results.db [top]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = c.execute('select sum(end-start)/1e6, count(*) from kernels').fetchone()
print(f'total {tot[0]:.2f} ms in {tot[1]} launches')
for name, n, ms, us in c.execute('select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels '
                                 'group by name order by 3 desc limit ?', (top,)):
  print(f'{ms:9.2f} ms {n:7d} x {us:8.1f} us  {name[:120]}')
