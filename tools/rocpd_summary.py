#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the per-kernel summary text kept
under profiles/.  Usage: python tools/rocpd_summary.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} "
      f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7}")
for n, c, tot, avg, mn, mx, vg, ag, sg, lds, scr in rows:
    print(f"{n[:100]:<100} {c:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} "
          f"{100 * tot / total:>6.2f} {vg:>5} {ag:>5} {sg:>5} {lds:>7} {scr:>7}")
