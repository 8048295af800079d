#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / mean / min / max (microseconds)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("""select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3,
                      min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                      group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':80s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[0][:80]:80s} {r[1]:6d} {r[2]:12.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100*r[2]/tot:6.1f}")
