#!/usr/bin/env python
"""Sums one PMC counter of a rocprofv3 rocpd database per kernel (FETCH_SIZE / WRITE_SIZE are in KiB)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
ev = "event_id" if "event_id" in dcols else "id"
rows = cur.execute(f"select s.{name_col}, count(*), sum(p.value), i.name from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.{ev} "
                   f"join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc i on p.pmc_id = i.id group by s.{name_col}, i.name order by 3 desc").fetchall()
out = [f"{'kernel':<80} {'counter':<12} {'calls':>6} {'sum_KiB':>14} {'per_call_MiB':>13}"]
for name, n, tot, cname in rows[:25]:
    out.append(f"{name[:80]:<80} {cname:<12} {n:>6} {tot:>14.1f} {tot / n / 1024:>13.3f}")
print("\n".join(out))
if len(sys.argv) > 2: open(sys.argv[2], "w").write("\n".join(out) + "\n")
