#!/usr/bin/env python
"""Timeline of the last LM iteration in a rocprofv3 --kernel-trace database: every kernel dispatch between the last two
back substitutions with start offset, duration and the idle gap in front of it (all streams merged).
  python tools/step_timeline.py <rocpd .db> [min_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
rows = cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_back_dataflow" in r[0] or "k_back_panel_diag" in r[0]]
end = idx[-1]; beg = idx[-2] + 1
t0 = rows[beg][1]; busy_until = t0; idle = 0.0
print("   start ms   dur ms   gap ms  kernel")
for name, st, en in rows[beg:end + 1]:
    gap = max(0, st - busy_until) / 1e6
    idle += gap
    if (en - st) / 1e3 >= min_us or gap * 1e3 >= min_us:
        print("%10.3f %8.3f %8.3f  %s" % ((st - t0) / 1e6, (en - st) / 1e6, gap, name.split("(")[0][:70]))
    busy_until = max(busy_until, en)
print("iteration %.3f ms, device idle %.3f ms" % ((rows[end][2] - t0) / 1e6, idle))
