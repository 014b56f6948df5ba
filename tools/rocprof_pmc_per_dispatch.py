#!/usr/bin/env python
"""PMC counters of a rocprofv3 rocpd database PER DISPATCH of the kernels whose name contains <pattern>, in dispatch order, with
the grid size and the duration of each dispatch (the three k_ldlt_tail launches of a factorisation -- super-panel, super-panel,
final -- differ only in those).  FETCH_SIZE / WRITE_SIZE are reported by the counter in KiB.
  python tools/rocprof_pmc_per_dispatch.py <rocpd .db> <pattern>[,<pattern>...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pats = sys.argv[2].split(",")
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
ev = "event_id" if "event_id" in dcols else "id"
grid = "d.grid_size_x" if "grid_size_x" in dcols else ("d.grid_size" if "grid_size" in dcols else "0")
wg = "d.workgroup_size_x" if "workgroup_size_x" in dcols else "1"
rows = cur.execute(
    f"select d.{ev}, s.{name_col}, d.start, d.end, {grid}, {wg}, i.name, sum(p.value) from rocpd_pmc_event p "
    f"join rocpd_kernel_dispatch d on p.event_id = d.{ev} join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
    f"join rocpd_info_pmc i on p.pmc_id = i.id group by d.{ev}, i.name order by d.start").fetchall()
seen = {}
order = []
for evid, name, st, en, g, w, cname, val in rows:
    if not any(p in name for p in pats):
        continue
    if evid not in seen:
        seen[evid] = {"name": name.split("(")[0][:44], "ms": (en - st) / 1e6, "wgs": (g // w) if w else g, "c": {}}
        order.append(evid)
    seen[evid]["c"][cname] = val
names = sorted({c for e in seen.values() for c in e["c"]})
print(f"{'#':>3} {'kernel':<44} {'workgroups':>10} {'ms':>8} " + " ".join(f"{c:>22}" for c in names))
for k, evid in enumerate(order):
    e = seen[evid]
    print(f"{k:>3} {e['name']:<44} {e['wgs']:>10} {e['ms']:>8.3f} " + " ".join(f"{e['c'].get(c, float('nan')):>22.1f}" for c in names))
