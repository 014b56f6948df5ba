#!/usr/bin/env python
"""Timeline of the LAST reduced-system factorisation in a rocprofv3 --kernel-trace database (tools/bin/bench_tail or bench.py):
every dispatch of the `span_ms` before the end of the last k_ldlt_tail launch, all streams, with start offset and duration.
  python tools/factor_timeline.py <rocpd .db> [span_ms] [min_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
span = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else "0")
rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
tails = [r for r in rows if "k_ldlt_tail" in r[0]]
if not tails: sys.exit("no k_ldlt_tail")
t1 = tails[-1][2]; t0 = t1 - span * 1e6
win = [r for r in rows if r[2] > t0 and r[1] <= t1]
base = win[0][1]
print("  start ms    dur ms  queue  kernel")
for name, st, en, q in win:
    if (en - st) / 1e3 >= min_us:
        short = name.split("(")[0]
        short = short.replace("void cba::", "").replace("cba::", "")[:64]
        print("%9.3f %9.3f  %5s  %s" % ((st - base) / 1e6, (en - st) / 1e6, q, short))
