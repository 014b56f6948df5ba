#!/usr/bin/env python
"""Prints busy-time statistics from a rocprofv3 rocpd database: per kernel totals inside a time window and
how much of the window is covered by any kernel / by the big GEMM kernel (overlap diagnostics)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else None)
rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
# window: from the first k_ldlt_diag to the last k_back_panel
idx = [i for i, r in enumerate(rows) if 'k_ldlt_diag' in r[0]]
if not idx: sys.exit("no k_ldlt_diag")
# use the LAST factorisation: find the last run of diag kernels (196 of them)
last = idx[-1]; first = idx[-1]
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 194
first = idx[-nblk]
t0, t1 = rows[first][1], max(r[2] for r in rows[first:last + 40] if 'k_gemm' in r[0] or 'k_ldlt' in r[0])
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
print(f"window {(t1 - t0) / 1e6:.3f} ms, {len(win)} dispatches, queues {sorted(set(r[3] for r in win))}")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
print(f"covered by any kernel: {union([(r[1], r[2]) for r in win]) / 1e6:.3f} ms")
big = [(r[1], r[2]) for r in win if 'ILi128ELi128' in r[0]]
print(f"covered by 128x128 GEMM: {union(big) / 1e6:.3f} ms ({len(big)} launches, sum {sum(e - s for s, e in big) / 1e6:.3f} ms)")
small = [(r[1], r[2]) for r in win if 'ILi128ELi128' not in r[0]]
print(f"covered by panel kernels: {union(small) / 1e6:.3f} ms (sum {sum(e - s for s, e in small) / 1e6:.3f} ms)")
by = {}
for r in win:
    k = r[0][:60]; by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += r[2] - r[1]
for k, (n, t) in sorted(by.items(), key=lambda x: -x[1][1]): print(f"  {k:<62} {n:5d} {t / 1e6:9.3f} ms  avg {t / n / 1e3:8.2f} us")
