#!/usr/bin/env python
"""Per-panel timeline of the last LDL^T factorisation in a rocprofv3 rocpd database (kernel trace).
For every panel (group of 4 k_ldlt_diag launches): chain span, gap to the next panel's first diagonal
kernel, and what the bulk GEMM was doing meanwhile."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_ldlt_diag' in r[0]]
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 196
per_panel = int(sys.argv[3]) if len(sys.argv) > 3 else 4
diag = [rows[i] for i in idx[-nblk:]]
t0 = diag[0][1]
tend = max(r[2] for r in rows if r[1] >= t0)
win = [r for r in rows if r[1] >= t0]
big = [r for r in win if 'ILi128ELi128' in r[0]]
print(f"factor window {(tend - t0) / 1e6:.3f} ms; {len(diag)} diag blocks")
print("panel  t_start   chain_us  gap_next_us  |  bulk gemm launches overlapping: (start, dur) us rel. to panel start")
for p in range(0, len(diag) // per_panel):
    d = diag[p * per_panel:(p + 1) * per_panel]
    ps, pe = d[0][1], d[-1][2]
    nxt = diag[(p + 1) * per_panel][1] if (p + 1) * per_panel < len(diag) else None
    ov = [(r[1] - ps, r[2] - r[1], r[3]) for r in big if r[2] > ps and r[1] < (nxt or pe)]
    between = [r for r in win if nxt and r[1] >= pe and r[2] <= nxt and r[3] == d[0][3]]
    btxt = " ".join(f"[{r[0][10:34]} {(r[1]-pe)/1e3:.0f}+{(r[2]-r[1])/1e3:.0f}]" for r in between)
    print(f"{p:4d} {(ps - t0) / 1e3:9.1f} {(pe - ps) / 1e3:9.1f} {((nxt - pe) / 1e3) if nxt else 0:9.1f}   | " +
          " ".join(f"q{q}({s / 1e3:.0f},{du / 1e3:.0f})" for s, du, q in ov) + "  || " + btxt)
if len(sys.argv) > 4:
    p = int(sys.argv[4])
    d = diag[p * per_panel:(p + 1) * per_panel]
    ps = d[0][1]
    nxt = diag[(p + 1) * per_panel][1]
    print(f"--- all dispatches in panel {p} (start us, dur us, queue, kernel)")
    for r in win:
        if r[1] >= ps - 5000 and r[1] < nxt:
            print(f"  {(r[1] - ps) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f}  q{r[3]}  {r[0][10:60]}")
