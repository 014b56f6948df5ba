#!/usr/bin/env python
"""Per-queue busy/idle breakdown of the last LDL^T factorisation in a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
rows = cur.execute(f"select s.{name_col}, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_ldlt_diag' in r[0]]
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 194
first = idx[-nblk]
t0 = rows[first][1]
t1 = max(r[2] for r in rows[first:] if 'k_gemm' in r[0] or 'k_ldlt' in r[0])
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
print(f"window {(t1 - t0) / 1e6:.3f} ms")
for q in sorted(set(r[3] for r in win)):
    qr = [r for r in win if r[3] == q]
    busy = sum(r[2] - r[1] for r in qr)
    by = {}
    for r in qr:
        k = r[0][10:52]; by.setdefault(k, [0, 0]); by[k][0] += 1; by[k][1] += r[2] - r[1]
    print(f"queue {q}: {len(qr)} kernels, busy {busy / 1e6:.3f} ms")
    for k, (n, t) in sorted(by.items(), key=lambda x: -x[1][1]): print(f"     {k:<44} {n:5d} {t / 1e6:8.3f} ms avg {t / n / 1e3:7.2f} us")
# the chain: print a sample of consecutive chain-queue kernels with gaps (middle of the factorisation)
cq = rows[first][3]
ch = [r for r in win if r[3] == cq]
mid = len(ch) // 2
print("chain sample (name, dur us, gap to previous end us):")
for i in range(mid, mid + 14):
    print(f"   {ch[i][0][10:50]:<42} {(ch[i][2] - ch[i][1]) / 1e3:7.2f} {(ch[i][1] - ch[i - 1][2]) / 1e3:8.2f}")
