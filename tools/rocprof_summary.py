#!/usr/bin/env python
"""Summarises a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table
(count, total / average / min / max duration), like `rocprofv3 --stats` prints."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         f"group by s.{name_col} order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for name, n, tot, mn, mx in rows:
        nm = name if len(name) <= 90 else name[:87] + "..."
        lines.append(f"{nm:<90} {n:>7} {tot / 1e6:>10.3f} {tot / n / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100.0 * tot / total:>6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
