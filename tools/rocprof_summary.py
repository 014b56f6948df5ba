#!/usr/bin/env python
"""Summarises a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table
(count, total / average / min / max duration), like `rocprofv3 --stats` prints.
  rocprof_summary.py <db> [out] [bygrid]     bygrid: one row per (kernel, grid size) -- a command that runs several problem sizes
                                             (bench.py's default line: the headline configuration and two side legs) launches the same
                                             kernel with different grids; their durations must not be averaged together"""
import sqlite3
import sys


def main(path, out=None, bygrid=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[1])
    gcol = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in cols), None) if bygrid else None
    name_expr = f"s.{name_col}" + (f" || ' [grid ' || d.{gcol} || ']'" if gcol else "")
    q = (f"select {name_expr}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by 1 order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for name, n, tot, mn, mx in rows:
        tail = name[name.rfind(" [grid "):] if " [grid " in name else ""
        base = name[:len(name) - len(tail)]
        nm = (base if len(base) + len(tail) <= 90 else base[:87 - len(tail)] + "...") + tail
        lines.append(f"{nm:<90} {n:>7} {tot / 1e6:>10.3f} {tot / n / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100.0 * tot / total:>6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "bygrid" else None, "bygrid" in sys.argv[2:])
