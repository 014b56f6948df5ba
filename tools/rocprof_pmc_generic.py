import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
ev = "event_id" if "event_id" in dcols else "id"
rows = cur.execute(f"select s.{name_col}, i.name, count(*), sum(p.value) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.{ev} "
                   f"join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc i on p.pmc_id = i.id group by s.{name_col}, i.name").fetchall()
pat = sys.argv[2].split(',')
for name, c, n, tot in rows:
    if any(p in name for p in pat): print(f"{name[10:48]:<40} {c:<28} calls {n:>5} per-call {tot / n:>16.1f}")
