"""Per-kernel statistics out of a rocprofv3 result database (rocpd / SQLite, the default output format of
ROCm 7): name, grid, calls, average / min / max duration in us.  Usage: python tools/rocpd_stats.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in cols else "display_name"
rows = c.execute(f"""select s.{name_col}, d.grid_size_x, d.workgroup_size_x, count(*), avg(d.end - d.start), min(d.end - d.start),
                     max(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id
                     group by s.{name_col}, d.grid_size_x order by sum(d.end - d.start) desc""").fetchall()
print(f"{'calls':>6} {'avg us':>9} {'min us':>9} {'max us':>9} {'WGs':>6}  kernel")
for name, grid, wg, n, avg, lo, hi in rows:
    print(f"{n:6d} {avg / 1e3:9.2f} {lo / 1e3:9.2f} {hi / 1e3:9.2f} {grid // max(wg, 1):6d}  {name[:150]}")
