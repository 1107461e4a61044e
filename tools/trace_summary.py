"""Summarises a rocprofv3 kernel trace (the rocpd .db it writes by default): calls, total, average, min, max per kernel."""
import glob, os, sqlite3, sys
path = sys.argv[1]
dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True) if os.path.isdir(path) else [path]
cur = sqlite3.connect(dbs[0]).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    name = r[0].replace("fx::(anonymous namespace)::", "")
    print(f"{name[:64]:64s} n={r[1]:5d} total={r[2]/1e3:9.1f}us avg={r[3]/1e3:7.2f} min={r[4]/1e3:6.2f} max={r[5]/1e3:7.2f}")
