"""Timeline of the last host-buffer call in a rocprofv3 database (--hip-trace --kernel-trace --memory-copy-trace):
python tools/hp_timeline.py results.db [window_ms] [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
window = float(sys.argv[2]) if len(sys.argv) > 2 else 4.5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ev = []
for n, s, e in db.execute("select name,start,end from regions"):
    ev.append((s, e, "api", n))
for n, s, e, st in db.execute("select name,start,end,stream_id from kernels"):
    ev.append((s, e, "kern", f"[s{st}] {n[:70]}"))
for n, s, e, sz in db.execute("select name,start,end,size from memory_copies"):
    ev.append((s, e, "copy", f"{n} {sz}"))
ev.sort()
last = [e for e in ev if e[2] != "api"][-1][1]
t0 = last - int(window * 1e6)
for s, e, k, n in ev:
    if s >= t0 and (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {k:5s} {n}")
