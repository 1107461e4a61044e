"""Diagnostic: where the one-shot cfg2 solve spends its wall time."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
b = workloads.large_sketch(5000)
ctx.system_solve_batch(b)
for rep in range(2):
    t0 = time.time(); a = abi.normalize_batch(b); t1 = time.time()
    db = ctx.upload(a); ctx.synchronize(); t2 = time.time()
    db.system_solve(); ctx.synchronize(); t3 = time.time()
    v = db.get_vars(); r = db.get_results(); t4 = time.time()
    db.free(); t5 = time.time()
    print(f"normalize {1e3*(t1-t0):.1f} ms, upload {1e3*(t2-t1):.1f} ms, solve {1e3*(t3-t2):.1f} ms, download {1e3*(t4-t3):.1f} ms, free {1e3*(t5-t4):.1f} ms")
