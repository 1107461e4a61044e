"""Diagnostic: cost of the host-buffer entry point (upload + solve + download) against the resident solve."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = fiksi_amd.Context(0)
b = abi.normalize_batch(workloads.ring16(n))
for rep in range(3):
    t0 = time.time(); db = ctx.upload(b); ctx.synchronize(); t1 = time.time()
    db.system_solve(); ctx.synchronize(); t2 = time.time()
    v = db.get_vars(); r = db.get_results(); t3 = time.time()
    db.free()
    t4 = time.time(); v2, r2 = ctx.system_solve_batch(b); t5 = time.time()
    print(f"n={n}: upload {1e3*(t1-t0):.1f} ms, resident solve {1e3*(t2-t1):.1f} ms, download {1e3*(t3-t2):.1f} ms; "
          f"fx_system_solve_batch end to end {1e3*(t5-t4):.1f} ms = {n/(t5-t4)/1e6:.2f} M systems/s")
