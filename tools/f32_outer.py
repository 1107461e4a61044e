"""f32 (cfg5): accepted-step limit against batch time and agreement with the f64 solve."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
for name, b in (("cfg5 share (125k inconsistent)", workloads.ring16(125000, inconsistent=True)), ("ring16 100k consistent", workloads.ring16(100000))):
    db = ctx.upload(b)
    db.system_solve(abi.solving_opts()); ctx.synchronize()
    r64 = db.get_results().copy()
    print(f"{name} f64: accepted max {r64['accepted'].max()} p99.9 {np.percentile(r64['accepted'], 99.9):.0f} trials max {r64['trials'].max()}")
    for mo in (100, 60, 40, 30, 20):
        o = abi.solving_opts(f32=True, max_outer=mo)
        db.system_solve(o); ctx.synchronize()
        ctx.timer_begin()
        for _ in range(3): db.system_solve(o)
        ms = ctx.timer_end() / 3
        r = db.get_results()
        rel = np.abs(r["sse"] - r64["sse"]) / np.maximum(r64["sse"], 1e-30)
        print(f"  max_outer {mo}: {ms:.3f} ms trials {r['trials'].sum()} max {r['trials'].max()} exits {np.bincount(r['exit'], minlength=6)} "
              f"rel SSE vs f64 median {np.median(rel):.2e} p95 {np.quantile(rel, 0.95):.2e} p99.9 {np.quantile(rel, 0.999):.2e} converged {(r['sse_unscaled'] < 1e-4).mean():.4f}", flush=True)
    db.free()
