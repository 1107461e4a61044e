"""Diagnostic: split the fused kernel's time into per-System fixed cost and per-trial cost."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads, abi
n = 100000
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(n))
def t(opts, reps=5):
    db.system_solve(opts); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(reps): db.system_solve(opts)
    return ctx.timer_end() / reps
full = t(abi.solving_opts()); res = db.get_results()
t0 = t(abi.solving_opts(max_outer=0))
t1 = t(abi.solving_opts(max_trials=1)); r1 = db.get_results()
t2 = t(abi.solving_opts(max_trials=2)); r2 = db.get_results()
print(f"full {full:.3f} ms ({res['trials'].mean():.2f} trials, {res['accepted'].mean():.2f} accepted per system)")
print(f"max_outer=0 (setup+eval+form+check): {t0:.3f} ms")
print(f"max_trials=1: {t1:.3f} ms (trials {r1['trials'].mean():.2f}, accepted {r1['accepted'].mean():.2f})")
print(f"max_trials=2: {t2:.3f} ms (trials {r2['trials'].mean():.2f}, accepted {r2['accepted'].mean():.2f})")
print(f"per-trial (full): {(full - t0) / res['trials'].mean():.3f} ms per 100k trials")
