"""Grouped kernel (four Systems per wavefront) against one System per wavefront by batch size, f64 and f32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
full = workloads.ring16(100000)
for n in (1000, 2000, 4000, 6250, 8192, 12500, 25000, 50000, 100000):
    b = workloads.shard(full, 0, 100000 // n) if n < 100000 else full
    db = ctx.upload(b)
    line = f"{len(b['var_off']) - 1:7d} Systems:"
    for f32 in (False, True):
        for g in (0, 1):
            ctx.set_routing(g)
            o = abi.solving_opts(f32=f32)
            db.system_solve(o); ctx.synchronize()
            ctx.timer_begin()
            for _ in range(5): db.system_solve(o)
            ms = ctx.timer_end() / 5
            line += f"  {'f32' if f32 else 'f64'} {'grouped' if g else 'one/wave'} {ms:7.3f} ms"
    print(line, flush=True)
    db.free()
ctx.set_routing(-1)
b5 = workloads.ring16(125000, inconsistent=True)
db = ctx.upload(b5)
for f32 in (False, True):
    o = abi.solving_opts(f32=f32)
    db.system_solve(o); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(5): db.system_solve(o)
    ms = ctx.timer_end() / 5
    r = db.get_results()
    print(f"cfg5 share {'f32' if f32 else 'f64'}: {ms:.3f} ms trials {r['trials'].sum()} accepted {r['accepted'].sum()} exits {np.bincount(r['exit'], minlength=6)} max trials {r['trials'].max()}", flush=True)
