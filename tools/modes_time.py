"""Diagnostic: the 100k ring16 batch in every solver mode."""
import sys; sys.path.insert(0, '.')
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(100000))
modes = {"LM": abi.solving_opts(), "LM refined": abi.solving_opts(solver=1), "LM f32": abi.solving_opts(f32=True),
         "LM SinglePass": abi.solving_opts(decomposer=1), "L-BFGS": abi.solving_opts(optimizer=1)}
for name, o in modes.items():
    db.system_solve(o); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(3): db.system_solve(o)
    ms = ctx.timer_end() / 3
    r = db.get_results()
    print(f"{name:14s}: {ms:7.2f} ms per 100k systems; converged (unscaled SSE < 1e-4) {float((r['sse_unscaled'] < 1e-4).mean()):.3f}; "
          f"iterations {r['accepted'].mean():.1f}, evaluations/trials {r['trials'].mean():.1f} per system")
