"""FX_STEP_QR against the oracle: how often the LM path is identical, bit for bit, and what the mode costs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g
g.build()
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle as O
from helpers import mixed_sketch, random_sketch, random_big_sketch

ctx = fiksi_amd.Context(0)

def report(name, b, **kw):
    opts = abi.solving_opts(solver=2, **kw)
    t0 = time.time()
    v, res = ctx.system_solve_batch(b, opts)
    t1 = time.time()
    if kw.get("decomposer") == 1:
        v_o, res_o = O.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    else:
        v_o, res_o = O.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    n = len(res)
    same = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"]) & (res["exit"] == res_o["exit"])
    bits = np.zeros(n, bool)
    for s in range(n):
        v0, v1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
        bits[s] = np.array_equal(v[v0:v1].view(np.uint64), v_o[v0:v1].view(np.uint64))
    sse_bits = np.array_equal(res["sse"].view(np.uint64), res_o["sse"].view(np.uint64))
    d = np.abs(res["sse"] - res_o["sse"])
    ok = np.isfinite(d)
    print(f"{name}: n={n} same counts {same.mean():.4f}  variables bit-identical {bits.mean():.4f}  sse bit-identical(all) {sse_bits} "
          f"max|dsse| {d[ok].max() if ok.any() else 0:.3e}  exits gpu {np.bincount(res['exit'], minlength=6)} ref {np.bincount(res_o['exit'], minlength=6)}  ({t1-t0:.2f}s)", flush=True)
    bad = np.nonzero(~same)[0][:5]
    for s in bad:
        print("   differs:", s, res[s], res_o[s])

report("hinged11 x64 (distance only)", workloads.hinged_triangles(64, 11))
report("quadrilaterals", workloads.concat([workloads.quadrilateral(), workloads.quadrilateral(False)]))
report("ring16 x2048", workloads.ring16(2048))
report("ring16 gauge-fixed x512", workloads.ring16(512, fix_gauge=True))
report("ring16 inconsistent x512", workloads.ring16(512, inconsistent=True))
report("mixed x64", fiksi_amd.flatten([mixed_sketch(100 + s, fix_some=(s % 3 == 0)) for s in range(64)]))
report("random x300", workloads.concat([random_sketch(s).flatten() for s in range(300)]))
report("random big 20..30 x24", workloads.concat([random_big_sketch(1000 + s, 20 + s % 10).flatten() for s in range(24)]))
report("hinged11 SinglePass x64", workloads.hinged_triangles(64, 11), decomposer=1)
report("random SinglePass x200", workloads.concat([random_sketch(s).flatten() for s in range(200)]), decomposer=1)

# cost on the headline batch
b = workloads.ring16(100000)
db = ctx.upload(b)
for name, o in (("cholesky", abi.solving_opts()), ("qr", abi.solving_opts(solver=2)), ("refined", abi.solving_opts(solver=1))):
    db.system_solve(o); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(3):
        db.system_solve(o)
    ms = ctx.timer_end() / 3
    r = db.get_results()
    print(f"100k ring16 {name}: {ms:.2f} ms  converged {(r['sse_unscaled'] < 1e-4).mean():.4f} trials {r['trials'].sum()}", flush=True)
pc = db.phase_cycles(abi.solving_opts(solver=2))
tot = sum(pc.values())
print("QR phase cycles per System:", {k: v // 100000 for k, v in pc.items()}, "total", tot // 100000, flush=True)
pc = db.phase_cycles(abi.solving_opts())
print("default-route phase cycles per System:", {k: v // 100000 for k, v in pc.items()}, flush=True)
