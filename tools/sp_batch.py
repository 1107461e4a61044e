"""Diagnostic: Decomposer::SinglePass on a resident batch — first call (builds the block plan) vs later calls."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = fiksi_amd.Context(0)
for name, b in (("ring16", workloads.ring16(n)), ("hinged_triangles(11)", workloads.hinged_triangles(n, 11))):
    db = ctx.upload(b)
    for opts, label in ((abi.solving_opts(), "None"), (abi.solving_opts(decomposer=1), "SinglePass")):
        for rep in range(3):
            t = time.time(); db.system_solve(opts); ctx.synchronize(); dt = time.time() - t
            print(f"{name} x{n} {label} call {rep}: {dt*1e3:.1f} ms")
    r = db.get_results()
    print("   converged (unscaled SSE < 1e-4):", float((r["sse_unscaled"] < 1e-4).mean()))
    db.free()
