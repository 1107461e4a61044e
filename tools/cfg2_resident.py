"""rocprofv3 target: cfg2 (one 5 000-point sketch) solved on a resident batch, plan warm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.large_sketch(5000))
opts = abi.solving_opts()
db.system_solve(opts)
ctx.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx.timer_begin()
for _ in range(reps):
    db.system_solve(opts)
ms = ctx.timer_end() / reps
res = db.get_results()
print(f"cfg2 resident: {ms:.3f} ms per solve, accepted {int(res['accepted'][0])} trials {int(res['trials'][0])}")
