"""Diagnostic driver for rocprofv3: a resident batch of the reference's bench sketch (fiksi_bench.rs:15-40) —
`n_batch` Systems of `n_tri` hinged triangles — solved `reps` times after a warm-up.
    python3 tools/hinged_batch.py [n_tri] [n_batch] [reps] [wide routing: 1 | 0 | -1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import abi, workloads
n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = fiksi_amd.Context(0)
if len(sys.argv) > 4:  # 4th argument: fx_ctx_set_wide_routing (1 wide kernel, 0 team kernels, -1 by cost)
    ctx.set_wide_routing(int(sys.argv[4]))
db = ctx.upload(workloads.hinged_triangles(n_batch, n_tri))
opts = abi.solving_opts()
db.system_solve(opts)
ctx.synchronize()
ctx.timer_begin()
for _ in range(reps):
    db.system_solve(opts)
ms = ctx.timer_end() / reps
res = db.get_results()
print(f"hinged_triangles({n_tri}) x {n_batch}: {ms:.3f} ms per solve, converged {(res['sse_unscaled'] < 1e-4).mean():.3f}")
