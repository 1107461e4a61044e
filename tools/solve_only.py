"""Diagnostic driver for rocprofv3: upload cfg3 (n ring16 sketches), run the fused solve `reps` times — nothing else, so
that a kernel-stats row of this run is the headline workload alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = fiksi_amd.Context(0)
db = ctx.upload(workloads.ring16(n))
for _ in range(reps):
    db.system_solve()
ctx.synchronize()
# (outside a profiler: the mean solve time by HIP events, one line)
ctx.timer_begin()
for _ in range(reps):
    db.system_solve()
print('{"systems": %d, "ms_per_solve": %.4f}' % (n, ctx.timer_end() / reps))
