"""A / B of the Jacobian-assembly kernel's store policy (run once per policy; FIKSI_AMD_K1_STORES=plain selects default-policy
stores instead of the shipped non-temporal ones): K1 on 100k and 500k ring16 sketches, HIP-event timed, as one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fiksi_amd
from fiksi_amd import workloads

ctx = fiksi_amd.Context(0)
out = {"stores": "plain" if os.environ.get("FIKSI_AMD_K1_STORES", "").startswith("p") else "non-temporal"}
for n in (100_000, 500_000):
    b = workloads.ring16(n, seed0=5_000_000)
    db = ctx.upload(b)
    for _ in range(3):
        db.eval_residual_jacobian(0)
    ctx.synchronize()
    reps = 40 if n == 100_000 else 16
    best = None
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(reps):
            db.eval_residual_jacobian(0)
        ms = ctx.timer_end() / reps
        best = ms if best is None else min(best, ms)
    nbytes = workloads.k1_algorithmic_bytes(b, db.nnz)
    out[str(n)] = {"avg_launch_us": best * 1e3, "GBps_by_8d_bytes": nbytes / (best * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (best * 1e-3) / 1e9 / 8000.0}
    db.free()
print(json.dumps(out))
