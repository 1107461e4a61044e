"""fx_system_solve_batch on host buffers (analysis + upload + solve + read back), the reference's ring16 workload.
    python3 tools/host_path.py [n_systems] [reps]           (FIKSI_AMD_TRACE=1: the phases of every call on stderr)
Prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ctx = abi.Context(0)
    b = workloads.ring16(n)
    ctx.system_solve_batch(b)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        v, res = ctx.system_solve_batch(b)
        times.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"workload": f"ring16 x {n}", "entry_point": "fx_system_solve_batch", "ms_per_call_min": min(times),
                      "ms_per_call_median": sorted(times)[len(times) // 2], "converged": int((res["sse_unscaled"] < 1e-4).sum())}))


if __name__ == "__main__":
    main()
