"""fx_system_solve_batch on host buffers (analysis + upload + solve + read back), the reference's ring16 workload.
    python3 tools/host_path.py [n_systems] [reps] [helped]  (FIKSI_AMD_TRACE=1: the phases of every call on stderr)
helped = 1: the arrays page-locked (fx_host_register) and the one-structure hint set; 2: page-locked only; 3: the hint only.
Prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from fiksi_amd import abi, workloads


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    helped = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import ctypes as C

    import numpy as np

    from fiksi_amd._lib import check, lib
    ctx = abi.Context(0)
    a = abi.normalize_batch(workloads.ring16(n))
    start = a["vars"].copy()
    a["vars"] = start.copy()
    res = np.zeros(n, dtype=abi.RESULT_DTYPE)
    st, o = abi.as_struct(a), abi.solving_opts()
    if helped in (1, 2):
        ctx.host_register(a["vars"], a["expr_param"], res)
    ctx.set_batch_hints(one_structure=helped in (1, 3))
    times = []
    for k in range(reps + 1):
        a["vars"][:] = start
        t0 = time.perf_counter()
        check(lib.fx_system_solve_batch(ctx.handle, C.byref(st), C.byref(o), res.ctypes.data), "fx_system_solve_batch")
        if k:
            times.append((time.perf_counter() - t0) * 1e3)
    if helped in (1, 2):
        ctx.host_unregister(a["vars"], a["expr_param"], res)
    print(json.dumps({"workload": f"ring16 x {n}", "entry_point": "fx_system_solve_batch", "helped": helped, "ms_per_call_min": min(times),
                      "ms_per_call_median": sorted(times)[len(times) // 2], "converged": int((res["sse_unscaled"] < 1e-4).sum())}))


if __name__ == "__main__":
    main()
