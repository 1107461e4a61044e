"""Mutation fuzz of the host-side entry points (no GPU): random corruptions of valid batches must come back as error codes,
never as a crash. Run against the sanitizer build:  make -C fiksi_amd/csrc asan &&
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0
FIKSI_AMD_LIBRARY=fiksi_amd/libfiksi_host_asan.so FIKSI_AMD_HIP_RUNTIME=system python tools/abi_fuzz.py 3000"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from fiksi_amd import abi, workloads
from fiksi_amd._lib import lib
from helpers import random_sketch, Lcg

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = Lcg(12345)
base = [workloads.ring16(3), workloads.hinged_triangles(2, 4), workloads.concat([random_sketch(s).flatten() for s in range(4)])]
codes = {}
for it in range(n_iter):
    b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base[it % len(base)].items()}
    for _ in range(1 + int(g.u(0, 3.99))):
        key = ["var_off", "expr_off", "expr_tag", "expr_idx", "var_fixed", "var_comp", "expr_comp", "vars", "expr_param"][int(g.u(0, 8.99))]
        a = b.get(key)
        if a is None or len(a) == 0:
            continue
        i = int(g.u(0, len(a) - 1e-9))
        kind = int(g.u(0, 4.99))
        if a.dtype.kind == "f":
            a[i] = [np.nan, np.inf, -np.inf, 1e308, 0.0][kind]
        else:
            info = np.iinfo(a.dtype)
            a[i] = [info.max, 0, int(a[i]) ^ 1, min(info.max, int(a[i]) + 7), max(0, int(a[i]) - 3)][kind]
    try:
        a = abi.normalize_batch(b)
    except Exception as e:  # the Python layer refused it (length checks)
        codes["python:" + type(e).__name__] = codes.get("python:" + type(e).__name__, 0) + 1
        continue
    st = abi.as_struct(a)
    rc = lib.fx_batch_validate(C.byref(st))
    codes[rc] = codes.get(rc, 0) + 1
    if rc == 0:  # still a valid batch: the other host-only entry points must cope with it
        nnz = C.c_uint64(0)
        ne = int(a["expr_off"][-1])
        rp = np.zeros(ne + 1, dtype=np.uint32)
        lib.fx_jacobian_structure(C.byref(st), C.byref(nnz), rp.ctypes.data, None)
        for s in range(len(a["var_off"]) - 1):
            try:
                abi.single_pass_blocks(a, s)
            except Exception:
                pass
print("return codes:", codes)
