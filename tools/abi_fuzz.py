"""Mutation fuzz of the library's host side (no GPU): random corruptions of valid inputs must come back as error codes,
never as a crash or a sanitizer report. Meant for the sanitizer build with its make-believe device
(fx_hip_shim.h, FIKSI_AMD_SHIM_FAKE_DEVICE=1: memory on the host heap, every copy real, every kernel launch "no device"),
so that the device entry points run their host analysis and their uploads under ASan / UBSan too:

    make -C fiksi_amd/csrc asan
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    FIKSI_AMD_LIBRARY=fiksi_amd/libfiksi_host_asan.so FIKSI_AMD_HIP_RUNTIME=system FIKSI_AMD_SHIM_FAKE_DEVICE=1 \
    python tools/abi_fuzz.py 300 [seed]

Targets: fx_batch_validate, fx_jacobian_structure, fx_single_pass_blocks (round 2); fx_qr_symbolic on corrupted column
patterns, fxs_recursive_plan on random sketches with odd budgets / capacities, and — through the make-believe device —
fx_batch_upload, fx_system_solve_batch, fx_system_prepare_batch, fx_system_solve_batch_multi, fx_cluster_solve_batch,
fx_pose_transform_points, fx_unscale_vars_strided; batches big enough for the structure classes and the one-structure programs
of the grouped kernel's builds (build_gc_program, build_gs_program, the class lists) (round 4). tests/test_host_sanitizers.py runs it with a fixed seed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from fiksi_amd import abi, workloads
from fiksi_amd._lib import lib
from helpers import Lcg, random_sketch

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = Lcg(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
codes = {}


def note(name, rc):
    codes.setdefault(name, {})
    codes[name][rc] = codes[name].get(rc, 0) + 1


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---- a context on the make-believe device (None on the real product without a GPU, or on the plain host-only build)
ctxs = []
for _ in range(3):
    h = C.c_void_p()
    if lib.fx_ctx_create(C.byref(h), 0) == 0:
        ctxs.append(h)
print("contexts:", len(ctxs))

sketches = [random_sketch(s) for s in range(6)]
base = [workloads.ring16(3), workloads.hinged_triangles(2, 4), workloads.concat([s.flatten() for s in sketches[:4]]),
        workloads.hinged_triangles(2, 16), workloads.large_sketch(40, seed=2)]


def mutate(b, n_mut):
    for _ in range(n_mut):
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


for it in range(n_iter):
    # ---------------- flat batches through every entry point that takes one ----------------
    b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base[it % len(base)].items()}
    mutate(b, int(g.u(0, 3.99)))  # (sometimes none: the valid batch goes through the device entry points as far as they get)
    try:
        a = abi.normalize_batch(b)
    except Exception as e:  # the Python layer refused it (length checks)
        note("python", type(e).__name__)
        a = None
    if a is not None:
        st = abi.as_struct(a)
        rc = lib.fx_batch_validate(C.byref(st))
        note("fx_batch_validate", rc)
        n = len(a["var_off"]) - 1
        if rc == 0:  # still a valid batch: the host-only entry points must cope with it
            nnz = C.c_uint64(0)
            ne = int(a["expr_off"][-1])
            rp = np.zeros(ne + 1, dtype=np.uint32)
            note("fx_jacobian_structure", lib.fx_jacobian_structure(C.byref(st), C.byref(nnz), rp.ctypes.data, None))
            for s in range(n):
                try:
                    abi.single_pass_blocks(a, s)
                except Exception:
                    pass
        if ctxs:  # valid or not: the device entry points validate for themselves, then analyse and upload
            res = np.zeros(max(n, 1), dtype=abi.RESULT_DTYPE)
            for dec in (0, 1):
                o = abi.solving_opts(decomposer=dec, solver=int(g.u(0, 2.99)))
                work = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.items()}
                note("fx_system_solve_batch", lib.fx_system_solve_batch(ctxs[0], C.byref(abi.as_struct(work)), C.byref(o), ptr(res)))
            work = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.items()}
            handles = (C.c_void_p * len(ctxs))(*ctxs)
            total = (C.c_uint64 * 4)()
            o = abi.solving_opts()
            note("fx_system_solve_batch_multi", lib.fx_system_solve_batch_multi(handles, int(g.u(1, len(ctxs) + 0.99)), C.byref(abi.as_struct(work)),
                                                                                 C.byref(o), ptr(res), total))
            nv, ne = len(a["vars"]), len(a["expr_tag"])
            ov, op, osc = np.zeros(max(nv, 1)), np.zeros(max(ne, 1)), np.zeros(max(n, 1))
            note("fx_system_prepare_batch", lib.fx_system_prepare_batch(ctxs[0], C.byref(st), int(g.u(0, 1.99)), ptr(ov), ptr(op), ptr(osc)))
            lo = abi.solving_opts().lm
            note("fx_cluster_solve_batch", lib.fx_cluster_solve_batch(ctxs[0], C.byref(st), C.byref(lo), ptr(res)))
            db = C.c_void_p()
            rc = lib.fx_batch_upload(ctxs[0], C.byref(st), C.byref(db))
            note("fx_batch_upload", rc)
            if rc == 0:
                lib.fx_batch_free(ctxs[0], db)
    # ---------------- fx_qr_symbolic on a corrupted column pattern ----------------
    nc, extra = int(g.u(1, 24)), int(g.u(0, 30))
    nr = nc + extra  # an augmented matrix [J; sqrt(lambda) I]: column j holds some rows of J and row extra + j
    cols = []
    for j in range(nc):
        k = int(g.u(0, min(extra, 6) + 0.99))
        cols.append(sorted(set(int(g.u(0, extra - 1e-9)) for _ in range(k) if extra)) + [extra + j])
    cp = np.zeros(nc + 1, dtype=np.int32)
    cp[1:] = np.cumsum([len(c) for c in cols])
    ri = np.array([r for c in cols for r in c] + [0], dtype=np.int32)
    if g.u(0, 1) < 0.6:
        for _ in range(int(g.u(1, 2.99))):
            tgt = cp if g.u(0, 1) < 0.5 else ri
            i = int(g.u(0, len(tgt) - 1e-9))
            tgt[i] = [-1, 2 ** 31 - 1, int(tgt[i]) + 1, int(tgt[i]) - 1, nr, 0][int(g.u(0, 5.99))]
    col_perm, row_perm = np.zeros(nc, dtype=np.int32), np.zeros(nr, dtype=np.int32)
    h_ptr, r_ptr = np.zeros(nc + 1, dtype=np.int32), np.zeros(nc + 1, dtype=np.int32)
    h_cap = [nr * nc + 1, nr * nc + 1, 1, 0, nr][int(g.u(0, 4.99))]  # sometimes too small: the call must say so
    r_cap = [nc * (nc + 1) // 2 + 1, nc * (nc + 1) // 2 + 1, 1, 0][int(g.u(0, 3.99))]
    h_rows, r_rows = np.zeros(max(h_cap, 1), dtype=np.int32), np.zeros(max(r_cap, 1), dtype=np.int32)
    note("fx_qr_symbolic", lib.fx_qr_symbolic([nr, -1, 0][int(g.u(0, 1.2))] if g.u(0, 1) < 0.05 else nr, nc, ptr(cp), ptr(ri), int(g.u(0, 1.99)),
                                               ptr(col_perm), ptr(row_perm), ptr(h_ptr), ptr(h_rows), h_cap, ptr(r_ptr), ptr(r_rows), r_cap))
    # ---------------- fxs_recursive_plan: odd budgets and capacities ----------------
    sk = sketches[it % len(sketches)]
    n_w, fl = C.c_uint32(0), C.c_uint32(0)
    budget = [0, 1, 7, 500, 200000][int(g.u(0, 4.99))]
    rc = lib.fxs_recursive_plan(sk._h, budget, None, 0, C.byref(n_w), C.byref(fl))
    note("fxs_recursive_plan", rc)
    if rc == 0:
        cap = [n_w.value, max(0, n_w.value - 1), 0, n_w.value // 2][int(g.u(0, 3.99))]
        out = np.zeros(max(cap, 1), dtype=np.uint32)
        note("fxs_recursive_plan", lib.fxs_recursive_plan(sk._h, budget, ptr(out), cap, C.byref(n_w), C.byref(fl)))
    # ---------------- fx_pose_transform_points / fx_unscale_vars_strided: index arguments ----------------
    if ctxs:
        n_vars, n_poses, n_pts = int(g.u(1, 30)), int(g.u(1, 4)), int(g.u(0, 8.99))
        poses, vars_ = np.zeros(3 * n_poses), np.zeros(n_vars)
        pose_of = np.array([int(g.u(0, n_poses + 0.3)) for _ in range(n_pts)] + [0], dtype=np.uint32)  # (sometimes one past the end)
        var_idx = np.array([int(g.u(0, n_vars + 0.5)) for _ in range(n_pts)] + [0], dtype=np.uint32)   # (overlaps, last variable, past the end)
        note("fx_pose_transform_points", lib.fx_pose_transform_points(ctxs[0], ptr(poses), n_poses, ptr(pose_of), ptr(var_idx), n_pts, ptr(vars_), n_vars))
        ns, nvs = int(g.u(0, 5.99)), int(g.u(0, 9.99))
        sc, sv, mk, vv = np.ones(max(ns, 1)), np.zeros(max(ns * nvs, 1)), np.ones(max(nvs, 1), dtype=np.uint8), np.zeros(max(ns * nvs, 1))
        note("fx_unscale_vars_strided", lib.fx_unscale_vars_strided(ctxs[0], ptr(sc), ns, nvs, ptr(sv), ptr(mk), ptr(vv)))

# ---------------- batches big enough for the structure-class analysis and the one-structure programs (round 4) ----------------
if ctxs:
    for it in range(3):
        parts = [workloads.ring16(2100 + 7 * it), workloads.hinged_triangles(2060, 5), workloads.ring16(900, fix_gauge=True),
                 workloads.hinged_triangles(40, 16), workloads.hinged_triangles(300, 11)]
        for b in (workloads.concat(parts[:3]), workloads.concat(parts), workloads.hinged_triangles(64, 16), workloads.hinged_triangles(40, 11),
                  workloads.ring_chords(33, 20, 7)):
            b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
            mutate(b, it)  # (the first round: the valid batch)
            try:
                a = abi.normalize_batch(b)
            except Exception as e:
                note("python", type(e).__name__)
                continue
            st = abi.as_struct(a)
            db = C.c_void_p()
            rc = lib.fx_batch_upload(ctxs[0], C.byref(st), C.byref(db))
            note("fx_batch_upload (classes / programs)", rc)
            if rc == 0:
                o = abi.solving_opts()
                note("fx_system_solve_device (classes / programs)", lib.fx_system_solve_device(ctxs[0], db, C.byref(o)))
                lib.fx_batch_free(ctxs[0], db)

# ---------------- host-buffer calls under the one-structure hint, on registered buffers (round 5) ----------------
if ctxs:
    lib.fx_ctx_set_batch_hints(ctxs[0], abi.HINT_ONE_STRUCTURE)
    for it in range(4):
        b = [workloads.ring16(700), workloads.hinged_triangles(300, 11), workloads.ring16(70000 if n_iter >= 100 else 5000),
             workloads.concat([workloads.ring16(40), workloads.hinged_triangles(40, 5)])][it]
        for wrong in (False, True):  # the claim holds / one System differs: the library must find out by itself
            a = abi.normalize_batch({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
            if wrong:
                s = len(a["var_off"]) // 2
                a["expr_idx"][4 * int(a["expr_off"][s]) + 1] ^= 1
                a["var_fixed"][int(a["var_off"][s])] ^= 1
            if hasattr(lib, "fx_test_verify_one_structure"):  # (the sanitizer build's hooks, fx_host_only.cpp)
                ok = lib.fx_test_verify_one_structure(C.byref(abi.as_struct(a)))
                assert ok == (0 if wrong or it == 3 else 1), ("verify_one_structure", it, wrong, ok)
                if not wrong and it != 3:
                    d = lib.fx_test_hinted_plan_differs(C.byref(abi.as_struct(a)))
                    assert d == 0, ("the plan from System 0 alone differs from the full analysis", it, d)
                    note("hinted plan == full plan", d)
            res = np.zeros(len(a["var_off"]) - 1, dtype=abi.RESULT_DTYPE)
            for arr in (a["vars"], a["expr_param"], res):
                note("fx_host_register", lib.fx_host_register(ctxs[0], ptr(arr), arr.nbytes))
            o = abi.solving_opts()
            note("fx_system_solve_batch (hinted)", lib.fx_system_solve_batch(ctxs[0], C.byref(abi.as_struct(a)), C.byref(o), ptr(res)))
            for arr in (a["vars"], a["expr_param"], res):
                note("fx_host_unregister", lib.fx_host_unregister(ctxs[0], ptr(arr)))
    note("fx_host_unregister (never registered)", lib.fx_host_unregister(ctxs[0], ptr(np.zeros(4))))
    note("fx_host_register (null)", lib.fx_host_register(ctxs[0], None, 64))
    lib.fx_ctx_set_batch_hints(ctxs[0], 0)

for h in ctxs:
    lib.fx_ctx_destroy(h)
print("return codes:", codes)
