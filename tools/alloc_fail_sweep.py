"""Allocation-failure sweep of the library's host side (no GPU): for every entry point scenario below, the n-th allocation the
call makes is made to fail, for n = 1 ... N (N = the allocations an undisturbed call makes; big N are sampled), and the call
must come back with an error code — FX_ERR_NOMEM, or success where the failure could be worked around (a thread that cannot
be started runs on the caller) — never with a signal, std::terminate or a sanitizer report; afterwards the library holds no more
host blocks and no more blocks of the make-believe device than an undisturbed call leaves behind.

Runs against the sanitizer build (`make -C fiksi_amd/csrc asan`), whose own operator new counts and fails on request
(fx_host_only.cpp: fx_test_fail_alloc_at) and whose make-believe device keeps its memory on the host heap (fx_hip_shim.h):

    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0 FIKSI_AMD_LIBRARY=fiksi_amd/libfiksi_host_asan.so \
    FIKSI_AMD_HIP_RUNTIME=system FIKSI_AMD_SHIM_FAKE_DEVICE=1 python tools/alloc_fail_sweep.py [max points per scenario]

What it pins is SURVEY 8b / include/fiksi_amd.h: "nothing throws, aborts or panics across this boundary" — the reference's
own conventions (System::solve returns, fiksi/src/lib.rs:464; numerical failure is silent, lm.rs:134-137). tests/test_host_sanitizers.py runs it."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from fiksi_amd import abi, workloads
from fiksi_amd._lib import lib
from helpers import random_sketch

MAX_POINTS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
FX_ERR_NOMEM, FX_ERR_HIP, FX_ERR_INTERNAL = -5, -3, -7

for name, res in (("fx_test_fail_alloc_at", None), ("fx_test_alloc_count", C.c_long), ("fx_test_live_allocations", C.c_long),
                  ("fx_test_live_device_blocks", C.c_long)):
    f = getattr(lib, name)  # (only the sanitizer build has them: AttributeError otherwise)
    f.restype = res
lib.fx_test_fail_alloc_at.argtypes = [C.c_long]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def sweep(name, call, ok_codes=(0,), settle=None):
    """call() -> rc. First undisturbed (twice: caches warm), then with the n-th allocation failing."""
    lib.fx_test_fail_alloc_at(0)
    for _ in range(2):
        rc0 = call()
        if settle:
            settle()
    assert rc0 in ok_codes, f"{name}: undisturbed call returned {rc0} ({lib.fx_last_error().decode()})"
    c0 = lib.fx_test_alloc_count()
    rc0 = call()
    n_allocs = lib.fx_test_alloc_count() - c0
    if settle:
        settle()
    live0, dev0 = lib.fx_test_live_allocations(), lib.fx_test_live_device_blocks()
    points = list(range(1, n_allocs + 1))
    if len(points) > MAX_POINTS:  # every one of the first allocations, then evenly spread
        head = points[:MAX_POINTS // 2]
        step = (n_allocs - len(head)) / (MAX_POINTS - len(head))
        points = head + sorted(set(int(len(head) + 1 + k * step) for k in range(MAX_POINTS - len(head))))
    seen = {}
    for n in points:
        lib.fx_test_fail_alloc_at(n)
        rc = call()
        lib.fx_test_fail_alloc_at(0)
        if settle:
            settle()
        seen[rc] = seen.get(rc, 0) + 1
        assert rc in ok_codes or rc in (FX_ERR_NOMEM,), f"{name}: allocation {n} of {n_allocs} failing gave {rc} ({lib.fx_last_error().decode()})"
        if rc == FX_ERR_NOMEM:
            assert b"memory" in lib.fx_last_error(), lib.fx_last_error()
        live, dev = lib.fx_test_live_allocations(), lib.fx_test_live_device_blocks()
        assert live <= live0, f"{name}: allocation {n} of {n_allocs} failing left {live - live0} host blocks behind"
        assert dev <= dev0, f"{name}: allocation {n} of {n_allocs} failing left {dev - dev0} device blocks behind"
    print(f"{name}: {n_allocs} allocations per call, {len(points)} failure points, return codes {seen}")
    return seen


ctx = C.c_void_p()
have_device = lib.fx_ctx_create(C.byref(ctx), 0) == 0
print("make-believe device:", have_device)

# ---- the host-only entry points ------------------------------------------------------------------------------------------
big = abi.normalize_batch(workloads.ring16(30000))
st_big = abi.as_struct(big)
sweep("fx_batch_validate (30 000 ring16: one structure)", lambda: lib.fx_batch_validate(C.byref(st_big)))
# (several structures, enough rows and variables for the analysis to cut the batch into ranges on threads of their own)
mixed = abi.normalize_batch(workloads.concat([workloads.ring16(2600), workloads.hinged_triangles(2300, 5), workloads.ring16(700, fix_gauge=True),
                                              workloads.concat([random_sketch(s).flatten() for s in range(5)])]))
assert int(mixed["expr_off"][-1]) + int(mixed["var_off"][-1]) > 200000
st_mixed = abi.as_struct(mixed)
sweep("fx_batch_validate (structure classes, threaded analysis)", lambda: lib.fx_batch_validate(C.byref(st_mixed)))
ne = int(mixed["expr_off"][-1])
nnz = C.c_uint64(0)
rp, ci = np.zeros(ne + 1, dtype=np.uint32), np.zeros(8 * ne, dtype=np.uint32)
sweep("fx_jacobian_structure", lambda: lib.fx_jacobian_structure(C.byref(st_mixed), C.byref(nnz), ptr(rp), ptr(ci)))

hinged = abi.normalize_batch(workloads.hinged_triangles(3, 16))
st_h = abi.as_struct(hinged)
nb = C.c_uint32(0)
ne_h = int(hinged["expr_off"][2] - hinged["expr_off"][1])
nv_h = int(hinged["var_off"][2] - hinged["var_off"][1])
bc, bro, brows, bvo = (np.zeros(4 * ne_h + 1, dtype=np.uint32) for _ in range(4))
bvars = np.zeros(nv_h, dtype=np.uint32)


def single_pass():
    return lib.fx_single_pass_blocks(C.byref(st_h), 1, C.byref(nb), ptr(bc), ptr(bro), ptr(brows), ptr(bvo), ptr(bvars))


sweep("fx_single_pass_blocks", single_pass)

# the augmented matrix [J; sqrt(lambda) I] of a ring16 sketch: 32 columns, 64 rows (column c: the rows that read variable c)
rp16, ci16 = abi.jacobian_structure(workloads.ring16(1))
cols = [[] for _ in range(32)]
for e in range(32):
    for c in ci16[rp16[e]:rp16[e + 1]]:
        cols[int(c)].append(e)
for c in range(32):
    cols[c].append(32 + c)
cp = np.zeros(33, dtype=np.int32)
cp[1:] = np.cumsum([len(c) for c in cols])
ri = np.array([r for c in cols for r in c], dtype=np.int32)
col_perm, row_perm = np.zeros(32, dtype=np.int32), np.zeros(64, dtype=np.int32)
h_ptr, r_ptr = np.zeros(33, dtype=np.int32), np.zeros(33, dtype=np.int32)
h_rows, r_rows = np.zeros(64 * 32, dtype=np.int32), np.zeros(33 * 16, dtype=np.int32)
sweep("fx_qr_symbolic (COLAMD)", lambda: lib.fx_qr_symbolic(64, 32, ptr(cp), ptr(ri), 1, ptr(col_perm), ptr(row_perm), ptr(h_ptr), ptr(h_rows),
                                                           len(h_rows), ptr(r_ptr), ptr(r_rows), len(r_rows)))

# ---- the builder ---------------------------------------------------------------------------------------------------------
import fiksi_amd  # noqa: E402


def build_and_plan():
    """a System built from scratch with a failure somewhere in it: every builder call either happens or returns a code"""
    h = C.c_void_p()
    rc = lib.fxs_system_new(C.byref(h))
    if rc:
        return rc
    try:
        pts = []
        for k in range(6):
            r = lib.fxs_point_create(h, C.c_double(float(k)), C.c_double(float(k * k % 5)))
            if r < 0:
                return int(r)
            pts.append(int(r))
        n_before = lib.fxs_num_constraints(h), lib.fxs_num_expressions(h)
        for a, b in ((0, 1), (1, 2), (2, 0), (2, 3), (3, 4), (4, 5), (5, 3)):
            el = (C.c_uint32 * 2)(pts[a], pts[b])
            r = lib.fxs_constraint_create(h, 1, el, 2, C.c_double(1.5))
            if r < 0:
                # all or nothing: the System is as it was before the failing call
                assert (lib.fxs_num_constraints(h), lib.fxs_num_expressions(h)) == n_before
                return int(r)
            n_before = lib.fxs_num_constraints(h), lib.fxs_num_expressions(h)
        r = lib.fxs_element_fix(h, pts[0])
        if r:
            return r
        n_w, fl = C.c_uint32(0), C.c_uint32(0)
        r = lib.fxs_recursive_plan(h, 200000, None, 0, C.byref(n_w), C.byref(fl))
        if r:
            return r
        ncomp = C.c_uint32(0)
        ec, cc = np.zeros(6, dtype=np.uint16), np.zeros(7, dtype=np.uint16)
        r = lib.fxs_components(h, C.byref(ncomp), ptr(ec), ptr(cc))
        if r:
            return r
        flat = C.c_void_p()
        arr = (C.c_void_p * 1)(h)
        r = lib.fxs_flatten(arr, 1, C.byref(flat))
        if r:
            return r
        lib.fxs_flat_free(flat)
        return 0
    finally:
        lib.fxs_system_free(h)


sweep("builder: System, elements, constraints, fix, recursive plan, components, flatten", build_and_plan)

# ---- the device entry points as far as the make-believe device takes them (analysis + uploads for real, launches "no device") --
if have_device:
    def upload(st):
        def run():
            db = C.c_void_p()
            rc = lib.fx_batch_upload(ctx, C.byref(st), C.byref(db))
            if rc == 0:
                lib.fx_batch_free(ctx, db)
            return rc
        return run

    small = abi.normalize_batch(workloads.ring16(40))
    st_small = abi.as_struct(small)
    sweep("fx_batch_upload (40 ring16: one-structure program, staged copy)", upload(st_small))
    sweep("fx_batch_upload (structure classes and their programs)", upload(st_mixed))
    wide = abi.normalize_batch(workloads.hinged_triangles(12, 16))
    st_wide = abi.as_struct(wide)
    sweep("fx_batch_upload (66-variable sketches: sparse one-structure program)", upload(st_wide))
    large = abi.normalize_batch(workloads.large_sketch(300, seed=3))
    st_large = abi.as_struct(large)
    sweep("fx_batch_upload (a System beyond one wavefront: host copy kept)", upload(st_large))

    res = np.zeros(40, dtype=abi.RESULT_DTYPE)
    o = abi.solving_opts()
    # the solve itself ends at the first launch ("no device" -> FX_ERR_HIP): what is walked here is analysis + upload + cleanup
    sweep("fx_system_solve_batch (host buffers)", lambda: lib.fx_system_solve_batch(ctx, C.byref(st_small), C.byref(o), ptr(res)), ok_codes=(FX_ERR_HIP,))
    o_sp = abi.solving_opts(decomposer=1)

    def resident_single_pass():
        db = C.c_void_p()
        rc = lib.fx_batch_upload(ctx, C.byref(st_h), C.byref(db))
        if rc:
            return rc
        rc = lib.fx_system_solve_device(ctx, db, C.byref(o_sp))  # ensure_units (threads, decomposition), then the launch fails
        lib.fx_batch_free(ctx, db)
        return rc

    sweep("fx_system_solve_device (SinglePass blocks built on first use)", resident_single_pass, ok_codes=(FX_ERR_HIP,))
    o_qr = abi.solving_opts(solver=2)

    def resident_qr():
        db = C.c_void_p()
        rc = lib.fx_batch_upload(ctx, C.byref(st_small), C.byref(db))
        if rc:
            return rc
        rc = lib.fx_system_solve_device(ctx, db, C.byref(o_qr))  # ensure_qr_plans: COLAMD, symbolic QR, the table program
        lib.fx_batch_free(ctx, db)
        return rc

    sweep("fx_system_solve_device (FX_STEP_QR plans built on first use)", resident_qr, ok_codes=(FX_ERR_HIP,))

    def eval_jacobian():
        r = np.zeros(int(small["expr_off"][-1]))
        return lib.fx_eval_residual_jacobian(ctx, C.byref(st_small), ptr(r), None)

    sweep("fx_eval_residual_jacobian (row-parallel arrays built on first use)", eval_jacobian, ok_codes=(FX_ERR_HIP,))

    ctx2 = C.c_void_p()
    assert lib.fx_ctx_create(C.byref(ctx2), 0) == 0
    handles = (C.c_void_p * 2)(ctx, ctx2)
    total = (C.c_uint64 * 4)()
    res2 = np.zeros(40, dtype=abi.RESULT_DTYPE)
    sweep("fx_system_solve_batch_multi (two contexts, a thread per shard)",
          lambda: lib.fx_system_solve_batch_multi(handles, 2, C.byref(st_small), C.byref(o), ptr(res2), total), ok_codes=(FX_ERR_HIP,))
    lib.fx_ctx_destroy(ctx2)
    lib.fx_ctx_destroy(ctx)
    assert lib.fx_test_live_device_blocks() == 0, f"{lib.fx_test_live_device_blocks()} device blocks left after the contexts were destroyed"
print("allocation-failure sweep: ok")
