"""The host-buffer call (fx_system_solve_batch) with the caller's help: page-locked buffers (fx_host_register) and the
one-structure hint (fx_ctx_set_batch_hints). The hint is never trusted — a batch that does not keep the promise must come
back exactly as without the hint — and a batch that keeps it must come back bit for bit as from the full analysis."""
import numpy as np
import pytest

import fiksi_amd
from fiksi_amd import abi, workloads

pytestmark = pytest.mark.gpu


def _solve(ctx, b, hint, register=False, lm_level=False, register_results=True, opts=None):
    """register: the value arrays page-locked — from 16 384 Systems of one structure on, the grouped kernel's one-structure build then
    reads and writes them in place (fx_solve.cpp: solve_host_in_place)."""
    import ctypes as C
    from fiksi_amd._lib import check, lib
    a = abi.normalize_batch({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
    res = np.zeros(len(a["var_off"]) - 1, dtype=abi.RESULT_DTYPE)
    ctx.set_batch_hints(one_structure=hint)
    locked = [a["vars"], a["expr_param"]] + ([res] if register_results else [])
    if register:
        ctx.host_register(*locked)
    try:
        if lm_level:
            o = opts if opts is not None else abi.lm_opts()
            check(lib.fx_lm_solve_batch(ctx.handle, C.byref(abi.as_struct(a)), C.byref(o), res.ctypes.data), "fx_lm_solve_batch")
        else:
            o = opts if opts is not None else abi.solving_opts()
            check(lib.fx_system_solve_batch(ctx.handle, C.byref(abi.as_struct(a)), C.byref(o), res.ctypes.data), "fx_system_solve_batch")
    finally:
        if register:
            ctx.host_unregister(*locked)
        ctx.set_batch_hints(one_structure=False)
    return a["vars"], res


def _same(x, y):
    assert np.array_equal(x[0].view(np.uint64), y[0].view(np.uint64))
    assert x[1].tobytes() == y[1].tobytes()


def test_in_place_transfers_give_the_bits_of_the_copied_call():
    """Every way through solve_host_in_place against the ordinary call: results page-locked or not, the LM-level entry point, options
    whose solve does not take the one-structure build (the values then go up by copies after all), gauge-fixed sketches."""
    ctx = fiksi_amd.Context(0)
    b = workloads.ring16(20000)
    plain = _solve(ctx, b, hint=False)
    _same(plain, _solve(ctx, b, hint=False, register=True))
    _same(plain, _solve(ctx, b, hint=False, register=True, register_results=False))
    _same(plain, _solve(ctx, b, hint=True, register=True, register_results=False))
    _same(_solve(ctx, b, hint=False, lm_level=True), _solve(ctx, b, hint=True, register=True, lm_level=True))
    for o in (abi.solving_opts(decomposer=1), abi.solving_opts(solver=1), abi.solving_opts(optimizer=1)):  # SinglePass, the refined step, L-BFGS
        _same(_solve(ctx, b, hint=False, opts=o), _solve(ctx, b, hint=True, register=True, opts=o))
    g = workloads.ring16(20000, fix_gauge=True)
    _same(_solve(ctx, g, hint=False), _solve(ctx, g, hint=True, register=True))
    h = workloads.hinged_triangles(17000, 11)  # (46 variables: the three-column build)
    _same(_solve(ctx, h, hint=False), _solve(ctx, h, hint=True, register=True))
    f = _solve(ctx, b, hint=False, opts=abi.solving_opts(f32=True))
    _same(f, _solve(ctx, b, hint=True, register=True, opts=abi.solving_opts(f32=True)))


@pytest.mark.parametrize("n", [3000, 70000])  # (one upload through the staging area; two chunks on two streams)
def test_hinted_call_equals_the_analysed_call_bit_for_bit(n):
    ctx = fiksi_amd.Context(0)
    for b in (workloads.ring16(n), workloads.hinged_triangles(n // 4, 11)):
        v0, r0 = _solve(ctx, b, hint=False)
        v1, r1 = _solve(ctx, b, hint=True, register=True)
        assert np.array_equal(v0.view(np.uint64), v1.view(np.uint64))
        assert r0.tobytes() == r1.tobytes()
        assert np.count_nonzero(r1["sse_unscaled"] < 1e-4) > 0.9 * len(r1)


@pytest.mark.parametrize("n", [3000, 70000])
def test_a_wrong_hint_is_found_out_and_the_batch_solved_the_ordinary_way(n):
    ctx = fiksi_amd.Context(0)
    b = workloads.ring16(n)
    b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    s = n - 7
    e = int(b["expr_off"][s])
    nv0 = int(b["var_off"][1])
    idx = b["expr_idx"].reshape(-1, 4)
    idx[e + 3, 1], idx[e + 5, 1] = idx[e + 5, 1], idx[e + 3, 1]  # two constraints of one System trade an operand: same sizes, other structure
    b["var_fixed"][(n // 2) * nv0 + 5] ^= 1                        # ... and another System pins one more coordinate
    v0, r0 = _solve(ctx, b, hint=False)
    v1, r1 = _solve(ctx, b, hint=True, register=True)
    assert np.array_equal(v0.view(np.uint64), v1.view(np.uint64))
    assert r0.tobytes() == r1.tobytes()


def test_hint_with_systems_beyond_one_wavefront_or_ragged_offsets_is_simply_not_taken():
    ctx = fiksi_amd.Context(0)
    for b in (workloads.hinged_triangles(3, 64), workloads.concat([workloads.ring16(50), workloads.hinged_triangles(50, 5)])):
        v0, r0 = _solve(ctx, b, hint=False)
        v1, r1 = _solve(ctx, b, hint=True)
        assert np.array_equal(v0.view(np.uint64), v1.view(np.uint64))
        assert r0.tobytes() == r1.tobytes()


def test_register_refuses_nonsense_and_unregister_of_unknown_memory_is_an_error_code():
    import ctypes as C
    from fiksi_amd._lib import lib
    ctx = fiksi_amd.Context(0)
    assert lib.fx_host_register(ctx.handle, None, 64) == -1
    a = np.zeros(1024)
    assert lib.fx_host_register(ctx.handle, a.ctypes.data, 0) == -1
    assert lib.fx_host_unregister(ctx.handle, a.ctypes.data) != 0  # never registered: the runtime's refusal as a code, no crash
    assert lib.fx_host_register(ctx.handle, a.ctypes.data, a.nbytes) == 0
    assert lib.fx_host_unregister(ctx.handle, a.ctypes.data) == 0


def test_in_place_with_the_tiny_build_and_across_shards():
    """The in-place transfers through the other kernel that honours them (fx_grouped_tiny.hip, with its hand-over of stragglers to the
    16-column build), and through fx_system_solve_batch_multi: every shard's slice of the registered arrays is solved in place."""
    from helpers import tiny_sketch_batches

    ctx = fiksi_amd.Context(0)
    for b in (workloads.hinged_triangles(20000, 1), dict(tiny_sketch_batches(17000))["quadrilateral_impossible"]):
        db = ctx.upload(b)
        assert db.grouped_build() == 4
        db.free()
        _same(_solve(ctx, b, hint=False), _solve(ctx, b, hint=True, register=True))
    b = workloads.ring16(70000)
    plain = _solve(ctx, b, hint=False)
    a = abi.normalize_batch({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
    res = np.zeros(len(a["var_off"]) - 1, dtype=abi.RESULT_DTYPE)
    ctxs = [ctx, fiksi_amd.Context(0)]
    for c in ctxs:
        c.set_batch_hints(one_structure=True)
    ctx.host_register(a["vars"], a["expr_param"], res)
    try:
        import ctypes as C
        from fiksi_amd._lib import check, lib
        handles = (C.c_void_p * 2)(*[c.handle for c in ctxs])
        total = (C.c_uint64 * 4)()
        o = abi.solving_opts()
        check(lib.fx_system_solve_batch_multi(handles, 2, C.byref(abi.as_struct(a)), C.byref(o), res.ctypes.data, total), "fx_system_solve_batch_multi")
    finally:
        ctx.host_unregister(a["vars"], a["expr_param"], res)
        for c in ctxs:
            c.set_batch_hints(one_structure=False)
    _same(plain, (a["vars"], res))
    assert int(total[0]) == 70000
