"""Thin Python objects over the C ABI (``include/fiksi_amd.h``): flat batches, device contexts,
HBM-resident batches. All numerics happen in the HIP kernels behind the ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import FxBatch, FxLmOpts, FxResult, FxSolvingOpts, check, lib

RESULT_DTYPE = np.dtype(
    [("accepted", "<u4"), ("trials", "<u4"), ("exit", "<u4"), ("ncomp", "<u4"),
     ("scale", "<f8"), ("sse0", "<f8"), ("sse", "<f8"), ("sse_unscaled", "<f8")]
)
assert RESULT_DTYPE.itemsize == C.sizeof(FxResult)

EXIT_SSE, EXIT_STEP, EXIT_FTOL, EXIT_MAX_OUTER, EXIT_TRIAL_CAP, EXIT_NAN = range(6)
NO_COMPONENT = 0xFFFF
HINT_ONE_STRUCTURE = 1  # (FX_HINT_ONE_STRUCTURE)

# fx_tag
(VARIABLE_VARIABLE_EQUALITY, POINT_POINT_DISTANCE, POINT_POINT_POINT_ANGLE, POINT_LINE_INCIDENCE,
 POINT_LINE_DISTANCE, POINT_CIRCLE_INCIDENCE, SEGMENT_SEGMENT_LENGTH_EQUALITY, LINE_LINE_ANGLE,
 LINE_LINE_PARALLELISM, LINE_LINE_PERPENDICULARITY, LINE_CIRCLE_TANGENCY) = range(11)

_FIELDS = {
    "var_off": np.uint32, "expr_off": np.uint32, "vars": np.float64, "var_fixed": np.uint8,
    "expr_tag": np.uint8, "expr_idx": np.uint32, "expr_param": np.float64,
    "var_comp": np.uint16, "expr_comp": np.uint16,
}


def normalize_batch(arrays: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Contiguous arrays of the ABI dtypes, keyed by the fx_batch field names. The C side trusts the
    pointers it is given, so the lengths are checked against the offsets here (ValueError otherwise)."""
    out = {}
    for k, dt in _FIELDS.items():
        a = arrays.get(k)
        if a is None:
            if k in ("var_comp", "expr_comp"):
                continue
            raise KeyError(f"batch is missing {k}")
        out[k] = np.ascontiguousarray(a, dtype=dt)
    if len(out["var_off"]) < 1 or len(out["var_off"]) != len(out["expr_off"]):
        raise ValueError("var_off and expr_off need n_systems + 1 entries each (at least one)")
    nv, ne = int(out["var_off"][-1]), int(out["expr_off"][-1])
    want = {"vars": nv, "var_fixed": nv, "var_comp": nv, "expr_tag": ne, "expr_param": ne, "expr_comp": ne, "expr_idx": 4 * ne}
    for k, n in want.items():
        if k in out and len(out[k]) != n:
            raise ValueError(f"{k} has {len(out[k])} entries, the offsets say {n}")
    return out


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def as_struct(arrays: Dict[str, np.ndarray]) -> FxBatch:
    """fx_batch struct pointing into the (kept alive by the caller) numpy arrays."""
    b = FxBatch()
    b.n_systems = len(arrays["var_off"]) - 1
    for k in _FIELDS:
        setattr(b, k, _ptr(arrays.get(k)))
    return b


def lm_opts(f32: bool = False, **kw) -> FxLmOpts:
    o = FxLmOpts()
    (lib.fx_lm_opts_default_f32 if f32 else lib.fx_lm_opts_default)(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solving_opts(perturb: bool = True, f32: bool = False, decomposer: int = 0, optimizer: int = 0, **lm_kw) -> FxSolvingOpts:
    """fx_solving_opts: decomposer 0 = None, 1 = SinglePass; optimizer 0 = LevenbergMarquardt, 1 = LBfgs."""
    o = FxSolvingOpts()
    lib.fx_solving_opts_default(C.byref(o))
    o.decomposer = decomposer
    o.optimizer = optimizer
    if f32:
        lib.fx_lm_opts_default_f32(C.byref(o.lm))
    o.perturb = 1 if perturb else 0
    for k, v in lm_kw.items():
        setattr(o.lm, k, v)
    return o


def device_count() -> int:
    n = C.c_int(0)
    rc = lib.fx_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def validate(arrays) -> None:
    a = normalize_batch(arrays)
    check(lib.fx_batch_validate(C.byref(as_struct(a))), "fx_batch_validate")


def jacobian_structure(arrays):
    """(row_ptr, col_idx) of the CSR Jacobian the device fills (host-side, no GPU needed)."""
    a = normalize_batch(arrays)
    st = as_struct(a)
    nnz = C.c_uint64(0)
    ne = int(a["expr_off"][-1]) if len(a["expr_off"]) else 0
    row_ptr = np.zeros(ne + 1, dtype=np.uint32)
    check(lib.fx_jacobian_structure(C.byref(st), C.byref(nnz), _ptr(row_ptr), None), "fx_jacobian_structure")
    col = np.zeros(nnz.value, dtype=np.uint32)
    check(lib.fx_jacobian_structure(C.byref(st), C.byref(nnz), _ptr(row_ptr), _ptr(col)), "fx_jacobian_structure")
    return row_ptr, col


def single_pass_blocks(arrays, system: int = 0):
    """The blocks Decomposer.SinglePass solves System ``system`` by, in solve order (host-side, no GPU
    needed): a list of (component, expression ids, free variable ids)."""
    a = normalize_batch(arrays)
    st = as_struct(a)
    if not 0 <= system < len(a["var_off"]) - 1:
        raise IndexError(f"system {system} out of range ({len(a['var_off']) - 1} systems)")
    ne = int(a["expr_off"][system + 1] - a["expr_off"][system])
    nv = int(a["var_off"][system + 1] - a["var_off"][system])
    nb = C.c_uint32(0)
    comp = np.zeros(4 * ne + 1, dtype=np.uint32)
    row_off = np.zeros(4 * ne + 1, dtype=np.uint32)
    rows = np.zeros(4 * ne + 1, dtype=np.uint32)
    var_off = np.zeros(4 * ne + 1, dtype=np.uint32)
    vs = np.zeros(max(nv, 1), dtype=np.uint32)
    check(lib.fx_single_pass_blocks(C.byref(st), system, C.byref(nb), _ptr(comp), _ptr(row_off), _ptr(rows),
                                    _ptr(var_off), _ptr(vs)), "fx_single_pass_blocks")
    return [(int(comp[k]), rows[row_off[k]:row_off[k + 1]].tolist(), vs[var_off[k]:var_off[k + 1]].tolist())
            for k in range(nb.value)]


def atan2_cr(y, x) -> np.ndarray:
    """The correctly rounded atan2 of the FX_STEP_QR kernels, host build (element-wise; no GPU needed)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert y.shape == x.shape
    out = np.zeros_like(y)
    lib.fx_atan2_cr_batch(y.size, _ptr(y), _ptr(x), _ptr(out))
    return out


def qr_symbolic(nrows: int, ncols: int, colptr, rowidx, colamd: bool = True):
    """SymbolicQr::build on a column pattern (host-side, no GPU needed): dict with col_perm, row_perm,
    h_ptr / h_rows (Householder vectors) and r_ptr / r_rows (columns of R)."""
    cp = np.ascontiguousarray(colptr, dtype=np.int32)
    ri = np.ascontiguousarray(rowidx, dtype=np.int32)
    col_perm = np.zeros(ncols, dtype=np.int32)
    row_perm = np.zeros(nrows, dtype=np.int32)
    h_ptr = np.zeros(ncols + 1, dtype=np.int32)
    r_ptr = np.zeros(ncols + 1, dtype=np.int32)
    h_cap, r_cap = nrows * ncols + 1, ncols * (ncols + 1) // 2 + 1
    h_rows = np.zeros(h_cap, dtype=np.int32)
    r_rows = np.zeros(r_cap, dtype=np.int32)
    check(lib.fx_qr_symbolic(nrows, ncols, _ptr(cp), _ptr(ri), 1 if colamd else 0, _ptr(col_perm), _ptr(row_perm),
                             _ptr(h_ptr), _ptr(h_rows), h_cap, _ptr(r_ptr), _ptr(r_rows), r_cap), "fx_qr_symbolic")
    return dict(col_perm=col_perm, row_perm=row_perm, h_ptr=h_ptr, h_rows=h_rows[: h_ptr[-1]].copy(), r_ptr=r_ptr,
                r_rows=r_rows[: r_ptr[-1]].copy())


class Context:
    """fx_ctx: one HIP device + stream. Raises FiksiError(FX_ERR_NO_DEVICE) without a gfx950 GPU."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(lib.fx_ctx_create(C.byref(h), device), "fx_ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib.fx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        check(lib.fx_ctx_device_name(self._h, buf, 256), "fx_ctx_device_name")
        return buf.value.decode()

    def set_routing(self, grouped: int = -1, grouped_min_systems: int = 0):
        """grouped: -1 by batch size (default), 0 never, 1 whenever the batch qualifies (fx_ctx_set_routing)."""
        check(lib.fx_ctx_set_routing(self._h, grouped, grouped_min_systems), "fx_ctx_set_routing")

    def set_presort(self, enable: bool = True, min_systems: int = 0):
        """Most-work-first hand-out of big batches from a scout pass (fx_ctx_set_presort); results unchanged."""
        check(lib.fx_ctx_set_presort(self._h, 1 if enable else 0, min_systems), "fx_ctx_set_presort")

    def set_ladder(self, enable: bool = True, tail_systems: int = 0xFFFFFFFF, min_trials: int = 8, spread: bool = True):
        """Grouped kernel: idle rows of a wavefront try the next lambdas of a running System side by side
        (fx_ctx_set_ladder); results unchanged."""
        check(lib.fx_ctx_set_ladder(self._h, 1 if enable else 0, tail_systems, min_trials, int(spread)), "fx_ctx_set_ladder")

    def set_one_structure_builds(self, enable: bool = True, tiny: bool = True):
        """Batches of one structure take the grouped kernel's builds made for them (fx_ctx_set_one_structure_builds); tiny = False
        keeps structures of at most eight variables / expressions on the 16-column build instead of fx_grouped_tiny.hip (same bits)."""
        check(lib.fx_ctx_set_one_structure_builds(self._h, (1 if tiny else 2) if enable else 0), "fx_ctx_set_one_structure_builds")

    def set_hold_passes(self, passes: int = 2):
        """Grouped kernel: passes a finished row waits for a second one (fx_ctx_set_hold_passes); results unchanged."""
        check(lib.fx_ctx_set_hold_passes(self._h, passes), "fx_ctx_set_hold_passes")

    def set_wide_routing(self, wide: int = -1):
        """Components of 65 ... 128 free variables: 1 wide kernel, 0 team kernels, -1 by cost (fx_ctx_set_wide_routing)."""
        check(lib.fx_ctx_set_wide_routing(self._h, wide), "fx_ctx_set_wide_routing")

    def set_sparse_fronts(self, enable: bool = True, ranks: int = 0):
        """Systems beyond one wavefront: the multifrontal build where the structure allows it (default), or the column walkers;
        ranks: lambda trials per launch of a large System alone (0: by the room on the chip, at most 8)."""
        check(lib.fx_ctx_set_sparse_fronts(self._h, 1 if enable else 0, ranks), "fx_ctx_set_sparse_fronts")

    def set_batch_hints(self, one_structure: bool = False):
        """Hints for the host-buffer calls to come (fx_ctx_set_batch_hints). one_structure: every System of a batch has System 0's
        structure — the library analyses System 0 alone and verifies the claim against every System while the device works; a
        batch that fails the check is solved again the ordinary way (the hint is never trusted)."""
        check(lib.fx_ctx_set_batch_hints(self._h, HINT_ONE_STRUCTURE if one_structure else 0), "fx_ctx_set_batch_hints")

    def host_register(self, *arrays):
        """Page-locks caller-owned numpy arrays (fx_host_register) so that the copies of a host-buffer call run at the link's rate
        without the staging buffer; unregister before freeing them."""
        for a in arrays:
            check(lib.fx_host_register(self._h, a.ctypes.data, a.nbytes), "fx_host_register")

    def host_unregister(self, *arrays):
        for a in arrays:
            check(lib.fx_host_unregister(self._h, a.ctypes.data), "fx_host_unregister")

    def set_host_threads(self, threads: int = 0):
        """Host threads for the sparse path's loops when a batch holds several large Systems (fx_ctx_set_host_threads)."""
        check(lib.fx_ctx_set_host_threads(self._h, threads), "fx_ctx_set_host_threads")

    def synchronize(self):
        check(lib.fx_ctx_synchronize(self._h), "fx_ctx_synchronize")

    def timer_begin(self):
        check(lib.fx_timer_begin(self._h), "fx_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float(0)
        check(lib.fx_timer_end(self._h, C.byref(ms)), "fx_timer_end")
        return float(ms.value)

    def upload(self, arrays) -> "DeviceBatch":
        return DeviceBatch(self, arrays)

    # ---- host-buffer entry points -------------------------------------------------------
    def system_solve_batch(self, arrays, opts: Optional[FxSolvingOpts] = None):
        """assemble::solve on host buffers; returns (solved vars, results)."""
        a = normalize_batch(arrays)
        a["vars"] = a["vars"].copy()
        res = np.zeros(len(a["var_off"]) - 1, dtype=RESULT_DTYPE)
        o = opts if opts is not None else solving_opts()
        check(lib.fx_system_solve_batch(self._h, C.byref(as_struct(a)), C.byref(o), _ptr(res)), "fx_system_solve_batch")
        return a["vars"], res

    @staticmethod
    def system_solve_batch_multi(contexts, arrays, opts: Optional[FxSolvingOpts] = None):
        """fx_system_solve_batch_multi: one batch sharded over several contexts (one per device, or several on one
        device), a host thread each. Returns (solved vars, results, {systems, converged, accepted, trials})."""
        a = normalize_batch(arrays)
        a["vars"] = a["vars"].copy()
        res = np.zeros(len(a["var_off"]) - 1, dtype=RESULT_DTYPE)
        o = opts if opts is not None else solving_opts()
        handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
        total = (C.c_uint64 * 4)()
        check(lib.fx_system_solve_batch_multi(handles, len(contexts), C.byref(as_struct(a)), C.byref(o), _ptr(res), total),
              "fx_system_solve_batch_multi")
        return a["vars"], res, dict(zip(("systems", "converged", "accepted", "trials"), (int(x) for x in total)))

    def lm_solve_batch(self, arrays, opts: Optional[FxLmOpts] = None):
        a = normalize_batch(arrays)
        a["vars"] = a["vars"].copy()
        res = np.zeros(len(a["var_off"]) - 1, dtype=RESULT_DTYPE)
        o = opts if opts is not None else lm_opts()
        check(lib.fx_lm_solve_batch(self._h, C.byref(as_struct(a)), C.byref(o), _ptr(res)), "fx_lm_solve_batch")
        return a["vars"], res

    def eval_residual_jacobian(self, arrays, want_jacobian: bool = True):
        a = normalize_batch(arrays)
        st = as_struct(a)
        ne = int(a["expr_off"][-1])
        r = np.zeros(ne, dtype=np.float64)
        if not want_jacobian:
            check(lib.fx_eval_residual_jacobian(self._h, C.byref(st), _ptr(r), None), "fx_eval_residual_jacobian")
            return r, None
        row_ptr, col = jacobian_structure(a)
        vals = np.zeros(len(col), dtype=np.float64)
        check(lib.fx_eval_residual_jacobian(self._h, C.byref(st), _ptr(r), _ptr(vals)), "fx_eval_residual_jacobian")
        return r, (row_ptr, col, vals)

    def eval_residual_dense_jacobian(self, arrays):
        """Problem::calculate_residuals_and_jacobian: (residuals, [dense row-major J of every System])."""
        a = normalize_batch(arrays)
        st = as_struct(a)
        n = len(a["var_off"]) - 1
        off = np.zeros(n + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        check(lib.fx_eval_residual_dense_jacobian(self._h, C.byref(st), None, None, _ptr(off), C.byref(total)),
              "fx_eval_residual_dense_jacobian")
        r = np.zeros(int(a["expr_off"][-1]) if n else 0, dtype=np.float64)
        jac = np.zeros(max(int(total.value), 1), dtype=np.float64)
        check(lib.fx_eval_residual_dense_jacobian(self._h, C.byref(st), _ptr(r), _ptr(jac), _ptr(off), C.byref(total)),
              "fx_eval_residual_dense_jacobian")
        blocks = []
        for s in range(n):
            m = int(a["expr_off"][s + 1] - a["expr_off"][s])
            blk = jac[int(off[s]):int(off[s + 1])]
            blocks.append(blk.reshape(m, -1) if m and len(blk) else blk.reshape(m, 0))
        return r, blocks

    def analyze_batch(self, arrays):
        """System::analyze per system: 1 per expression that over-constrains (does not increase the rank)."""
        a = normalize_batch(arrays)
        dep = np.zeros(max(int(a["expr_off"][-1]), 1), dtype=np.uint8)
        check(lib.fx_analyze_batch(self._h, C.byref(as_struct(a)), _ptr(dep)), "fx_analyze_batch")
        return dep[: int(a["expr_off"][-1])]

    def constraint_residuals(self, arrays):
        a = normalize_batch(arrays)
        r = np.zeros(int(a["expr_off"][-1]), dtype=np.float64)
        check(lib.fx_constraint_residuals(self._h, C.byref(as_struct(a)), _ptr(r)), "fx_constraint_residuals")
        return r


class DeviceBatch:
    """fx_dbatch: a batch resident in HBM, solved repeatedly from its start values."""

    def __init__(self, ctx: Context, arrays):
        self.ctx = ctx
        a = normalize_batch(arrays)
        self.n_systems = len(a["var_off"]) - 1
        self.n_vars = int(a["var_off"][-1]) if self.n_systems else 0
        self.n_exprs = int(a["expr_off"][-1]) if self.n_systems else 0
        h = C.c_void_p()
        check(lib.fx_batch_upload(ctx.handle, C.byref(as_struct(a)), C.byref(h)), "fx_batch_upload")
        self._h = h
        self.nnz = int(lib.fx_batch_nnz(h))

    def free(self):
        if self._h and self.ctx.handle:
            lib.fx_batch_free(self.ctx.handle, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def system_solve(self, opts: Optional[FxSolvingOpts] = None):
        o = opts if opts is not None else solving_opts()
        check(lib.fx_system_solve_device(self.ctx.handle, self._h, C.byref(o)), "fx_system_solve_device")

    def lm_solve(self, opts: Optional[FxLmOpts] = None):
        o = opts if opts is not None else lm_opts()
        check(lib.fx_lm_solve_device(self.ctx.handle, self._h, C.byref(o)), "fx_lm_solve_device")

    def phase_cycles(self, opts: Optional[FxSolvingOpts] = None):
        """Diagnostic: shader cycles per phase of the fused kernel, summed over wavefronts."""
        o = opts if opts is not None else solving_opts()
        c = (C.c_uint64 * 6)()
        check(lib.fx_debug_phase_cycles(self.ctx.handle, self._h, C.byref(o), c), "fx_debug_phase_cycles")
        return dict(zip(("setup", "eval", "form", "factor", "solve", "tail"), [int(x) for x in c]))

    def solve_route(self, opts=None) -> int:
        """0: one System per wavefront, 1: the grouped kernel (diagnostic; launches nothing)."""
        o = opts if opts is not None else solving_opts()
        r = C.c_int(0)
        check(lib.fx_debug_solve_route(self.ctx.handle, self._h, C.byref(o), C.byref(r)), "fx_debug_solve_route")
        return int(r.value)

    def grouped_build(self, opts=None) -> int:
        """-1: not the grouped kernel, 0: its general build, 1: its build for batches of one structure, 2: the sparse build for
        such batches, 3: build 1 over the big structure classes of a batch of several structures (diagnostic)."""
        o = opts if opts is not None else solving_opts()
        r = C.c_int(0)
        check(lib.fx_debug_grouped_build(self.ctx.handle, self._h, C.byref(o), C.byref(r)), "fx_debug_grouped_build")
        return int(r.value)

    def schedule_by_last_solve(self, enable: bool = True):
        """Later solves start the Systems that took the most LM trials in the last solve first (results unchanged)."""
        check(lib.fx_batch_schedule_by_last_solve(self.ctx.handle, self._h, 1 if enable else 0), "fx_batch_schedule_by_last_solve")

    def eval_residual_jacobian(self, which: int = 0):
        check(lib.fx_eval_residual_jacobian_device(self.ctx.handle, self._h, which), "fx_eval_residual_jacobian_device")

    def eval_residual(self, which: int = 0):
        check(lib.fx_eval_residual_device(self.ctx.handle, self._h, which), "fx_eval_residual_device")

    def set_params(self, expr_param):
        """New constraint targets (distances / angles) for the resident batch; structure unchanged."""
        v = np.ascontiguousarray(expr_param, dtype=np.float64)
        assert v.size == self.n_exprs
        check(lib.fx_batch_set_params(self.ctx.handle, self._h, _ptr(v)), "fx_batch_set_params")

    def set_vars(self, vars_):
        v = np.ascontiguousarray(vars_, dtype=np.float64)
        assert v.size == self.n_vars
        check(lib.fx_batch_set_vars(self.ctx.handle, self._h, _ptr(v)), "fx_batch_set_vars")

    def get_vars(self) -> np.ndarray:
        v = np.zeros(self.n_vars, dtype=np.float64)
        check(lib.fx_batch_get_vars(self.ctx.handle, self._h, _ptr(v)), "fx_batch_get_vars")
        return v

    def get_results(self) -> np.ndarray:
        r = np.zeros(self.n_systems, dtype=RESULT_DTYPE)
        check(lib.fx_batch_get_results(self.ctx.handle, self._h, _ptr(r)), "fx_batch_get_results")
        return r

    def get_residuals(self) -> np.ndarray:
        r = np.zeros(self.n_exprs, dtype=np.float64)
        check(lib.fx_batch_get_residuals(self.ctx.handle, self._h, _ptr(r)), "fx_batch_get_residuals")
        return r

    def get_jacobian_values(self) -> np.ndarray:
        v = np.zeros(self.nnz, dtype=np.float64)
        check(lib.fx_batch_get_jacobian_values(self.ctx.handle, self._h, _ptr(v)), "fx_batch_get_jacobian_values")
        return v
