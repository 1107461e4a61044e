//! `bindings/ffi.rs` — the Rust side of the drop-in boundary: what a `fiksi` crate
//! built with `--features amd` declares to link `libfiksi_amd.so`. Source only: the image this library is built in has
//! no Rust toolchain, so this file is checked WITHOUT cargo — `tools/check_ffi_layout.py` (run by
//! `tests/test_ffi_binding.py`) parses the `#[repr(C)]` structs below, lays them out by the C rules and has the C
//! compiler `_Static_assert` every offset and size against `include/fiksi_amd.h`, and checks that every function
//! declared here is declared there with the same argument count and is exported by the library.
//! The `extern "C"` block is generated from the header's prototypes (`python3 tools/check_ffi_layout.py --regen`).
#![allow(non_camel_case_types, dead_code)]
use core::ffi::{c_char, c_int, c_void};

/// Opaque: device + stream + scratch (one per host thread / device).
#[repr(C)] pub struct fx_ctx { _private: [u8; 0] }
/// Opaque: a batch resident in HBM.
#[repr(C)] pub struct fx_dbatch { _private: [u8; 0] }

pub const FX_OK: c_int = 0;
pub const FX_ERR_INVALID: c_int = -1;
pub const FX_ERR_NO_DEVICE: c_int = -2;
pub const FX_ERR_HIP: c_int = -3;
pub const FX_ERR_TOO_LARGE: c_int = -4;
pub const FX_ERR_NOMEM: c_int = -5;           // out of memory inside a call: a code, never an abort (the process is intact)
pub const FX_ERR_UNSUPPORTED: c_int = -6;
pub const FX_ERR_INTERNAL: c_int = -7;        // any other C++ exception, caught at the boundary: nothing unwinds into Rust
pub const FX_NO_COMPONENT: u16 = 0xFFFF;
pub const FX_STEP_CHOLESKY: u32 = 0;          // fx_lm_opts.solver
pub const FX_STEP_CHOLESKY_REFINED: u32 = 1;
pub const FX_STEP_QR: u32 = 2;                // the reference's own numerics (bit-identical iterates)
pub const FX_HINT_ONE_STRUCTURE: u32 = 1;     // fx_ctx_set_batch_hints: every System of a batch has System 0's structure (verified, never trusted)

/// One batch of independent Systems, struct-of-arrays == the numeric state of `fiksi::System` (lib.rs:256-303).
#[repr(C)] #[derive(Clone, Copy)]
pub struct fx_batch {
    pub n_systems: u32,
    pub var_off: *const u32,
    pub expr_off: *const u32,
    pub vars: *mut f64,
    pub var_fixed: *const u8,
    pub expr_tag: *const u8,
    pub expr_idx: *const u32,
    pub expr_param: *const f64,
    pub var_comp: *const u16,
    pub expr_comp: *const u16,
}

/// Levenberg-Marquardt constants; `fx_lm_opts_default` == the literals of lm.rs:108-189.
#[repr(C)] #[derive(Clone, Copy)]
pub struct fx_lm_opts {
    pub lambda0: f64,
    pub sse_tol: f64,
    pub step_tol: f64,
    pub ftol: f64,
    pub accept_factor: f64,
    pub reject_factor: f64,
    pub singular_factor: f64,
    pub lambda_min: f64,
    pub max_outer: u32,
    pub max_trials: u32,
    pub solver: u32,
    pub precision: u32,
}

/// `SolvingOptions` (lib.rs:205-237).
#[repr(C)] #[derive(Clone, Copy)]
pub struct fx_solving_opts {
    pub optimizer: u32,
    pub decomposer: u32,
    pub perturb: u32,
    pub plan_budget: u32,
    pub lm: fx_lm_opts,
}

/// Per-System outcome.
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct fx_result {
    pub accepted: u32,
    pub trials: u32,
    pub exit: u32,
    pub ncomp: u32,
    pub scale: f64,
    pub sse0: f64,
    pub sse: f64,
    pub sse_unscaled: f64,
}

/// What a sharded solve adds up to (`fx_system_solve_batch_multi`).
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct fx_throughput {
    pub systems: u64,
    pub converged: u64,
    pub accepted: u64,
    pub trials: u64,
}

#[link(name = "fiksi_amd")]
extern "C" {
    pub fn fx_abi_version() -> c_int;
    pub fn fx_last_error() -> *const c_char;
    pub fn fx_device_count(count: *mut c_int) -> c_int;
    pub fn fx_ctx_create(ctx: *mut *mut fx_ctx, device: c_int) -> c_int;
    pub fn fx_ctx_destroy(ctx: *mut fx_ctx);
    pub fn fx_ctx_set_routing(ctx: *mut fx_ctx, grouped: c_int, grouped_min_systems: u32) -> c_int;
    pub fn fx_ctx_set_presort(ctx: *mut fx_ctx, enable: c_int, min_systems: u32) -> c_int;
    pub fn fx_ctx_set_hold_passes(ctx: *mut fx_ctx, passes: u32) -> c_int;
    pub fn fx_ctx_set_one_structure_builds(ctx: *mut fx_ctx, enable: c_int) -> c_int;
    pub fn fx_ctx_set_ladder(ctx: *mut fx_ctx, enable: c_int, tail_systems: u32, min_trials: u32, spread: c_int) -> c_int;
    pub fn fx_ctx_set_wide_routing(ctx: *mut fx_ctx, wide: c_int) -> c_int;
    pub fn fx_ctx_set_sparse_fronts(ctx: *mut fx_ctx, enable: c_int, ranks: u32) -> c_int;
    pub fn fx_ctx_set_host_threads(ctx: *mut fx_ctx, threads: u32) -> c_int;
    pub fn fx_ctx_synchronize(ctx: *mut fx_ctx) -> c_int;
    pub fn fx_ctx_device_name(ctx: *mut fx_ctx, buf: *mut c_char, len: usize) -> c_int;
    pub fn fx_lm_opts_default(opts: *mut fx_lm_opts);
    pub fn fx_lm_opts_default_f32(opts: *mut fx_lm_opts);
    pub fn fx_solving_opts_default(opts: *mut fx_solving_opts);
    pub fn fx_batch_validate(batch: *const fx_batch) -> c_int;
    pub fn fx_jacobian_structure(batch: *const fx_batch, nnz: *mut u64, row_ptr: *mut u32, col_idx: *mut u32) -> c_int;
    pub fn fx_batch_upload(ctx: *mut fx_ctx, batch: *const fx_batch, out: *mut *mut fx_dbatch) -> c_int;
    pub fn fx_batch_free(ctx: *mut fx_ctx, db: *mut fx_dbatch);
    pub fn fx_batch_set_vars(ctx: *mut fx_ctx, db: *mut fx_dbatch, vars: *const f64) -> c_int;
    pub fn fx_batch_set_params(ctx: *mut fx_ctx, db: *mut fx_dbatch, expr_param: *const f64) -> c_int;
    pub fn fx_batch_schedule_by_last_solve(ctx: *mut fx_ctx, db: *mut fx_dbatch, enable: c_int) -> c_int;
    pub fn fx_batch_get_vars(ctx: *mut fx_ctx, db: *mut fx_dbatch, vars: *mut f64) -> c_int;
    pub fn fx_batch_get_results(ctx: *mut fx_ctx, db: *mut fx_dbatch, results: *mut fx_result) -> c_int;
    pub fn fx_batch_nnz(db: *const fx_dbatch) -> u64;
    pub fn fx_system_solve_device(ctx: *mut fx_ctx, db: *mut fx_dbatch, opts: *const fx_solving_opts) -> c_int;
    pub fn fx_lm_solve_device(ctx: *mut fx_ctx, db: *mut fx_dbatch, opts: *const fx_lm_opts) -> c_int;
    pub fn fx_eval_residual_jacobian_device(ctx: *mut fx_ctx, db: *mut fx_dbatch, which: c_int) -> c_int;
    pub fn fx_eval_residual_device(ctx: *mut fx_ctx, db: *mut fx_dbatch, which: c_int) -> c_int;
    pub fn fx_batch_get_residuals(ctx: *mut fx_ctx, db: *mut fx_dbatch, r: *mut f64) -> c_int;
    pub fn fx_batch_get_jacobian_values(ctx: *mut fx_ctx, db: *mut fx_dbatch, jvals: *mut f64) -> c_int;
    pub fn fx_timer_begin(ctx: *mut fx_ctx) -> c_int;
    pub fn fx_timer_end(ctx: *mut fx_ctx, milliseconds: *mut f32) -> c_int;
    pub fn fx_debug_phase_cycles(ctx: *mut fx_ctx, db: *mut fx_dbatch, opts: *const fx_solving_opts, cycles: *mut u64) -> c_int;
    pub fn fx_debug_solve_route(ctx: *mut fx_ctx, db: *mut fx_dbatch, opts: *const fx_solving_opts, route: *mut c_int) -> c_int;
    pub fn fx_debug_grouped_build(ctx: *mut fx_ctx, db: *mut fx_dbatch, opts: *const fx_solving_opts, build: *mut c_int) -> c_int;
    pub fn fx_system_solve_batch(ctx: *mut fx_ctx, batch: *const fx_batch, opts: *const fx_solving_opts, results: *mut fx_result) -> c_int;
    pub fn fx_host_register(ctx: *mut fx_ctx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn fx_host_unregister(ctx: *mut fx_ctx, ptr: *mut c_void) -> c_int;
    pub fn fx_ctx_set_batch_hints(ctx: *mut fx_ctx, hints: u32) -> c_int;
    pub fn fx_lm_solve_batch(ctx: *mut fx_ctx, batch: *const fx_batch, opts: *const fx_lm_opts, results: *mut fx_result) -> c_int;
    pub fn fx_system_solve_batch_multi(ctxs: *const *mut fx_ctx, n_ctx: u32, batch: *const fx_batch, opts: *const fx_solving_opts, results: *mut fx_result, total: *mut fx_throughput) -> c_int;
    pub fn fx_eval_residual_jacobian(ctx: *mut fx_ctx, batch: *const fx_batch, r: *mut f64, jvals: *mut f64) -> c_int;
    pub fn fx_eval_residual_dense_jacobian(ctx: *mut fx_ctx, batch: *const fx_batch, r: *mut f64, jac: *mut f64, jac_off: *mut u64, total: *mut u64) -> c_int;
    pub fn fx_analyze_batch(ctx: *mut fx_ctx, batch: *const fx_batch, dependent: *mut u8) -> c_int;
    pub fn fx_constraint_residuals(ctx: *mut fx_ctx, batch: *const fx_batch, r: *mut f64) -> c_int;
    pub fn fx_system_prepare_batch(ctx: *mut fx_ctx, batch: *const fx_batch, perturb: u32, out_vars: *mut f64, out_params: *mut f64, out_scale: *mut f64) -> c_int;
    pub fn fx_cluster_solve_batch(ctx: *mut fx_ctx, batch: *const fx_batch, opts: *const fx_lm_opts, results: *mut fx_result) -> c_int;
    pub fn fx_pose_transform_points(ctx: *mut fx_ctx, poses: *const f64, n_poses: u32, pose_of: *const u32, var_idx: *const u32, n_points: u32, vars: *mut f64, n_vars: u32) -> c_int;
    pub fn fx_unscale_vars(ctx: *mut fx_ctx, scale: f64, scaled: *const f64, mask: *const u8, vars: *mut f64, n: u32) -> c_int;
    pub fn fx_unscale_vars_strided(ctx: *mut fx_ctx, scales: *const f64, n_systems: u32, nvars: u32, scaled: *const f64, mask: *const u8, vars: *mut f64) -> c_int;
    pub fn fx_single_pass_blocks(batch: *const fx_batch, system: u32, n_blocks: *mut u32, block_comp: *mut u32, row_off: *mut u32, rows: *mut u32, var_off: *mut u32, vars: *mut u32) -> c_int;
    pub fn fx_atan2_cr_batch(n: u64, y: *const f64, x: *const f64, out: *mut f64);
    pub fn fx_qr_symbolic(nrows: i32, ncols: i32, colptr: *const i32, rowidx: *const i32, use_colamd: c_int, col_perm: *mut i32, row_perm: *mut i32, h_ptr: *mut i32, h_rows: *mut i32, h_cap: i32, r_ptr: *mut i32, r_rows: *mut i32, r_cap: i32) -> c_int;
}
