/*
 * fiksi_amd — C ABI of the MI355X-native batched geometric-constraint solver.
 *
 * Drop-in boundary for fiksi's numeric hot path (reference = endoli/fiksi, paths relative to the
 * reference tree). The reference has no FFI layer; its internal seams are
 *
 *   trait Problem                                   fiksi/src/solve/mod.rs:29-49
 *   levenberg_marquardt(problem, variables)         fiksi/src/solve/lm.rs:21
 *   Subsystem::new(vars, exprs, free, expr_ids)     fiksi/src/subsystem.rs:26-38
 *   assemble::solve(system, opts)  (None arm)       fiksi/src/assemble/mod.rs:46-167
 *   System::solve(&mut self, SolvingOptions)        fiksi/src/lib.rs:464-466
 *
 * and this header exports exactly those, batched over independent Systems:
 *
 *   fx_eval_residual_jacobian*  == Problem::calculate_residuals_and_sparse_jacobian (subsystem.rs:126-166)
 *   fx_eval_residual*           == Problem::calculate_residuals                    (subsystem.rs:93-104)
 *   fx_eval_residual_dense_jacobian == Problem::calculate_residuals_and_jacobian   (subsystem.rs:106-124)
 *   fx_lm_solve*                == levenberg_marquardt(Subsystem)                  (lm.rs:21-193)
 *   fx_system_solve*            == assemble::solve, Decomposer::None, LM           (assemble/mod.rs:46-167)
 *   fx_constraint_residuals*    == ConstraintHandle::calculate_residual            (constraints/mod.rs:88-110)
 *   fx_analyze_batch            == System::analyze (over-constraint detection)     (analyze/numerical/mod.rs:123-163)
 *   fx_single_pass_blocks       == find_strongly_connected_expressions             (analyze/graph/equations.rs:186-221)
 *
 * Conventions
 *  - Plain C: pointers + sizes, no C++/torch types. All functions return 0 (FX_OK) or a negative
 *    fx_status; nothing throws, aborts or panics across this boundary. A Rust shim turns codes
 *    into the reference's panics where the reference panics.
 *  - Every pointer in fx_batch is caller-owned HOST memory; the library copies. Objects created
 *    by fx_*_create / fx_batch_upload are freed with the matching destroy/free.
 *  - An fx_ctx is bound to one HIP device + one stream and is NOT thread-safe; use one ctx per
 *    host thread / GPU. Calls on different contexts are independent.
 *  - There is no CPU fallback: without a usable gfx950 device fx_ctx_create fails with
 *    FX_ERR_NO_DEVICE and every compute entry point needs a ctx.
 */
#ifndef FIKSI_AMD_H
#define FIKSI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_ABI_VERSION 1

typedef enum fx_status {
    FX_OK = 0,
    FX_ERR_INVALID = -1,     /* NULL pointer, inconsistent offsets, index out of range, bad tag */
    FX_ERR_NO_DEVICE = -2,   /* no usable HIP device (no CPU fallback exists)                   */
    FX_ERR_HIP = -3,         /* a HIP runtime call failed; see fx_last_error                    */
    FX_ERR_TOO_LARGE = -4,   /* a system exceeds FX_MAX_LARGE_SYSTEM_VARS                       */
    FX_ERR_NOMEM = -5,       /* host or device memory ran out; the call had no effect that must be undone */
    FX_ERR_UNSUPPORTED = -6,
    FX_ERR_INTERNAL = -7     /* anything else that went wrong inside the library (a C++ exception caught at
                                the boundary); the process is intact, see fx_last_error                    */
} fx_status;

/* Variant order of `enum Expression`, fiksi/src/constraints/expressions.rs:28-40. */
typedef enum fx_tag {
    FX_VARIABLE_VARIABLE_EQUALITY = 0,
    FX_POINT_POINT_DISTANCE = 1,
    FX_POINT_POINT_POINT_ANGLE = 2,
    FX_POINT_LINE_INCIDENCE = 3,
    FX_POINT_LINE_DISTANCE = 4,
    FX_POINT_CIRCLE_INCIDENCE = 5,
    FX_SEGMENT_SEGMENT_LENGTH_EQUALITY = 6,
    FX_LINE_LINE_ANGLE = 7,
    FX_LINE_LINE_PARALLELISM = 8,
    FX_LINE_LINE_PERPENDICULARITY = 9,
    FX_LINE_CIRCLE_TANGENCY = 10,
    FX_NUM_TAGS = 11,
    /* Rows of a RecursiveAssembly cluster problem (fiksi/src/assemble/mod.rs:547-588), accepted by
     * fx_cluster_solve_batch only: a point of an already solved cluster, moved by the cluster's pose
     * (Pose2D, constraints/expressions.rs:1094-1159), has to land on the point's new position.
     * expr_idx: {pose (rotation, tx, ty: three consecutive variables), the point as solved so far (x; y is +1;
     * fixed variables), the x (_X) or y (_Y) variable of the new position, 0}. */
    FX_POSE_COINCIDENCE_X = 11,
    FX_POSE_COINCIDENCE_Y = 12
} fx_tag;

/* One-wavefront limits of the fused solve kernel (one wavefront per System). Systems within them
 * are solved thousands at a time; up to FX_MAX_WIDE_FREE_VARS free variables per component a second
 * LDS-resident kernel takes over (f64 LM without a decomposer); a System beyond that (more free
 * variables or expressions in a component, or more variables) is solved by the sparse large-sketch
 * path, one at a time. */
#define FX_MAX_FREE_VARS 64u          /* free variables (Jacobian columns) per component          */
#define FX_MAX_WIDE_FREE_VARS 128u    /* ... per component, wide kernel                            */
#define FX_MAX_ROWS 256u              /* expressions (Jacobian rows) per component                */
#define FX_MAX_SYSTEM_VARS 512u       /* variables (free + fixed) per System, fused kernel        */
#define FX_MAX_LARGE_SYSTEM_VARS 65535u /* variables per System, sparse path (16-bit local indices) */
#define FX_NO_COMPONENT 0xFFFFu

/*
 * One batch of independent Systems in flat, struct-of-arrays form == the numeric state of
 * fiksi::System after building (fiksi/src/lib.rs:256-303): `variables`, `expressions`,
 * `fixed_variables`, and the connected components of `graph` (fiksi/src/graph.rs).
 *
 *  expr_idx: 4 entries per expression = the element fields of the Rust expression struct in
 *  declaration order, each the SYSTEM-LOCAL variable index of a point's x (y is +1) or of a
 *  length; unused entries 0. E.g. PointPointDistance {p1, p2}; PointCircleIncidence {point,
 *  center, radius}; LineCircleTangency {line p1, line p2, center, radius};
 *  VariableVariableEquality {variable1, variable2}.
 *  expr_param: `distance` / `angle` of the variant, 0 if none.
 *  var_comp / expr_comp: index of the connected component (order of
 *  Graph::connected_components(), empties skipped) the variable's element / the expression's
 *  constraint belongs to; FX_NO_COMPONENT for variables of unconstrained elements. Either may be
 *  NULL: then every variable and expression is in component 0.
 *  Row order inside a component = ascending expression index; column order = ascending index of
 *  its non-fixed variables (assemble/mod.rs:91-111, :132-146).
 */
typedef struct fx_batch {
    uint32_t n_systems;
    const uint32_t* var_off;   /* [n_systems+1] offsets into vars / var_fixed / var_comp        */
    const uint32_t* expr_off;  /* [n_systems+1] offsets into expr_*                             */
    double* vars;              /* in: start values; out (host entry points): solved values      */
    const uint8_t* var_fixed;  /* 1 = member of System::fixed_variables                         */
    const uint8_t* expr_tag;   /* fx_tag                                                        */
    const uint32_t* expr_idx;  /* [4 * n_exprs]                                                 */
    const double* expr_param;  /* [n_exprs]                                                     */
    const uint16_t* var_comp;  /* [n_vars] or NULL                                              */
    const uint16_t* expr_comp; /* [n_exprs] or NULL                                             */
} fx_batch;

/* Which linear solve the LM step uses. */
typedef enum fx_step_solver {
    FX_STEP_CHOLESKY = 0, /* (JtJ + lambda I) delta = -Jt r, dense Cholesky per wavefront       */
    FX_STEP_CHOLESKY_REFINED = 1, /* + one refinement step on the least-squares problem itself
                                    (corrected semi-normal equations: the residual -r - J delta comes
                                    from the Jacobian rows, not from JtJ) — the step then has the
                                    accuracy of the reference's QR on ill-conditioned sketches, for
                                    about a fifth more time. Every path honours it (fused, wide, walker,
                                    sparse) except the grouped kernel, whose batches then run one System
                                    per wavefront.                                                     */
    FX_STEP_QR = 2 /* the reference's own numerics (lm.rs:98-132, solvi qr.rs:226-356): Householder QR of
                      [J; sqrt(lambda) I] in the reference's column order (COLAMD, computed on the host) and row
                      order, every sum taken in the reference's order, nothing fused. The LM path — trial counts,
                      every iterate — is bit-identical to the reference algorithm's on all eleven expression
                      kinds, relative to a correctly rounded atan2: the QR kernels evaluate the two angle
                      residuals with fx_atan2.h's correctly rounded routine (a platform libm's atan2 — the
                      reference's — may be an ulp off on ~7e-4 of arguments; see fx_atan2_cr_batch). f64,
                      Levenberg-Marquardt. Components of up to FX_MAX_FREE_VARS free variables on the one-wavefront
                      kernels (batches of one structure with at most 32 columns: four Systems per wavefront);
                      Decomposer::None Systems of at most 512 variables whose components have up to 128 free variables
                      and 256 expressions on the wide kernel's QR build; larger Systems, and SinglePass blocks
                      beyond one wavefront, run FX_STEP_CHOLESKY_REFINED instead.
                      Two to six times the cost of FX_STEP_CHOLESKY.                                        */
} fx_step_solver;

/* Levenberg-Marquardt constants; fx_lm_opts_default() == the literals of lm.rs:108-189. */
typedef struct fx_lm_opts {
    double lambda0;       /* 0.5     lm.rs:108                                                  */
    double sse_tol;       /* 1e-8    lm.rs:110   stop when SSE < sse_tol                        */
    double step_tol;      /* 1e-12   lm.rs:140   stop when |delta|^2 < step_tol                 */
    double ftol;          /* 1e-6    lm.rs:164   stop when relative SSE decrease <= ftol        */
    double accept_factor; /* 0.125   lm.rs:153                                                  */
    double reject_factor; /* 2       lm.rs:189                                                  */
    double singular_factor; /* 8     lm.rs:135                                                  */
    double lambda_min;    /* 1e-50   lm.rs:154-156                                              */
    uint32_t max_outer;   /* 100     lm.rs:109                                                  */
    uint32_t max_trials;  /* 4096    (reference: unbounded) total LM trials    */
    uint32_t solver;      /* fx_step_solver                                                     */
    uint32_t precision;   /* 0 or 64: f64 (the reference's arithmetic); 32: f32 compute (cfg5) —
                             HBM arrays stay f64, scale + perturbation stay f64. Applies to Systems
                             within the one-wavefront limits; larger ones are computed in f64 on
                             their own device paths with the same options                         */
} fx_lm_opts;

/* SolvingOptions (fiksi/src/lib.rs:205-237). optimizer: 0 = LevenbergMarquardt, 1 = LBfgs
 * (solve/lbfgs.rs; f64 only; fx_result.accepted counts its
 * iterations, .trials its residual+Jacobian evaluations); decomposer: 0 = None, 1 = SinglePass (assemble/mod.rs:169-210:
 * maximum matching + strongly connected blocks, solved one after the other); 2 =
 * RecursiveAssembly: through the builder only (fxs_system_solve, include/fiksi_amd_builder.h) — its plan needs the
 * System's elements, which a flat batch does not carry; the batch entry points answer FX_ERR_UNSUPPORTED. */
typedef struct fx_solving_opts {
    uint32_t optimizer;
    uint32_t decomposer;
    uint32_t perturb;  /* 1 = LCG perturbation, seed 42 (assemble/mod.rs:47,113-124) */
    uint32_t plan_budget; /* RecursiveAssembly only: subgraphs its plan search may grow, in thousands (0 = 200, the
                             default; the reference's search is unbounded). Otherwise ignored — leave it 0.   */
    fx_lm_opts lm;
} fx_solving_opts;

/* Why a component's LM loop ended. */
typedef enum fx_exit {
    FX_EXIT_SSE = 0,       /* SSE < sse_tol                                                     */
    FX_EXIT_STEP = 1,      /* |delta|^2 < step_tol                                              */
    FX_EXIT_FTOL = 2,      /* relative decrease <= ftol                                         */
    FX_EXIT_MAX_OUTER = 3, /* max_outer accepted steps used                                     */
    FX_EXIT_TRIAL_CAP = 4, /* max_trials reached (the reference would keep looping)            */
    FX_EXIT_NAN = 5        /* non-finite SSE or step (the reference would loop forever)        */
} fx_exit;

/* Per-System outcome. Counters are summed over the System's components. */
typedef struct fx_result {
    uint32_t accepted; /* accepted LM steps == Gauss-Newton iterations (one J evaluation each) */
    uint32_t trials;   /* factor + solve + trial-residual evaluations                          */
    uint32_t exit;     /* fx_exit of the last component                                        */
    uint32_t ncomp;    /* components solved                                                    */
    double scale;      /* system scale (assemble/mod.rs:32-44); 1 for fx_lm_solve*             */
    double sse0;       /* SSE at the start point (scaled space), summed over components        */
    double sse;        /* SSE at the returned point (scaled space), summed over components     */
    double sse_unscaled; /* sum of squared expression residuals on the solved, UNSCALED variables
                            (constraints/mod.rs:96-109; the bench's `converged` test,
                            fiksi/benches/fiksi_bench.rs:65-72)                                 */
} fx_result;

typedef struct fx_ctx fx_ctx;       /* device + stream + scratch                               */
typedef struct fx_dbatch fx_dbatch; /* a batch resident in HBM (+ its CSR Jacobian structure)  */

/* ---- library / lifecycle ------------------------------------------------------------------ */
int fx_abi_version(void);
const char* fx_last_error(void);           /* thread-local text of the last failure           */
int fx_device_count(int* count);           /* HIP devices visible to this process              */
int fx_ctx_create(fx_ctx** ctx, int device);
void fx_ctx_destroy(fx_ctx* ctx);
/* Which kernel batches of small Systems (components of at most 48 free variables) take: grouped = -1 (default): the
 * grouped kernel — four Systems per wavefront — from grouped_min_systems Systems on (default 8; 0 keeps the current value),
 * one wavefront per System below; 0: never the grouped kernel; 1: whenever the batch qualifies. (Round 4: since the rows of a
 * wavefront that have run out of Systems help the ones still running — fx_ctx_set_ladder — a small batch of sketches of
 * uneven difficulty is as slow as its slowest System's LADDER, not its trial count: 8 ring16 sketches 0.19 -> 0.15 ms,
 * 256: 0.75 -> 0.33 ms, 1000: 0.80 -> 0.40 ms, and ONE sketch that takes 97 trials 0.92 -> 0.39 ms — but one that takes 3
 * trials 0.051 -> 0.067 ms, which is why a handful of Systems stay on the lighter kernel; a small batch of identical Systems,
 * which has no slowest one, loses up to a third — 500 four-triangle sketches 0.041 -> 0.058 ms — and may pin 0. Up to round 3
 * the threshold was 1024.) Results do not
 * depend on it beyond the last bits of sums of LDS float atomics on sketches where several rows add into one entry.
 * A new context starts from FIKSI_AMD_GROUPED=0|1 if that is set in the environment.
 * A batch whose Systems all have ONE structure (one component of at most 48 free variables) runs the grouped kernel's build for
 * such batches (fx_grouped_c.hip: four / two wavefronts per SIMD up to 16 / 32 free variables; 100 000 ring16 sketches 2.81 -> 1.76 ms, same bits) — nothing to set;
 * fx_debug_grouped_build tells, FIKSI_AMD_GROUPED_C=0 in the environment of fx_ctx_create keeps the general build. */
int fx_ctx_set_routing(fx_ctx* ctx, int grouped, uint32_t grouped_min_systems);
/* Batches the grouped kernel takes, of min_systems Systems or more (default 8192; 0 keeps the current value): a scout
 * pass (residuals at the start values) and a ranking of strided chunks hand the Systems out most-work-first, because a batch is as slow
 * as its slowest System plus the time before that System was started (~0.035 ms per 100 000 Systems; enable = 0 turns it
 * off). Scheduling only: every System's result is the same bits either way. */
int fx_ctx_set_presort(fx_ctx* ctx, int enable, uint32_t min_systems);
/* Grouped kernel: a row of a wavefront that has finished its System (SinglePass: its block) waits up to `passes` trial passes (default 2) for a
 * second row to finish, so that the two take their next Systems side by side — the hand-over blocks cost the wavefront
 * the same for one row as for four. 0: never wait. Scheduling only: every System's result is the same bits either way. */
int fx_ctx_set_hold_passes(fx_ctx* ctx, uint32_t passes);
/* Batches whose Systems all have ONE structure (one component) run builds of the grouped kernel made for them: up to 48 free
 * variables fx_grouped_c.hip (the structure's lists shared by a wavefront, Jt J by its pattern, up to four wavefronts per SIMD:
 * the same bits as the general build); 33 ... 255 free variables — from 33 on when the Cholesky factor has at most a quarter of the
 * dense triangle's entries, always from 49 on; at most 255 variables and 255 expressions, a factor of at most 1023 slots, at most
 * 1023 compact Jacobian entries — fx_grouped_s.hip
 * (the factorisation as a level schedule over tables in LDS: the normal-equation step in a minimum-degree order — the same
 * counters as the team / wide kernels such batches took before, variables to round-off). enable = 0 keeps such batches on the
 * general paths (default 1; a context created under FIKSI_AMD_GROUPED_C=0 starts with 0); enable = 2: as 1, but structures of at
 * most eight variables and eight expressions stay on fx_grouped_c.hip's 16-column build instead of fx_grouped_tiny.hip (same bits
 * either way: A / B measurements). fx_debug_grouped_build tells. */
int fx_ctx_set_one_structure_builds(fx_ctx* ctx, int enable);
/* (These builds need a batch — or, under fx_system_solve_batch_multi, a SHARD — of at least 8 Systems (2 for fx_grouped_c.hip's
 * one-structure detection). fx_grouped_c.hip gives the general build's bits, so its routing never shows; fx_grouped_s.hip sums in a
 * minimum-degree order and agrees with the team / wide kernels to round-off only, so a shard of fewer than 8 Systems of 33 ... 255
 * variables each takes another kernel than its siblings and differs from them in the last bits. Shards that small mean more
 * devices than work; pin fx_ctx_set_one_structure_builds(ctx, 0) on every context if the last bits must not depend on it.) */
/* Grouped kernel, the lambda ladder. The trials that follow a rejected trial of the reference's loop (lm.rs:187-190) read
 * the same point, Jacobian and residuals and differ in lambda only (x 2 each), so a row of a wavefront that has no System
 * of its own tries the next lambda of a System that is still running in its wavefront, in the same pass: up to four
 * lambdas side by side. The verdicts are read in the loop's order and the first one that is not a plain reject decides, as it
 * would have in the sequential loop; `trials` counts what that loop would have counted. enable = 0: off. Rows help once
 * the queue of Systems is empty; with tail_systems > 0 a wavefront that holds a System past min_trials trials also stops
 * taking new Systems when at most tail_systems are left in the queue (its rows go over to the straggler as they finish;
 * FX_LADDER_TAIL_AUTO: eight Systems per row the device holds at once, 32 768 for the headline shape).
 * spread != 0: with a scheduled hand-out (presort, fx_batch_schedule_by_last_solve) the first round of Systems is dealt one
 * per wavefront instead of four, so that the likely stragglers do not share a wavefront.
 * Defaults: on, FX_LADDER_TAIL_AUTO, 8, on (100 000 ring16 Systems 3.1 -> 2.9 ms, a shard of 12 500 1.1 -> 0.6 ms).
 * Scheduling only: every System's result is the same bits either way (tests/test_gpu_grouped.py). */
#define FX_LADDER_TAIL_AUTO 0xFFFFFFFFu
int fx_ctx_set_ladder(fx_ctx* ctx, int enable, uint32_t tail_systems, uint32_t min_trials, int spread);
/* Where components of 65 ... 128 free variables are solved: 1 = the wide kernel (one wavefront per System, dense factor in LDS:
 * the faster one for thousands of them), 0 = the team kernels (a workgroup per System, sparse factor: half the latency of
 * one solve, and faster at any count from ~112 columns on), -1 (default) = by the measured cost of either for the batch at
 * hand. Both follow the reference's iteration path; they add in different orders, so the choice shows in the last bits of
 * a result — pin it when results must not depend on how many such Systems share a batch. */
int fx_ctx_set_wide_routing(fx_ctx* ctx, int wide);
/* Systems beyond one wavefront (more than 128 free variables in a component, or any System the team kernels take): the
 * MULTIFRONTAL build (fx_front.h) where the structure allows it — the elimination tree cut into fronts of at most 15
 * columns, a front factored in the registers of one row of 16 lanes, four fronts per wavefront; chain-like sketches
 * (BASELINE's 5 000-point sketch, the reference's hinged triangles) qualify, a structure with a larger front keeps the
 * column walkers of fx_sparse_team.h. The same normal-equation step (lm.rs:28-63) in another summation order: the
 * reference's iteration path, last bits differ. enable = 0 keeps the walkers (default 1; FIKSI_AMD_FRONTS=0 likewise).
 * ranks: a large System alone (its tree cut into parts + top, two launches per trial) leaves most of the chip idle, and the
 * trials that follow a rejected trial of lm.rs:114-190 differ in lambda only — a launch makes `ranks` of them side by side
 * (lambda x reject_factor^k) and the decision reads their verdicts in order: every counter, lambda and accepted point is the
 * sequential loop's, in fewer launches (BASELINE's large sketch: 89 trials in 31 rounds). 0 = as many as the chip has
 * room for (at most 8: one and a half workgroups per CU over the parts x ranks), 1 = one trial per launch. */
int fx_ctx_set_sparse_fronts(fx_ctx* ctx, int enable, uint32_t ranks);
/* Systems beyond the one-wavefront kernels are grouped by structure, and every launch carries a whole group (fx_sparse_team.h).
 * A batch with SEVERAL structures solves its groups side by side on this many host threads, a stream each (default 8; 0 restores
 * the default, 1 = one after the other on the caller's thread and the context's stream). Results do not depend on it.
 * (Profilers that intercept launches may not cope with concurrent launching threads: profile with 1.) */
int fx_ctx_set_host_threads(fx_ctx* ctx, uint32_t threads);
int fx_ctx_synchronize(fx_ctx* ctx);
int fx_ctx_device_name(fx_ctx* ctx, char* buf, size_t len);

void fx_lm_opts_default(fx_lm_opts* opts);           /* lm.rs:108-189 literals                */
/* f32 variant: same schedule, ftol 1e-4, lambda_min 1e-7 and max_outer 40 (what f32 round-off can resolve). With precision = 32
 * ftol also ends a solve on a REJECTED trial whose SSE exceeds the current one by no more than ftol * SSE: that is
 * round-off, not a worse point (the f64 path keeps the reference's rule: only accepted steps test ftol). */
void fx_lm_opts_default_f32(fx_lm_opts* opts);
void fx_solving_opts_default(fx_solving_opts* opts); /* SolvingOptions::DEFAULT, lib.rs:232-236 */

/* ---- host-side validation / structure (no device needed) ----------------------------------- */
/* Validate offsets, tags, indices and the per-wavefront limits. */
int fx_batch_validate(const fx_batch* batch);
/* CSR structure of the batch Jacobian that fx_eval_residual_jacobian* fills: global rows (row of
 * system s start at expr_off[s]), system-local free-column indices ascending inside a row,
 * duplicate columns merged (== TripletMat -> SparseColMat::from_triplet_mat semantics,
 * solvi/src/sparse_col_mat.rs:690-737, transposed), fixed variables dropped
 * (subsystem.rs:159-164). row_ptr has n_exprs+1 entries. Pass col_idx == NULL to query nnz. */
int fx_jacobian_structure(const fx_batch* batch, uint64_t* nnz, uint32_t* row_ptr, uint32_t* col_idx);

/* ---- device-resident batches ---------------------------------------------------------------- */
int fx_batch_upload(fx_ctx* ctx, const fx_batch* batch, fx_dbatch** out);
void fx_batch_free(fx_ctx* ctx, fx_dbatch* db);
/* Replace the start values of an uploaded batch (n_vars doubles). */
int fx_batch_set_vars(fx_ctx* ctx, fx_dbatch* db, const double* vars);
/* Replace the expression parameters (distances / angles; n_exprs doubles) of an uploaded batch — the
 * resident counterpart of ConstraintHandle::update_parameter (constraints/mod.rs:992-1046): the
 * structure stays, only the targets change (dragging a dimension). */
int fx_batch_set_params(fx_ctx* ctx, fx_dbatch* db, const double* expr_param);
/* A resident batch that is solved again and again with nearly the same data (dragging a dimension: fx_batch_set_params +
 * solve, many times) is as slow as its slowest System plus the time before that System was started. enable != 0:
 * later solves of this batch hand the Systems out in descending order of the LM trials each took in the LAST solve
 * (read back here, once), so yesterday's stragglers start first; enable == 0: index order again. Scheduling only:
 * every System's result is bit-identical either way (Systems are independent). Used by the grouped kernel. */
int fx_batch_schedule_by_last_solve(fx_ctx* ctx, fx_dbatch* db, int enable);
/* Copy the current (last solved) values / results back. */
int fx_batch_get_vars(fx_ctx* ctx, fx_dbatch* db, double* vars);
int fx_batch_get_results(fx_ctx* ctx, fx_dbatch* db, fx_result* results);
uint64_t fx_batch_nnz(const fx_dbatch* db);

/* Asynchronous on the ctx stream; inputs are the batch's start values, outputs stay in HBM. */
int fx_system_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts);
int fx_lm_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_lm_opts* opts);
/* r (n_exprs) and jvals (nnz) are DEVICE buffers owned by the dbatch; fetch with the getters.
 * which: 0 = evaluate at the start values, 1 = at the last solved values. */
int fx_eval_residual_jacobian_device(fx_ctx* ctx, fx_dbatch* db, int which);
int fx_eval_residual_device(fx_ctx* ctx, fx_dbatch* db, int which);
int fx_batch_get_residuals(fx_ctx* ctx, fx_dbatch* db, double* r);
int fx_batch_get_jacobian_values(fx_ctx* ctx, fx_dbatch* db, double* jvals);

/* HIP-event timing on the ctx stream (bench.py: "measured live ... on the stream the kernel is
 * launched on"). begin/end bracket any number of *_device calls; end synchronizes. */
int fx_timer_begin(fx_ctx* ctx);
int fx_timer_end(fx_ctx* ctx, float* milliseconds);

/* Diagnostic only: shader cycles per phase {setup, eval, form, factor, solve, tail}, summed over all
 * wavefronts, from a stamped build of the fused kernel (32-free-variable shape only). For a batch the
 * grouped kernel takes, the sums are lane 0's: the first System row's view of its wavefront's time. */
int fx_debug_phase_cycles(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, uint64_t cycles[6]);
/* Diagnostic only: which kernel fx_system_solve_device would launch for the Systems of up to 64 free
 * variables of this batch with these options: 0 = lm_solve_kernel (one System per wavefront),
 * 1 = the grouped kernel (four Systems per wavefront, fx_grouped.hip). Launches nothing. */
int fx_debug_solve_route(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* route);
/* Diagnostic only: which BUILD of the grouped kernel such a launch would be: -1 = not the grouped kernel, 0 = the general build
 * (one wavefront per SIMD for components of 17 ... 32 free variables), 1 = the build for batches of one structure
 * (fx_grouped_c.hip: the structure's lists shared by a wavefront's four Systems, Jt J by its pattern, two wavefronts per
 * SIMD for that shape; same bits), 2 = the sparse build for batches of one structure with a component of 33 ... 255 free
 * variables and a small Cholesky factor (fx_grouped_s.hip: the factorisation as a level schedule over tables in LDS; the limits:
 * fx_ctx_set_one_structure_builds).
 * 3 = a batch of SEVERAL structures whose big structure classes (256 Systems and more, up to eight, when they hold three quarters of
 * the batch between them) run build 1 in one launch,
 * everyone else the general build. 4 = build 1's arithmetic for structures of at most eight variables and eight expressions
 * (fx_grouped_tiny.hip: eight lanes per System, eight Systems per wavefront; same bits as build 1;
 * fx_ctx_set_one_structure_builds(ctx, 2) keeps such batches on build 1). A context created under FIKSI_AMD_GROUPED_C=0 takes
 * none of 1, 2, 3, 4. Launches nothing. */
int fx_debug_grouped_build(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* build);

/* ---- host-buffer entry points (upload -> run -> download; PCIe inclusive) ------------------- */
/* == assemble::solve: batch->vars in: unscaled values, out: solved values. results may be NULL.
 * Systems beyond the one-wavefront kernels need a plan (ordering, symbolic factorisation, gather lists): the context keeps
 * the plans of the last eight structures + solve modes these one-shot entry points have seen, so that System::solve on the
 * same large sketch with new values pays for it once (a plan's index arrays stay in device memory until it is dropped;
 * fx_ctx_destroy frees them all). */
int fx_system_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* opts, fx_result* results);
/* What a host that calls the entry points above again and again on the same buffers can do for them. Both are optional and
 * change no result.
 *  - fx_host_register / fx_host_unregister: page-locks caller-owned memory (the arrays a batch points to, the result array) for
 *    the device's copy engines: copies to and from registered memory run at the bus's rate instead of through a staging
 *    buffer (100 000 ring16 sketches: 82 MB per call). The memory stays the caller's; unregister before freeing it.
 *  - fx_ctx_set_batch_hints(ctx, FX_HINT_ONE_STRUCTURE): the caller says that every System of the batches to come has the
 *    structure of the first (one sketch, many parameter sets). The library does not believe it: the comparison of every
 *    System with the first — a pass of the host over 100 MB that otherwise comes before anything is uploaded — runs while
 *    the device already works, and a batch that turns out not to be of one structure is solved again the ordinary way before
 *    anything is written to the caller's arrays. 0 clears the hints. */
#define FX_HINT_ONE_STRUCTURE 1u
int fx_host_register(fx_ctx* ctx, void* ptr, size_t bytes);
int fx_host_unregister(fx_ctx* ctx, void* ptr);
int fx_ctx_set_batch_hints(fx_ctx* ctx, uint32_t hints);
/* == levenberg_marquardt(Subsystem): values used as given (already scaled/perturbed). */
int fx_lm_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results);

/* What a sharded solve adds up to (the counters one all-reduce would carry). */
typedef struct fx_throughput {
    uint64_t systems;   /* Systems solved                                                          */
    uint64_t converged; /* ... with sum r^2 < 1e-4 on the unscaled variables (fiksi_bench.rs:65-72) */
    uint64_t accepted;  /* Gauss-Newton iterations                                                 */
    uint64_t trials;    /* LM trials                                                               */
} fx_throughput;
/* == fx_system_solve_batch over several devices: shard r of n_ctx = Systems [r N / n_ctx, (r + 1) N / n_ctx), solved by
 * context r on its own device from its own host thread (independent Systems, no data-path collective — xGMI
 * carries nothing); batch->vars is solved in place shard by shard; results[N] (may be NULL) and *total (may be NULL: the
 * counters summed on the host) are filled. Every result is the bits a one-device solve gives: the shard rule does not
 * enter a System's arithmetic. The contexts must be distinct (one per device, or several on one device). */
int fx_system_solve_batch_multi(fx_ctx* const* ctxs, uint32_t n_ctx, const fx_batch* batch, const fx_solving_opts* opts,
                                fx_result* results, fx_throughput* total);
/* == Problem::calculate_residuals_and_sparse_jacobian at batch->vars; jvals in
 * fx_jacobian_structure order (may be NULL for residuals only). */
int fx_eval_residual_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jvals);
/* == Problem::calculate_residuals_and_jacobian (subsystem.rs:106-124), the dense variant L-BFGS and
 * analyze use: per System a row-major [n_exprs_s x n_free_s] block at jac[jac_off[s]], columns = the
 * free rank of fx_jacobian_structure; partials of fixed variables are dropped and — unlike the sparse
 * variant, which sums — a later partial of the same column overwrites an earlier one
 * (expressions.rs:993-1008). jac_off (n_systems + 1 entries) and total (doubles in jac) are outputs and
 * may be NULL; with r == jac == NULL only they are filled in (size query, no device needed). */
int fx_eval_residual_dense_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jac, uint64_t* jac_off,
                                    uint64_t* total);
/* == System::analyze -> analyze::numerical::find_overconstraints (analyze/numerical/mod.rs:123-163):
 * dependent[e] = 1 for every expression that does not increase the rank of the dense Jacobian at
 * batch->vars (all variables free, no scaling, no perturbation), n_exprs entries. */
int fx_analyze_batch(fx_ctx* ctx, const fx_batch* batch, uint8_t* dependent);
/* == calculate_residual of every expression at batch->vars with all variables as given
 * (IdentityVariableMap, constraints/mod.rs:96-109). */
int fx_constraint_residuals(fx_ctx* ctx, const fx_batch* batch, double* r);

/* ---- Decomposer::RecursiveAssembly: the device steps of assemble/mod.rs:212-277 ----------------------------
 * The plan (analyze/graph/recursive_assembly.rs) and the make-up of each cluster problem are host bookkeeping in
 * the builder; these four calls are everything numeric the arm does. */
/* What assemble::solve does before any decomposer arm (assemble/mod.rs:58-124): out_vars = variables divided by the
 * System's scale, free variables of each component nudged by the LCG (perturb != 0); out_params = expression
 * parameters after Expression::transform (distances divided by the scale); out_scale[s] = the scale. */
int fx_system_prepare_batch(fx_ctx* ctx, const fx_batch* batch, uint32_t perturb, double* out_vars, double* out_params,
                            double* out_scale);
/* fx_lm_solve_batch for cluster problems: the batch may hold FX_POSE_COINCIDENCE_X / _Y rows. f64, Levenberg-Marquardt
 * (lm.rs:21-193, as the arm calls it, assemble/mod.rs:224-227), any fx_step_solver. Problems within the one-wavefront
 * limits (FX_MAX_FREE_VARS unknowns, FX_MAX_ROWS rows, FX_MAX_SYSTEM_VARS variables) run on the fused kernel, larger ones on
 * the sparse path (where FX_STEP_QR means FX_STEP_CHOLESKY_REFINED, as everywhere). */
int fx_cluster_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results);
/* Pose2D::transform_point (expressions.rs:1120-1134) in place on points (vars[var_idx[i]], vars[var_idx[i] + 1]) with
 * pose poses[3 * pose_of[i] ..] = (rotation, tx, ty): the points a moved cluster carries along (assemble/mod.rs:238-275). */
int fx_pose_transform_points(fx_ctx* ctx, const double* poses, uint32_t n_poses, const uint32_t* pose_of, const uint32_t* var_idx,
                             uint32_t n_points, double* vars, uint32_t n_vars);
/* vars[i] = scale * scaled[i] where mask[i] != 0 (assemble/mod.rs:234-235, 259-262); other entries keep their bits. */
int fx_unscale_vars(fx_ctx* ctx, double scale, const double* scaled, const uint8_t* mask, double* vars, uint32_t n);
/* The same for n_systems Systems of one structure solved side by side (the batched RecursiveAssembly arm): System k owns
 * entries [k * nvars, (k + 1) * nvars) of scaled / vars and is un-scaled by scales[k]; mask[nvars] is shared. */
int fx_unscale_vars_strided(fx_ctx* ctx, const double* scales, uint32_t n_systems, uint32_t nvars, const double* scaled,
                            const uint8_t* mask, double* vars);

/* == find_strongly_connected_expressions per connected component (analyze/graph/equations.rs:186-221),
 * the structural plan Decomposer::SinglePass solves by: the blocks of System `system` in solve order.
 * Host-only (no device needed). block_comp[k] = component of block k; its expressions are
 * rows[row_off[k] .. row_off[k+1]) (system-local ids, block order), its free variables
 * vars[var_off[k] .. var_off[k+1]) (ascending). Capacities the caller provides, with E / V the
 * System's expression / variable counts: block_comp[4E], row_off[4E+1], rows[4E], var_off[4E+1],
 * vars[V] — an expression joins at most one block per component it touches (at most four: the
 * stale-label quirk of graph.rs:211-222 lets a component's matching claim a neighbour's expression),
 * a free variable exactly one. Any output pointer may be NULL. */
int fx_single_pass_blocks(const fx_batch* batch, uint32_t system, uint32_t* n_blocks, uint32_t* block_comp,
                          uint32_t* row_off, uint32_t* rows, uint32_t* var_off, uint32_t* vars);

/* Host build of the correctly rounded atan2 the FX_STEP_QR kernels evaluate angle residuals with
 * (fiksi_amd/csrc/fx_atan2.h; the reference: f64::atan2 at expressions.rs:393, :665 — the platform libm's, within
 * an ulp of this value). Element-wise over n arguments; host-only, for verification against an independent
 * high-precision atan2 (tests/test_atan2.py). */
void fx_atan2_cr_batch(uint64_t n, const double* y, const double* x, double* out);

/* == SymbolicQr::build (solvi/src/decomposition/sparse/qr.rs:118-206), the host-side phase FX_STEP_QR replays the
 * reference's numeric QR from: the COLAMD column order (use_colamd != 0; colamd_rs with its default knobs) or the
 * natural one, the elimination tree, the row permutation of Davis section 5.3 and the row patterns of the Householder
 * vectors (H) and of R, on the pattern of an nrows x ncols matrix given by columns (rows ascending inside a column).
 * Host-only (no device needed). col_perm[ncols]: position -> column; row_perm[nrows]: row -> permuted row;
 * h_ptr / r_ptr [ncols + 1] into h_rows / r_rows (capacities h_cap / r_cap; nrows * ncols and
 * ncols * (ncols + 1) / 2 always suffice). Any output pointer may be NULL. */
int fx_qr_symbolic(int32_t nrows, int32_t ncols, const int32_t* colptr, const int32_t* rowidx, int use_colamd,
                   int32_t* col_perm, int32_t* row_perm, int32_t* h_ptr, int32_t* h_rows, int32_t h_cap, int32_t* r_ptr,
                   int32_t* r_rows, int32_t r_cap);

#ifdef __cplusplus
}
#endif
#endif /* FIKSI_AMD_H */
