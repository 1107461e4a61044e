// Host <-> device plumbing shared by the host sources (fx_analyze / fx_programs / fx_upload / fx_solve / fx_entry .cpp) and the kernels (not part of the public ABI).
#pragma once
#ifdef FX_HOST_ONLY
#include "fx_hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/fiksi_amd.h"

namespace fx {

// var_info encoding (one u16 per variable): low 15 bits = component (0x7FFF = none), bit 15 = fixed.
constexpr uint16_t VAR_COMP_MASK = 0x7FFF;
constexpr uint16_t VAR_COMP_NONE = 0x7FFF;
constexpr uint16_t VAR_FIXED_BIT = 0x8000;

// Per 256-row block of the row-parallel kernels.
struct BlockInfo {
    uint32_t jbase;   // first CSR value of the block
    uint32_t jcount;  // CSR values of the block
    uint32_t sys0;    // System of the block's first row
    uint32_t flags;   // bit 0: "simple" — every row has only free, distinct variables
};

// One block of the SinglePass decomposition (fx_decompose.h), in solve order per System.
struct UnitDesc {
    uint32_t row_off;  // into unit_rows
    uint32_t var_off;  // into unit_vars
    uint16_t nrows, nvars;
    uint16_t comp;     // connected component the block belongs to
    uint16_t flags;
};
constexpr uint16_t UNIT_FIRST = 1;  // first block of its component: the component is perturbed before it
constexpr uint16_t UNIT_EMPTY = 2;  // placeholder of a component without any block
constexpr uint16_t UNIT_RESTORE = 4;  // the block is a whole component of Decomposer::None: afterwards the working vector goes
                                      // back to the pre-solve snapshot (quirk Q2) instead of taking the solved values
// Reference-numerics LM step (FX_STEP_QR): what the host's symbolic analysis (fx_qrplan.h) hands the kernel for one
// component / SinglePass block. u16 blob at u16_off: colperm[n] (position -> free column), rowperm[m + n] (row of the
// augmented matrix [J; sqrt(lambda) I] -> permuted row), hptr[n + 1], hrows[nnzh] (rows of Householder vector k,
// ascending, first = k). u64 blob at u64_off: colmask[n] (bit k: vector k touches the column at position j),
// rowmask[n] (bit j: R[k][j] is structurally non-zero, j > k).
struct QrDesc {
    uint32_t u16_off, u64_off;
    uint16_t n, m, nnzh, ok;
};
struct QrPlans {
    QrDesc* desc = nullptr;
    uint16_t* u16 = nullptr;
    unsigned long long* u64 = nullptr;
    uint32_t* index = nullptr;  // Decomposer::None: [n_systems] first desc of the System (+ component); SinglePass: [units] desc of the block
    uint32_t max_m = 0;         // rows of the largest augmented matrix (expressions + free variables)
    uint32_t max_h = 0;         // entries of the largest Householder structure
    // the wide kernel's QR build (fx_wide.hip): Systems beyond one wavefront whose components have at most 128 columns —
    // qrw_list[i] is solved with the programs words + prog_off[comp_first[i] + c] of its components c (0xFFFFFFFF: none)
    uint32_t* qrw_list = nullptr;
    uint32_t* qrw_comp_first = nullptr;
    uint32_t* qrw_prog_off = nullptr;
    uint32_t* qrw_words = nullptr;
    uint32_t n_qrw = 0, qrw_nx = 0, qrw_free = 0, qrw_vars = 0, qrw_rows = 0;
    // the grouped build's program (fx_programs.cpp: build_qrg_program), for a batch of one structure; null otherwise
    uint32_t* qrg = nullptr;
    uint32_t qrg_words = 0, qrg_small = 0, qrg_nx = 0, qrg_n = 0, qrg_m = 0, qrg_ng = 0;
};
constexpr uint32_t MODE_UNITS = 4;  // LmParams::mode bit: solve block by block
constexpr uint32_t MODE_LBFGS = 8;  // LmParams::mode bit: Optimizer::LBfgs instead of Levenberg-Marquardt

// The program of the grouped kernel's one-structure build (fx_grouped_c.hip; written by fx_programs.cpp: build_gc_program): byte
// offsets of its tables of fixed size; the right-hand-side list and, behind it, the product list follow at PE
template <int NC, int RC> struct GcTable {  // NC columns per lane, RC chunks of 16 rows: Systems of at most NV = 16 NC variables and NR = 16 RC expressions
    static constexpr uint32_t NV = 16u * NC, NR = 16u * RC;
    static constexpr uint32_t VCOL = 80, FIDX = VCOL + NV, RTAG = FIDX + NV, GBASE = RTAG + NR, GVAR = GBASE + 2 * NR, LT = GVAR + 8 * NR,
                              PE = LT + 16 * NC * NV;
};

// One structure class of a batch of several structures, for a launch of fx_grouped_c.hip over the batch's big classes: its program
// (words into gc_tab), its members (entries into order)
struct GcClass {
    uint32_t prog_off, words, words_all, list_off, count;  // (words: what the f64 builds copy of the program; words_all: the f32 build)
};

// A batch resident in HBM. All arrays are struct-of-arrays over the concatenated Systems.
struct DeviceBatch {
    uint32_t n_systems, n_vars, n_exprs;
    uint64_t nnz;
    // maxima over the batch (host-computed; size LDS and pick the kernel instantiation)
    uint32_t max_free;   // free variables per component
    uint32_t max_rows;   // expressions per component
    uint32_t max_vars;   // variables per System
    uint32_t max_exprs;  // expressions per System
    uint32_t max_vars_all, max_exprs_all;  // the same maxima over ALL Systems (large ones included)
    uint32_t max_pairs, max_ents;          // per component: sum over rows of (free entries)^2 / of free entries
    uint32_t max_pairs_g, max_ents_g;      // the same over the Systems of the GLOBAL block walker
    uint32_t max_pairs_tri;                // per component: products of the lower triangle only (grouped kernel)
    uint32_t uniform;                      // 1: every System has the same structure (variables, fixed flags, expressions)
    uint32_t u_nvars, u_nexprs, u_ncomp;   // ... and then its sizes: offsets are multiples, no table look-up needed

    uint32_t* var_off;     // [n_systems+1]
    uint32_t* expr_off;    // [n_systems+1]
    uint16_t* sys_ncomp;   // [n_systems] number of components
    uint8_t* sys_large;    // [n_systems] 0 = fused one-wavefront kernel; 2 = medium (65..128 free variables per
                           // component): the wide kernel (fx_wide.hip); 1 = beyond that: the sparse path
    double* vars0;         // [n_vars] start values (never written by solves)
    double* vars;          // [n_vars] last solved values
    uint16_t* var_info;    // [n_vars]
    uint8_t* expr_tag;     // [n_exprs] fx_tag | 0x80 if the row reads one free variable twice
    uint8_t* row_perm;     // [n_exprs] per 256-row block: local row handled by thread t (tag-sorted)
    uint16_t* expr_comp;   // [n_exprs]
    uint16_t* expr_idx;    // [4*n_exprs] system-local element fields (ushort4 per expression)
    double* expr_param;    // [n_exprs]
    uint32_t* expr_var0;   // [n_exprs] var_off of the owning System (row-parallel kernels, non-simple blocks)
    uint8_t* row_sysoff;   // [n_exprs] owning System minus the block's first System
    BlockInfo* blk_info;   // [ceil(n_exprs / 256)]
    // CSR Jacobian (fixed pattern): rows global, columns = system-local free rank
    uint32_t* jrow_ptr;    // [n_exprs+1]
    uint32_t* jcol;        // [nnz]
    uint32_t* jslot;       // [n_exprs] 8 x 4-bit: CSR slot of gradient entry e, 0xF = dropped
    double* jvals;         // [nnz]
    double* resid;         // [n_exprs]
    fx_result* results;    // [n_systems]
    double* sse_unscaled;  // [n_systems] sum r^2 on the solved, unscaled variables
    uint32_t* work_counter;  // [1] next System of the grouped kernel's device-side queue (reset before each launch)
    uint32_t* sys_class;     // [n_systems] or null: first System with the same structure (batches of several sketches)
    uint32_t* order;         // [n_systems] or null: the System ticket t of the grouped kernel's queue stands for
                             // (fx_batch_schedule_by_last_solve: last solve's longest Systems first)
    // medium Systems (sys_large == 2)
    uint32_t n_wide, w_max_free, w_max_vars, w_max_rows;
    uint32_t* w_list;         // [n_wide] System ids
    // SinglePass blocks, built on first use (null until then)
    uint32_t max_unit_free;   // free variables per block
    uint32_t max_unit_rows;   // expressions per block (a block may hold a neighbouring component's rows)
    uint32_t* sys_unit_off;   // [n_systems+1] into unit_desc (large Systems own no entries)
    UnitDesc* unit_desc;
    uint32_t* unit_rows;      // system-local expression ids
    uint16_t* unit_vars;      // system-local variable ids, ascending per block
    // large Systems whose blocks all fit one wavefront: walked by the GLOBAL kernel instantiation
    uint32_t n_g, max_unit_free_g, max_unit_rows_g;
    uint32_t* g_list;         // [n_g] System ids
    uint32_t* g_off;          // [n_g] offset of the System's slice in the scratch arrays (in variables)
    double* g_xs;             // [2 * total] working variables, two halves per System
    double* g_vout;           // [total] unscaled output values
    int16_t* g_colof;         // [total] variable -> free column of the block in flight
    // FX_STEP_QR plans, built on first use (null until then)
    QrPlans qr_none, qr_units;
    // the program of the grouped kernel's one-structure build (fx_grouped_c.hip; fx_programs.cpp: build_gc_program): null unless the
    // batch is uniform with one component of at most 48 free variables
    uint32_t* gc_tab;
    uint32_t gc_words, gc_nslots, gc_ng;  // words of the program (the f64 builds' part); slots of Jt J's pattern (+ the zero slot), compact Jacobian entries
    uint32_t gc_words_all;                // ... with the f32 build's gather tables
    uint32_t gc_nc;                       // columns per lane of the build the program is for: 1 (up to 16 free variables), 2 (17 ... 32) or 3 (33 ... 48)
    uint32_t gc_rc;                       // ... and its chunks of 16 expressions: gc_nc, or twice that for an over-constrained structure
    const GcClass* gc_classes;            // a launch over several structure classes (null: one program, the whole batch)
    uint32_t gc_nclasses;
    // the program of its sparse build (fx_grouped_s.hip; build_gs_program): uniform batches with one component of 33 ... 255 free
    // variables (33 ... 48: a factor of at most a quarter of the dense triangle), at most 255 variables / expressions, a Cholesky
    // factor of at most 1023 slots and at most 1023 compact Jacobian entries; null otherwise
    uint32_t* gs_tab;
    uint32_t gs_words, gs_nl, gs_ng, gs_nfree;  // words; slots of the factor (even); compact Jacobian entries (even); free variables
    // 1: the batch holds pose rows (FX_TAG_POSE_X / _Y, cluster problems of Decomposer::RecursiveAssembly):
    // only the pose instantiations of the solve kernel may run it
    uint32_t has_pose;
    // A host-buffer call on page-locked arrays (fx_host_register), one-structure build of the grouped kernel only: the kernel reads a
    // System's start values and parameters from the caller's arrays when the System's turn comes and writes its solved free variables
    // and its result record there when it is done — the transfers ride inside the solve instead of before and after it. The device
    // arrays are still filled (vars0 / vars / expr_param / results stay what they are for every other consumer). Null otherwise.
    // The tiny build's hand-over (fx_grouped_tiny.hip): with `order` and `queue_len` both set, a System still running after the
    // build's trial budget is appended to `order` (count in *queue_len) and the 16-column build, whose ladder is made for stragglers,
    // solves those Systems from their start values — same arithmetic, same bits; its queue is then *queue_len long.
    uint32_t* queue_len;
    const double* vars_in;     // [n_vars] device-visible address of the caller's vars
    const double* param_in;    // [n_exprs] ... of the caller's expr_param
    double* vars_out;          // [n_vars] ... of the caller's vars again (written: free variables of a finished System)
    fx_result* results_out;    // [n_systems] ... of the caller's results
};

struct LmParams {
    fx_lm_opts lm;
    uint32_t mode;  // bit0: scale by system RMS, bit1: LCG perturbation, bit2: MODE_UNITS, bit3: MODE_LBFGS
    unsigned long long* prof = nullptr;  // diagnostic build only: 6 per-phase cycle sums
    // routing of batches of small Systems (fx_ctx_set_routing): -1 = by batch size, 0 = never the grouped kernel,
    // 1 = whenever the batch qualifies; the size from which a batch takes it
    int route_grouped = -1;
    int grouped_one_structure = 1;  // 0: batches of one structure stay on the general build (fx_grouped_c.hip is never taken)
    uint32_t grouped_min_systems = 8u;
    uint32_t hold_passes = 2u;  // grouped kernel: passes a finished row waits for a second one before its set-up blocks (fx_ctx_set_hold_passes)
    // grouped kernel, the lambda ladder (fx_ctx_set_ladder): rows without a System of their own try the next lambdas of a
    // running System of their wavefront side by side. ladder_tail / ladder_k: with at most ladder_tail Systems left in the
    // queue, a wavefront that holds a System past ladder_k trials takes no further Systems (0: rows only help once the
    // queue is empty; 0xFFFFFFFF: the launcher's default, eight Systems per resident row). spread: nonzero = the first round
    // of tickets of a scheduled hand-out is dealt one per wavefront (the launcher puts the number of wavefronts there).
    uint32_t ladder = 1u, ladder_k = 8u, ladder_tail = 0xFFFFFFFFu, spread = 1u;
    // Systems beyond one wavefront: the multifrontal build (fx_front.h) where the structure's fronts fit a row of 16 lanes;
    // 0 keeps the column walkers of fx_sparse_team.h (fx_ctx_set_sparse_fronts)
    uint32_t sparse_fronts = 1u;
    uint32_t sparse_front_ranks = 0u;  // lambda trials a launch of its parts + top kernels makes side by side (0: as many as the chip has room for, at most 4)
};

#ifndef FX_HOST_ONLY
// A kernel's dynamic-LDS limit raised to the 160 KB any layout may ask for, once per kernel and device: a set per launch with
// the launch's own size costs a small solve microseconds, and two host threads on one device (several contexts, the sparse
// path's workers) could interleave set(X), set(Y < X), launch(X). One slot per call site (`done`: bit d = device d).
inline hipError_t raise_lds_limit_once(const void* fn, unsigned int* done_bits) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned int bit = 1u << (dev & 31);
    if (__atomic_load_n(done_bits, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) __atomic_fetch_or(done_bits, bit, __ATOMIC_RELEASE);
    return e;
}
#endif

// Kernel launchers (fx_kernels.hip). All asynchronous on `stream`.
hipError_t launch_solve(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
hipError_t launch_eval(const DeviceBatch& b, const double* x, bool want_jacobian, hipStream_t stream);
hipError_t launch_identity_residuals(const DeviceBatch& b, const double* x, double* out, hipStream_t stream);
hipError_t launch_dense_jacobian(const DeviceBatch& b, const double* x, const uint16_t* var_rank, const uint32_t* expr_sys,
                                 const uint16_t* sys_nfree, const uint64_t* dense_off, double* resid, double* jac,
                                 hipStream_t stream);
hipError_t launch_solve_wide(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
// FX_STEP_QR on the Systems of b.qr_none.qrw_list (components of at most 128 columns)
hipError_t launch_solve_wide_qr(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
size_t wide_qr_lds_bytes(uint32_t max_free, uint32_t max_vars, uint32_t max_rows, uint32_t nx);
// several Systems per wavefront (fx_grouped.hip): batches of components with at most 32 free variables
bool grouped_applies(const DeviceBatch& b, const LmParams& p);
hipError_t launch_solve_grouped(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
// ... its general build whatever programs the batch carries (the Systems of a batch of several structures that belong to no big class)
hipError_t launch_solve_grouped_general(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
size_t grouped_lds_bytes(const DeviceBatch& b, uint32_t element_size, bool single_pass_blocks);
// ... its build for batches of one structure, two wavefronts per SIMD (fx_grouped_c.hip)
bool grouped_c_applies(const DeviceBatch& b, const LmParams& p);
hipError_t launch_solve_grouped_c(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
size_t grouped_c_lds_bytes(const DeviceBatch& b, uint32_t element_size);
// ... and its sparse build for batches of one structure of 33 ... 255 free variables with a small factor (fx_grouped_s.hip)
bool grouped_s_applies(const DeviceBatch& b, const LmParams& p);
bool grouped_qr_class_applies(const DeviceBatch& b, const LmParams& p);
hipError_t launch_grouped_qr_class(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
// fx_grouped_tiny.hip: the one-structure build for Systems of at most eight variables and expressions (eight lanes per System)
bool grouped_tiny_applies(const DeviceBatch& b, const LmParams& p);
hipError_t launch_solve_tiny(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
hipError_t launch_solve_grouped_s(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
// the GLOBAL block walker on the lists of `b` (g_list / unit arrays): SinglePass blocks or None-mode components
hipError_t launch_solve_walk(const DeviceBatch& b, const LmParams& p, hipStream_t stream);
// dst[0 .. bytes) = src[0 .. bytes), 16 bytes per thread (both 16-byte aligned, bytes a multiple of 16): pulls a one-shot
// batch's image out of host-coherent memory in one burst (fx_cluster.hip)
hipError_t launch_pull(void* dst, const void* src, size_t bytes, hipStream_t stream);
hipError_t launch_replicate(void* dst, size_t period_bytes, size_t total_bytes, hipStream_t stream);
// scout + sort for the longest-first hand-out of the grouped kernel (fx_presort.hip)
size_t presort_temp_bytes(uint32_t n);
hipError_t launch_presort(const DeviceBatch& b, float* keys, uint32_t* ids, void* temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_presort_lists(const DeviceBatch& b, float* keys, const uint32_t* lists, const uint32_t* offs, const uint32_t* counts, uint32_t n_lists,
                                uint32_t* out, hipStream_t stream);
// fx_cluster.hip — the device side of the RecursiveAssembly arm around its cluster solves:
// scale + LCG perturbation of whole Systems (assemble/mod.rs:58-124) without a solve, one wavefront per System
// (out_params: the expression parameters as Expression::transform leaves them, expressions.rs:195-211)
hipError_t launch_prepare(const DeviceBatch& b, uint32_t mode, double* out_vars, double* out_params, double* out_scale, hipStream_t stream);
// vars[i] = scale * scaled[i] where mask[i] (assemble/mod.rs:234-235, 259-262)
hipError_t launch_unscale(double scale, const double* scaled, const uint8_t* mask, double* vars, uint32_t n, hipStream_t stream);
hipError_t launch_unscale_strided(const double* scales, uint32_t n_systems, uint32_t nvars, const double* scaled, const uint8_t* mask, double* vars,
                                  hipStream_t stream);
// Pose2D::transform_point (expressions.rs:1120-1134) of n points in place: vars[idx[i]], vars[idx[i]+1] by pose[3*pose_of[i]..]
hipError_t launch_pose_transform(const double* poses, const uint32_t* pose_of, const uint32_t* idx, uint32_t n, double* vars,
                                 hipStream_t stream);
size_t wide_lds_bytes(const DeviceBatch& b);
size_t solve_lds_bytes(const DeviceBatch& b);
size_t solve_lds_bytes_units(const DeviceBatch& b);
size_t solve_lds_bytes_qr(const DeviceBatch& b, bool units);
size_t analyze_lds_bytes(uint32_t max_vars, uint32_t max_exprs);
hipError_t launch_analyze(const DeviceBatch& b, const double* x, uint32_t max_vars, uint32_t max_exprs,
                          uint8_t* dependent, hipStream_t stream);

}  // namespace fx
