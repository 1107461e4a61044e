// HIP kernels for gfx950 (MI355X, wave64). Hand-written; no CUDA dual path.
//
//   lm_solve_kernel<N>   one wavefront per System: scale + LCG perturbation (K0), then per connected
//                        component: residual/Jacobian rows (K1), JtJ + lambda I in LDS via ds_add_f64
//                        (K3), register-resident Cholesky + triangular solves with v_readlane
//                        broadcasts (K4), LM control (K5), write-back (K6), unscaled residual check.
//                        Reference: fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193.
//   eval_rows_kernel     one thread per expression over the whole batch: residual + CSR Jacobian
//                        values (subsystem.rs:126-166) — the HBM-streaming kernel.
//   identity_residual_kernel  calculate_residual with IdentityVariableMap (constraints/mod.rs:96-109).
//
// Compiled with -ffp-contract=off: products and sums stay separate exactly as in the reference;
// fused multiply-adds are written explicitly (fma) where the algorithm is ours (Cholesky).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <atomic>
#include <stdint.h>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_lbfgs.h"
#include "fx_chol.h"
#include "fx_wave.h"

namespace fx {

// ------------------------------------------------------------------------------------------
// LDS layout of the fused solve kernel (byte offsets, computed on the host)
// ------------------------------------------------------------------------------------------
struct SolveLayout {
    uint32_t vt;   // padded variables per System
    uint32_t mr;   // padded rows per component
    uint32_t off_xs, off_a, off_rhs, off_g, off_r, off_p, off_gvar, off_gcol, off_rtag, off_fidx, off_colof, off_vout;
    uint32_t off_pw, off_pe;  // packed work lists of the normal-equation assembly
    uint32_t pw_cap, pe_cap;  // their capacities in entries (0: the row-by-row assembly is used)
    uint32_t off_qx, off_qp, off_qs;  // FX_STEP_QR: augmented matrix, plan (u16), per-vector scalars
    uint32_t qr_m, qr_h;              // its row / Householder-entry capacities (0: not a QR launch)
    uint32_t total;
};

static SolveLayout make_layout(uint32_t n_pad, uint32_t max_vars, uint32_t max_rows, uint32_t es /* sizeof(T) */,
                               bool lbfgs = false, uint32_t max_pairs = 0, uint32_t max_ents = 0, uint32_t qr_m = 0,
                               uint32_t qr_h = 0) {
    SolveLayout L;
    L.vt = (max_vars + 7u) & ~7u;
    L.mr = (max_rows + 7u) & ~7u;
    if (L.vt == 0) L.vt = 8;
    if (L.mr == 0) L.mr = 8;
    uint32_t o = 0;
    auto take = [&](uint32_t bytes) { uint32_t at = o; o += (bytes + 15u) & ~15u; return at; };
    L.off_xs = take(2u * L.vt * es);
    // LM: the N x LD normal matrix. L-BFGS: 5 + 5 history vectors, a dot-product scratch vector, rho[5]
    // (padded to 16 doubles) and one "overwritten entry" byte mask per row.
    // (an FX_STEP_QR launch never forms the normal equations: no matrix, no product lists)
    L.off_a = take(qr_m ? 16u : lbfgs ? (11u * n_pad + 16u) * 8u + L.mr : n_pad * (n_pad + 16u / es) * es);
    L.off_rhs = take(n_pad * es);
    L.off_g = take(2u * L.mr * 8u * es);
    L.off_r = take(2u * L.mr * es);
    L.off_p = take(L.mr * es);
    L.off_gvar = take(L.mr * 8u * 2u);
    L.off_gcol = take(L.mr * 8u);
    L.off_rtag = take(L.mr);
    L.off_fidx = take(n_pad * 2u);
    L.off_colof = take(L.vt * 2u);
    L.off_vout = take(L.vt * 8u);  // unscaled output values (f64) for the post-solve check
    // one u32 per product g_a * g_b of the assembly (and per g * r of the right-hand side); worth its
    // LDS only while it stays small (ring16: 672 + 144 entries = 3.3 KB)
    const bool packed = !lbfgs && !qr_m && max_pairs > 0 && max_pairs <= 4096u;
    L.pw_cap = packed ? ((max_pairs + 63u) & ~63u) : 0u;
    L.pe_cap = packed ? ((max_ents + 63u) & ~63u) : 0u;
    L.off_pw = take(L.pw_cap * 4u);
    L.off_pe = take(L.pe_cap * 4u);
    // FX_STEP_QR: the dense (rows + columns) x (columns + 1) augmented matrix [J | -r; sqrt(lambda) I | 0] in the
    // permuted order, the component's plan, v0 / beta of every Householder vector
    L.qr_m = qr_m;
    L.qr_h = qr_h;
    L.off_qx = take(qr_m ? (qr_m + 1u) * (n_pad + 1u) * 8u : 0u);  // + one row of zeros (the padding of the register window reads it)
    L.off_qp = take(qr_m ? (qr_m + (n_pad + 1u) + qr_h + 32u + n_pad) * 2u : 0u);  // 32: slack of the register window
    L.off_qs = take(qr_m ? 2u * n_pad * 8u : 0u);
    L.total = o;
    return L;
}

static uint32_t pad_n(uint32_t max_free) {
    uint32_t n = (max_free + 7u) & ~7u;
    return n < 8u ? 8u : n;
}

size_t solve_lds_bytes(const DeviceBatch& b) {
    return make_layout(pad_n(b.max_free), b.max_vars, b.max_rows, 8u, false, b.max_pairs, b.max_ents).total;
}
size_t solve_lds_bytes_qr(const DeviceBatch& b, bool units) {
    const QrPlans& Q = units ? b.qr_units : b.qr_none;
    const uint32_t rows = (units && b.max_unit_rows > b.max_rows) ? b.max_unit_rows : b.max_rows;
    return make_layout(pad_n(units ? b.max_unit_free : b.max_free), b.max_vars, rows, 8u, false, b.max_pairs, b.max_ents, Q.max_m, Q.max_h).total;
}
size_t solve_lds_bytes_units(const DeviceBatch& b) {
    return make_layout(pad_n(b.max_unit_free), b.max_vars, b.max_rows > b.max_unit_rows ? b.max_rows : b.max_unit_rows, 8u, false,
                       b.max_pairs, b.max_ents).total;
}

// ------------------------------------------------------------------------------------------
// fused per-System solve
// ------------------------------------------------------------------------------------------
// PROF = true is the diagnostic build of the same kernel: s_memtime stamps at the phase boundaries,
// summed per phase into prm.prof (never launched by the product entry points).
enum Phase { PH_SETUP = 0, PH_EVAL = 1, PH_FORM = 2, PH_FACTOR = 3, PH_SOLVE = 4, PH_TAIL = 5, PH_COUNT = 6 };

// T = double: the reference precision. T = float: BASELINE cfg5 (the HBM arrays stay f64; scale and
// perturbation are computed in f64, everything after in f32).
//
// UNITS = true is `Decomposer::SinglePass` (assemble/mod.rs:169-210): the loop runs over the blocks the
// host decomposition produced (fx_decompose.h) instead of over whole components, the component's
// perturbation happens before its first block, and a solved block is written through to the working
// vectors so later blocks see it.
//
// OPT = 1 is `Optimizer::LBfgs` (fiksi/src/solve/lbfgs.rs) in place of the LM loop: same scaling,
// perturbation, row lists and write-back; the optimizer keeps one variable per lane, the 5 + 5
// history vectors in LDS, and sums every dot product in index order (the reference's order) by
// reading the lane products back from LDS. The line search is the fx::HzMachine state machine
// around the single evaluation site.
//
// GLOBAL = true (with UNITS) serves Systems too large for LDS whose SinglePass blocks are all small:
// the System-wide vectors (working variables, output values, variable -> column map) live in an
// HBM scratch area instead of LDS — L2-resident for the one wavefront that walks the blocks — while
// everything per block stays in LDS and registers as before.
// QR = true adds the reference-numerics step (FX_STEP_QR) — an instantiation of its own, so that the register
// budget of the plain kernel (two wavefronts per SIMD) does not pay for the QR step's register window.
template <int N, typename T, bool PROF, bool UNITS, int OPT = 0, bool GLOBAL = false, bool QR = false, bool POSE = false>
__device__ __forceinline__ void lm_solve_body(const DeviceBatch& b, const LmParams& prm, const SolveLayout& L, unsigned char* smem) {
    unsigned long long ph[PH_COUNT] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_last = 0;
    auto stamp = [&](int phase) {
        if (PROF) {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[phase] += t - t_last;
            t_last = t;
        }
    };
    if (PROF) t_last = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x;
    constexpr bool CR_ATAN2 = QR && sizeof(T) == 8 && OPT == 0 && !GLOBAL;  // FX_STEP_QR: correctly rounded atan2 (fx_atan2.h)
    const uint32_t s = GLOBAL ? b.g_list[blockIdx.x] : blockIdx.x;
    if (!GLOBAL && b.sys_large[s]) return;  // handled by the GLOBAL launch or the sparse path (fx_sparse.hip)
    constexpr int LD = N + Vec16<T>::n;  // 16-byte aligned columns, conflict-free ds_read_b128

    T* XS = reinterpret_cast<T*>(smem + L.off_xs);       // [2][vt] full variable vectors (GLOBAL: re-pointed below)
    T* Amat = reinterpret_cast<T*>(smem + L.off_a);      // [N][LD] JtJ (lambda on demand)
    T* rhsv = reinterpret_cast<T*>(smem + L.off_rhs);    // [N] -Jt r
    T* G = reinterpret_cast<T*>(smem + L.off_g);         // [2][mr][8] Jacobian rows
    T* R = reinterpret_cast<T*>(smem + L.off_r);         // [2][mr] residuals
    T* P = reinterpret_cast<T*>(smem + L.off_p);         // [mr] scaled parameters
    uint16_t* gvar = reinterpret_cast<uint16_t*>(smem + L.off_gvar);  // [mr][8] variable of entry e
    int8_t* gcol = reinterpret_cast<int8_t*>(smem + L.off_gcol);      // [mr][8] free column or -1
    uint8_t* rtag = reinterpret_cast<uint8_t*>(smem + L.off_rtag);    // [mr]
    uint16_t* fidx = reinterpret_cast<uint16_t*>(smem + L.off_fidx);  // [N] free column -> variable
    double* VOUT = reinterpret_cast<double*>(smem + L.off_vout);       // [vt] unscaled values as written back
    const uint32_t mr = L.mr;

    const uint32_t v0 = b.var_off[s], nvt = b.var_off[s + 1] - v0;
    int16_t* colof = reinterpret_cast<int16_t*>(smem + L.off_colof);  // [vt] variable -> free column
    const uint32_t vt = GLOBAL ? nvt : L.vt;  // distance between the two halves of XS
    if constexpr (GLOBAL) {
        const size_t go = b.g_off[blockIdx.x];
        XS = reinterpret_cast<T*>(b.g_xs + 2 * go);
        VOUT = b.g_vout + go;
        colof = b.g_colof + go;
    }
    const uint32_t e0 = b.expr_off[s], net = b.expr_off[s + 1] - e0;
    const uint32_t ncomp = b.sys_ncomp[s];
    const fx_lm_opts o = prm.lm;

    // ---- the first 64 variables / expressions of the System, fetched up front in one round trip (for
    // the headline shape that is all of them). Element `lane` of every per-variable / per-expression
    // array sits in a register; the accessors fall back to a load for any other index.
    // (not in the block-walking instantiations: their rows come in block order, and the many small
    // blocks run better with the registers left to occupancy)
    constexpr bool PF = !UNITS;
    const bool pf_hv = PF && (uint32_t)lane < nvt, pf_he = PF && (uint32_t)lane < net;
    const double pf_var = pf_hv ? b.vars0[v0 + lane] : 0.0;
    const uint16_t pf_info = pf_hv ? b.var_info[v0 + lane] : (uint16_t)0;
    const int pf_tag = pf_he ? (int)(b.expr_tag[e0 + lane] & 0x7F) : 0;
    const double pf_param = pf_he ? b.expr_param[e0 + lane] : 0.0;
    const uint16_t pf_comp = pf_he ? b.expr_comp[e0 + lane] : (uint16_t)0xFFFF;
    const ushort4 pf_idx = pf_he ? reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + lane] : make_ushort4(0, 0, 0, 0);
    auto ld_var = [&](uint32_t i) -> double { return (PF && i == (uint32_t)lane) ? pf_var : b.vars0[v0 + i]; };
    auto ld_info = [&](uint32_t i) -> uint16_t { return (PF && i == (uint32_t)lane) ? pf_info : b.var_info[v0 + i]; };
    auto ld_tag = [&](uint32_t i) -> int { return (PF && i == (uint32_t)lane) ? pf_tag : (int)(b.expr_tag[e0 + i] & 0x7F); };
    auto ld_param = [&](uint32_t i) -> double { return (PF && i == (uint32_t)lane) ? pf_param : b.expr_param[e0 + i]; };
    auto ld_comp = [&](uint32_t i) -> uint16_t { return (PF && i == (uint32_t)lane) ? pf_comp : b.expr_comp[e0 + i]; };
    auto ld_idx = [&](uint32_t i) -> ushort4 {
        return (PF && i == (uint32_t)lane) ? pf_idx : reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + i];
    };

    // ---- K0a: system scale = sqrt((sum v^2 + sum d^2) / count), summed strictly in reference
    // order (assemble/mod.rs:32-44, utils.rs:11-33) so the scale is bit-identical. ------------
    double scale = 1.0, scale_recip = 1.0;
    if (prm.mode & 1u) {
        scale = system_scale_wave(nvt, net, lane, ld_var, ld_tag, ld_param);
        scale_recip = 1.0 / scale;
    }

    // ---- scaled snapshot of all variables (both halves of XS), output defaults to input ------
    for (uint32_t i = lane; i < nvt; i += 64) {
        double v = ld_var(i);
        double xsv = (prm.mode & 1u) ? v * scale_recip : v;
        XS[i] = (T)xsv;
        XS[vt + i] = (T)xsv;
        VOUT[i] = v;
        b.vars[v0 + i] = v;  // fixed / unconstrained variables stay bit-identical
    }
    __syncthreads();

    uint32_t rng = 42u;  // one Rng::from_seed(42) per solve, shared by the components (:47)
    uint32_t tot_accept = 0, tot_trials = 0, last_exit = FX_EXIT_SSE, comps_done = 0;
    double tot_sse0 = 0.0, tot_sse = 0.0;
    if constexpr (UNITS) {  // blocks set and clear their own entries
        for (uint32_t i = lane; i < nvt; i += 64) colof[i] = (int16_t)-1;
        __syncthreads();
    }

    const uint32_t unit0 = UNITS ? b.sys_unit_off[s] : 0u;
    const uint32_t n_iter = UNITS ? b.sys_unit_off[s + 1] - unit0 : ncomp;
    for (uint32_t c = 0; c < n_iter; ++c) {
        uint32_t nfree = 0, m_rows = 0;
        uint32_t unit_flags = 0;
        if constexpr (UNITS) {
        const UnitDesc ud = b.unit_desc[unit0 + c];
        unit_flags = ud.flags;
        if (ud.flags & UNIT_FIRST) {  // the component's perturbation comes before its first block (:91-111)
            comps_done += 1;
            last_exit = FX_EXIT_SSE;
            if (prm.mode & 2u) {
                uint32_t rank0 = 0;
                for (uint32_t base = 0; base < nvt; base += 64) {
                    uint32_t i = base + lane;
                    bool in = false;
                    if (i < nvt) {
                        uint16_t info = ld_info(i);
                        in = ((info & VAR_COMP_MASK) == ud.comp) && !(info & VAR_FIXED_BIT);
                    }
                    uint64_t mk = __ballot(in);
                    if (in) {
                        uint32_t st = lcg_jump(rng, 2u * (rank0 + (uint32_t)__popcll(mk & lanemask_lt(lane))));
                        st = st * 1664525u + 1013904223u;
                        double f1 = (1.0 / 4294967295.0) * (double)st;
                        st = st * 1664525u + 1013904223u;
                        double f2 = (1.0 / 4294967295.0) * (double)st;
                        double x = ld_var(i);
                        if (prm.mode & 1u) x = x * scale_recip;
                        x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                        XS[i] = (T)x;
                        XS[vt + i] = (T)x;
                    }
                    rank0 += (uint32_t)__popcll(mk);
                }
                rng = lcg_jump(rng, 2u * rank0);
            }
        }
        if (ud.flags & UNIT_EMPTY) continue;  // a component no expression could be matched in
        nfree = ud.nvars;
        m_rows = ud.nrows;
        if ((uint32_t)lane < nfree) {
            uint32_t vi = b.unit_vars[ud.var_off + lane];
            fidx[lane] = (uint16_t)vi;
            colof[vi] = (int16_t)lane;
        }
        __syncthreads();
        for (uint32_t pos = lane; pos < m_rows; pos += 64) {  // rows in block order
            uint32_t i = b.unit_rows[ud.row_off + pos];
            int tag = ld_tag(i);
            const ushort4 f4 = ld_idx(i);
            uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
            uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int k = expand_vars<POSE>(tag, ff, vars8);
            double prm_e = ld_param(i);
            if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
            rtag[pos] = (uint8_t)tag;
            P[pos] = (T)prm_e;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gvar[pos * 8 + e] = (uint16_t)vars8[e];
                gcol[pos * 8 + e] = (e < k) ? (int8_t)colof[vars8[e]] : (int8_t)-1;
            }
        }
        __syncthreads();
        } else {
        // ---- free variables of the component, ascending (BTreeSet order, :91-111) -----------
        for (uint32_t base = 0; base < nvt; base += 64) {
            uint32_t i = base + lane;
            bool in = false;
            if (i < nvt) {
                uint16_t info = ld_info(i);
                in = ((info & VAR_COMP_MASK) == c) && !(info & VAR_FIXED_BIT);
            }
            uint64_t m = __ballot(in);
            uint32_t pos = nfree + (uint32_t)__popcll(m & lanemask_lt(lane));
            if (i < nvt) colof[i] = in ? (int16_t)pos : (int16_t)-1;
            if (in && pos < (uint32_t)N) fidx[pos] = (uint16_t)i;
            nfree += (uint32_t)__popcll(m);
        }
        // a component without variables is skipped by the reference (`elements.is_empty()`)
        bool any_var = false;
        for (uint32_t base = 0; base < nvt; base += 64) {
            uint32_t i = base + lane;
            bool in = (i < nvt) && ((ld_info(i) & VAR_COMP_MASK) == c);
            any_var = any_var || (__ballot(in) != 0ull);
        }
        if (!uniform(any_var)) continue;
        __syncthreads();

        // ---- K0b: perturbation of the free variables, 2 LCG draws each, ascending order -----
        if (prm.mode & 2u) {
            uint32_t st = lcg_jump(rng, 2u * (uint32_t)lane);
            st = st * 1664525u + 1013904223u;
            double f1 = (1.0 / 4294967295.0) * (double)st;
            st = st * 1664525u + 1013904223u;
            double f2 = (1.0 / 4294967295.0) * (double)st;
            if ((uint32_t)lane < nfree) {
                uint32_t vi = fidx[lane];
                // recomputed from the f64 input so the f64 start point is bit-identical to the reference
                double x = ld_var(vi);
                if (prm.mode & 1u) x = x * scale_recip;
                x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                XS[vi] = (T)x;
                XS[vt + vi] = (T)x;
            }
            if (nfree > 0) rng = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)(nfree - 1));
        }

        // ---- rows of the component: ascending expression id (:139-145) ----------------------
        for (uint32_t base = 0; base < net; base += 64) {
            uint32_t i = base + lane;
            bool in = (i < net) && (ld_comp(i) == c);
            uint64_t mk = __ballot(in);
            uint32_t pos = m_rows + (uint32_t)__popcll(mk & lanemask_lt(lane));
            if (in) {
                int tag = ld_tag(i);
                const ushort4 f4 = ld_idx(i);
                uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
                uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int k = expand_vars<POSE>(tag, ff, vars8);
                double prm_e = ld_param(i);
                if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                rtag[pos] = (uint8_t)tag;
                P[pos] = (T)prm_e;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gvar[pos * 8 + e] = (uint16_t)vars8[e];
                    gcol[pos * 8 + e] = (e < k) ? (int8_t)colof[vars8[e]] : (int8_t)-1;
                }
            }
            m_rows += (uint32_t)__popcll(mk);
        }
        __syncthreads();
        }  // !UNITS

        // ---- packed work lists of the normal-equation assembly: one u32 per product g_a * g_b with both
        // columns free (row << 19 | a << 16 | b << 13 | address in A) and one per g * r of the right-hand
        // side (row << 9 | a << 6 | column). Built once per component / block; every later assembly then
        // is a flat loop over them instead of 8 x 8 lanes per row with most of them idle.
        uint32_t* PW = reinterpret_cast<uint32_t*>(smem + L.off_pw);
        uint32_t* PE = reinterpret_cast<uint32_t*>(smem + L.off_pe);
        uint32_t n_pw = 0, n_pe = 0;
        bool use_packed = false;
        if (OPT == 0 && L.pw_cap && m_rows > 8u) {  // a handful of rows is cheaper on the lane grid than listing them
            for (uint32_t base = 0; base < m_rows; base += 64) {
                const uint32_t row = base + lane;
                uint32_t mask = 0;
                uint64_t cols = 0;  // the row's eight columns, one byte each (registers only from here on)
                if (row < m_rows) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int cc = gcol[row * 8 + e];
                        mask |= (cc >= 0) ? (1u << e) : 0u;
                        cols |= (uint64_t)(uint8_t)cc << (8 * e);
                    }
                }
                const uint32_t kf = (uint32_t)__popc(mask);
                uint32_t inc2 = kf * kf, inc1 = kf;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t t2 = __shfl_up(inc2, off, 64), t1 = __shfl_up(inc1, off, 64);
                    if (lane >= off) {
                        inc2 += t2;
                        inc1 += t1;
                    }
                }
                uint32_t at2 = n_pw + inc2 - kf * kf, at1 = n_pe + inc1 - kf;
                if (at2 + kf * kf <= L.pw_cap && at1 + kf <= L.pe_cap) {
                    for (uint32_t m1 = mask; m1; m1 &= m1 - 1u) {
                        const uint32_t a = (uint32_t)__ffs(m1) - 1u;
                        const uint32_t ca = (uint32_t)(cols >> (8u * a)) & 0xFFu;
                        PE[at1++] = (row << 9) | (a << 6) | ca;
                        for (uint32_t m2 = mask; m2; m2 &= m2 - 1u) {
                            const uint32_t bb = (uint32_t)__ffs(m2) - 1u;
                            const uint32_t cb = (uint32_t)(cols >> (8u * bb)) & 0xFFu;
                            PW[at2++] = (row << 19) | (a << 16) | (bb << 13) | (ca * (uint32_t)LD + cb);
                        }
                    }
                }
                n_pw += (uint32_t)__builtin_amdgcn_readlane((int)inc2, 63);
                n_pe += (uint32_t)__builtin_amdgcn_readlane((int)inc1, 63);
            }
            use_packed = n_pw <= L.pw_cap && n_pe <= L.pe_cap;
            __syncthreads();
        }

        // ---- FX_STEP_QR (reference numerics): the component's plan from the host's symbolic analysis ----
        // The LM step is then the reference's own computation, operation by operation: Householder QR of the
        // augmented matrix [J; sqrt(lambda) I] in the reference's column order (COLAMD) and row order (Davis 5.3),
        // every inner product summed over the Householder vector's rows in ascending order, multiplications
        // and additions unfused (solvi/src/decomposition/sparse/qr.rs:226-356). The matrix is kept dense in LDS,
        // one column per lane; entries outside the symbolic patterns are exact zeros that no operation reads.
        constexpr bool QR_BUILD = QR && sizeof(T) == 8 && OPT == 0 && !GLOBAL;
        const bool qr = QR_BUILD && o.solver == FX_STEP_QR && L.qr_m != 0u;
        constexpr uint32_t LDX = (uint32_t)N + 1u;
        double* QX = reinterpret_cast<double*>(smem + L.off_qx);           // [m + n][LDX]
        uint16_t* q_rowperm = reinterpret_cast<uint16_t*>(smem + L.off_qp);  // [qr_m]
        uint16_t* q_hptr = q_rowperm + L.qr_m;                                // [N + 1]
        uint16_t* q_hoff = q_hptr + (N + 1);                                  // [qr_h] row of a vector entry, times LDX
        uint16_t* q_cpos = q_hoff + L.qr_h + 32u;                             // [N] free column -> position
        double* QV0 = reinterpret_cast<double*>(smem + L.off_qs);             // [N] first entry of vector k
        double* QBETA = QV0 + N;                                              // [N]
        unsigned long long q_colmask = 0ull, q_rowmask = 0ull;
        uint32_t q_cp = 0;
        bool qr_bad_plan = false;
        if constexpr (QR_BUILD) {
            if (qr) {
                const QrPlans& Q = UNITS ? b.qr_units : b.qr_none;
                const QrDesc qd = Q.desc[UNITS ? Q.index[unit0 + c] : Q.index[s] + c];
                qr_bad_plan = !qd.ok || qd.n != nfree || qd.m != m_rows || (uint32_t)qd.m + qd.n > L.qr_m || qd.nnzh > L.qr_h;
                if (!qr_bad_plan) {
                    const uint16_t* pu = Q.u16 + qd.u16_off;
                    const unsigned long long* pm = Q.u64 + qd.u64_off;
                    const uint32_t Mq = m_rows + nfree;
                    if ((uint32_t)lane < nfree) {
                        q_cp = pu[lane];
                        q_cpos[q_cp] = (uint16_t)lane;
                        q_colmask = pm[lane];
                        q_rowmask = pm[nfree + lane];
                    }
                    for (uint32_t i = lane; i < Mq; i += 64) q_rowperm[i] = pu[nfree + i];
                    for (uint32_t i = lane; i <= nfree; i += 64) q_hptr[i] = pu[nfree + Mq + i];
                    for (uint32_t i = lane; i < qd.nnzh; i += 64) q_hoff[i] = (uint16_t)(pu[nfree + Mq + nfree + 1u + i] * LDX);
                    __syncthreads();
                }
            }
        }
        // sum of squares in index order (lm.rs:195-197): every lane adds the same numbers in the same order
        auto seq_sse_lm = [&](int buf) -> double {
            double acc = 0.0;
            const T* rb = R + buf * mr;
            for (uint32_t row0 = 0; row0 < m_rows; row0 += 16) {  // 16 loads in flight, then their squares in order
                double rr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) rr[u] = (row0 + u < m_rows) ? (double)rb[row0 + u] : 0.0;  // + 0.0: exact
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += rr[u] * rr[u];
            }
            return bcast(acc, 0);
        };
        // One LM trial's step by the reference's QR. Returns false when R has an exactly zero diagonal entry
        // (sparse_col_mat.rs:800-810). delta_out: this lane's free column; dn2_out: |delta|^2 summed in index order.
        auto qr_step = [&](double lam, int buf, T& delta_out, T& dn2_out) -> bool {
            bool ok = true;
            if constexpr (QR_BUILD) {
                const uint32_t Mq = m_rows + nfree;
                const double sl = ::sqrt(lam);  // lm.rs:119
                __syncthreads();
                for (uint32_t i = lane; i < (Mq + 1u) * LDX; i += 64) QX[i] = 0.0;  // row Mq stays zero
                __syncthreads();
                // J (duplicates of a row summed in gradient order, sparse_col_mat.rs:710-711) and b = -r (lm.rs:86-91,130)
                for (uint32_t row = lane; row < m_rows; row += 64) {
                    const uint32_t pr = q_rowperm[row];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int cc = gcol[row * 8 + e];
                        // (the matrix was just zeroed: an LDS add is the store, and the second entry of a repeated
                        // column is added to the first; the adds of one lane execute in order)
                        if (cc >= 0) lds_add(&QX[pr * LDX + q_cpos[cc]], (double)G[(buf * mr + row) * 8 + e]);
                    }
                    QX[pr * LDX + N] = -(double)R[buf * mr + row];
                }
                // the damping entry of every column (lm.rs:92-96,119-125)
                if ((uint32_t)lane < nfree) QX[q_rowperm[m_rows + lane] * LDX + q_cpos[lane]] = sl;
                __syncthreads();
                stamp(PH_FORM);

                // apply_householder (qr.rs:226-240) of vector k to column cx, entry by entry (vectors beyond the
                // register window below, and the right-hand side pass of a 64-column component)
                auto apply_h = [&](uint32_t k, uint32_t cx, double v0n, double beta, uint32_t hb, uint32_t he) {
                    double tau = 0.0;
                    tau = tau + v0n * QX[k * LDX + cx];
                    for (uint32_t t = hb + 1u; t < he; ++t) {
                        const uint32_t r = q_hoff[t];
                        tau = tau + QX[r + k] * QX[r + cx];
                    }
                    tau = tau * beta;
                    QX[k * LDX + cx] = QX[k * LDX + cx] - v0n * tau;
                    for (uint32_t t = hb + 1u; t < he; ++t) {
                        const uint32_t r = q_hoff[t];
                        QX[r + cx] = QX[r + cx] - QX[r + k] * tau;
                    }
                };
                // the right-hand side rides along as column N: on the lane after the last column, or (64 columns) in a
                // pass of its own below
                const bool is_b = (uint32_t)lane == nfree;
                const uint32_t cx = is_b ? (uint32_t)N : (uint32_t)lane;
                const int my_hptr = (int)q_hptr[lane];          // hptr[0..63], one per lane
                const uint32_t hptr_end = q_hptr[nfree];
                constexpr int CHT = 32;  // entries of a Householder vector (below the diagonal) kept in registers
                for (uint32_t k = 0; k < nfree; ++k) {
                    const uint32_t hb = (uint32_t)__builtin_amdgcn_readlane(my_hptr, (int)k);
                    const uint32_t he = (k + 1u < nfree) ? (uint32_t)__builtin_amdgcn_readlane(my_hptr, (int)(k + 1u)) : hptr_end;
                    const uint32_t len1 = he - hb - 1u;
                    const bool act = is_b || ((uint32_t)lane < nfree && ((q_colmask >> k) & 1ull));
                    const double v0 = QX[k * LDX + k];
                    const bool windowed = len1 <= (uint32_t)CHT;
                    // the whole vector and this lane's column entries under it, fetched once: row offsets, then the two
                    // columns — two LDS round trips per vector instead of two per entry and pass. The host pads every
                    // vector to whole blocks of eight with the row of zeros (build_qr_plan), so a block is loaded,
                    // multiplied and stored with no per-entry test.
                    // calculate_householder (qr.rs:244-275) on column k below the diagonal; every lane computes it
                    auto householder = [&](double sigma, double& norm, double& beta, double& v0n) {
                        norm = ::fabs(v0);
                        beta = (v0 >= 0.0) ? 0.0 : 2.0;
                        v0n = 1.0;
                        if (sigma != 0.0) {
                            norm = ::sqrt(sigma + v0 * v0);
                            v0n = (v0 <= 0.0) ? v0 - norm : -sigma / (v0 + norm);
                            beta = -(1.0 / (norm * v0n));
                        }
                    };
                    // a vector of NB blocks of eight entries below the diagonal, start to finish (one copy of the code per
                    // block count, picked once per vector: the same body with a test in front of every block of every pass
                    // spent a tenth of the loop's instructions on scalar compares and branches)
                    double norm = 0.0, beta = 0.0, v0n = 1.0;
                    auto window = [&](auto nb_c) {
                        constexpr int NE = 8 * decltype(nb_c)::value;
                        uint32_t ro[NE > 0 ? NE : 1];
                        double vk[NE > 0 ? NE : 1], xj[NE > 0 ? NE : 1];
                        const double xk0 = QX[k * LDX + cx];
#pragma unroll
                        for (int u = 0; u < NE; ++u) ro[u] = q_hoff[hb + 1u + (uint32_t)u];
#pragma unroll
                        for (int u = 0; u < NE; ++u) {
                            vk[u] = QX[ro[u] + k];
                            xj[u] = QX[ro[u] + cx];
                        }
                        double sigma = 0.0;
#pragma unroll
                        for (int u = 0; u < NE; ++u) sigma = sigma + vk[u] * vk[u];
                        householder(sigma, norm, beta, v0n);
                        double tau = 0.0;
                        tau = tau + v0n * xk0;
#pragma unroll
                        for (int u = 0; u < NE; ++u) tau = tau + vk[u] * xj[u];
                        tau = tau * beta;
                        if (act) {
                            QX[k * LDX + cx] = xk0 - v0n * tau;
#pragma unroll
                            for (int u = 0; u < NE; ++u) QX[ro[u] + cx] = xj[u] - vk[u] * tau;  // (padding: 0 - 0 tau)
                        }
                    };
                    if (windowed) {
                        switch (len1 >> 3) {  // (the host pads len1 to a multiple of eight)
                            case 0: window(std::integral_constant<int, 0>{}); break;
                            case 1: window(std::integral_constant<int, 1>{}); break;
                            case 2: window(std::integral_constant<int, 2>{}); break;
                            case 3: window(std::integral_constant<int, 3>{}); break;
                            default: window(std::integral_constant<int, 4>{}); break;
                        }
                    } else {
                        double sigma = 0.0;
                        for (uint32_t t = hb + 1u; t < he; ++t) {
                            const double x = QX[q_hoff[t] + k];
                            sigma = sigma + x * x;
                        }
                        householder(sigma, norm, beta, v0n);
                        if (act) apply_h(k, cx, v0n, beta, hb, he);
                    }
                    if constexpr (N == 64) {  // (kept for the right-hand side's own pass below, which only a 64-column component needs)
                        if (lane == 0) {
                            QV0[k] = v0n;
                            QBETA[k] = beta;
                        }
                    }
                    __syncthreads();
                    if (lane == 0) QX[k * LDX + k] = norm;  // R's diagonal (qr.rs:319)
                }
                stamp(PH_FACTOR);
                if (nfree == 64u) {  // Q^T b (qr.rs:328-346) with the stored vectors
                    if (lane == 0)
                        for (uint32_t k = 0; k < nfree; ++k) apply_h(k, (uint32_t)N, QV0[k], QBETA[k], q_hptr[k], q_hptr[k + 1]);
                }
                __syncthreads();
                // back substitution with R (sparse_col_mat.rs:788-826), lane r holds entry r of the vector and row r of R
                // (fetched up front: the loop itself is a chain of divisions with no memory access in it)
                double yv = ((uint32_t)lane < nfree) ? QX[lane * LDX + N] : 0.0;
                double rrow[N];
#pragma unroll
                for (int i = 0; i < N; ++i) rrow[i] = ((uint32_t)lane < nfree && (uint32_t)i < nfree) ? QX[lane * LDX + i] : 1.0;
                double dg = 1.0;  // this lane's diagonal entry
#pragma unroll
                for (int i = 0; i < N; ++i) dg = (lane == i) ? rrow[i] : dg;
                ok = __ballot((uint32_t)lane < nfree && dg == 0.0) == 0ull;
                if (ok) {
#pragma unroll
                    for (int i = N - 1; i >= 0; --i) {
                        if ((uint32_t)i < nfree) {
                            const double coeff = bcast(yv, i) / bcast(dg, i);
                            if (lane == i) yv = coeff;
                            if (lane < i && ((q_rowmask >> i) & 1ull)) yv = yv - coeff * rrow[i];
                        }
                    }
                }
                // undo the column permutation (qr.rs:354): delta[colperm[j]] = x_j
                double* QD = QV0;
                __syncthreads();
                if ((uint32_t)lane < nfree) QD[q_cp] = yv;
                __syncthreads();
                delta_out = ((uint32_t)lane < nfree) ? (T)QD[lane] : T(0);
                double acc = 0.0;
#pragma unroll
                for (int c0 = 0; c0 < N; c0 += 16) {
                    if ((uint32_t)c0 < nfree) {
                        double dd[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) dd[u] = ((uint32_t)(c0 + u) < nfree) ? QD[c0 + u] : 0.0;
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc += dd[u] * dd[u];
                    }
                }
                dn2_out = (T)bcast(acc, 0);
                __syncthreads();
            }
            return ok;
        };

        // evaluates all rows at XS[buf] into G[buf], R[buf]; returns SSE (wave-uniform)
        auto eval_rows = [&](int buf) -> T {
            const T* xs = XS + buf * vt;
            T part = T(0);
            for (uint32_t row = lane; row < m_rows; row += 64) {
                T v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xs[gvar[row * 8 + e]];
                T r = eval_expression<T, true, CR_ATAN2, POSE>(rtag[row], v, P[row], g);
                R[buf * mr + row] = r;
#pragma unroll
                for (int e = 0; e < 8; ++e) G[(buf * mr + row) * 8 + e] = g[e];
                part += r * r;
            }
            return wave_sum(part);
        };

        // K3: A = Jt J (full symmetric), rhs = -Jt r, accumulated row by row with LDS f64 atomics
        // (one row per wave instruction, so the summation order is the row order).
        auto form_normal = [&](int buf) {
            for (uint32_t i = lane; i < (uint32_t)(N * LD); i += 64) Amat[i] = T(0);
            if (lane < N) rhsv[lane] = T(0);
            __syncthreads();
            if (use_packed) {
                const T* Gb = G + (size_t)buf * mr * 8;
                for (uint32_t t = lane; t < n_pw; t += 64) {
                    const uint32_t w = PW[t];
                    const uint32_t gb = (w >> 19) * 8u;
                    lds_add(&Amat[w & 0x1FFFu], Gb[gb + ((w >> 16) & 7u)] * Gb[gb + ((w >> 13) & 7u)]);
                }
                for (uint32_t t = lane; t < n_pe; t += 64) {
                    const uint32_t w = PE[t];
                    const uint32_t row = w >> 9;
                    lds_add(&rhsv[w & 63u], Gb[row * 8u + ((w >> 6) & 7u)] * -R[buf * mr + row]);
                }
                if (lane < N && (uint32_t)lane >= nfree) Amat[lane * LD + lane] = T(1);  // identity padding
                __syncthreads();
                return;
            }
            const int e1 = lane >> 3, e2 = lane & 7;
            constexpr int RB = 4;  // rows per batch: all loads of a batch are issued before its atomics
            for (uint32_t row0 = 0; row0 < m_rows; row0 += RB) {
                int c1[RB], c2[RB];
                T g1[RB], g2[RB], rr[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    uint32_t row = min(row0 + q, m_rows - 1);
                    c1[q] = gcol[row * 8 + e1];
                    c2[q] = gcol[row * 8 + e2];
                    g1[q] = G[(buf * mr + row) * 8 + e1];
                    g2[q] = G[(buf * mr + row) * 8 + e2];
                    rr[q] = -R[buf * mr + row];
                    if (row0 + q >= m_rows) c1[q] = -1;
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    if (c1[q] >= 0 && c2[q] >= 0) lds_add(&Amat[c1[q] * LD + c2[q]], g1[q] * g2[q]);
                    if (e2 == 0 && c1[q] >= 0) lds_add(&rhsv[c1[q]], g1[q] * rr[q]);
                }
            }
            if (lane < N && (uint32_t)lane >= nfree) Amat[lane * LD + lane] = T(1);  // identity padding
            __syncthreads();
        };

        int cur = 0;
        const T xstart = ((uint32_t)lane < nfree) ? XS[fidx[lane]] : T(0);  // perturbed start of this lane's variable
        stamp(PH_SETUP);
        T sse = eval_rows(0);
        if (qr) sse = (T)seq_sse_lm(0);
        T sse_start = sse;
        uint32_t accepted = 0, trials = 0, exit_code = FX_EXIT_MAX_OUTER;
        if constexpr (OPT == 1) {
        // ================= Optimizer::LBfgs (lbfgs.rs:20-193) =================
        double* SH = reinterpret_cast<double*>(Amat);  // [5][N] s_k ring
        double* YH = SH + 5 * N;                        // [5][N] y_k ring
        double* DOT = YH + 5 * N;                       // [N] lane products of the dot product in flight
        double* RHO = DOT + N;                          // [5] (+ padding)
        uint8_t* rdead = reinterpret_cast<uint8_t*>(RHO + 16);  // [mr] bit e: entry e is overwritten by a later one
        double* GR = reinterpret_cast<double*>(rhsv);   // [N] gradient accumulator
        for (uint32_t i = lane; i < (uint32_t)(11 * N + 16); i += 64) SH[i] = 0.0;
        for (uint32_t row = lane; row < m_rows; row += 64) {
            // dense Jacobian scatter (expressions.rs:993-1008): a later entry of the same column wins
            uint32_t dead = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int e2 = e + 1; e2 < 8; ++e2)
                    if (gcol[row * 8 + e] >= 0 && gcol[row * 8 + e] == gcol[row * 8 + e2]) dead |= 1u << e;
            }
            rdead[row] = (uint8_t)dead;
        }
        __syncthreads();
        // sum_k u_k w_k in index order (lbfgs.rs:213-216 and the explicit loops of :86-133)
        auto seq_dot = [&](double u, double w) -> double {
            if (lane < N) DOT[lane] = u * w;
            __syncthreads();
            double acc = 0.0;
            for (uint32_t j = 0; j < nfree; ++j) acc += DOT[j];
            __syncthreads();
            return bcast(acc, 0);  // every lane holds the same sum: tell the compiler it is wave-uniform
        };
        // sum of squared residuals in row order (utils.rs:11-19)
        auto seq_sse = [&](int buf) -> double {
            double acc = 0.0;
            for (uint32_t row = 0; row < m_rows; ++row) {
                double r = (double)R[buf * mr + row];
                acc += r * r;
            }
            return bcast(acc, 0);
        };
        // gradient J^T r, rows added in order (lbfgs.rs:199-210); zero entries of the dense J add nothing
        auto form_gradient = [&](int buf) -> double {
            if (lane < N) GR[lane] = 0.0;
            __syncthreads();
            const int e = lane & 7;
            constexpr int RB = 4;
            for (uint32_t row0 = 0; row0 < m_rows; row0 += RB) {
                int col[RB];
                double val[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    uint32_t row = min(row0 + q, m_rows - 1);
                    col[q] = gcol[row * 8 + e];
                    bool dead = (rdead[row] >> e) & 1;
                    val[q] = (double)G[(buf * mr + row) * 8 + e] * (double)R[buf * mr + row];
                    if (row0 + q >= m_rows || lane >= 8 || dead) col[q] = -1;
                }
#pragma unroll
                for (int q = 0; q < RB; ++q)
                    if (col[q] >= 0) lds_add(&GR[col[q]], val[q]);  // one row per instruction: row order
            }
            __syncthreads();
            double g = (lane < N) ? GR[lane] : 0.0;
            __syncthreads();
            return g;
        };

        trials = 1;
        double prev = seq_sse(0);
        sse = (T)prev;
        sse_start = sse;
        double x = ((uint32_t)lane < nfree) ? (double)XS[fidx[lane]] : 0.0;
        if (!(prev == prev)) {
            exit_code = FX_EXIT_NAN;
        } else if (prev < LbfgsConst::START_THRESHOLD) {  // :54-56
            exit_code = FX_EXIT_SSE;
        } else {
            double grad = form_gradient(0);
            for (uint32_t k = 0; k < LbfgsConst::MAX_ITERATIONS; ++k) {
                const uint32_t hl = k < 5u ? k : 5u;
                // two-loop recursion (:86-139); note the ring index (k + i) % 5 of the reference
                double dir = grad;
                double alpha[5] = {0., 0., 0., 0., 0.};
#pragma unroll
                for (int i = 4; i >= 0; --i) {
                    if ((uint32_t)i < hl) {
                        const uint32_t h = (k + (uint32_t)i) % 5u;
                        double dp = seq_dot((lane < N) ? SH[h * N + lane] : 0.0, dir);
                        alpha[i] = RHO[h] * dp;
                        dir -= alpha[i] * ((lane < N) ? YH[h * N + lane] : 0.0);
                    }
                }
                if (k > 0) {
                    const uint32_t h = (k - 1u) % 5u;
                    double sv = (lane < N) ? SH[h * N + lane] : 0.0, yv = (lane < N) ? YH[h * N + lane] : 0.0;
                    double s_dot_y = seq_dot(sv, yv);
                    double y_dot_y = seq_dot(yv, yv);
                    if (y_dot_y > 0.) dir *= s_dot_y / y_dot_y;
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if ((uint32_t)i < hl) {
                        const uint32_t h = (k + (uint32_t)i) % 5u;
                        double dp = seq_dot((lane < N) ? YH[h * N + lane] : 0.0, dir);
                        double beta = RHO[h] * dp;
                        dir += ((lane < N) ? SH[h * N + lane] : 0.0) * (alpha[i] - beta);
                    }
                }
                dir *= -1.;
                if ((uint32_t)lane >= nfree) dir = 0.0;
                const uint32_t h = k % 5u;
                if (lane < N) YH[h * N + lane] = grad;  // the old gradient, turned into y_k below (:143-146)

                // Hager-Zhang line search (:148-163) around the one evaluation site
                HzMachine hz;
                double step = hz.start(prev, seq_dot(grad, dir));
                HzParam acc_pt{0., 0., 0.};
                for (;;) {
                    if ((uint32_t)lane < nfree) XS[vt + fidx[lane]] = (T)(x + step * dir);  // calculate_phi (:270-284)
                    __syncthreads();
                    eval_rows(1);
                    trials += 1;
                    double phi = seq_sse(1);
                    grad = form_gradient(1);
                    double dphi = seq_dot(grad, dir);
                    if (hz.feed(HzParam{step, phi, dphi}, step, acc_pt)) break;
                }
                x = x + acc_pt.p * dir;  // == the scratch vector of the last evaluation (:166)
                double sk = acc_pt.p * dir;
                double yk = grad - ((lane < N) ? YH[h * N + lane] : 0.0);
                if (lane < N) {
                    SH[h * N + lane] = sk;
                    YH[h * N + lane] = yk;
                }
                double s_dot_y = seq_dot(sk, yk);
                if (lane == 0) RHO[h] = 1.0 / s_dot_y;
                __syncthreads();
                accepted += 1;
                sse = (T)acc_pt.phi;
                if (hz.capped) {
                    exit_code = FX_EXIT_TRIAL_CAP;
                    break;
                }
                if (!(acc_pt.phi == acc_pt.phi)) {
                    exit_code = FX_EXIT_NAN;
                    break;
                }
                if (::fabs(prev - acc_pt.phi) < LbfgsConst::CONVERGENCE_THRESHOLD) {  // :183-185
                    exit_code = FX_EXIT_FTOL;
                    break;
                }
                if (acc_pt.phi < LbfgsConst::RESIDUAL_THRESHOLD) {  // :186-188
                    exit_code = FX_EXIT_SSE;
                    break;
                }
                prev = acc_pt.phi;
            }
            if ((uint32_t)lane < nfree) XS[fidx[lane]] = (T)x;
            __syncthreads();
        }
        } else {
        // ================= Optimizer::LevenbergMarquardt (lm.rs:21-193) =================
        stamp(PH_EVAL);
        if (!qr) form_normal(0);
        stamp(PH_FORM);
        T diag = (lane < N && !qr) ? Amat[lane * LD + lane] : T(1);
        T rhs_l = (lane < N && !qr) ? rhsv[lane] : T(0);

        double lambda = o.lambda0;
        bool done = false;
        if (!(sse == sse) || !(sse < Lim<T>::huge()) || qr_bad_plan) {
            exit_code = FX_EXIT_NAN;
            done = true;
        }

        for (uint32_t outer = 0; outer < o.max_outer && !done; ++outer) {
            if (sse < (T)o.sse_tol) {  // lm.rs:110-112
                exit_code = FX_EXIT_SSE;
                break;
            }
            for (;;) {  // lambda trials, lm.rs:115-191
                if (trials >= o.max_trials) {
                    exit_code = FX_EXIT_TRIAL_CAP;
                    done = true;
                    break;
                }
                trials += 1;
                T delta = T(0), dn2 = T(0);
                if (qr) {
                    if (!uniform(qr_step(lambda, cur, delta, dn2))) {  // lm.rs:134-137
                        lambda *= o.singular_factor;
                        continue;
                    }
                } else {
                // K4: factor (JtJ + lambda I) and solve for delta
                T a[N];
                if (lane < N) {
                    Amat[lane * LD + lane] = diag + (T)lambda;  // same lane reads it back: LDS is in order
                    using V = typename Vec16<T>::type;
                    constexpr int VN = Vec16<T>::n;
                    const V* col = reinterpret_cast<const V*>(Amat + lane * LD);
#pragma unroll
                    for (int i = 0; i < N; i += VN) {
                        V t = col[i / VN];
                        const T* tv = reinterpret_cast<const T*>(&t);
#pragma unroll
                        for (int q = 0; q < VN; ++q) a[i + q] = tv[q];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) a[i] = T(0);
                }
                T invd = T(1);
                stamp(PH_TAIL);
                bool solved = chol_factor<N, T>(a, invd, lane);
                stamp(PH_FACTOR);
                if (!uniform(solved)) {  // lm.rs:134-137
                    lambda *= o.singular_factor;
                    continue;
                }
                delta = chol_solve<N, T>(a, invd, rhs_l, lane);
                if ((uint32_t)lane >= nfree) delta = T(0);
                if (o.solver == FX_STEP_CHOLESKY_REFINED) {
                    // One step of refinement on the least-squares problem itself (corrected semi-normal
                    // equations): t = -r - J delta from the Jacobian rows, not from JtJ, then
                    // (JtJ + lambda I) e = Jt t - lambda delta with the factor at hand. Brings the step to
                    // the accuracy of the reference's QR on ill-conditioned sketches for ~15 % more work.
                    T* dls = rhsv;                  // delta by column, then the refinement right-hand side
                    T* tr = R + (cur ^ 1) * mr;     // per-row t (the trial buffer is overwritten later anyway)
                    __syncthreads();
                    if (lane < N) dls[lane] = delta;
                    __syncthreads();
                    for (uint32_t row = lane; row < m_rows; row += 64) {
                        T acc = -R[cur * mr + row];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int cc = gcol[row * 8 + e];
                            if (cc >= 0) acc -= G[(cur * mr + row) * 8 + e] * dls[cc];
                        }
                        tr[row] = acc;
                    }
                    __syncthreads();
                    if (lane < N) dls[lane] = T(0);
                    __syncthreads();
                    if (use_packed) {  // Jt t over the packed entry list
                        const T* Gb = G + (size_t)cur * mr * 8;
                        for (uint32_t t = lane; t < n_pe; t += 64) {
                            const uint32_t w = PE[t];
                            const uint32_t row = w >> 9;
                            lds_add(&dls[w & 63u], Gb[row * 8u + ((w >> 6) & 7u)] * tr[row]);
                        }
                    } else {
                        for (uint32_t row = 0; row < m_rows; ++row) {  // one row per instruction
                            if (lane < 8) {
                                const int cc = gcol[row * 8 + lane];
                                if (cc >= 0) lds_add(&dls[cc], G[(cur * mr + row) * 8 + lane] * tr[row]);
                            }
                        }
                    }
                    __syncthreads();
                    T g2 = (lane < N) ? dls[lane] - (T)lambda * delta : T(0);
                    T corr = chol_solve<N, T>(a, invd, g2, lane);
                    if ((uint32_t)lane < nfree) delta += corr;
                    __syncthreads();
                }
                dn2 = wave_sum(delta * delta);
                }  // normal-equation step
                stamp(PH_SOLVE);
                if (!(dn2 == dn2)) {
                    exit_code = FX_EXIT_NAN;
                    done = true;
                    break;
                }
                if (dn2 < (T)o.step_tol) {  // lm.rs:139-142
                    exit_code = FX_EXIT_STEP;
                    done = true;
                    break;
                }
                // K2/K1 at the trial point (gradient kept: it becomes J on acceptance)
                const int trial = cur ^ 1;
                if ((uint32_t)lane < nfree) {
                    uint32_t vi = fidx[lane];
                    XS[trial * vt + vi] = XS[cur * vt + vi] + delta;
                }
                __syncthreads();
                stamp(PH_TAIL);
                T sse_t = eval_rows(trial);
                if (qr) sse_t = (T)seq_sse_lm(trial);
                stamp(PH_EVAL);
                if (sse_t < sse) {  // accept, lm.rs:151-186
                    lambda *= o.accept_factor;
                    if (lambda < o.lambda_min) lambda = o.lambda_min;
                    cur = trial;
                    accepted += 1;
                    T rel = (sse - sse_t) / sse;
                    sse = sse_t;  // the returned point's SSE (the reference leaves it stale, quirk Q9)
                    if (rel <= (T)o.ftol) {
                        exit_code = FX_EXIT_FTOL;
                        done = true;
                        break;
                    }
                    __syncthreads();
                    stamp(PH_TAIL);
                    if (!qr) {
                        form_normal(cur);
                        diag = (lane < N) ? Amat[lane * LD + lane] : T(1);
                        rhs_l = (lane < N) ? rhsv[lane] : T(0);
                    }
                    stamp(PH_FORM);
                    break;
                } else {  // reject, lm.rs:187-190
                    lambda *= o.reject_factor;
                    if (!(sse_t == sse_t) && !(lambda < 1.0e300)) {
                        // NaN trial point: the reference would double lambda forever
                        exit_code = FX_EXIT_NAN;
                        done = true;
                        break;
                    }
                    if (sizeof(T) == 4 && sse_t - sse <= (T)o.ftol * sse) {
                        // f32 only: a rejected trial within ftol of the current SSE is round-off — stagnated
                        // (see fx_lm_opts_default_f32, and the same branch in fx_grouped.hip)
                        exit_code = FX_EXIT_FTOL;
                        done = true;
                        break;
                    }
                }
            }
        }

        }  // optimizer

        // ---- K6: write back scale * x for the free variables (:161-166) ----------------------
        if ((uint32_t)lane < nfree) {
            uint32_t vi = fidx[lane];
            double x = (double)XS[cur * vt + vi];
            double xo = (prm.mode & 1u) ? scale * x : x;
            b.vars[v0 + vi] = xo;
            VOUT[vi] = xo;
            if (UNITS) colof[vi] = (int16_t)-1;
            if (UNITS && !(unit_flags & UNIT_RESTORE)) {  // SinglePass also updates the working vector (:201-207)
                T xv = XS[cur * vt + vi];
                XS[vi] = xv;
                XS[vt + vi] = xv;
            } else {
                // later components are solved against the PRE-solve snapshot (only `system.variables` is
                // written back, quirk Q2): restore the perturbed start value in both halves
                XS[vi] = xstart;
                XS[vt + vi] = xstart;
            }
        }
        __syncthreads();
        tot_accept += accepted;
        tot_trials += trials;
        last_exit = exit_code;
        tot_sse0 += (double)sse_start;
        tot_sse += (double)sse;
        if (!UNITS) comps_done += 1;
    }

    // ---- post-solve check on unscaled variables (constraints/mod.rs:96-109) ------------------
    __syncthreads();
    const double* XD = VOUT;
    double part = 0.0;
    for (uint32_t i = lane; i < net; i += 64) {
        int tag = ld_tag(i);
        const ushort4 f4 = ld_idx(i);
        uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        expand_vars<POSE>(tag, ff, vars8);
        double v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = XD[vars8[e]];
        double r = eval_expression<double, false, CR_ATAN2, POSE>(tag, v, ld_param(i), g);
        part += r * r;
    }
    double sse_u = wave_sum(part);

    if (lane == 0) {
        fx_result res;
        res.accepted = tot_accept;
        res.trials = tot_trials;
        res.exit = last_exit;
        res.ncomp = comps_done;
        res.scale = scale;
        res.sse0 = tot_sse0;
        res.sse = tot_sse;
        res.sse_unscaled = sse_u;
        b.results[s] = res;
    }
    if (PROF) {
        stamp(PH_TAIL);
        if (lane == 0 && prm.prof) {
            for (int i = 0; i < PH_COUNT; ++i) atomicAdd(&prm.prof[i], ph[i]);
        }
    }
}

template <int N, typename T, bool PROF, bool UNITS, int OPT = 0, bool GLOBAL = false>
__global__ __launch_bounds__(64) void lm_solve_kernel(DeviceBatch b, LmParams prm, SolveLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    lm_solve_body<N, T, PROF, UNITS, OPT, GLOBAL, false>(b, prm, L, smem);
}
// the FX_STEP_QR instantiations: up to 32 columns the register window of the QR step fits a 256-register budget
// (two wavefronts per SIMD hide its LDS round trips); wider builds take what they need
template <int N, bool PROF, bool UNITS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void lm_solve_qr_kernel_w2(DeviceBatch b, LmParams prm, SolveLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    lm_solve_body<N, double, PROF, UNITS, 0, false, true>(b, prm, L, smem);
}
template <int N, bool PROF, bool UNITS>
__global__ __launch_bounds__(64) void lm_solve_qr_kernel(DeviceBatch b, LmParams prm, SolveLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    lm_solve_body<N, double, PROF, UNITS, 0, false, true>(b, prm, L, smem);
}

// cluster problems of Decomposer::RecursiveAssembly (fx_recursive.h): the same solve with the two pose rows
// of fx_expr.h switched on. Builds of 16, 32 and 64 columns: a step of one System is a single small problem, but the
// batched arm solves that step for every System of a group at once (10 000 triangles: 10 000 problems of 9 columns,
// which the 64-column build runs at a quarter of the 16-column build's occupancy); FX_STEP_QR keeps to one build.
template <bool QR, int N = 64>
__global__ __launch_bounds__(64) void lm_solve_pose_kernel(DeviceBatch b, LmParams prm, SolveLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    lm_solve_body<N, double, false, false, 0, false, QR, true>(b, prm, L, smem);
}

__global__ __launch_bounds__(256) void identity_residual_kernel(DeviceBatch b, const double* __restrict__ x,
                                                                double* __restrict__ out) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= b.n_exprs) return;
    uint32_t v0 = b.expr_var0[row];
    int tag = b.expr_tag[row] & 0x7F;
    ushort4 f4 = reinterpret_cast<const ushort4*>(b.expr_idx)[row];
    uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    expand_vars(tag, ff, vars8);
    double v[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = x[v0 + vars8[e]];
    out[row] = eval_expression<double, false>(tag, v, b.expr_param[row], g);
}

// ------------------------------------------------------------------------------------------
// System::analyze — over-constraint detection (fiksi/src/analyze/numerical/mod.rs:33-163)
// ------------------------------------------------------------------------------------------
// One wavefront per System. Dense Jacobian of ALL expressions w.r.t. ALL variables (all free,
// unscaled, duplicates overwrite: expressions.rs:1003-1007) in LDS, then the reference's row-wise
// incremental Gauss-Jordan elimination with column swaps. Lanes own columns; every matrix element
// sees exactly the reference's sequence of operations (no reductions), so the verdict is
// bit-for-bit the reference's. dependent[e] = 1 when expression e does not increase the rank.
__global__ __launch_bounds__(64) void analyze_kernel(DeviceBatch b, const double* __restrict__ x, uint32_t ld_m,
                                                     uint32_t ld_n, uint8_t* __restrict__ dependent) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t s = blockIdx.x;
    const uint32_t v0 = b.var_off[s], n = b.var_off[s + 1] - v0;
    const uint32_t e0 = b.expr_off[s], m = b.expr_off[s + 1] - e0;
    double* M = reinterpret_cast<double*>(smem);                       // [m][n] row-major
    uint16_t* colidx = reinterpret_cast<uint16_t*>(M + (size_t)ld_m * ld_n);  // [n]
    uint8_t* inc = reinterpret_cast<uint8_t*>(colidx + ld_n);          // [m]
    const double EPSILON = 1e-8;  // numerical/mod.rs:8

    for (uint32_t i = lane; i < m * n; i += 64) M[i] = 0.0;
    for (uint32_t i = lane; i < n; i += 64) colidx[i] = (uint16_t)i;
    for (uint32_t i = lane; i < m; i += 64) inc[i] = 0;
    __syncthreads();
    for (uint32_t row = lane; row < m; row += 64) {  // numerical/mod.rs:135-140
        int tag = b.expr_tag[e0 + row] & 0x7F;
        const uint16_t* f = b.expr_idx + 4 * (size_t)(e0 + row);
        uint16_t ff[4] = {f[0], f[1], f[2], f[3]};
        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int k = expand_vars(tag, ff, vars8);
        double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[v0 + vars8[e]];
        eval_expression<double, true>(tag, v, b.expr_param[e0 + row], g);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < k) M[row * n + vars8[e]] = g[e];  // later duplicates overwrite
    }
    __syncthreads();

    uint32_t current_col = 0;
    const uint32_t steps = min(m, n);
    for (uint32_t row = 0; row < steps; ++row) {
        uint32_t rank = 0;
        for (uint32_t row_idx = 0; row_idx < row; ++row_idx) {
            const uint32_t column_idx = colidx[rank];
            const double factor = M[row * n + column_idx];  // read by every lane before anyone writes it
            __syncthreads();
            for (uint32_t col = lane; col < n; col += 64) M[row * n + col] = M[row * n + col] - factor * M[row_idx * n + col];
            __syncthreads();
            if (inc[row_idx]) rank += 1;
        }
        // first column (in pivot order) with |value| > eps
        uint32_t found = 0xFFFFFFFFu;
        for (uint32_t base = current_col; base < n && found == 0xFFFFFFFFu; base += 64) {
            uint32_t idx = base + lane;
            bool hit = idx < n && ::fabs(M[row * n + colidx[min(idx, n - 1)]]) > EPSILON;
            uint64_t mask = __ballot(hit);
            if (mask) found = base + (uint32_t)__builtin_ctzll(mask);
        }
        if (found == 0xFFFFFFFFu) continue;  // all-zero row: dependent
        __syncthreads();
        if (lane == 0) {
            uint16_t t = colidx[current_col];
            colidx[current_col] = colidx[found];
            colidx[found] = t;
        }
        __syncthreads();
        const uint32_t pc = colidx[current_col];
        const double pivot = M[row * n + pc];
        const double inv = 1. / pivot;
        __syncthreads();
        for (uint32_t col = lane; col < n; col += 64) M[row * n + col] = M[row * n + col] * inv;
        __syncthreads();
        for (uint32_t row_idx = 0; row_idx < row; ++row_idx) {
            const double fct = M[row_idx * n + pc];
            __syncthreads();
            for (uint32_t col = lane; col < n; col += 64) M[row_idx * n + col] = M[row_idx * n + col] - fct * M[row * n + col];
            __syncthreads();
        }
        current_col += 1;
        if (lane == 0) inc[row] = 1;
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t i = lane; i < m; i += 64) dependent[e0 + i] = inc[i] ? 0 : 1;
}

size_t analyze_lds_bytes(uint32_t max_vars, uint32_t max_exprs) {
    return (size_t)max_vars * max_exprs * 8u + (size_t)max_vars * 2u + max_exprs + 64u;
}

hipError_t launch_analyze(const DeviceBatch& b, const double* x, uint32_t max_vars, uint32_t max_exprs,
                          uint8_t* dependent, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    uint32_t ld_n = (max_vars + 3u) & ~3u, ld_m = max_exprs;
    size_t lds = (size_t)ld_m * ld_n * 8u + (size_t)ld_n * 2u + ld_m + 64u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    static unsigned int raised = 0;
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&analyze_kernel), &raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(analyze_kernel, dim3(b.n_systems), dim3(64), lds, stream, b, x, ld_m, ld_n, dependent);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// the GLOBAL instantiation: one wavefront per listed large System
template <int N>
static hipError_t launch_solve_global_n(const DeviceBatch& b, const LmParams& p, const SolveLayout& L, hipStream_t stream) {
    if (L.total > 160u * 1024u) return hipErrorInvalidValue;
    static unsigned int raised = 0;  // (one per N: this is a template)
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&lm_solve_kernel<N, double, false, true, 0, true>), &raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((lm_solve_kernel<N, double, false, true, 0, true>), dim3(b.n_g), dim3(64), L.total, stream, b, p, L);
    return hipGetLastError();
}

static hipError_t launch_solve_global(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    uint32_t n = pad_n(b.max_unit_free_g);
    SolveLayout L = make_layout(n, 8u, b.max_unit_rows_g, 8u, false, b.max_pairs_g, b.max_ents_g);  // System-wide vectors are not in LDS
    switch (n) {
        case 8: return launch_solve_global_n<8>(b, p, L, stream);
        case 16: return launch_solve_global_n<16>(b, p, L, stream);
        case 24: return launch_solve_global_n<24>(b, p, L, stream);
        case 32: return launch_solve_global_n<32>(b, p, L, stream);
        case 40: return launch_solve_global_n<40>(b, p, L, stream);
        case 48: return launch_solve_global_n<48>(b, p, L, stream);
        case 56: return launch_solve_global_n<56>(b, p, L, stream);
        case 64: return launch_solve_global_n<64>(b, p, L, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_solve_walk(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (b.n_g == 0) return hipSuccess;
    return launch_solve_global(b, p, stream);
}

template <int N, typename T, bool PROF, bool UNITS, int OPT, bool QR = false>
static hipError_t launch_solve_n(const DeviceBatch& b, const LmParams& p, const SolveLayout& L, hipStream_t stream) {
    if (L.total > 160u * 1024u) return hipErrorInvalidValue;
    void (*fn)(DeviceBatch, LmParams, SolveLayout);
    if constexpr (QR && N <= 32) fn = &lm_solve_qr_kernel_w2<N, PROF, UNITS>;
    else if constexpr (QR) fn = &lm_solve_qr_kernel<N, PROF, UNITS>;
    else fn = &lm_solve_kernel<N, T, PROF, UNITS, OPT>;
    // the attribute is raised once per instantiation and device (a call per launch costs a single small solve a few
    // microseconds of its ~50): 160 KB is what any layout may ask for
    static std::atomic<uint32_t> raised_on{0};  // bit d: done on device d
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    if (!(raised_on.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised_on.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(fn, dim3(b.n_systems), dim3(64), L.total, stream, b, p, L);
    return hipGetLastError();
}

template <typename T, bool UNITS, int OPT>
static hipError_t launch_solve_t(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    uint32_t n = pad_n(UNITS ? b.max_unit_free : b.max_free);
    const uint32_t rows = (UNITS && b.max_unit_rows > b.max_rows) ? b.max_unit_rows : b.max_rows;
    // block walking: blocks of up to 8 rows never list their products (see the kernel), so no LDS for it
    const bool lists = !UNITS || b.max_unit_rows > 8u;
    // FX_STEP_QR: the plans of the host's symbolic analysis size the augmented matrix
    const QrPlans& Q = UNITS ? b.qr_units : b.qr_none;
    const bool qr = p.lm.solver == FX_STEP_QR;
    if (qr && (sizeof(T) != 8 || OPT != 0 || !Q.desc)) return hipErrorInvalidValue;
    SolveLayout L = make_layout(n, b.max_vars, rows, (uint32_t)sizeof(T), OPT == 1, lists ? b.max_pairs : 0u, lists ? b.max_ents : 0u,
                                qr ? Q.max_m : 0u, qr ? Q.max_h : 0u);
    if constexpr (sizeof(T) == 8 && OPT == 0) {
        if (qr) {
            switch (n) {
                case 8: return launch_solve_n<8, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 16: return launch_solve_n<16, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 24: return launch_solve_n<24, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 32: return launch_solve_n<32, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 40: return launch_solve_n<40, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 48: return launch_solve_n<48, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 56: return launch_solve_n<56, T, false, UNITS, OPT, true>(b, p, L, stream);
                case 64: return launch_solve_n<64, T, false, UNITS, OPT, true>(b, p, L, stream);
                default: return hipErrorInvalidValue;
            }
        }
    }
    switch (n) {
        case 8: return launch_solve_n<8, T, false, UNITS, OPT>(b, p, L, stream);
        case 16: return launch_solve_n<16, T, false, UNITS, OPT>(b, p, L, stream);
        case 24: return launch_solve_n<24, T, false, UNITS, OPT>(b, p, L, stream);
        case 32: return launch_solve_n<32, T, false, UNITS, OPT>(b, p, L, stream);
        case 40: return launch_solve_n<40, T, false, UNITS, OPT>(b, p, L, stream);
        case 48: return launch_solve_n<48, T, false, UNITS, OPT>(b, p, L, stream);
        case 56: return launch_solve_n<56, T, false, UNITS, OPT>(b, p, L, stream);
        case 64: return launch_solve_n<64, T, false, UNITS, OPT>(b, p, L, stream);
        default: return hipErrorInvalidValue;
    }
}

// Batches holding pose rows (b.has_pose, set by the upload when the RecursiveAssembly arm asks for it)
static hipError_t launch_solve_pose(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (p.lm.precision == 32 || (p.mode & (MODE_UNITS | MODE_LBFGS)) || b.max_free > 64u) return hipErrorInvalidValue;
    const bool qr = p.lm.solver == FX_STEP_QR;
    if (qr && !b.qr_none.desc) return hipErrorInvalidValue;
    const uint32_t n = qr ? 64u : b.max_free <= 16u ? 16u : b.max_free <= 32u ? 32u : 64u;
    SolveLayout L = make_layout(n, b.max_vars, b.max_rows, 8u, false, b.max_pairs, b.max_ents, qr ? b.qr_none.max_m : 0u,
                                qr ? b.qr_none.max_h : 0u);
    if (L.total > 160u * 1024u) return hipErrorInvalidValue;
    void (*fn)(DeviceBatch, LmParams, SolveLayout) = qr ? &lm_solve_pose_kernel<true> : n == 16u ? &lm_solve_pose_kernel<false, 16> :
                                                     n == 32u ? &lm_solve_pose_kernel<false, 32> : &lm_solve_pose_kernel<false, 64>;
    static unsigned int raised[4] = {0, 0, 0, 0};  // one slot per pose build
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(fn), &raised[qr ? 0 : n == 16u ? 1 : n == 32u ? 2 : 3]);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(b.n_systems), dim3(64), L.total, stream, b, p, L);
    return hipGetLastError();
}

hipError_t launch_solve(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    if (b.has_pose) return launch_solve_pose(b, p, stream);
    if (p.prof && grouped_applies(b, p)) return launch_solve_grouped(b, p, stream);
    if (p.prof) {  // diagnostic build, instantiated for the headline shape only
        uint32_t n = pad_n(b.max_free);
        if (n != 32 || p.lm.precision == 32 || (p.mode & (MODE_UNITS | MODE_LBFGS))) return hipErrorInvalidValue;
        if (p.lm.solver == FX_STEP_QR) {
            if (!b.qr_none.desc) return hipErrorInvalidValue;
            SolveLayout L = make_layout(n, b.max_vars, b.max_rows, 8u, false, b.max_pairs, b.max_ents, b.qr_none.max_m, b.qr_none.max_h);
            return launch_solve_n<32, double, true, false, 0, true>(b, p, L, stream);
        }
        SolveLayout L = make_layout(n, b.max_vars, b.max_rows, 8u, false, b.max_pairs, b.max_ents);
        return launch_solve_n<32, double, true, false, 0>(b, p, L, stream);
    }
    const bool units = (p.mode & MODE_UNITS) != 0;
    if (units && !b.sys_unit_off) return hipErrorInvalidValue;
    if (p.mode & MODE_LBFGS) {  // f64 only (checked at the ABI)
        if (p.lm.precision == 32) return hipErrorInvalidValue;
        return units ? launch_solve_t<double, true, 1>(b, p, stream) : launch_solve_t<double, false, 1>(b, p, stream);
    }
    if (units) {
        // Systems beyond one wavefront: the f64 walker; reference numerics stop at one wavefront, they take the refined step
        // (as fx_step_solver documents)
        LmParams pg = p;
        if (pg.lm.solver == FX_STEP_QR) pg.lm.solver = FX_STEP_CHOLESKY_REFINED;
        if (p.lm.precision == 32) {
            hipError_t e = grouped_applies(b, p) ? launch_solve_grouped(b, p, stream) : launch_solve_t<float, true, 0>(b, p, stream);
            if (e == hipSuccess && b.n_g) e = launch_solve_global(b, pg, stream);
            return e;
        }
        hipError_t e = grouped_applies(b, p) ? launch_solve_grouped(b, p, stream) : launch_solve_t<double, true, 0>(b, p, stream);
        if (e == hipSuccess && b.n_g) e = launch_solve_global(b, pg, stream);
        return e;
    }
    // components of at most 32 free variables: several Systems per wavefront (fx_grouped.hip)
    if (grouped_applies(b, p)) return launch_solve_grouped(b, p, stream);
    return p.lm.precision == 32 ? launch_solve_t<float, false, 0>(b, p, stream) : launch_solve_t<double, false, 0>(b, p, stream);
}

// Problem::calculate_residuals_and_jacobian (subsystem.rs:106-124): the dense row-major Jacobian of
// every System, [n_exprs_s x n_free_s] at dense_off[s]; one thread per row zeroes its row and then writes
// the partials in gradient order, so a later entry of the same column overwrites an earlier one
// (expressions.rs:993-1008, quirk Q4). var_rank: system-local free rank of every variable or 0xFFFF.
__global__ __launch_bounds__(256) void dense_jacobian_kernel(DeviceBatch b, const double* __restrict__ x,
                                                             const uint16_t* __restrict__ var_rank,
                                                             const uint32_t* __restrict__ expr_sys,
                                                             const uint16_t* __restrict__ sys_nfree,
                                                             const uint64_t* __restrict__ dense_off,
                                                             double* __restrict__ resid, double* __restrict__ jac) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= b.n_exprs) return;
    const uint32_t s = expr_sys[row];
    const uint32_t v0 = b.var_off[s], e0 = b.expr_off[s];
    const uint32_t nfree = sys_nfree[s];
    const int tag = b.expr_tag[row] & 0x7F;
    ushort4 f4 = reinterpret_cast<const ushort4*>(b.expr_idx)[row];
    uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int k = expand_vars(tag, ff, vars8);
    double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = x[v0 + vars8[e]];
    resid[row] = eval_expression<double, true>(tag, v, b.expr_param[row], g);
    double* out = jac + dense_off[s] + (uint64_t)(row - e0) * nfree;
    for (uint32_t c = 0; c < nfree; ++c) out[c] = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < k) {
            const uint16_t c = var_rank[v0 + vars8[e]];
            if (c != 0xFFFFu) out[c] = g[e];
        }
    }
}

hipError_t launch_dense_jacobian(const DeviceBatch& b, const double* x, const uint16_t* var_rank, const uint32_t* expr_sys,
                                 const uint16_t* sys_nfree, const uint64_t* dense_off, double* resid, double* jac,
                                 hipStream_t stream) {
    if (b.n_exprs == 0) return hipSuccess;
    dim3 grid((b.n_exprs + 255u) / 256u), block(256);
    hipLaunchKernelGGL(dense_jacobian_kernel, grid, block, 0, stream, b, x, var_rank, expr_sys, sys_nfree, dense_off, resid, jac);
    return hipGetLastError();
}

hipError_t launch_identity_residuals(const DeviceBatch& b, const double* x, double* out, hipStream_t stream) {
    if (b.n_exprs == 0) return hipSuccess;
    dim3 grid((b.n_exprs + 255u) / 256u), block(256);
    hipLaunchKernelGGL(identity_residual_kernel, grid, block, 0, stream, b, x, out);
    return hipGetLastError();
}

}  // namespace fx
