// Grouped fused solve (gfx950, wave64): FOUR Systems per wavefront, one per row of 16 lanes, for batches
// whose components have at most 32 free variables — the headline shape (BASELINE cfg3: 32 variables /
// 32 expressions per System) and everything smaller — and up to 48 (the reference's own bench sketch,
// fiksi_bench.rs:15-40: 46 variables); Decomposer::None and SinglePass, f64 and f32.
//
// Same algorithm and the same arithmetic as lm_solve_kernel (fx_kernels.hip; reference:
// fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193), reorganised around what that kernel's
// profile shows: with N <= 32 columns only half of a wavefront's lanes hold a column of the
// register-resident Cholesky, and every multiply-add of it pays two v_readlane for its broadcast. Here
//   * a System lives in one DPP row: lane r of the row holds columns r, r + 16 (and r + 32) of the matrix
//     (NC = 1, 2 or 3 columns per lane), and a broadcast is the DPP control `row_newbcast:k` — on gfx90a+
//     also legal on 64-bit operations, so the factor's inner step is ONE instruction,
//     `v_fmac_f64_dpp a, -src row_newbcast:k, mul`, serving the four Systems of the wavefront at once, with no
//     SGPR and no LDS round trip (NC = 3 leaves the DPP to the compiler: `v_mov_b64_dpp` + three fma);
//   * what was wave-uniform (lambda, SSE, trial counts, exit code) is row-uniform and kept per lane; the LM
//     control flow is a per-row state machine (NEXT System -> COMPonent set-up -> RUN trials -> FINISH
//     component), so rows only wait for each other inside one wave instruction;
//   * rows take Systems from a device-side counter: a row that finishes early starts its next System
//     while its neighbours are still iterating (trial counts are very uneven: ring16 3 .. 30+);
//   * LDS per System is cut to 10 KB for the headline shape (JtJ as a packed lower triangle, one
//     Jacobian-row buffer — the trial point's rows are only read after it was accepted —, the current point
//     in registers, set-up scratch aliased with the triangle), so a CU still holds 16 Systems;
//   * a batch whose Systems all share one structure (DeviceBatch::uniform) keeps a row's lists between Systems.
// Sums (SSE, |delta|^2, the system scale) are taken in the order lm_solve_kernel takes them, and the
// factorization does the same operations on the same operands: results are bit-identical to that kernel's
// on the ring16 and hinged-triangle shapes (tests/test_gpu_grouped.py). DESIGN.md section 3.1a.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_grouped_rows.h"
#include "fx_wave.h"

#ifndef FX_GROUPED_LADDER
#define FX_GROUPED_LADDER 1  // 0: a build without the lambda ladder (A/B measurements, tools/ab_build.sh)
#endif

namespace fx {

struct GroupLayout {
    uint32_t vt, mr;  // padded variables per System / rows per component
    uint32_t off_xs, off_a, off_rhs, off_g, off_r, off_p, off_gvar, off_rtag, off_vout, off_pw, off_pe;
    uint32_t off_colof, off_gcol, off_fidx;  // set-up scratch inside the triangle's bytes
    uint32_t pw_cap, pe_cap;
    uint32_t stride;  // bytes per System
    // the FX_STEP_QR build: the program of the batch's one structure (shared by the wavefront, at the start of its LDS) and a
    // System's matrix stored by its symbolic patterns
    uint32_t tab_bytes, off_qx;
    uint32_t ent_in_lds;  // the steps' per-entry words are part of the LDS copy (they fit beside four wavefronts per CU)
};

static GroupLayout make_group_layout(uint32_t n, uint32_t max_vars, uint32_t max_rows, uint32_t es, uint32_t max_pairs_tri,
                                     uint32_t max_ents, uint32_t qrg_words = 0, uint32_t qrg_nx = 0, uint32_t qrg_ng = 0) {
    GroupLayout L;
    L.vt = (max_vars + 7u) & ~7u;
    L.mr = (max_rows + 7u) & ~7u;
    if (L.vt == 0) L.vt = 8;
    if (L.mr == 0) L.mr = 8;
    uint32_t o = 0;
    auto al = [](uint32_t bytes) { return (bytes + 15u) & ~15u; };
    auto take = [&](uint32_t bytes) { uint32_t at = o; o += al(bytes); return at; };
    L.off_xs = take(L.vt * es);
    // (the QR build forms no normal equations: no triangle, no right-hand side vector; its set-up scratch lies in the bytes of
    // the matrix, which is written for the first time after the set-up)
    const uint32_t tri = qrg_words ? qrg_nx * 8u : n * (n + 1u) / 2u * es;
    const uint32_t scratch = al(L.vt * 2u) + al(L.mr * 8u) + al(n * 2u);
    L.off_a = take(tri > scratch ? tri : scratch);
    L.off_qx = L.off_a;
    L.off_colof = L.off_a;
    L.off_gcol = L.off_colof + al(L.vt * 2u);
    L.off_fidx = L.off_gcol + al(L.mr * 8u);
    L.off_rhs = take(qrg_words ? 0u : n * es);
    L.off_g = take(qrg_words ? qrg_ng * es : L.mr * 8u * es);  // (QR build: compact rows, one entry per variable of the kind)
    L.off_r = take(L.mr * es);
    L.off_p = take(L.mr * es);
    L.off_gvar = take(L.mr * 16u);
    L.off_rtag = take(L.mr);
    L.off_vout = take(L.vt * 8u);
    L.pw_cap = (max_pairs_tri + 3u) & ~3u;
    L.pe_cap = (max_ents + 7u) & ~7u;
    L.off_pw = take(L.pw_cap * 4u);
    L.off_pe = take(L.pe_cap * 2u);
    L.tab_bytes = al(qrg_words * 4u);
    L.ent_in_lds = 0u;
    L.stride = o;
    return L;
}


// PROF = true is the diagnostic build (fx_debug_phase_cycles): s_memtime stamps at the phase boundaries, summed
// per phase over the wavefront (all four rows) into prm.prof — same six phases as lm_solve_kernel's.
enum GPhase { GH_SETUP = 0, GH_EVAL = 1, GH_FORM = 2, GH_FACTOR = 3, GH_SOLVE = 4, GH_TAIL = 5, GH_COUNT = 6 };

// UNITS = true is `Decomposer::SinglePass` (assemble/mod.rs:169-210), as in lm_solve_kernel: the loop runs over the
// blocks of the host decomposition (fx_decompose.h), a component is perturbed before its first block, and a solved
// block is written through to the working vector so later blocks see it.
// QRG = true is FX_STEP_QR for batches of one structure (one component of at most 32 columns): the LM step is the
// reference's Householder QR of [J; sqrt(lambda) I] (solvi/src/decomposition/sparse/qr.rs:226-356) with the operations and
// their order of the one-wavefront QR kernel (fx_kernels.hip: qr_step), driven by the host's program (build_qrg_program) —
// the lanes of a row are the ACTIVE columns of the Householder step at hand (about seven of 33 for the headline shape), and
// the wavefront serves four Systems. Sums are the reference's sequential ones, the angle residuals use the correctly rounded
// atan2: every bit is that kernel's, and so the reference algorithm's.
template <int NC, typename T, bool PROF, bool UNITS, bool QRG = false>
__device__ __forceinline__ void grouped_body(const DeviceBatch& b, const LmParams& prm, const GroupLayout& L,
                                             uint32_t* __restrict__ next_system, unsigned char* smem) {
    unsigned long long ph[GH_COUNT] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_last = 0;
    // (the stamps sit in divergent control flow, so the compiler keeps these sums per lane: a lane only sees the
    // stamps of the blocks its row takes part in, and the time its row sits out lands on its next stamp. Lane 0's
    // sums are reported: row 0's view of the wavefront's time.)
    auto stamp = [&](int phase_id) {
        if (PROF) {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[phase_id] += t - t_last;
            t_last = t;
        }
    };
    if (PROF) t_last = __builtin_amdgcn_s_memtime();
    constexpr int N = RS * NC;
    constexpr int CB = (NC == 1) ? 4 : (NC == 2) ? 5 : 6;  // column bits of a packed entry
    constexpr uint32_t CMASK = (1u << CB) - 1u;
    const int lane = threadIdx.x;
    const int hl = lane & (RS - 1);
    const int gbase = lane & ~(RS - 1);
    const uint32_t below = (1u << hl) - 1u;
    unsigned char* const rows0 = smem + L.tab_bytes;  // the four Systems' blocks, behind the shared program of the QR build
    unsigned char* base = rows0 + (uint32_t)(lane / RS) * L.stride;
    if constexpr (QRG) {  // the program of the batch's one structure, once per wavefront
        // (the per-entry offsets of the steps — 13 KB for the headline shape — stay in global memory when they do not fit beside
        // four wavefronts per CU: read by every wavefront of the device, they are at home in the L1 / L2 caches)
        const uint32_t nw = L.ent_in_lds ? b.qr_none.qrg_words : b.qr_none.qrg_small;
        const uint4* src = reinterpret_cast<const uint4*>(b.qr_none.qrg);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = lane; i < nw / 4u; i += 64) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    T* XS = reinterpret_cast<T*>(base + L.off_xs);       // [vt] working variables: snapshot, trial point on the free ones
    T* At = reinterpret_cast<T*>(base + L.off_a);        // packed lower triangle of JtJ (+ lambda on the diagonal per trial)
    T* rhsv = reinterpret_cast<T*>(base + L.off_rhs);    // [N] -Jt r
    T* G = reinterpret_cast<T*>(base + L.off_g);         // [mr][8] Jacobian rows of the last evaluated point
    T* R = reinterpret_cast<T*>(base + L.off_r);         // [mr]
    T* P = reinterpret_cast<T*>(base + L.off_p);         // [mr] scaled parameters
    uint16_t* gvar = reinterpret_cast<uint16_t*>(base + L.off_gvar);  // [mr][8]
    uint8_t* rtag = reinterpret_cast<uint8_t*>(base + L.off_rtag);    // [mr]
    double* VOUT = reinterpret_cast<double*>(base + L.off_vout);       // [vt] unscaled values as written back
    uint32_t* PW = reinterpret_cast<uint32_t*>(base + L.off_pw);       // products of the triangle: row<<19 | a<<16 | b<<13 | address
    uint16_t* PE = reinterpret_cast<uint16_t*>(base + L.off_pe);       // entries of the right-hand side: row<<(3+CB) | a<<CB | column
    // set-up scratch, dead before the first assembly zeroes the triangle
    int16_t* colof = reinterpret_cast<int16_t*>(base + L.off_colof);   // [vt] variable -> free column
    int8_t* gcol = reinterpret_cast<int8_t*>(base + L.off_gcol);       // [mr][8] free column or -1
    uint16_t* fidx = reinterpret_cast<uint16_t*>(base + L.off_fidx);   // [N] free column -> variable

    const fx_lm_opts o = prm.lm;
    auto gballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> gbase) & 0xFFFFu; };
    auto tri_at = [](uint32_t r, uint32_t cc) -> uint32_t { return r * (r + 1u) / 2u + cc; };  // r >= cc

    // per-row state (identical in every lane of the row unless noted)
    int phase = GP_NEXT;
    uint32_t s = 0, v0 = 0, nvt = 0, e0 = 0, net = 0, ncomp = 0, c = 0;  // UNITS: ncomp / c count blocks
    uint32_t unit0 = 0, unit_flags = 0;
    double scale = 1.0, scale_recip = 1.0;
    uint32_t rng = 42u;
    uint32_t tot_accept = 0, tot_trials = 0, last_exit = FX_EXIT_SSE, comps_done = 0;
    double tot_sse0 = 0.0, tot_sse = 0.0;
    uint32_t nfree = 0, m_rows = 0, n_pw = 0, n_pe = 0;
    // per lane and column q (column hl + 16 q): its variable, current point, perturbed start, diagonal, right-hand side
    uint32_t my_vi[NC];
    T xc[NC], xstart[NC], diag[NC], rhs_l[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        my_vi[q] = 0;
        xc[q] = xstart[q] = rhs_l[q] = T(0);
        diag[q] = T(1);
    }
    T sse = T(0), sse_start = T(0);
    double lambda = 0.0;
    uint32_t accepted = 0, trials = 0, outer = 0, exit_code = FX_EXIT_MAX_OUTER;
    bool fresh = false;  // RUN evaluates the component's start point instead of a trial point
    uint32_t held = 0;   // passes this row has waited, done, for company (see FINISH)
    // The lambda ladder (prm.ladder). The trials that follow a plain reject read the same point, Jacobian and residuals
    // and differ in lambda only (x reject_factor each, lm.rs:187-190), so they can be made side by side: a row without a
    // System of its own (the queue is empty, or the wavefront holds a straggler near the end of the queue) joins a
    // running row of its wavefront as rank 1, 2 or 3 of that System's group and tries lambda x reject_factor^rank in
    // the same pass. The verdicts are read in rank order and the first that is not a plain reject decides for the whole
    // group exactly as it would have decided in the sequential loop; `trials` advances by the trials the sequential loop
    // would have made up to it. Every row of a group holds the same LM state at the top of every pass.
    constexpr bool LADDER = (NC <= 2) && !PROF && FX_GROUPED_LADDER;
    const int myrow = lane / RS;
    int lad_rank = 0, lad_width = 1, lad_lead = myrow;
    uint32_t lad_members = (uint32_t)myrow * 0x55u;  // row of rank k at bits 2k, 2k + 1
    int win_row = myrow;  // the row whose trial decided this pass (its Jacobian rows are the accepted point's)
    bool qdone = false;   // this row has seen the end of the queue
    uint32_t last_tk = 0; // the last ticket this row drew
    // The product list of a component does not change between its assemblies: the 32-column f64 build (one wavefront
    // per SIMD, registers to spare) keeps each lane's first PWR / PER list words in registers, which takes the list
    // read — one of three dependent LDS round trips per batch of products — out of every assembly.
    constexpr int PWR = (NC == 2) ? 28 : 0;
    constexpr int PER = (NC == 2) ? 10 : 0;
    uint32_t pw_reg[PWR > 0 ? PWR : 1], pe_reg[PER > 0 ? PER : 1];
#pragma unroll
    for (int u = 0; u < (PWR > 0 ? PWR : 1); ++u) pw_reg[u] = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < (PER > 0 ? PER : 1); ++u) pe_reg[u] = 0xFFFFFFFFu;
    // Batches whose Systems all have the same structure (one sketch, many parameter sets): the row lists, the
    // product lists and the free-variable map of a single-component System are built for the first System a row
    // takes and kept for the following ones — only values change. Per lane: the column of variable hl + 16 k and
    // the row of expression hl + 16 k (or -1).
    bool built = false;
    uint32_t cls = 0xFFFFFFFFu, built_cls = 0xFFFFFFFFu;  // structure class of the System / of the System the lists are from
    int c_col[3] = {-1, -1, -1}, c_row[3] = {-1, -1, -1};

    // The System's own arrays, fetched in one round trip when the row takes the System: lane r keeps
    // elements r and r + 16 of every per-variable / per-expression array (all of them for the headline
    // shape); elements past 32 are loaded where they are used.
    constexpr int PF = NC >= 3 ? 3 : 2;  // chunks of 16 kept in registers (the selects below pick among up to three)
    double c_var[PF], c_param[PF];
    uint32_t c_info[PF], c_tag[PF], c_comp[PF];
    ushort4 c_idx[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        c_var[k] = c_param[k] = 0.0;
        c_info[k] = c_tag[k] = c_comp[k] = 0;
        c_idx[k] = make_ushort4(0, 0, 0, 0);
    }
    // f(i, value, info) over all variables / f(i, tag, param, comp, idx) over all expressions of the System,
    // 16 at a time in ascending order; i may lie past the end (then the other arguments are zero / 0xFFFF)
    // (one dynamic loop each, the cached chunks picked by selects: one copy of every loop body in the code)
    auto pick = [&](uint32_t at, const auto& arr) {
        auto r = arr[PF - 1];
        if constexpr (PF >= 3) r = (at == (uint32_t)RS) ? arr[1] : r;
        return (at == 0u) ? arr[0] : r;
    };
    auto for_vars = [&](auto&& f) {
#pragma unroll 1
        for (uint32_t at = 0; at < nvt; at += RS) {
            const uint32_t i = at + (uint32_t)hl;
            double v = pick(at, c_var);
            uint32_t info = pick(at, c_info);
            if (at >= (uint32_t)(RS * PF)) {
                const bool have = i < nvt;
                v = have ? b.vars0[v0 + i] : 0.0;
                info = have ? (uint32_t)b.var_info[v0 + i] : 0xFFFFu;
            }
            f(i, v, info);
        }
    };
    auto for_exprs = [&](auto&& f) {
#pragma unroll 1
        for (uint32_t at = 0; at < net; at += RS) {
            const uint32_t i = at + (uint32_t)hl;
            int tag = (int)pick(at, c_tag);
            double param = pick(at, c_param);
            uint32_t comp = pick(at, c_comp);
            ushort4 idx = pick(at, c_idx);
            if (at >= (uint32_t)(RS * PF)) {
                const bool have = i < net;
                tag = have ? (int)(b.expr_tag[e0 + i] & 0x7F) : 0;
                param = have ? b.expr_param[e0 + i] : 0.0;
                comp = have ? (uint32_t)b.expr_comp[e0 + i] : 0xFFFFu;
                idx = have ? reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + i] : make_ushort4(0, 0, 0, 0);
            }
            f(i, tag, param, comp, idx);
        }
    };

    // FULL: residuals and Jacobian rows of the point in XS, stored (a component's start point; in the QR build also the point
    // of an accepted trial, evaluated a second time — its trial only needed the residuals, and so the rows of the CURRENT
    // point, which every one of the QR step's trials reads, are never overwritten by a rejected one)
    auto eval_rows = [&](auto full_c) -> T {
        constexpr bool FULL = decltype(full_c)::value;
        T part[4] = {T(0), T(0), T(0), T(0)};
        for (uint32_t row = hl; row < m_rows; row += RS) {
            T v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = XS[gvar[row * 8 + e]];
            T r = eval_expression<T, FULL, QRG>(rtag[row], v, P[row], g);  // (QR build: the correctly rounded atan2)
            if constexpr (FULL) {
                R[row] = r;
                if constexpr (QRG) {  // compact: the entries of the row's kind only (the program's gbase)
                    const uint32_t* TBq = reinterpret_cast<const uint32_t*>(smem);
                    const uint32_t gb = reinterpret_cast<const uint16_t*>(TBq + TBq[15])[row];
                    const int kk = tag_nvars<true>((int)rtag[row]);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < kk) G[gb + (uint32_t)e] = g[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) G[row * 8 + e] = g[e];
                }
            }
            const T r2 = r * r;
            const uint32_t blk = (row >> 4) & 3u;
#pragma unroll
            for (int q = 0; q < 4; ++q) part[q] += (blk == (uint32_t)q) ? r2 : T(0);
        }
        group_sync();
        if constexpr (QRG) {  // the reference's sum of squares, in index order (lm.rs:195-197): row 16 q + lane of block q
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((uint32_t)(RS * q) < m_rows) seq_add(acc, (double)part[q]);  // (rows past the end: + 0.0, exact)
            return (T)acc;
        } else {
            return block_sum4(part);
        }
    };
    // One LM trial's step by the reference's QR, table-driven (build_qrg_program). Returns false when R has an exactly zero
    // diagonal entry (sparse_col_mat.rs:800-810); delta: this lane's free columns hl, hl + 16.
    auto qr_step = [&](double lam, T (&delta)[NC]) -> bool {
        bool ok = true;
        if constexpr (QRG) {
            const uint32_t* TB = reinterpret_cast<const uint32_t*>(smem);
            const uint32_t qn = TB[0], qm = TB[1], nx = TB[2], rhsbase = TB[12];  // (offsets into the matrix are in BYTES)
            const uint16_t* scat = reinterpret_cast<const uint16_t*>(TB + TB[4]);
            const uint16_t* rhs_off = reinterpret_cast<const uint16_t*>(TB + TB[5]);
            const uint16_t* damp = reinterpret_cast<const uint16_t*>(TB + TB[6]);
            const uint32_t* stepT = TB + TB[8];
            const uint16_t* bptr = reinterpret_cast<const uint16_t*>(TB + TB[9]);
            const uint32_t* bent = TB + TB[10];
            const uint32_t* ENT = L.ent_in_lds ? TB : b.qr_none.qrg;
            unsigned char* X = base + L.off_qx;
            auto xat = [&](uint32_t byte_off) -> double& { return *reinterpret_cast<double*>(X + byte_off); };
            const double sl = ::sqrt(lam);  // lm.rs:119
            {
                double2 z;
                z.x = z.y = 0.0;
                for (uint32_t i = hl; i < nx / 2u; i += RS) reinterpret_cast<double2*>(X)[i] = z;
            }
            group_sync();
            // J (duplicates of a row summed in gradient order, sparse_col_mat.rs:710-711) and b = -r (lm.rs:86-91,130);
            // the damping entry of every column (lm.rs:92-96,119-125)
            const uint16_t* gbaseT = reinterpret_cast<const uint16_t*>(TB + TB[15]);
            for (uint32_t row = hl; row < qm; row += RS) {
                const uint32_t gb = gbaseT[row];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t off = scat[row * 8 + e];
                    if (off != 0xFFFFu) lds_add(&xat(off), (double)G[gb + (uint32_t)e]);
                }
                xat(rhs_off[row]) = -(double)R[row];
            }
            for (uint32_t c = hl; c < qn; c += RS) xat(damp[c]) = sl;
            group_sync();
            for (uint32_t k = 0; k < qn; ++k) {
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)stepT[3 * k]);
                const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)stepT[3 * k + 1]);
                const uint32_t nact = (uint32_t)__builtin_amdgcn_readfirstlane((int)stepT[3 * k + 2]);
                const uint32_t vdiag = w0 & 0xFFFFu, len = w0 >> 16;
                const double v0 = xat(vdiag);
                double norm = 0.0, beta = 0.0, v0n = 1.0;
                // calculate_householder (qr.rs:244-275) on column k below the diagonal; every lane computes it
                auto householder = [&](double sigma) {
                    norm = ::fabs(v0);
                    beta = (v0 >= 0.0) ? 0.0 : 2.0;
                    v0n = 1.0;
                    if (sigma != 0.0) {
                        norm = ::sqrt(sigma + v0 * v0);
                        v0n = (v0 <= 0.0) ? v0 - norm : -sigma / (v0 + norm);
                        beta = -(1.0 / (norm * v0n));
                    }
                };
                for (uint32_t i0 = 0; i0 < nact; i0 += RS) {
                    const bool act = i0 + (uint32_t)hl < nact;
                    const uint32_t* e = ENT + ent + (act ? i0 + (uint32_t)hl : 0u);  // (idle lanes follow column 0 and store nothing)
                    // the vector and this lane's column under it, NB blocks of four entries, fetched once (one copy of the
                    // body per block count, picked once per vector)
                    auto window = [&](auto nb_c) {
                        constexpr int NE = 4 * decltype(nb_c)::value;
                        uint32_t w[NE > 0 ? NE : 1];
                        double vk[NE > 0 ? NE : 1], xj[NE > 0 ? NE : 1];
                        const uint32_t wk0 = e[0];
#pragma unroll
                        for (int u = 0; u < NE; ++u) w[u] = e[(uint32_t)(1 + u) * nact];
                        const double xk0 = xat(wk0);
#pragma unroll
                        for (int u = 0; u < NE; ++u) {
                            vk[u] = xat(w[u] >> 16);
                            xj[u] = xat(w[u] & 0xFFFFu);
                        }
                        double sigma = 0.0;
#pragma unroll
                        for (int u = 0; u < NE; ++u) sigma = sigma + vk[u] * vk[u];
                        householder(sigma);
                        // apply_householder (qr.rs:226-240)
                        double tau = 0.0;
                        tau = tau + v0n * xk0;
#pragma unroll
                        for (int u = 0; u < NE; ++u) tau = tau + vk[u] * xj[u];
                        tau = tau * beta;
                        if (act) {
                            xat(wk0) = xk0 - v0n * tau;
#pragma unroll
                            for (int u = 0; u < NE; ++u) xat(w[u] & 0xFFFFu) = xj[u] - vk[u] * tau;  // (padding: 0 - 0 tau into the zero slot)
                        }
                    };
                    switch (len >> 2) {
                        case 0: window(std::integral_constant<int, 0>{}); break;
                        case 1: window(std::integral_constant<int, 1>{}); break;
                        case 2: window(std::integral_constant<int, 2>{}); break;
                        case 3: window(std::integral_constant<int, 3>{}); break;
                        case 4: window(std::integral_constant<int, 4>{}); break;
                        case 5: window(std::integral_constant<int, 5>{}); break;
                        case 6: window(std::integral_constant<int, 6>{}); break;
                        case 7: window(std::integral_constant<int, 7>{}); break;
                        default: window(std::integral_constant<int, 8>{}); break;
                    }
                    group_sync();
                }
                if (hl == 0) xat(vdiag) = norm;  // R's diagonal (qr.rs:319)
                group_sync();
            }
            // back substitution with R (sparse_col_mat.rs:788-826), in place on the right-hand side: column i of R from the
            // last to the first, one lane per entry above the diagonal
            {
                bool zero_diag = false;
                for (uint32_t c = hl; c < qn; c += RS) zero_diag = zero_diag || xat(stepT[3 * c] & 0xFFFFu) == 0.0;
                ok = gballot(zero_diag) == 0u;
            }
            if (ok) {
                for (uint32_t ii = qn; ii > 0; --ii) {
                    const uint32_t i = ii - 1u;
                    const uint32_t dgo = (uint32_t)__builtin_amdgcn_readfirstlane((int)stepT[3 * i]) & 0xFFFFu;
                    const uint32_t eb = (uint32_t)__builtin_amdgcn_readfirstlane((int)bptr[i]);
                    const uint32_t ee = (uint32_t)__builtin_amdgcn_readfirstlane((int)bptr[i + 1]);
                    const double coeff = xat(rhsbase + 8u * i) / xat(dgo);
                    for (uint32_t t = eb + (uint32_t)hl; t < ee; t += RS) {
                        const uint32_t wv = bent[t];
                        const uint32_t yo = rhsbase + (wv >> 16);
                        xat(yo) = xat(yo) - coeff * xat(wv & 0xFFFFu);
                    }
                    if (hl == 0) xat(rhsbase + 8u * i) = coeff;
                    group_sync();
                }
            }
            // undo the column permutation (qr.rs:354): delta[colperm[j]] = x_j
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const uint32_t c = (uint32_t)(hl + RS * q);
                const uint16_t* cposT = reinterpret_cast<const uint16_t*>(TB + TB[7]);
                delta[q] = (ok && c < qn) ? (T)xat(rhsbase + 8u * (uint32_t)cposT[c < qn ? c : 0u]) : T(0);
            }
            group_sync();
        }
        return ok;
    };
    // K3: the lower triangle of Jt J and -Jt r from the packed lists (ds_add_f64 / ds_add_f32)
    auto form_normal = [&]() {
        {  // zero the triangle, 16 bytes per lane and instruction
            using V = typename Vec16<T>::type;
            constexpr uint32_t NV = (uint32_t)(N * (N + 1) / 2) / (uint32_t)Vec16<T>::n;
            static_assert((N * (N + 1) / 2) % Vec16<T>::n == 0, "triangle is a whole number of 16-byte vectors");
            V z;
            for (int q = 0; q < Vec16<T>::n; ++q) reinterpret_cast<T*>(&z)[q] = T(0);
            for (uint32_t i = hl; i < NV; i += RS) reinterpret_cast<V*>(At)[i] = z;
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) rhsv[hl + RS * q] = T(0);
        group_sync();
        constexpr int U = 4;
        if constexpr (PWR > 0) {  // the list words held in registers: seven products' factors in flight at a time
            constexpr int UB = 7;
#pragma unroll
            for (int u0 = 0; u0 < PWR; u0 += UB) {
                T g1[UB], g2[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const uint32_t ww = (pw_reg[u0 + u] == 0xFFFFFFFFu) ? 0u : pw_reg[u0 + u];
                    const uint32_t gb = (ww >> 19) * 8u;
                    g1[u] = G[gb + ((ww >> 16) & 7u)];
                    g2[u] = G[gb + ((ww >> 13) & 7u)];
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (pw_reg[u0 + u] != 0xFFFFFFFFu) lds_add(&At[pw_reg[u0 + u] & 0x1FFFu], g1[u] * g2[u]);
            }
            {
                T g1[PER], rr[PER];
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const uint32_t ww = (pe_reg[u] == 0xFFFFFFFFu) ? 0u : pe_reg[u];
                    const uint32_t row = ww >> (3 + CB);
                    g1[u] = G[row * 8u + ((ww >> CB) & 7u)];
                    rr[u] = -R[row];
                }
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    if (pe_reg[u] != 0xFFFFFFFFu) lds_add(&rhsv[pe_reg[u] & CMASK], g1[u] * rr[u]);
            }
        }
        // the rest of the lists (all of them in the other builds) from LDS, four products per lane at a time: their
        // list words first, then their eight factors, then the four atomics (three LDS round trips per batch)
        for (uint32_t t0 = (uint32_t)(RS * PWR); t0 < n_pw; t0 += RS * U) {
            uint32_t w[U];
            T g1[U], g2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t t = t0 + (uint32_t)(u * RS + hl);
                w[u] = (t < n_pw) ? PW[t] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                const uint32_t gb = (ww >> 19) * 8u;
                g1[u] = G[gb + ((ww >> 16) & 7u)];
                g2[u] = G[gb + ((ww >> 13) & 7u)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (w[u] != 0xFFFFFFFFu) lds_add(&At[w[u] & 0x1FFFu], g1[u] * g2[u]);
        }
        for (uint32_t t0 = (uint32_t)(RS * PER); t0 < n_pe; t0 += RS * U) {
            uint32_t w[U];
            T g1[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t t = t0 + (uint32_t)(u * RS + hl);
                w[u] = (t < n_pe) ? (uint32_t)PE[t] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                const uint32_t row = ww >> (3 + CB);
                g1[u] = G[row * 8u + ((ww >> CB) & 7u)];
                rr[u] = -R[row];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (w[u] != 0xFFFFFFFFu) lds_add(&rhsv[w[u] & CMASK], g1[u] * rr[u]);
        }
        group_sync();
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const uint32_t j = (uint32_t)(hl + RS * q);
            if (j >= nfree) At[tri_at(j, j)] = T(1);  // identity padding
        }
        group_sync();
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const uint32_t j = (uint32_t)(hl + RS * q);
            diag[q] = At[tri_at(j, j)];
            rhs_l[q] = rhsv[j];
        }
    };

    for (;;) {
        // Ladder: near the end of the queue a wavefront that holds a straggler — a System past prm.ladder_k trials —
        // stops taking Systems: its rows go idle as they finish and join the straggler's ladder (below). Without a
        // straggler a row that went idle that way goes back to the queue.
        bool park = false;
        if constexpr (LADDER) {
            if (prm.ladder && prm.ladder_tail != 0u) {
                const bool straggler =
                    __ballot(phase == GP_RUN && lad_rank == 0 && !fresh && trials >= prm.ladder_k) != 0ull;
                if (phase == GP_EXIT && !qdone && !straggler) phase = GP_NEXT;
                park = straggler && last_tk < b.n_systems && b.n_systems - last_tk <= prm.ladder_tail;
            }
        }
        // ================= NEXT: take a System, scale it, snapshot its variables =================
        if (phase == GP_NEXT && park) phase = GP_EXIT;
        if (phase == GP_NEXT) {
            uint32_t nxt;
            for (;;) {
                uint32_t tk = 0;
                if (hl == 0) {
                    tk = atomicAdd(next_system, 1u);
                    last_tk = tk;
                    if (tk >= b.n_systems) {
                        tk = 0xFFFFFFFFu;  // the queue is empty (b.n_systems is the QUEUE's length: an order list over a part of the
                                           // batch — launch_class_solves — holds System numbers beyond it)
                    } else if (b.order) {  // a schedule: presort, or an earlier solve of this batch; or a part of the batch
                        // the first round of tickets transposed, so that the Systems at the head of the schedule — the
                        // likely stragglers — go to different wavefronts (a straggler's ladder is the rows of ITS wavefront)
                        uint32_t pos = tk;
                        if (tk < 4u * prm.spread) pos = (tk & 3u) * prm.spread + (tk >> 2);
                        tk = b.order[pos];
                    }
                }
                nxt = (uint32_t)__shfl((int)tk, 0, RS);
                last_tk = (uint32_t)__shfl((int)last_tk, 0, RS);
                // large Systems belong to the other paths (a batch of one shared structure has none here)
                if (nxt == 0xFFFFFFFFu || b.uniform || !b.sys_large[nxt]) break;
            }
            if (nxt == 0xFFFFFFFFu) {
                phase = GP_EXIT;
                qdone = true;
            } else {
                s = nxt;
                if (b.uniform) {  // offsets are multiples of the common sizes: one round trip less
                    nvt = b.u_nvars;
                    net = b.u_nexprs;
                    v0 = s * nvt;
                    e0 = s * net;
                    ncomp = b.u_ncomp;
                } else {
                    v0 = b.var_off[s];
                    nvt = b.var_off[s + 1] - v0;
                    e0 = b.expr_off[s];
                    net = b.expr_off[s + 1] - e0;
                    ncomp = b.sys_ncomp[s];
                    cls = b.sys_class ? b.sys_class[s] : 0xFFFFFFFFu;
                }
                if constexpr (UNITS) {
                    unit0 = b.sys_unit_off[s];
                    ncomp = b.sys_unit_off[s + 1] - unit0;
                }
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const uint32_t i = (uint32_t)(RS * k + hl);
                    const bool hv_ = i < nvt, he_ = i < net;
                    c_var[k] = hv_ ? b.vars0[v0 + i] : 0.0;
                    c_info[k] = hv_ ? (uint32_t)b.var_info[v0 + i] : 0xFFFFu;
                    c_tag[k] = he_ ? (uint32_t)(b.expr_tag[e0 + i] & 0x7F) : 0u;
                    c_param[k] = he_ ? b.expr_param[e0 + i] : 0.0;
                    c_comp[k] = he_ ? (uint32_t)b.expr_comp[e0 + i] : 0xFFFFu;
                    c_idx[k] = he_ ? reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + i] : make_ushort4(0, 0, 0, 0);
                }
                // All of these are needed right away, so wait for them here, explicitly: otherwise the loads only
                // used by the COMP block stay "possibly in flight" for the compiler on the path that skips it, and
                // the trial code gets vmcnt(0) waits that in fact wait for this block's global STORES to land.
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                // K0a: system scale, summed strictly in reference order (assemble/mod.rs:32-44, utils.rs:11-33)
                scale = 1.0;
                scale_recip = 1.0;
                if (prm.mode & 1u) {
                    double sum = 0.0;
                    uint32_t count = nvt;
                    for_vars([&](uint32_t i, double v, uint32_t) {
                        seq_add(sum, i < nvt ? v * v : 0.0);  // lanes past the end add +0.0: exact
                    });
                    for_exprs([&](uint32_t i, int tag, double d, uint32_t, ushort4) {
                        const bool isd = i < net && ((tag == FX_TAG_PPD) || (tag == FX_TAG_PLD));
                        count += (uint32_t)__popc(gballot(isd));
                        seq_add(sum, isd ? d * d : 0.0);
                    });
                    scale = ::sqrt(sum / (double)count);
                    scale_recip = 1.0 / scale;
                }
                for_vars([&](uint32_t i, double v, uint32_t) {
                    if (i < nvt) {
                        XS[i] = (T)((prm.mode & 1u) ? v * scale_recip : v);
                        VOUT[i] = v;
                        b.vars[v0 + i] = v;  // fixed / unconstrained variables stay bit-identical
                    }
                });
                group_sync();
                rng = 42u;  // one Rng::from_seed(42) per solve, shared by the components (assemble/mod.rs:47)
                tot_accept = 0;
                tot_trials = 0;
                last_exit = FX_EXIT_SSE;
                comps_done = 0;
                tot_sse0 = 0.0;
                tot_sse = 0.0;
                c = 0;
                phase = GP_COMP;
            }
            stamp(GH_SETUP);
        }

        // ================= COMP: set up component (SinglePass: block) c =================
        if (phase == GP_COMP) {
            if (c < ncomp) {
                const bool reuse = !UNITS && (b.uniform != 0u || (cls != 0xFFFFFFFFu && cls == built_cls)) && built && ncomp == 1u && nvt <= (uint32_t)(RS * PF) && net <= (uint32_t)(RS * PF);
                if (reuse) {
                    // same structure as the System before: perturb and re-scale the parameters, nothing else
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const uint32_t i = (uint32_t)(RS * k + hl);
                        if (c_col[k] >= 0 && (prm.mode & 2u)) {
                            uint32_t st = lcg_jump(rng, 2u * (uint32_t)c_col[k]);
                            st = st * 1664525u + 1013904223u;
                            const double f1 = (1.0 / 4294967295.0) * (double)st;
                            st = st * 1664525u + 1013904223u;
                            const double f2 = (1.0 / 4294967295.0) * (double)st;
                            double x = (prm.mode & 1u) ? c_var[k] * scale_recip : c_var[k];
                            x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                            XS[i] = (T)x;
                        }
                        if (c_row[k] >= 0) {
                            const int tag = (int)c_tag[k];
                            double prm_e = c_param[k];
                            if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                            P[c_row[k]] = (T)prm_e;
                        }
                    }
                    if (prm.mode & 2u) rng = lcg_jump(rng, 2u * nfree);
                    group_sync();
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        xstart[q] = ((uint32_t)(hl + RS * q) < nfree) ? XS[my_vi[q]] : T(0);
                        xc[q] = xstart[q];
                    }
                }
                bool have_comp = reuse;
                bool rows_listed = false;
                if constexpr (UNITS) {
                    const UnitDesc ud = b.unit_desc[unit0 + c];
                    unit_flags = ud.flags;
                    if (ud.flags & UNIT_FIRST) {  // the component's perturbation comes before its first block (:91-111)
                        comps_done += 1;
                        last_exit = FX_EXIT_SSE;
                        if (prm.mode & 2u) {
                            uint32_t rank0 = 0;
                            for_vars([&](uint32_t i, double v, uint32_t info) {
                                const bool in = i < nvt && (info & VAR_COMP_MASK) == ud.comp && !(info & VAR_FIXED_BIT);
                                const uint32_t m = gballot(in);
                                if (in) {
                                    uint32_t st = lcg_jump(rng, 2u * (rank0 + (uint32_t)__popc(m & below)));
                                    st = st * 1664525u + 1013904223u;
                                    const double f1 = (1.0 / 4294967295.0) * (double)st;
                                    st = st * 1664525u + 1013904223u;
                                    const double f2 = (1.0 / 4294967295.0) * (double)st;
                                    double x = (prm.mode & 1u) ? v * scale_recip : v;
                                    x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                                    XS[i] = (T)x;
                                }
                                rank0 += (uint32_t)__popc(m);
                            });
                            rng = lcg_jump(rng, 2u * rank0);
                        }
                    }
                    if (ud.flags & UNIT_EMPTY) {
                        c += 1;  // a component no expression could be matched in
                    } else {
                        have_comp = true;
                        rows_listed = true;
                        nfree = ud.nvars;
                        m_rows = ud.nrows;
                        for_vars([&](uint32_t i, double, uint32_t) {
                            if (i < nvt) colof[i] = (int16_t)-1;
                        });
                        group_sync();
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const uint32_t j = (uint32_t)(hl + RS * q);
                            my_vi[q] = 0u;
                            if (j < nfree) {
                                my_vi[q] = (uint32_t)b.unit_vars[ud.var_off + j];
                                colof[my_vi[q]] = (int16_t)j;
                            }
                        }
                        group_sync();
                        for (uint32_t pos = hl; pos < m_rows; pos += RS) {  // rows in block order
                            const uint32_t i = b.unit_rows[ud.row_off + pos];
                            const int tag = (int)(b.expr_tag[e0 + i] & 0x7F);
                            const ushort4 f4 = reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + i];
                            uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
                            uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                            const int k = expand_vars(tag, ff, vars8);
                            double prm_e = b.expr_param[e0 + i];
                            if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                            rtag[pos] = (uint8_t)tag;
                            P[pos] = (T)prm_e;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                gvar[pos * 8 + e] = (uint16_t)vars8[e];
                                gcol[pos * 8 + e] = (e < k) ? (int8_t)colof[vars8[e]] : (int8_t)-1;
                            }
                        }
                    }
                }
                if (!UNITS && !reuse) {
                    // free variables of the component, ascending (BTreeSet order, assemble/mod.rs:91-111)
                    // (perturbed where they are found — K0b: 2 LCG draws each, in that order)
                    nfree = 0;
                    bool any_var = false;
                    for_vars([&](uint32_t i, double v, uint32_t info) {
                        const bool member = i < nvt && (info & VAR_COMP_MASK) == c;
                        const bool in = member && !(info & VAR_FIXED_BIT);
                        const uint32_t m = gballot(in);
                        const uint32_t pos = nfree + (uint32_t)__popc(m & below);
                        if (i < nvt) colof[i] = in ? (int16_t)pos : (int16_t)-1;
                        if (in && pos < (uint32_t)N) fidx[pos] = (uint16_t)i;
                        if (i < (uint32_t)(RS * PF)) {
                            const int cc = in ? (int)pos : -1;
                            if (i < (uint32_t)RS) c_col[0] = cc; else if (i < 2u * RS) c_col[1] = cc; else c_col[2] = cc;
                        }
                        if (in && (prm.mode & 2u)) {
                            uint32_t st = lcg_jump(rng, 2u * pos);
                            st = st * 1664525u + 1013904223u;
                            const double f1 = (1.0 / 4294967295.0) * (double)st;
                            st = st * 1664525u + 1013904223u;
                            const double f2 = (1.0 / 4294967295.0) * (double)st;
                            // from the f64 input, so the f64 start point is bit-identical to the reference
                            double x = (prm.mode & 1u) ? v * scale_recip : v;
                            x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                            XS[i] = (T)x;
                        }
                        nfree += (uint32_t)__popc(m);
                        any_var = any_var || gballot(member) != 0u;
                    });
                    if (!any_var) {
                        c += 1;  // a component without variables is skipped by the reference (`elements.is_empty()`)
                    } else {
                        have_comp = true;
                        rows_listed = true;
                        built = true;
                        built_cls = cls;
                        if (prm.mode & 2u) rng = lcg_jump(rng, 2u * nfree);
                        group_sync();
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const uint32_t j = (uint32_t)(hl + RS * q);
                            my_vi[q] = (j < nfree) ? (uint32_t)fidx[j] : 0u;
                        }
                        // rows of the component: ascending expression id (assemble/mod.rs:139-145)
                        m_rows = 0;
                        for_exprs([&](uint32_t i, int tag, double prm_e, uint32_t comp, ushort4 f4) {
                            const bool in = (i < net) && (comp == c);
                            const uint32_t mk = gballot(in);
                            const uint32_t pos = m_rows + (uint32_t)__popc(mk & below);
                            if (i < (uint32_t)(RS * PF)) {
                                const int rr = in ? (int)pos : -1;
                                if (i < (uint32_t)RS) c_row[0] = rr; else if (i < 2u * RS) c_row[1] = rr; else c_row[2] = rr;
                            }
                            if (in) {
                                uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
                                uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                                const int k = expand_vars(tag, ff, vars8);
                                if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                                rtag[pos] = (uint8_t)tag;
                                P[pos] = (T)prm_e;
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    gvar[pos * 8 + e] = (uint16_t)vars8[e];
                                    gcol[pos * 8 + e] = (e < k) ? (int8_t)colof[vars8[e]] : (int8_t)-1;
                                }
                            }
                            m_rows += (uint32_t)__popc(mk);
                        });
                    }
                }  // lists of a component
                if (rows_listed) {
                    group_sync();
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        xstart[q] = ((uint32_t)(hl + RS * q) < nfree) ? XS[my_vi[q]] : T(0);
                        xc[q] = xstart[q];
                    }
                    // packed work lists: one u32 per product g_a g_b (a <= b, both columns free) of the lower
                    // triangle, one u16 per g r of the right-hand side. Two entries of a row on one column
                    // (an expression reading a variable twice) add their cross product twice, as the full
                    // symmetric assembly does.
                    n_pw = 0;
                    n_pe = 0;
                    for (uint32_t at = 0; !QRG && at < m_rows; at += RS) {
                        const uint32_t row = at + (uint32_t)hl;
                        uint32_t mask = 0;
                        uint64_t cols = 0;
                        if (row < m_rows) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int cc = gcol[row * 8 + e];
                                mask |= (cc >= 0) ? (1u << e) : 0u;
                                cols |= (uint64_t)(uint8_t)cc << (8 * e);
                            }
                        }
                        const uint32_t kf = (uint32_t)__popc(mask);
                        uint32_t np = kf * (kf + 1u) / 2u;
                        for (uint32_t m1 = mask; m1; m1 &= m1 - 1u) {
                            const uint32_t a = (uint32_t)__ffs(m1) - 1u;
                            for (uint32_t m2 = m1 & (m1 - 1u); m2; m2 &= m2 - 1u) {
                                const uint32_t bb = (uint32_t)__ffs(m2) - 1u;
                                np += (((cols >> (8u * a)) ^ (cols >> (8u * bb))) & 0xFFu) == 0u;
                            }
                        }
                        uint32_t inc2 = np, inc1 = kf;
#pragma unroll
                        for (int off = 1; off < RS; off <<= 1) {
                            const uint32_t t2 = (uint32_t)__shfl_up((int)inc2, off, RS), t1 = (uint32_t)__shfl_up((int)inc1, off, RS);
                            if (hl >= off) {
                                inc2 += t2;
                                inc1 += t1;
                            }
                        }
                        uint32_t at2 = n_pw + inc2 - np, at1 = n_pe + inc1 - kf;
                        if (at2 + np <= L.pw_cap && at1 + kf <= L.pe_cap) {
                            for (uint32_t m1 = mask; m1; m1 &= m1 - 1u) {
                                const uint32_t a = (uint32_t)__ffs(m1) - 1u;
                                const uint32_t ca = (uint32_t)(cols >> (8u * a)) & 0xFFu;
                                PE[at1++] = (uint16_t)((row << (3 + CB)) | (a << CB) | ca);
                                for (uint32_t m2 = m1; m2; m2 &= m2 - 1u) {
                                    const uint32_t bb = (uint32_t)__ffs(m2) - 1u;
                                    const uint32_t cb = (uint32_t)(cols >> (8u * bb)) & 0xFFu;
                                    const uint32_t hi = ca > cb ? ca : cb, lo = ca > cb ? cb : ca;
                                    const uint32_t w = (row << 19) | (a << 16) | (bb << 13) | tri_at(hi, lo);
                                    PW[at2++] = w;
                                    if (a != bb && ca == cb) PW[at2++] = w;
                                }
                            }
                        }
                        n_pw += (uint32_t)__shfl((int)inc2, RS - 1, RS);
                        n_pe += (uint32_t)__shfl((int)inc1, RS - 1, RS);
                    }
                    group_sync();
                    if constexpr (PWR > 0) {
#pragma unroll
                        for (int u = 0; u < PWR; ++u) {
                            const uint32_t t = (uint32_t)(u * RS + hl);
                            pw_reg[u] = (t < n_pw && t < L.pw_cap) ? PW[t] : 0xFFFFFFFFu;
                        }
#pragma unroll
                        for (int u = 0; u < PER; ++u) {
                            const uint32_t t = (uint32_t)(u * RS + hl);
                            pe_reg[u] = (t < n_pe && t < L.pe_cap) ? (uint32_t)PE[t] : 0xFFFFFFFFu;
                        }
                    }
                }  // rows_listed
                if (have_comp) {
                    stamp(GH_SETUP);
                    // the start point is evaluated and assembled by the RUN block (one copy of that code)
                    lambda = o.lambda0;
                    accepted = 0;
                    trials = 0;
                    outer = 0;
                    exit_code = FX_EXIT_MAX_OUTER;
                    fresh = true;
                    phase = GP_RUN;
                    if (n_pw > L.pw_cap || n_pe > L.pe_cap) {  // the host sizes the lists; never expected
                        sse = sse_start = T(0);
                        exit_code = FX_EXIT_NAN;
                        phase = GP_FINISH;
                    }
                }
            }
        }

        // ================= LADDER: idle rows join a running row of their wavefront =================
        if constexpr (LADDER) {
            if (prm.ladder) {
                const unsigned long long bcand = __ballot(phase == GP_RUN && !fresh && lad_rank == 0);
                const unsigned long long bidle = __ballot(phase == GP_EXIT);
                if (bcand != 0ull && bidle != 0ull) {
                    // wave-uniform bookkeeping over the four rows, packed into words (no indexed arrays: they would live in
                    // scratch): every idle row goes to the running row with the smallest group so far; rows already helping
                    // stay where they are. wid: 4 bits per leader row; mem: its members, 8 bits per leader row; newlead /
                    // newrank: 4 bits per joining row (0xF: does not join)
                    uint32_t wid = 0, mem = 0, newlead = 0xFFFFu, newrank = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        wid |= ((uint32_t)__builtin_amdgcn_readlane(lad_width, RS * r) & 15u) << (4 * r);
                        mem |= ((uint32_t)__builtin_amdgcn_readlane((int)lad_members, RS * r) & 255u) << (8 * r);
                    }
                    bool anyjoin = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!((bidle >> (RS * r)) & 1ull)) continue;
                        uint32_t best = 15u, bw = 4u;
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            const uint32_t w = (wid >> (4 * l)) & 15u;
                            if (((bcand >> (RS * l)) & 1ull) && w < bw) {
                                best = (uint32_t)l;
                                bw = w;
                            }
                        }
                        if (best != 15u) {
                            newlead = (newlead & ~(15u << (4 * r))) | (best << (4 * r));
                            newrank |= bw << (4 * r);
                            const uint32_t at = 8u * best + 2u * bw;
                            mem = (mem & ~(3u << at)) | ((uint32_t)r << at);
                            wid += 1u << (4u * best);
                            anyjoin = true;
                        }
                    }
                    if (anyjoin) {
                        const uint32_t nl = (newlead >> (4 * myrow)) & 15u;
                        const bool joining = nl != 15u;
                        const int grp = joining ? (int)nl : lad_lead;  // the row whose group this row belongs to from now on
                        // the leader's registers (every lane runs the shuffles: the source lanes must be active)
                        const int srcl = grp * RS + hl;
                        auto cp = [&](auto& v) {
                            const auto t = lane_get(v, srcl);
                            if (joining) v = t;
                        };
                        cp(nfree); cp(m_rows); cp(n_pw); cp(n_pe); cp(trials); cp(accepted); cp(outer); cp(exit_code);
                        cp(sse); cp(lambda);
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            cp(my_vi[q]); cp(xc[q]); cp(diag[q]); cp(rhs_l[q]);
                        }
                        if constexpr (PWR > 0) {
#pragma unroll
                            for (int u = 0; u < PWR; ++u) cp(pw_reg[u]);
#pragma unroll
                            for (int u = 0; u < PER; ++u) cp(pe_reg[u]);
                        }
                        lad_width = (int)((wid >> (4 * grp)) & 15u);
                        lad_members = (mem >> (8 * grp)) & 255u;
                        if (joining) {
                            lad_lead = (int)nl;
                            lad_rank = (int)((newrank >> (4 * myrow)) & 15u);
                            built = false;  // this row's lists in registers are the leader's now
                            // ... and its LDS block: working variables, the triangle of Jt J as last assembled (every row
                            // writes its own diagonal per trial), right-hand side, row lists, parameters, product lists
                            using V = typename Vec16<T>::type;
                            const V* lb = reinterpret_cast<const V*>(rows0 + (uint32_t)nl * L.stride);
                            V* mine = reinterpret_cast<V*>(base);
                            for (uint32_t i = hl; i < L.stride / 16u; i += RS) mine[i] = lb[i];
                            fresh = false;
                            phase = GP_RUN;
                        }
                        group_sync();
                    }
                }
            }
        }

        // the most columns any row of the wavefront factors in this pass (wave-uniform: the four row leaders)
        int kmax;
        {
            const int km = (phase == GP_RUN && !fresh) ? (int)nfree : 0;
            const int k0 = __builtin_amdgcn_readlane(km, 0), k1 = __builtin_amdgcn_readlane(km, 16);
            const int k2 = __builtin_amdgcn_readlane(km, 32), k3 = __builtin_amdgcn_readlane(km, 48);
            kmax = max(max(k0, k1), max(k2, k3));
        }
        // ================= RUN: one lambda trial (lm.rs:115-191) =================
        if (phase == GP_RUN) {
            stamp(GH_TAIL);
            // --- the trial: what it finds is written down as a verdict (`code`); the row's LM state is only changed
            // below, by the verdict that decides (the row's own, or — on the ladder — its group's first decisive one)
            int code = LC_FRESH;
            bool go = true;
            T delta[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) delta[q] = T(0);
            if (!fresh) {
                code = LC_REJECT;
                // this row's lambda: the group's, after `rank` plain rejects
                double lam_k = lambda;
                if constexpr (LADDER) {
                    if (lad_rank > 0)
                        for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                }
                if (trials + (uint32_t)lad_rank >= o.max_trials) {
                    code = LC_CAP;
                    go = false;
                }
                if constexpr (QRG) {
                    if (go && !qr_step(lam_k, delta)) {  // lm.rs:134-137
                        code = LC_SINGULAR;
                        go = false;
                    }
                }
                if (!QRG && go) {
                    // K4: factor (JtJ + lambda I) and solve for delta; columns hl and hl + 16 of the symmetric
                    // matrix from the triangle (the lane id goes through an opaque move so that the addresses are
                    // recomputed per trial instead of being hoisted out of the loop into dozens of long-lived
                    // registers)
                    int hv = hl;
                    asm volatile("" : "+v"(hv));
#pragma unroll
                    for (int q = 0; q < NC; ++q) At[tri_at((uint32_t)(hv + RS * q), (uint32_t)(hv + RS * q))] = diag[q] + (T)lam_k;
                    group_sync();
                    // Element (i, j) of the symmetric matrix sits at row max(i, j) of the packed triangle. For a lane's
                    // column j = hv + 16 q that is its own row for i <= j (contiguous: one base register, the element
                    // index an instruction immediate) and row i, position j, for i > j (another base register, again an
                    // immediate). Which of the two holds is known at compile time for most elements — i <= 16 q: the
                    // lane's row, whatever hv; i >= 16 (q + 1): row i — and those loads take no address arithmetic at
                    // all; only the 15 elements per array in between select between the two bases per lane. (All 64
                    // used to compute both indices and select: ~450 instructions per trial, a tenth of the trial.)
                    T a[NC][N];
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int j = hv + RS * q;
                        const T* own = At + (uint32_t)(j * (j + 1) / 2);  // the lane's row: elements (j, 0 .. j)
                        const T* col = At + (uint32_t)j;                   // + i (i + 1) / 2: element (i, j) of row i
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            const int ti = i * (i + 1) / 2;
                            if (i <= RS * q) a[q][i] = own[i];
                            else if (i >= RS * (q + 1)) a[q][i] = col[ti];
                            else a[q][i] = ((hv >= i - RS * q) ? own : col + (ti - i))[i];
                        }
                    }
                    T invd[NC];
#pragma unroll
                    for (int q = 0; q < NC; ++q) invd[q] = T(1);
                    bool bad = false;
                    RBlock<NC, T, 0, UNITS>::factor(a, invd, bad, hl, kmax);
                    stamp(GH_FACTOR);
                    if (bad) {  // lm.rs:134-137
                        code = LC_SINGULAR;
                        go = false;
                    } else {
                        T acc[NC], invd2[NC];
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            acc[q] = rhs_l[q];
                            invd2[q] = invd[q] * invd[q];
                        }
                        RBlock<NC, T, 0, UNITS>::forward(a, invd, acc, hl, kmax);
                        RBlock<NC, T, N / 8 - 1, UNITS>::backward(a, invd2, acc, hl, kmax);
#pragma unroll
                        for (int q = 0; q < NC; ++q) delta[q] = ((uint32_t)(hl + RS * q) < nfree) ? acc[q] * invd2[q] : T(0);
                    }
                }
                if (go) {
                    // |delta|^2 block by block of 16 columns, as wave_sum adds it (QR build: in index order, lm.rs:195-197)
                    T dn2;
                    if constexpr (QRG) {
                        double acc = 0.0;
#pragma unroll
                        for (int q = 0; q < NC; ++q)
                            if ((uint32_t)(RS * q) < nfree) seq_add(acc, (double)delta[q] * (double)delta[q]);
                        dn2 = (T)acc;
                    } else {
                        dn2 = row_sum(delta[0] * delta[0]);
                        if constexpr (NC >= 2) dn2 = dn2 + row_sum(delta[1] * delta[1]);
                        if constexpr (NC >= 3) dn2 = dn2 + row_sum(delta[2] * delta[2]);
                    }
                    if (!(dn2 == dn2)) {
                        code = LC_NAN;
                        go = false;
                    } else if (dn2 < (T)o.step_tol) {  // lm.rs:139-142
                        code = LC_STEP;
                        go = false;
                    }
                    stamp(GH_SOLVE);
                }
                if (go) {
                    // K2/K1 at the trial point (its rows become J on acceptance)
#pragma unroll
                    for (int q = 0; q < NC; ++q)
                        if ((uint32_t)(hl + RS * q) < nfree) XS[my_vi[q]] = xc[q] + delta[q];
                    group_sync();
                }
            }
            T sse_t = T(0);
            if (go) {
                if constexpr (QRG) {
                    if (fresh) sse_t = eval_rows(std::true_type{});
                    else sse_t = eval_rows(std::false_type{});
                } else {
                    sse_t = eval_rows(std::true_type{});
                }
                stamp(GH_EVAL);
                if (!fresh) {
                    if (sse_t < sse) {
                        code = LC_ACCEPT;  // lm.rs:151-186
                    } else {               // lm.rs:187-190
                        double lam_k = lambda * o.reject_factor;
                        if constexpr (LADDER) {
                            if (lad_rank > 0)
                                for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                        }
                        if (!(sse_t == sse_t) && !(lam_k < 1.0e300)) {
                            code = LC_REJ_NAN;  // NaN trial point: the reference would double lambda forever
                        } else if (sizeof(T) == 4 && sse_t - sse <= (T)o.ftol * sse) {
                            // f32 only (fx_lm_opts_default_f32): a rejected trial whose SSE is within ftol of the current one
                            // is round-off, not a worse point — the solve has stagnated at what f32 can resolve. Without this
                            // exit such Systems double lambda dozens of times until |delta|^2 < step_tol, and a batch waits
                            // for them.
                            code = LC_REJ_FTOL;
                        }
                    }
                }
            }
            // --- the verdict that decides. Alone: the row's own. On the ladder: the verdicts of the group in rank order,
            // the first that is not a plain reject — what the sequential loop would have met first.
            int kw = (code != LC_REJECT) ? 0 : 1;  // plain rejects in front of the deciding trial (== width: all of them)
            int code_w = code;
            T sse_w = sse_t;
            T delta_w[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) delta_w[q] = delta[q];
            win_row = myrow;
            if constexpr (LADDER) {
                if (__ballot(lad_width > 1) != 0ull) {
                    int ck[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ck[k] = lane_get(code, (int)((lad_members >> (2 * k)) & 3u) * RS + hl);
                    kw = lad_width;
                    code_w = LC_REJECT;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        if (k < lad_width && ck[k] != LC_REJECT) {
                            kw = k;
                            code_w = ck[k];
                        }
                    }
                    const int wrow = (int)((lad_members >> (2 * (kw < lad_width ? kw : 0))) & 3u);
                    const int wl = wrow * RS + hl;
                    sse_w = lane_get(sse_t, wl);
#pragma unroll
                    for (int q = 0; q < NC; ++q) delta_w[q] = lane_get(delta[q], wl);
                    win_row = wrow;
                }
            }
            bool assemble = false, fin = false;
            if (fresh) {  // the component's start point
                sse = sse_t;
                sse_start = sse_t;
                assemble = true;
            } else {
                if (kw > 0) {  // the plain rejects in front (lm.rs:189)
                    lambda *= o.reject_factor;
                    if constexpr (LADDER)
                        for (int k = 1; k < kw; ++k) lambda *= o.reject_factor;
                }
                if (kw == lad_width) {
                    trials += (uint32_t)kw;  // nothing but plain rejects: on with the next lambdas
                } else {
                    trials += (uint32_t)kw + (code_w != LC_CAP ? 1u : 0u);
                    if (code_w == LC_CAP) {
                        exit_code = FX_EXIT_TRIAL_CAP;
                        fin = true;
                    } else if (code_w == LC_SINGULAR) {  // lm.rs:134-137
                        lambda *= o.singular_factor;
                    } else if (code_w == LC_NAN) {
                        exit_code = FX_EXIT_NAN;
                        fin = true;
                    } else if (code_w == LC_STEP) {  // lm.rs:139-142
                        exit_code = FX_EXIT_STEP;
                        fin = true;
                    } else if (code_w == LC_ACCEPT) {  // lm.rs:151-186
                        lambda *= o.accept_factor;
                        if (lambda < o.lambda_min) lambda = o.lambda_min;
#pragma unroll
                        for (int q = 0; q < NC; ++q)
                            if ((uint32_t)(hl + RS * q) < nfree) xc[q] = xc[q] + delta_w[q];
                        accepted += 1;
                        const T rel = (sse - sse_w) / sse;
                        sse = sse_w;  // the returned point's SSE (the reference leaves it stale, quirk Q9)
                        if (rel <= (T)o.ftol) {
                            exit_code = FX_EXIT_FTOL;
                            fin = true;
                        } else {
                            assemble = true;
                            outer += 1;
                        }
                    } else {  // a reject that ends the solve
                        lambda *= o.reject_factor;
                        exit_code = (code_w == LC_REJ_NAN) ? FX_EXIT_NAN : FX_EXIT_FTOL;
                        fin = true;
                    }
                }
            }
            if (assemble) {
                if constexpr (QRG) {
                    if (!fresh) {  // the accepted point once more, with its Jacobian rows (every row of a ladder group for itself)
#pragma unroll
                        for (int q = 0; q < NC; ++q)
                            if ((uint32_t)(hl + RS * q) < nfree) XS[my_vi[q]] = xc[q];
                        group_sync();
                        (void)eval_rows(std::true_type{});
                    }
                } else if constexpr (LADDER) {
                    if (win_row != myrow) {  // the accepted point's Jacobian rows and residuals are another row's
                        using V = typename Vec16<T>::type;
                        const unsigned char* wb = rows0 + (uint32_t)win_row * L.stride;
                        const V* gs = reinterpret_cast<const V*>(wb + L.off_g);
                        V* gd = reinterpret_cast<V*>(G);
                        const uint32_t ng = m_rows * 8u / (uint32_t)Vec16<T>::n;
                        for (uint32_t i = hl; i < ng; i += RS) gd[i] = gs[i];
                        const T* rs = reinterpret_cast<const T*>(wb + L.off_r);
                        for (uint32_t i = hl; i < m_rows; i += RS) R[i] = rs[i];
                        group_sync();
                    }
                }
                if constexpr (!QRG) form_normal();  // (the QR step takes J and r as they are)
                stamp(GH_FORM);
                // top of the next outer iteration (lm.rs:108-112)
                if (fresh && (!(sse == sse) || !(sse < Lim<T>::huge()))) {
                    exit_code = FX_EXIT_NAN;
                    fin = true;
                } else if (outer >= o.max_outer) {
                    fin = true;  // exit_code is still FX_EXIT_MAX_OUTER
                } else if (sse < (T)o.sse_tol) {
                    exit_code = FX_EXIT_SSE;
                    fin = true;
                }
            }
            fresh = false;
            if (fin) {
                phase = GP_FINISH;
                if constexpr (LADDER) {
                    if (lad_rank > 0) {  // a helper goes back to being an idle row; the leader writes the System back
                        phase = GP_EXIT;
                    }
                    lad_rank = 0;
                    lad_width = 1;
                    lad_lead = myrow;
                    lad_members = (uint32_t)myrow * 0x55u;
                }
            }
        }

        // A row that is done waits up to prm.hold_passes passes for a second row of the wavefront to get done, so that the
        // two go through FINISH / CLOSE / NEXT / COMP side by side: those blocks cost the wavefront the same whether one
        // row or four execute them, and a row alone in them stalls the other three (scheduling). SinglePass
        // blocks wait the same way (a block's FINISH / COMP are the hand-over there).
        bool finish_now = phase == GP_FINISH;
        if (prm.hold_passes) {
            const int n_done = __popcll(__ballot(phase == GP_FINISH)) / RS;
            const bool any_running = __ballot(phase == GP_RUN) != 0ull;
            if (phase == GP_FINISH) {
                if (n_done >= 2 || !any_running || held >= prm.hold_passes) {
                    held = 0;
                } else {
                    held += 1;
                    finish_now = false;
                }
            }
        }
        // ================= FINISH: K6 write back scale * x for the free variables (assemble/mod.rs:161-166) ===
        if (finish_now) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if ((uint32_t)(hl + RS * q) < nfree) {
                    const double x = (double)xc[q];
                    const double xo = (prm.mode & 1u) ? scale * x : x;
                    b.vars[v0 + my_vi[q]] = xo;
                    VOUT[my_vi[q]] = xo;
                    // later components are solved against the PRE-solve snapshot (only `system.variables` is
                    // written back, quirk Q2): the working vector goes back to the perturbed start value.
                    // SinglePass blocks update it instead (assemble/mod.rs:201-207).
                    XS[my_vi[q]] = (UNITS && !(unit_flags & UNIT_RESTORE)) ? xc[q] : xstart[q];
                }
            }
            group_sync();
            tot_accept += accepted;
            tot_trials += trials;
            last_exit = exit_code;
            tot_sse0 += (double)sse_start;
            tot_sse += (double)sse;
            if (!UNITS) comps_done += 1;
            c += 1;
            phase = GP_COMP;
            stamp(GH_TAIL);
        }

        // ================= CLOSE: all components done (in the pass of the last FINISH, so that the row takes its next
        // System at the top of the next pass instead of sitting one out) =================
        if (phase == GP_COMP && c >= ncomp) {
                // post-solve check on unscaled variables (constraints/mod.rs:96-109)
                double part[4] = {0.0, 0.0, 0.0, 0.0};
                for_exprs([&](uint32_t i, int tag, double param, uint32_t, ushort4 f4) {
                    if (i < net) {
                        uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
                        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        expand_vars(tag, ff, vars8);
                        double v[8], g[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = VOUT[vars8[e]];
                        const double r = eval_expression<double, false, QRG>(tag, v, param, g);
                        const double r2 = r * r;
                        const uint32_t blk = (i >> 4) & 3u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) part[q] += (blk == (uint32_t)q) ? r2 : 0.0;
                    }
                });
                const double sse_u = block_sum4(part);
                if (hl == 0) {
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
                group_sync();
                phase = GP_NEXT;
                stamp(GH_SETUP);
        }

        // (a row that went idle for a straggler's sake has not seen the end of the queue: it is sent back to it above)
        if (__ballot(phase != GP_EXIT || (LADDER && prm.ladder && !qdone)) == 0ull) break;
    }
    if (PROF) {
        stamp(GH_TAIL);
        if (lane == 0 && prm.prof) {
            for (int i = 0; i < GH_COUNT; ++i) atomicAdd(&prm.prof[i], ph[i]);
        }
    }
}

// The occupancy hint is part of the kernel's signature: the 32-column f64 build is LDS-bound to one wavefront
// per SIMD anyway (4 x 10 KB per wavefront) and may use the registers that frees; the others run two.
template <int NC, typename T, bool PROF, bool UNITS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_solve_grouped_kernel_w1(
    DeviceBatch b, LmParams prm, GroupLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_body<NC, T, PROF, UNITS>(b, prm, L, next_system, smem);
}
// FX_STEP_QR, batches of one structure
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_solve_grouped_qr_kernel(
    DeviceBatch b, LmParams prm, GroupLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_body<2, double, false, false, true>(b, prm, L, next_system, smem);
}
template <int NC, typename T, bool PROF, bool UNITS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void lm_solve_grouped_kernel_w2(
    DeviceBatch b, LmParams prm, GroupLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_body<NC, T, PROF, UNITS>(b, prm, L, next_system, smem);
}

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
template <int NC, typename T, bool PROF = false, bool UNITS = false>
static hipError_t launch_grouped_t(const DeviceBatch& b, const LmParams& p, uint32_t* counter, hipStream_t stream) {
    // SinglePass: rows and products per BLOCK (a row holds at most 8 entries: 36 products of the triangle plus 28
    // repeats, 8 right-hand-side entries)
    const uint32_t rows = UNITS ? b.max_unit_rows : b.max_rows;
    const uint32_t pairs = UNITS ? (b.max_unit_rows * 64u < b.max_pairs_tri ? b.max_unit_rows * 64u : b.max_pairs_tri) : b.max_pairs_tri;
    const uint32_t ents = UNITS ? (b.max_unit_rows * 8u < b.max_ents ? b.max_unit_rows * 8u : b.max_ents) : b.max_ents;
    const GroupLayout L = make_group_layout((uint32_t)(RS * NC), b.max_vars, rows, (uint32_t)sizeof(T), pairs, ents);
    constexpr uint32_t groups = 64u / (uint32_t)RS;
    const uint32_t per_wave = groups * L.stride;
    constexpr bool one_wave = (NC >= 2 && sizeof(T) == 8);
    const void* fn;
    if constexpr (one_wave) fn = reinterpret_cast<const void*>(&lm_solve_grouped_kernel_w1<NC, T, PROF, UNITS>);
    else fn = reinterpret_cast<const void*>(&lm_solve_grouped_kernel_w2<NC, T, PROF, UNITS>);
    if (per_wave > 160u * 1024u) return hipErrorInvalidValue;
    static unsigned int raised = 0;  // (one per instantiation of this template)
    hipError_t e = raise_lds_limit_once(fn, &raised);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // enough wavefronts to fill the chip a few times over; every row keeps taking Systems until none is left
    uint32_t waves = (b.n_systems + groups - 1u) / groups;
    const uint32_t cap = 256u * 16u;
    if (waves > cap) waves = cap;
    // the wavefronts that are resident at once draw the first tickets: with a schedule (longest first) those are dealt
    // one per wavefront — the likely stragglers then sit in different wavefronts, whose other rows can help them
    LmParams pl = p;
    pl.spread = 0u;
    if (p.ladder) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t by_lds = (160u * 1024u) / (per_wave ? per_wave : 1u), by_simd = one_wave ? 4u : 8u;
        uint32_t resident = (uint32_t)cus * (by_lds < by_simd ? by_lds : by_simd);
        if (resident > waves) resident = waves;
        if (b.order && p.spread) pl.spread = resident < b.n_systems / 4u ? resident : b.n_systems / 4u;
        // the default tail: eight Systems per resident row (measured on ring16 shards of 12 500 ... 100 000, tools/ladder_probe.py)
        if (p.ladder_tail == 0xFFFFFFFFu) pl.ladder_tail = 32u * resident;
    }
    if constexpr (one_wave) {
        hipLaunchKernelGGL((lm_solve_grouped_kernel_w1<NC, T, PROF, UNITS>), dim3(waves), dim3(64), per_wave, stream, b, pl, L, counter);
    } else {
        hipLaunchKernelGGL((lm_solve_grouped_kernel_w2<NC, T, PROF, UNITS>), dim3(waves), dim3(64), per_wave, stream, b, pl, L, counter);
    }
    return hipGetLastError();
}

// FX_STEP_QR on a batch of one structure: the program of ensure_qr_plans is there, and four Systems fit a wavefront's LDS
// (as_class: a launch over ONE structure class of a batch of several — the members come through b.order, the program is the
// class's; the batch's maxima size the LDS layout, the class's own structure may be smaller)
static bool grouped_qr_applies(const DeviceBatch& b, const LmParams& p, GroupLayout* out, bool as_class = false) {
    const QrPlans& Q = b.qr_none;
    if (p.lm.solver != FX_STEP_QR || !Q.qrg || p.lm.precision == 32 || p.prof) return false;
    if (as_class ? (b.uniform || !b.order || !b.sys_class) : (!b.uniform || b.u_ncomp != 1u)) return false;
    if ((p.mode & (MODE_UNITS | MODE_LBFGS)) || b.max_free > 32u || b.max_rows > 64u || b.max_vars > 64u || !b.work_counter) return false;
    if (as_class ? (Q.qrg_n > b.max_free || Q.qrg_m > b.max_rows) : (Q.qrg_n != b.max_free || Q.qrg_m != b.max_rows)) return false;
    GroupLayout L = make_group_layout(32u, b.max_vars, b.max_rows, 8u, 0u, 0u, Q.qrg_small, Q.qrg_nx, Q.qrg_ng);
    if ((size_t)L.tab_bytes + 4u * (size_t)L.stride > 160u * 1024u / 2u) return false;  // two wavefronts per CU at least
    {  // the whole program in LDS when that costs no wavefront (FIKSI_AMD_QR_TABLES=lds|global forces either: measurements)
        const GroupLayout Lw = make_group_layout(32u, b.max_vars, b.max_rows, 8u, 0u, 0u, Q.qrg_words, Q.qrg_nx, Q.qrg_ng);
        const size_t small_bytes = (size_t)L.tab_bytes + 4u * (size_t)L.stride, whole_bytes = (size_t)Lw.tab_bytes + 4u * (size_t)Lw.stride;
        static const char* force = getenv("FIKSI_AMD_QR_TABLES");
        bool whole = whole_bytes <= 160u * 1024u && (160u * 1024u) / whole_bytes >= std::min<size_t>(4, (160u * 1024u) / small_bytes);
        if (force && force[0] == 'l' && whole_bytes <= 80u * 1024u) whole = true;
        if (force && force[0] == 'g') whole = false;
        if (whole) {
            L = Lw;
            L.ent_in_lds = 1u;
        }
    }
    if (out) *out = L;
    return true;
}

static hipError_t launch_grouped_qr(const DeviceBatch& b, const LmParams& p, hipStream_t stream, bool as_class = false) {
    GroupLayout L;
    if (!grouped_qr_applies(b, p, &L, as_class)) return hipErrorInvalidValue;
    const uint32_t per_wave = L.tab_bytes + 4u * L.stride;
    static const bool trace = getenv("FIKSI_AMD_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[fiksi_amd] grouped QR: %u B of LDS per wavefront (tables %u, 4 x %u per System: matrix %u doubles), program %u words\n",
                per_wave, L.tab_bytes, L.stride, b.qr_none.qrg_nx, b.qr_none.qrg_words);
    static unsigned int raised = 0;
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&lm_solve_grouped_qr_kernel), &raised);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(b.work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    uint32_t waves = (b.n_systems + 3u) / 4u;
    if (waves > 256u * 16u) waves = 256u * 16u;
    LmParams pl = p;
    pl.spread = 0u;
    if (p.ladder) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t by_lds = (160u * 1024u) / per_wave;
        uint32_t resident = (uint32_t)cus * (by_lds < 4u ? by_lds : 4u);
        if (resident > waves) resident = waves;
        if (b.order && p.spread) pl.spread = resident < b.n_systems / 4u ? resident : b.n_systems / 4u;
        if (p.ladder_tail == 0xFFFFFFFFu) pl.ladder_tail = 32u * resident;
    }
    hipLaunchKernelGGL(lm_solve_grouped_qr_kernel, dim3(waves), dim3(64), per_wave, stream, b, pl, L, b.work_counter);
    return hipGetLastError();
}

// FX_STEP_QR on one structure class of a batch of several structures: b.qr_none.qrg* the class's program, b.order / b.n_systems its
// member list, b.work_counter a queue head of its own (fx_solve.cpp: launch_class_qr)
bool grouped_qr_class_applies(const DeviceBatch& b, const LmParams& p) { return grouped_qr_applies(b, p, nullptr, true); }
hipError_t launch_grouped_qr_class(const DeviceBatch& b, const LmParams& p, hipStream_t stream) { return launch_grouped_qr(b, p, stream, true); }

static uint32_t grouped_columns(const DeviceBatch& b, bool units) {
    const uint32_t f = units ? b.max_unit_free : b.max_free;
    return f <= 16u ? 1u : f <= 32u ? 2u : 3u;
}

// LDS bytes per wavefront of the grouped kernel for this batch, 0 when the batch does not qualify
size_t grouped_lds_bytes(const DeviceBatch& b, uint32_t es, bool units) {
    const uint32_t free_ = units ? b.max_unit_free : b.max_free, rows = units ? b.max_unit_rows : b.max_rows;
    if (free_ == 0 || free_ > 48u || rows > 256u || b.max_vars > 4096u) return 0;
    if (free_ > 32u && rows > 128u) return 0;  // 7 row bits in a packed right-hand-side entry of the 48-column build
    const uint32_t pairs = units ? (rows * 64u < b.max_pairs_tri ? rows * 64u : b.max_pairs_tri) : b.max_pairs_tri;
    const uint32_t ents = units ? (rows * 8u < b.max_ents ? rows * 8u : b.max_ents) : b.max_ents;
    const GroupLayout L = make_group_layout(RS * grouped_columns(b, units), b.max_vars, rows, es, pairs, ents);
    return (size_t)(64u / (uint32_t)RS) * L.stride;
}

bool grouped_applies(const DeviceBatch& b, const LmParams& p) {
    // The context's routing option (fx_ctx_set_routing; its default comes from FIKSI_AMD_GROUPED when the context is
    // created): 0 keeps every batch on the one-System-per-wavefront kernel, 1 sends every batch that qualifies here
    // (tests, A/B measurements). By default a batch must be big enough to fill the chip four Systems per wavefront:
    // below that one wavefront per System finishes sooner.
    if (p.route_grouped == 0 || b.has_pose) return false;  // (cluster problems: the pose builds of the one-wavefront kernel)
    if (p.route_grouped < 0 && b.n_systems < p.grouped_min_systems) {
        // (the tiny one-structure build runs eight Systems per wavefront without a queue: it pays from the first System on)
        return b.uniform && p.lm.solver == FX_STEP_CHOLESKY && !(p.mode & MODE_LBFGS) && grouped_c_applies(b, p) && grouped_tiny_applies(b, p);
    }
    if (p.lm.solver == FX_STEP_QR) return grouped_qr_applies(b, p, nullptr);
    if ((p.mode & MODE_LBFGS) || p.lm.solver != FX_STEP_CHOLESKY) return false;
    if (b.uniform && grouped_c_applies(b, p)) return true;  // (the one-structure build needs a fraction of the general build's LDS)
    const bool units = (p.mode & MODE_UNITS) != 0;
    if (units && (!b.sys_unit_off || p.prof)) return false;
    if (p.prof && p.lm.precision == 32) return false;
    if (grouped_columns(b, units) == 3u && p.prof) return false;
    if (!b.work_counter) return false;
    const size_t lds = grouped_lds_bytes(b, p.lm.precision == 32 ? 4u : 8u, units);
    // four wavefronts (16 Systems) per CU or more; two (8 Systems) for the 48-column build, whose Systems would
    // otherwise sit five to a CU, one per wavefront
    return lds != 0 && lds <= (160u * 1024u) / (grouped_columns(b, units) == 3u ? 2u : 4u);
}

hipError_t launch_solve_grouped(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    if (p.lm.solver == FX_STEP_QR) return launch_grouped_qr(b, p, stream);
    if (b.uniform && grouped_c_applies(b, p)) return launch_solve_grouped_c(b, p, stream);  // one structure: two wavefronts per SIMD
    return launch_solve_grouped_general(b, p, stream);
}

hipError_t launch_solve_grouped_general(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    const bool units = (p.mode & MODE_UNITS) != 0;
    const uint32_t nc = grouped_columns(b, units);
    const bool f32 = p.lm.precision == 32;
    if (units) {
        if (f32)
            return nc == 1u   ? launch_grouped_t<1, float, false, true>(b, p, b.work_counter, stream)
                   : nc == 2u ? launch_grouped_t<2, float, false, true>(b, p, b.work_counter, stream)
                              : launch_grouped_t<3, float, false, true>(b, p, b.work_counter, stream);
        return nc == 1u   ? launch_grouped_t<1, double, false, true>(b, p, b.work_counter, stream)
               : nc == 2u ? launch_grouped_t<2, double, false, true>(b, p, b.work_counter, stream)
                          : launch_grouped_t<3, double, false, true>(b, p, b.work_counter, stream);
    }
    const bool two = nc == 2u;
    if (nc == 3u) return f32 ? launch_grouped_t<3, float>(b, p, b.work_counter, stream) : launch_grouped_t<3, double>(b, p, b.work_counter, stream);
    if (p.prof) return two ? launch_grouped_t<2, double, true>(b, p, b.work_counter, stream)
                           : launch_grouped_t<1, double, true>(b, p, b.work_counter, stream);
    if (f32)
        return two ? launch_grouped_t<2, float>(b, p, b.work_counter, stream) : launch_grouped_t<1, float>(b, p, b.work_counter, stream);
    return two ? launch_grouped_t<2, double>(b, p, b.work_counter, stream) : launch_grouped_t<1, double>(b, p, b.work_counter, stream);
}

}  // namespace fx
