// Grouped fused solve, the build for batches of ONE structure (gfx950, wave64): four Systems per wavefront as in
// fx_grouped.hip — same algorithm, same arithmetic on the same operands, same order of every sum and of every addition
// into the normal equations, so the same bits (reference: fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193) —
// cut down to what TWO wavefronts per SIMD leave: 256 registers and 20 KB of LDS per wavefront (fx_grouped.hip's 32-column
// build takes 392 and 40 KB and runs one wavefront per SIMD, with the VALU busy half of the time).
//
// What makes it fit is that nothing about the structure is per System any more. The host writes one PROGRAM for the batch
// (fx_programs.cpp: build_gc_program; every System has one component, at most 32 variables and 32 expressions — two matrix columns per
// lane; 16 / 16: one column, four wavefronts per SIMD; 48 / 48, the reference's own bench sketch: three columns, a wavefront on
// every SIMD; an over-constrained structure — up to twice the shape's rows — the same bodies with twice the row chunks), the
// wavefront copies it into LDS once, and the four Systems share it:
//   * the row lists (variables and kind of every expression), the free-variable map, the product lists of Jt J and Jt r —
//     4 KB once per wavefront instead of 2.4 KB per System (and no list building when a row takes a System);
//   * Jt J by its PATTERN: a slot per structural non-zero of the lower triangle (ring16: 144 of 528), addressed through the
//     program — the product lists name slots, and a lane's 64 matrix elements are loaded through a table of slot numbers
//     (64 bytes per lane; what is not in the pattern reads the zero slot). The factor's fill exists in registers only;
//   * Jacobian rows compact (an expression's own entries instead of eight).
// ring16: 3.5 KB per System, 18.2 KB per wavefront, eight wavefronts per CU. The f32 instantiation (cfg5) assembles by gather
// instead of LDS float atomics (ds_add_f32 runs at a fourteenth of ds_add_f64's rate on gfx950).
// A batch of SEVERAL structures brings a program per big structure class; one launch works through all of them (a wavefront
// loads the next class's program when its own class's queue is empty).
// The per-row state machine, the device-side queue, the lambda ladder, the hold passes are fx_grouped.hip's.
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

namespace fx {

// A System's block, bytes: everything of fixed size first, at offsets the instructions carry as immediates (one base register per
// row), then Jt J's slots and behind them the compact Jacobian rows
// (NV = 16 NC: the most variables of a System in the build with NC columns per lane; NR = 16 RC: the most expressions — an
// over-constrained structure takes the instantiation with twice the rows; ES: bytes of the compute type)
template <int NV, int NR, int ES> struct GcBlock {
    static constexpr uint32_t XS = 0, RHS = ES * NV, R = 2 * ES * NV, P = R + ES * NR, VOUT = P + ES * NR, STASH = VOUT + 8 * NV, A = STASH + 16;
};
struct GcLayout {
    uint32_t tab_bytes, off_g, stride;
};

static GcLayout make_gc_layout(const DeviceBatch& b, uint32_t es) {
    GcLayout L;
    L.tab_bytes = ((es == 4u ? b.gc_words_all : b.gc_words) * 4u + 15u) & ~15u;
    L.off_g = (2u * es + 8u) * 16u * b.gc_nc + 2u * es * 16u * b.gc_rc + 16u + b.gc_nslots * es;  // (slots are a multiple of four: 16-byte aligned)
    L.stride = L.off_g + b.gc_ng * es;
    return L;
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int NC, int RC, typename T>
__device__ __forceinline__ void grouped_c_body(const DeviceBatch& b, const LmParams& prm, const GcLayout& L, uint32_t* __restrict__ next_system,
                                               unsigned char* smem) {
    constexpr int N = RS * NC;
    using BK = GcBlock<N, RS * RC, (int)sizeof(T)>;
    using V16 = typename Vec16<T>::type;
    using TK = GcTable<NC, RC>;
    const int lane = threadIdx.x;
    const int hl = lane & (RS - 1);
    const int gbase = lane & ~(RS - 1);
    const int myrow = lane / RS;
    // The program and the queue at hand. A batch of one structure has one of each; a batch of several structures brings a
    // program, a member list and a queue head per structure CLASS (b.gc_classes; fx_solve.cpp: launch_class_solves): a wavefront
    // starts on the class its place in the grid falls into — the grid is dealt in proportion to the classes' sizes — and, when
    // that queue is empty and its Systems are done, loads the next class's program and goes on there: one launch, every
    // wavefront busy until every queue is empty.
    const uint32_t* TB = reinterpret_cast<const uint32_t*>(smem);
    uint32_t nvt = 0, net = 0, nfree = 0, n_pw = 0, n_pe = 0, nslots = 0;
    uint32_t qn = 0;                  // Systems in the queue
    const uint32_t* qlist = nullptr;  // ... their numbers (null: the ticket is the number)
    uint32_t* qhead = next_system;    // ... its head
    // (the tables of fixed size sit at fixed places: fx_device.h, GcTable)
    const int8_t* vcol = reinterpret_cast<const int8_t*>(smem + TK::VCOL);         // [N] variable -> free column or -1
    const uint8_t* fidx = smem + TK::FIDX;                                         // [N] free column -> variable
    const uint8_t* rtag = smem + TK::RTAG;                                         // [N] kind of expression i
    const uint16_t* gbaseT = reinterpret_cast<const uint16_t*>(smem + TK::GBASE);  // [N] first compact Jacobian entry of row i
    const uint2* gvar = reinterpret_cast<const uint2*>(smem + TK::GVAR);           // [N] eight variable numbers, a byte each
    const uint4* LT = reinterpret_cast<const uint4*>(smem + TK::LT + (uint32_t)hl * (uint32_t)(NC * N));  // this lane's NC x N slot numbers
    const uint32_t* PE = reinterpret_cast<const uint32_t*>(smem + TK::PE);         // right-hand side: entry | row << 8 | column << 16
    const uint32_t* PW = PE;                                                       // products: entry a | entry b << 8 | slot << 16 (behind PE)

    unsigned char* const rows0 = smem + L.tab_bytes;
    unsigned char* base = rows0 + (uint32_t)myrow * L.stride;
    T* XS = reinterpret_cast<T*>(base + BK::XS);          // [N] working variables: trial point on the free ones
    T* At = reinterpret_cast<T*>(base + BK::A);           // Jt J by slots (+ lambda on the diagonal per trial)
    T* rhsv = reinterpret_cast<T*>(base + BK::RHS);       // [N] -Jt r
    T* G = At;                                            // compact Jacobian rows of the last evaluated point (behind the slots)
    T* R = reinterpret_cast<T*>(base + BK::R);            // [N]
    T* P = reinterpret_cast<T*>(base + BK::P);            // [N] scaled parameters
    double* VOUT = reinterpret_cast<double*>(base + BK::VOUT);    // [N] unscaled values as written back
    double* STASH = reinterpret_cast<double*>(base + BK::STASH);  // [2] the System's scale, the SSE of its start point

    const fx_lm_opts o = prm.lm;
    auto gballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> gbase) & 0xFFFFu; };

    // per lane, fixed for a program: the variables of its columns and the slots of their diagonal entries
    uint32_t my_vi[NC], dslot[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) my_vi[q] = dslot[q] = 0u;

    // per-row state (identical in every lane of the row unless noted)
    int phase = GP_NEXT;
    uint32_t s = 0;
    T xc[NC], diag[NC], rhs_l[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        xc[q] = rhs_l[q] = T(0);
        diag[q] = T(1);
    }
    T sse = T(0);
    double lambda = 0.0;
    uint32_t accepted = 0, trials = 0, outer = 0, exit_code = FX_EXIT_MAX_OUTER;
    bool fresh = false;
    uint32_t held = 0;
    // the lambda ladder (fx_grouped.hip)
    int lad_rank = 0, lad_width = 1, lad_lead = myrow;
    uint32_t lad_members = (uint32_t)myrow * 0x55u;
    int win_row = myrow;
    bool qdone = false;
    uint32_t last_tk = 0;

    auto row_vars = [&](uint32_t row, const T* from, T (&v)[8]) {
        const uint2 gv = gvar[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = from[(gv.x >> (8 * e)) & 0xFFu];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 + e] = from[(gv.y >> (8 * e)) & 0xFFu];
    };
    // sum over a vector laid out 16 entries per accumulator, as wave_sum adds its blocks: (b0 + b1) + (b2 + b3) with the blocks
    // past the end left out (+ 0.0 of a sum of squares: exact)
    auto chunk_sum = [&](const auto (&part)[NC]) {
        auto s01 = row_sum(part[0]);
        if constexpr (NC >= 2) s01 = s01 + row_sum(part[1]);
        if constexpr (NC >= 3) s01 = s01 + row_sum(part[2]);
        return s01;
    };
    // ... over the expressions' chunks: (b0 + b1) + (b2 + b3)
    auto rows_sum = [&](const auto (&part)[RC]) {
        auto s01 = row_sum(part[0]);
        if constexpr (RC >= 2) s01 = s01 + row_sum(part[1]);
        if constexpr (RC == 3) s01 = s01 + row_sum(part[2]);
        if constexpr (RC == 4) s01 = s01 + (row_sum(part[2]) + row_sum(part[3]));
        return s01;
    };
    // residuals and Jacobian rows of the point in XS
    auto eval_rows = [&]() -> T {
        T part[RC];
#pragma unroll
        for (int k = 0; k < RC; ++k) part[k] = T(0);
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const uint32_t row = (uint32_t)(hl + RS * k);
            if (row < net) {
                T v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                row_vars(row, XS, v);
                const int tag = (int)rtag[row];
                const T r = eval_expression<T, true, false>(tag, v, P[row], g);
                R[row] = r;
                const uint32_t gb = gbaseT[row];
                const int kk = tag_nvars(tag);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < kk) G[gb + (uint32_t)e] = g[e];
                part[k] = r * r;
            }
        }
        group_sync();
        return rows_sum(part);
    };
    // K3: Jt J into its slots and -Jt r from the program's lists (ds_add_f64; entry t is lane t % 16's, 16 consecutive entries
    // per instruction, in list order — fx_grouped.hip's order)
    auto form_normal = [&]() {
        if constexpr (sizeof(T) == 4) {
            // f32: every slot's and every column's sum by a gather, in list order — the order the atomics below arrive in — instead
            // of ds_add_f32, which gfx950 executes at a fourteenth of ds_add_f64's rate (tools/probes/lds_atomic_f32_probe.hip)
            const uint16_t* sptr = reinterpret_cast<const uint16_t*>(smem + rfl(TB[9]));
            const uint16_t* SPW = reinterpret_cast<const uint16_t*>(smem + rfl(TB[10]));
            const uint16_t* cptr = reinterpret_cast<const uint16_t*>(smem + rfl(TB[11]));
            const uint16_t* CPE = reinterpret_cast<const uint16_t*>(smem + rfl(TB[12]));
            for (uint32_t sl = hl; sl < nslots; sl += RS) {
                const uint32_t t0 = sptr[sl], t1 = sptr[sl + 1];
                T acc = T(0);
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint32_t w = SPW[t];
                    acc += G[w & 0xFFu] * G[w >> 8];
                }
                At[sl] = acc;
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const uint32_t j = (uint32_t)(hl + RS * q);
                const uint32_t t0 = cptr[j], t1 = cptr[j + 1];
                T acc = T(0);
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint32_t w = CPE[t];
                    acc += G[w & 0xFFu] * -R[w >> 8];
                }
                rhsv[j] = acc;
            }
            group_sync();
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if ((uint32_t)(hl + RS * q) >= nfree) At[dslot[q]] = T(1);  // identity padding
            group_sync();
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                diag[q] = At[dslot[q]];
                rhs_l[q] = rhsv[hl + RS * q];
            }
            return;
        }
        {
            V16 z;
            for (int q = 0; q < Vec16<T>::n; ++q) reinterpret_cast<T*>(&z)[q] = T(0);
            for (uint32_t i = hl; i < nslots / (uint32_t)Vec16<T>::n; i += RS) reinterpret_cast<V16*>(At)[i] = z;
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) rhsv[hl + RS * q] = T(0);
        group_sync();
        constexpr int U = 4;
        for (uint32_t t0 = 0; t0 < n_pw; t0 += RS * U) {
            uint32_t w[U];
            T g1[U], g2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = PW[t0 + (uint32_t)(u * RS + hl)];  // (padded to a multiple of 64 with 0xFFFFFFFF)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                g1[u] = G[ww & 0xFFu];
                g2[u] = G[(ww >> 8) & 0xFFu];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (w[u] != 0xFFFFFFFFu) lds_add(&At[w[u] >> 16], g1[u] * g2[u]);
        }
        for (uint32_t t0 = 0; t0 < n_pe; t0 += RS * U) {
            uint32_t w[U];
            T g1[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = PE[t0 + (uint32_t)(u * RS + hl)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                g1[u] = G[ww & 0xFFu];
                rr[u] = -R[(ww >> 8) & 0xFFu];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (w[u] != 0xFFFFFFFFu) lds_add(&rhsv[w[u] >> 16], g1[u] * rr[u]);
        }
        group_sync();
#pragma unroll
        for (int q = 0; q < NC; ++q)
            if ((uint32_t)(hl + RS * q) >= nfree) At[dslot[q]] = T(1);  // identity padding
        group_sync();
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            diag[q] = At[dslot[q]];
            rhs_l[q] = rhsv[hl + RS * q];
        }
    };

    // a launch over what the tiny build handed over (DeviceBatch::queue_len): mostly nothing, or a few dozen stragglers — a wavefront
    // whose first four tickets would lie past the end of that queue leaves before it has copied the program
    if (b.queue_len && !b.gc_nclasses && blockIdx.x * 4u >= rfl(*b.queue_len)) return;
    const uint32_t ncls = b.gc_nclasses ? b.gc_nclasses : 1u;
    uint32_t home = 0;
    if (b.gc_nclasses > 1u) {  // the class this wavefront's place in the grid falls into
        const unsigned long long at = (unsigned long long)blockIdx.x * b.n_systems;  // (b.n_systems: the classes' Systems in all)
        unsigned long long acc = 0;
        for (uint32_t c = 0; c < b.gc_nclasses; ++c) {
            acc += b.gc_classes[c].count;
            if (at < acc * gridDim.x) break;
            home = c + 1u < b.gc_nclasses ? c + 1u : c;
        }
    }
    for (uint32_t ci = 0; ci < ncls; ++ci) {
    {
        const uint32_t c = home + ci < ncls ? home + ci : home + ci - ncls;
        const uint32_t* prog = b.gc_tab;
        uint32_t words = sizeof(T) == 4 ? b.gc_words_all : b.gc_words;
        qn = b.n_systems;
        qlist = b.order;
        qhead = next_system;
        if (b.queue_len && !b.gc_nclasses) qn = rfl(*b.queue_len);  // (what the tiny build handed over)
        if (b.gc_nclasses) {
            const GcClass k = b.gc_classes[c];
            prog = b.gc_tab + k.prog_off;
            words = sizeof(T) == 4 ? k.words_all : k.words;
            qn = k.count;
            qlist = b.order + k.list_off;
            qhead = next_system + c;
        }
        group_sync();
        const uint4* src = reinterpret_cast<const uint4*>(prog);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = lane; i < words / 4u; i += 64) dst[i] = src[i];
        group_sync();
        nvt = rfl(TB[1]);
        net = rfl(TB[2]);
        nfree = rfl(TB[3]);
        n_pw = rfl(TB[4]);
        n_pe = rfl(TB[5]);
        nslots = rfl(TB[6]);
        PW = PE + n_pe;
        G = At + nslots;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const uint32_t j = (uint32_t)(hl + RS * q);
            my_vi[q] = j < nfree ? (uint32_t)fidx[j] : 0u;
            dslot[q] = (uint32_t)reinterpret_cast<const uint8_t*>(LT)[(uint32_t)(N * q) + j];
        }
        phase = GP_NEXT;
        fresh = false;
        held = 0;
        lad_rank = 0;
        lad_width = 1;
        lad_lead = myrow;
        lad_members = (uint32_t)myrow * 0x55u;
        qdone = false;
        last_tk = 0;
    }
    for (;;) {
        // near the end of the queue a wavefront that holds a straggler stops taking Systems (fx_grouped.hip)
        bool park = false;
        if (prm.ladder && prm.ladder_tail != 0u) {
            const bool straggler = __ballot(phase == GP_RUN && lad_rank == 0 && !fresh && trials >= prm.ladder_k) != 0ull;
            if (phase == GP_EXIT && !qdone && !straggler) phase = GP_NEXT;
            park = straggler && last_tk < qn && qn - last_tk <= prm.ladder_tail;
        }
        // ================= NEXT: take a System, scale and perturb it (assemble/mod.rs:32-44, 91-111) =================
        if (phase == GP_NEXT && park) phase = GP_EXIT;
        if (phase == GP_NEXT) {
            uint32_t tk = 0;
            if (hl == 0) {
                tk = atomicAdd(qhead, 1u);
                last_tk = tk;
                if (tk >= qn) {
                    tk = 0xFFFFFFFFu;  // the queue is empty (a list — a schedule, or the members of a structure class — may hold
                                       // System numbers beyond the queue's length)
                } else if (qlist) {
                    uint32_t pos = tk;
                    if (tk < 4u * prm.spread) pos = (tk & 3u) * prm.spread + (tk >> 2);
                    tk = qlist[pos];
                }
            }
            const uint32_t nxt = (uint32_t)__shfl((int)tk, 0, RS);
            last_tk = (uint32_t)__shfl((int)last_tk, 0, RS);
            if (nxt == 0xFFFFFFFFu) {
                phase = GP_EXIT;
                qdone = true;
            } else {
                s = nxt;
                // (a launch over ONE structure class of a batch of several — b.uniform == 0 — reads the System's offsets)
                const uint32_t v0 = b.uniform ? s * nvt : b.var_off[s], e0 = b.uniform ? s * net : b.expr_off[s];
                double c_var[NC], c_param[RC];
                int tagk[RC], colk[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const uint32_t i = (uint32_t)(RS * k + hl);
                    c_var[k] = i < nvt ? (b.vars_in ? b.vars_in : b.vars0)[v0 + i] : 0.0;
                    colk[k] = i < nvt ? (int)vcol[i] : -1;
                }
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    const uint32_t i = (uint32_t)(RS * k + hl);
                    c_param[k] = i < net ? (b.param_in ? b.param_in : b.expr_param)[e0 + i] : 0.0;
                    tagk[k] = i < net ? (int)rtag[i] : 0;
                }
                if (b.param_in) {  // (the closing check reads them again: from the device's copy, not over the link)
#pragma unroll
                    for (int k = 0; k < RC; ++k)
                        if ((uint32_t)(RS * k + hl) < net) b.expr_param[e0 + (uint32_t)(RS * k + hl)] = c_param[k];
                }
                if (b.vars_in) {  // (start values stay on the device: a refused hint puts them back, fx_solve.cpp)
#pragma unroll
                    for (int k = 0; k < NC; ++k)
                        if ((uint32_t)(RS * k + hl) < nvt) b.vars0[v0 + (uint32_t)(RS * k + hl)] = c_var[k];
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                // K0a: system scale, summed strictly in reference order (utils.rs:11-33)
                double scale = 1.0, scale_recip = 1.0;
                if (prm.mode & 1u) {
                    double sum = 0.0;
                    uint32_t count = nvt;
#pragma unroll
                    for (int k = 0; k < NC; ++k)
                        if ((uint32_t)(RS * k) < nvt) seq_add(sum, c_var[k] * c_var[k]);  // (past the end: + 0.0, exact)
#pragma unroll
                    for (int k = 0; k < RC; ++k) {
                        if ((uint32_t)(RS * k) < net) {
                            const bool isd = (uint32_t)(RS * k + hl) < net && (tagk[k] == FX_TAG_PPD || tagk[k] == FX_TAG_PLD);
                            count += (uint32_t)__popc(gballot(isd));
                            seq_add(sum, isd ? c_param[k] * c_param[k] : 0.0);
                        }
                    }
                    scale = ::sqrt(sum / (double)count);
                    scale_recip = 1.0 / scale;
                }
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const uint32_t i = (uint32_t)(RS * k + hl);
                    if (i < nvt) {
                        double x = (prm.mode & 1u) ? c_var[k] * scale_recip : c_var[k];
                        if (colk[k] >= 0 && (prm.mode & 2u)) {  // K0b: two draws of the LCG per free variable, in column order
                            uint32_t st = lcg_jump(42u, 2u * (uint32_t)colk[k]);
                            st = st * 1664525u + 1013904223u;
                            const double f1 = (1.0 / 4294967295.0) * (double)st;
                            st = st * 1664525u + 1013904223u;
                            const double f2 = (1.0 / 4294967295.0) * (double)st;
                            x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                        }
                        XS[i] = (T)x;  // (perturbed from the f64 input: the f64 start point is bit-identical to the reference)
                        VOUT[i] = c_var[k];
                        b.vars[v0 + i] = c_var[k];  // fixed variables stay bit-identical
                    }
                }
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    const uint32_t i = (uint32_t)(RS * k + hl);
                    if (i < net) {
                        double prm_e = c_param[k];
                        if ((prm.mode & 1u) && (tagk[k] == FX_TAG_PPD || tagk[k] == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                        P[i] = (T)prm_e;
                    }
                }
                if (hl == 0) STASH[0] = scale;
                group_sync();
#pragma unroll
                for (int q = 0; q < NC; ++q) xc[q] = ((uint32_t)(hl + RS * q) < nfree) ? XS[my_vi[q]] : T(0);
                lambda = o.lambda0;
                accepted = 0;
                trials = 0;
                outer = 0;
                exit_code = FX_EXIT_MAX_OUTER;
                fresh = true;
                phase = GP_RUN;
            }
        }

        // ================= LADDER: idle rows join a running row of their wavefront (fx_grouped.hip) =================
        if (prm.ladder) {
            const unsigned long long bcand = __ballot(phase == GP_RUN && !fresh && lad_rank == 0);
            const unsigned long long bidle = __ballot(phase == GP_EXIT);
            if (bcand != 0ull && bidle != 0ull) {
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
                    const int grp = joining ? (int)nl : lad_lead;
                    const int srcl = grp * RS + hl;
                    auto cp = [&](auto& v) {
                        const auto t = lane_get(v, srcl);
                        if (joining) v = t;
                    };
                    cp(trials); cp(accepted); cp(outer); cp(exit_code);
                    cp(sse); cp(lambda);
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        cp(xc[q]); cp(diag[q]); cp(rhs_l[q]);
                    }
                    lad_width = (int)((wid >> (4 * grp)) & 15u);
                    lad_members = (mem >> (8 * grp)) & 255u;
                    if (joining) {
                        lad_lead = (int)nl;
                        lad_rank = (int)((newrank >> (4 * myrow)) & 15u);
                        const uint4* lb = reinterpret_cast<const uint4*>(rows0 + (uint32_t)nl * L.stride);
                        uint4* mine = reinterpret_cast<uint4*>(base);
                        for (uint32_t i = hl; i < L.stride / 16u; i += RS) mine[i] = lb[i];
                        fresh = false;
                        phase = GP_RUN;
                    }
                    group_sync();
                }
            }
        }

        // ================= RUN: one lambda trial (lm.rs:115-191) =================
        if (phase == GP_RUN) {
            int code = LC_FRESH;
            bool go = true;
            T delta[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) delta[q] = T(0);
            if (!fresh) {
                code = LC_REJECT;
                double lam_k = lambda;
                if (lad_rank > 0)
                    for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                if (trials + (uint32_t)lad_rank >= o.max_trials) {
                    code = LC_CAP;
                    go = false;
                }
                if (go) {
                    // K4: factor (Jt J + lambda I) and solve for delta; columns hl and hl + 16 of the symmetric matrix through
                    // the lane's table of slots
#pragma unroll
                    for (int q = 0; q < NC; ++q) At[dslot[q]] = diag[q] + (T)lam_k;
                    group_sync();
                    T a[NC][N];
#pragma unroll
                    for (int cch = 0; cch < NC * N / 16; ++cch) {
                        const uint4 w4 = LT[cch];
                        const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int el = 16 * cch + e;
                            a[el / N][el % N] = At[(ws[e / 4] >> (8 * (e % 4))) & 0xFFu];
                        }
                    }
                    T invd[NC];
#pragma unroll
                    for (int q = 0; q < NC; ++q) invd[q] = T(1);
                    bool bad = false;
                    RBlock<NC, T, 0, false>::factor(a, invd, bad, hl, N);
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
                        RBlock<NC, T, 0, false>::forward(a, invd, acc, hl, N);
                        RBlock<NC, T, N / 8 - 1, false>::backward(a, invd2, acc, hl, N);
#pragma unroll
                        for (int q = 0; q < NC; ++q) delta[q] = ((uint32_t)(hl + RS * q) < nfree) ? acc[q] * invd2[q] : T(0);
                    }
                }
                if (go) {
                    T dsq[NC];
#pragma unroll
                    for (int q = 0; q < NC; ++q) dsq[q] = delta[q] * delta[q];
                    const T dn2 = chunk_sum(dsq);
                    if (!(dn2 == dn2)) {
                        code = LC_NAN;
                        go = false;
                    } else if (dn2 < (T)o.step_tol) {  // lm.rs:139-142
                        code = LC_STEP;
                        go = false;
                    }
                }
                if (go) {
#pragma unroll
                    for (int q = 0; q < NC; ++q)
                        if ((uint32_t)(hl + RS * q) < nfree) XS[my_vi[q]] = xc[q] + delta[q];
                    group_sync();
                }
            }
            T sse_t = T(0);
            if (go) {
                sse_t = eval_rows();
                if (!fresh) {
                    if (sse_t < sse) {
                        code = LC_ACCEPT;  // lm.rs:151-186
                    } else {               // lm.rs:187-190
                        double lam_k = lambda * o.reject_factor;
                        if (lad_rank > 0)
                            for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                        if (!(sse_t == sse_t) && !(lam_k < 1.0e300)) code = LC_REJ_NAN;  // the reference would double lambda forever
                        else if (sizeof(T) == 4 && sse_t - sse <= (T)o.ftol * sse) code = LC_REJ_FTOL;  // f32: stagnated at round-off (fx_grouped.hip)
                    }
                }
            }
            // --- the verdicts of a ladder group in rank order: the first that is not a plain reject decides
            int kw = (code != LC_REJECT) ? 0 : 1;
            int code_w = code;
            T sse_w = sse_t;
            T delta_w[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) delta_w[q] = delta[q];
            win_row = myrow;
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
            bool assemble = false, fin = false;
            if (fresh) {  // the start point
                sse = sse_t;
                if (hl == 0) STASH[1] = (double)sse_t;
                assemble = true;
            } else {
                if (kw > 0) {  // the plain rejects in front (lm.rs:189)
                    lambda *= o.reject_factor;
                    for (int k = 1; k < kw; ++k) lambda *= o.reject_factor;
                }
                if (kw == lad_width) {
                    trials += (uint32_t)kw;
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
                        sse = sse_w;
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
                if (win_row != myrow) {  // the accepted point's Jacobian rows and residuals are another row's
                    const unsigned char* wb = rows0 + (uint32_t)win_row * L.stride;
                    const uint32_t off_g = BK::A + nslots * (uint32_t)sizeof(T);
                    const uint4* gs = reinterpret_cast<const uint4*>(wb + off_g);
                    uint4* gd = reinterpret_cast<uint4*>(G);
                    const uint32_t ng2 = (L.stride - off_g) / 16u;
                    for (uint32_t i = hl; i < ng2; i += RS) gd[i] = gs[i];
                    const T* rs = reinterpret_cast<const T*>(wb + BK::R);
#pragma unroll
                    for (int k = 0; k < RC; ++k) R[hl + RS * k] = rs[hl + RS * k];
                    group_sync();
                }
                form_normal();
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
                if (lad_rank > 0) phase = GP_EXIT;  // a helper goes back to being an idle row; the leader writes the System back
                lad_rank = 0;
                lad_width = 1;
                lad_lead = myrow;
                lad_members = (uint32_t)myrow * 0x55u;
            }
        }

        // a row that is done waits up to prm.hold_passes passes for company (fx_grouped.hip)
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
        // ================= FINISH: write back scale * x (assemble/mod.rs:161-166), the closing check
        // (constraints/mod.rs:96-109), the result record =================
        if (finish_now) {
            const uint32_t v0 = b.uniform ? s * nvt : b.var_off[s], e0 = b.uniform ? s * net : b.expr_off[s];
            double c_param[RC];  // the unscaled parameters of expressions hl, hl + 16, ...
#pragma unroll
            for (int k = 0; k < RC; ++k) c_param[k] = (uint32_t)(RS * k + hl) < net ? b.expr_param[e0 + (uint32_t)(RS * k + hl)] : 0.0;
            const double scale = STASH[0];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if ((uint32_t)(hl + RS * q) < nfree) {
                    const double x = (double)xc[q];
                    const double xo = (prm.mode & 1u) ? scale * x : x;
                    b.vars[v0 + my_vi[q]] = xo;
                    if (b.vars_out) b.vars_out[v0 + my_vi[q]] = xo;
                    VOUT[my_vi[q]] = xo;
                }
            }
            group_sync();
            double part[RC];
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                const uint32_t i = (uint32_t)(hl + RS * k);
                part[k] = 0.0;
                if (i < net) {
                    double v[8], g[8];
                    const uint2 gv = gvar[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = VOUT[(gv.x >> (8 * e)) & 0xFFu];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 + e] = VOUT[(gv.y >> (8 * e)) & 0xFFu];
                    const double r = eval_expression<double, false, false>((int)rtag[i], v, c_param[k], g);
                    part[k] = r * r;
                }
            }
            const double sse_u = rows_sum(part);
            if (hl == 0) {
                fx_result res;
                res.accepted = accepted;
                res.trials = trials;
                res.exit = exit_code;
                res.ncomp = 1;
                res.scale = scale;
                res.sse0 = STASH[1];
                res.sse = (double)sse;
                res.sse_unscaled = sse_u;
                b.results[s] = res;
                if (b.results_out) b.results_out[s] = res;
            }
            group_sync();
            phase = GP_NEXT;
        }

        if (__ballot(phase != GP_EXIT || (prm.ladder && !qdone)) == 0ull) break;
    }
    }  // the next class's queue
}

// 16 free variables and fewer (the reference's bench sketches of one to three triangles): one column per lane, four wavefronts
// per SIMD
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void lm_solve_grouped_c1_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<1, 1, double>(b, prm, L, next_system, smem);
}
// 17 ... 32 free variables: two columns per lane, 256 registers, two wavefronts per SIMD
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lm_solve_grouped_c_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<2, 2, double>(b, prm, L, next_system, smem);
}
// ... in f32 (fx_lm_opts_default_f32): 178 registers (three wavefronts per SIMD — 168 registers, 24 bytes of scratch — measured
// the same: 2.79 against 2.77 ms on 125 000 inconsistent ring16 sketches)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lm_solve_grouped_c_f32_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<2, 2, float>(b, prm, L, next_system, smem);
}
// 33 ... 48 free variables (the reference's own bench sketch, fiksi_bench.rs:15-40: 46): three columns per lane are 288
// registers of matrix alone — one wavefront per SIMD, but on every SIMD (the general build's 16 KB of LDS per System leave two
// wavefronts per CU)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_solve_grouped_c3_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<3, 3, double>(b, prm, L, next_system, smem);
}

// Over-constrained structures — more expressions than the shape's 16 / 32 rows, up to twice as many: the same bodies with twice
// the row chunks (cfg5's theme: least squares over more constraints than unknowns)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void lm_solve_grouped_c1r_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<1, 2, double>(b, prm, L, next_system, smem);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lm_solve_grouped_cr_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<2, 4, double>(b, prm, L, next_system, smem);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lm_solve_grouped_cr_f32_kernel(
    DeviceBatch b, LmParams prm, GcLayout L, uint32_t* __restrict__ next_system) {
    extern __shared__ __align__(16) unsigned char smem[];
    grouped_c_body<2, 4, float>(b, prm, L, next_system, smem);
}

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
typedef void (*GcKernel)(DeviceBatch, LmParams, GcLayout, uint32_t*);
struct GcBuild {
    GcKernel fn;
    unsigned int* raised;   // (raise_lds_limit_once's per-device bits)
    uint32_t waves_per_cu;  // by registers
};
// the instantiation for a program's shape (columns per lane, row chunks) and the compute type; fn == nullptr: none
static GcBuild gc_build_for(uint32_t nc, uint32_t rc, bool f32) {
    static unsigned int r1 = 0, r1r = 0, r2 = 0, r2r = 0, r3 = 0, rf = 0, rfr = 0;
    if (f32) {
        if (nc == 2u && rc == 2u) return {&lm_solve_grouped_c_f32_kernel, &rf, 8u};
        if (nc == 2u && rc == 4u) return {&lm_solve_grouped_cr_f32_kernel, &rfr, 8u};
        return {nullptr, nullptr, 0u};
    }
    if (nc == 1u && rc == 1u) return {&lm_solve_grouped_c1_kernel, &r1, 16u};
    if (nc == 1u && rc == 2u) return {&lm_solve_grouped_c1r_kernel, &r1r, 16u};
    if (nc == 2u && rc == 2u) return {&lm_solve_grouped_c_kernel, &r2, 8u};
    if (nc == 2u && rc == 4u) return {&lm_solve_grouped_cr_kernel, &r2r, 8u};
    if (nc == 3u && rc == 3u) return {&lm_solve_grouped_c3_kernel, &r3, 4u};
    return {nullptr, nullptr, 0u};
}

// LDS bytes per wavefront, 0 when the batch has no program
size_t grouped_c_lds_bytes(const DeviceBatch& b, uint32_t es) {
    if (!b.gc_tab || !b.gc_words) return 0;
    const GcLayout L = make_gc_layout(b, es);
    return (size_t)L.tab_bytes + 4u * (size_t)L.stride;
}

bool grouped_c_applies(const DeviceBatch& b, const LmParams& p) {
    if (!p.grouped_one_structure) return false;  // (a context created under FIKSI_AMD_GROUPED_C=0: A / B measurements, tests)
    if (!b.gc_tab || !(b.uniform ? b.u_ncomp == 1u : b.gc_nclasses != 0u) || !b.work_counter || b.has_pose) return false;
    if (p.prof || p.lm.solver != FX_STEP_CHOLESKY || (p.mode & (MODE_UNITS | MODE_LBFGS))) return false;
    const bool f32 = p.lm.precision == 32;
    const GcBuild k = gc_build_for(b.gc_nc, b.gc_rc, f32);
    if (!k.fn) return false;  // (f32: the 32-column instantiations only)
    // one column per lane: eight wavefronts per CU or more; two: six (a SIMD with two is what the build is for); three: one per SIMD
    const size_t lds = grouped_c_lds_bytes(b, f32 ? 4u : 8u);
    return lds != 0 && lds <= (160u * 1024u) / (b.gc_nc == 1u ? 8u : b.gc_nc == 2u ? 6u : 4u);
}

hipError_t launch_solve_grouped_c(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (grouped_tiny_applies(b, p)) return launch_solve_tiny(b, p, stream);  // (at most eight variables and expressions: eight Systems per wavefront)
    const bool f32 = p.lm.precision == 32;
    const GcBuild k = gc_build_for(b.gc_nc, b.gc_rc, f32);
    if (!k.fn) return hipErrorInvalidValue;
    const GcLayout L = make_gc_layout(b, f32 ? 4u : 8u);
    const uint32_t per_wave = L.tab_bytes + 4u * L.stride;
    static const bool trace = getenv("FIKSI_AMD_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[fiksi_amd] grouped kernel, one-structure build (%u columns per lane, %u row chunks): %u B of LDS per wavefront (program %u, 4 x %u per System: %u slots of Jt J, %u Jacobian entries)\n",
                b.gc_nc, b.gc_rc, per_wave, L.tab_bytes, L.stride, b.gc_nslots, b.gc_ng);
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(k.fn), k.raised);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(b.work_counter, 0, sizeof(uint32_t) * (b.gc_nclasses ? b.gc_nclasses : 1u), stream);  // the queue heads
    if (e != hipSuccess) return e;
    uint32_t waves = (b.n_systems + 3u) / 4u;
    if (waves > 256u * 16u) waves = 256u * 16u;
    LmParams pl = p;
    pl.spread = 0u;
    if (p.ladder) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t by_lds = (160u * 1024u) / per_wave;
        uint32_t resident = (uint32_t)cus * (by_lds < k.waves_per_cu ? by_lds : k.waves_per_cu);
        if (resident > waves) resident = waves;
        if (b.order && p.spread && !b.gc_nclasses) pl.spread = resident < b.n_systems / 4u ? resident : b.n_systems / 4u;
        if (p.ladder_tail == 0xFFFFFFFFu) pl.ladder_tail = 32u * resident / (b.gc_nclasses ? b.gc_nclasses : 1u);
    }
    hipLaunchKernelGGL(k.fn, dim3(waves), dim3(64), per_wave, stream, b, pl, L, b.work_counter);
    return hipGetLastError();
}

}  // namespace fx
