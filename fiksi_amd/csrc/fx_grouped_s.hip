// Grouped fused solve, the SPARSE build for batches of one structure (gfx950, wave64): four Systems per wavefront, one per row
// of 16 lanes, for components too wide for a register-resident factor (33 ... 255 free variables: from 33 on when the factor is sparse, always from 49 on) whose Cholesky factor is small —
// the reference's bench sketch of 16 hinged triangles (fiksi_bench.rs:15-40, 46-73) has 66 variables, 48 distances, and a factor
// of 291 entries under a minimum-degree order: a dense 66 x 66 factorisation would do eight times the work, and the team kernels
// of the sparse path (one workgroup per System, a wavefront walking a column at a time through barriers) take 54 us per trial of it.
//
// Same algorithm as the other LM kernels (reference: fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193; the step is
// the normal-equation step of fx_grouped.hip), everything a walk over the tables of the batch's one PROGRAM (fx_programs.cpp:
// build_gs_program), copied into LDS once per wavefront and shared by its four Systems:
//   * a System lives in LDS entirely — working point, current point, step, the factor's slots, compact Jacobian rows, residuals,
//     parameters (5.5 KB for the 66-variable sketch) — and a lane holds a handful of scalars;
//   * a trial: zero the factor's slots, add the products of Jt J into them (the product list names slots; ds_add_f64) and -Jt r
//     into the step vector, lambda on the diagonal, factor in place LEVEL by level of the elimination tree (the columns of a level are independent: a lane takes a
//     column, scales it by 1 / sqrt(pivot) — the diagonal slot keeps that reciprocal —, then the level's update triples
//     L(i, j) -= L(i, k) L(j, k) go sixteen at a time), forward substitution by levels (entries of a level's columns, atomically
//     into the vector), backward substitution by levels (a lane gathers its column);
//   * the per-row state machine, the device-side queue, the lambda ladder and the hold passes are fx_grouped_c.hip's.
// Because a rejected trial must not overwrite the Jacobian rows the next trial assembles from, a trial evaluates residuals only and
// an accepted point is evaluated once more with its rows (as in the grouped QR build).
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

// a System's block, bytes (all arrays of doubles; VOUT — the unscaled values of the closing check — lies over the factor's slots)
struct GsLayout {
    uint32_t tab_bytes, off_xc, off_d, off_r, off_p, off_l, off_g, stride;
};

static GsLayout make_gs_layout(const DeviceBatch& b) {
    GsLayout L;
    const uint32_t nv = (b.u_nvars + 1u) & ~1u, m = (b.u_nexprs + 1u) & ~1u, n = (b.gs_nfree + 1u) & ~1u;
    const uint32_t lv = std::max(b.gs_nl, nv);  // (VOUT over the slots)
    uint32_t o = nv * 8u;  // XS at 0
    auto take = [&](uint32_t doubles) { uint32_t at = o; o += doubles * 8u; return at; };
    L.tab_bytes = (b.gs_words * 4u + 15u) & ~15u;
    L.off_xc = take(n);
    L.off_d = take(n);
    L.off_r = take(m);
    L.off_p = take(m);
    L.off_l = take(lv + 2u);  // + the System's scale and the SSE of its start point
    L.off_g = take(b.gs_ng);
    L.stride = o;
    return L;
}

__device__ __forceinline__ uint32_t rfl_s(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// A workgroup is several wavefronts that share ONE copy of the program (the program is a fifth of a wavefront's LDS for the
// 66-variable sketch: five wavefronts per CU instead of four); after the copy they never meet again.
__global__ __launch_bounds__(1024) void lm_solve_grouped_s_kernel(DeviceBatch b, LmParams prm, GsLayout L, uint32_t* __restrict__ next_system) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int hl = lane & (RS - 1);
    const int gbase = lane & ~(RS - 1);
    const int myrow = (int)(threadIdx.x / RS);  // the row's place in the workgroup (a wavefront's four rows are neighbours)
    {
        const uint4* src = reinterpret_cast<const uint4*>(b.gs_tab);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < b.gs_words / 4u; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t* TB = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t nvt = rfl_s(TB[1]), net = rfl_s(TB[2]), nfree = rfl_s(TB[3]), n_pw = rfl_s(TB[4]), n_pe = rfl_s(TB[5]), nl = rfl_s(TB[6]);
    const uint32_t nlev = rfl_s(TB[8]);
    const int16_t* vcol = reinterpret_cast<const int16_t*>(smem + rfl_s(TB[11]));
    const uint16_t* fidx = reinterpret_cast<const uint16_t*>(smem + rfl_s(TB[12]));
    const uint8_t* rtag = smem + rfl_s(TB[13]);
    const uint16_t* gbaseT = reinterpret_cast<const uint16_t*>(smem + rfl_s(TB[14]));
    const uint2* gvar = reinterpret_cast<const uint2*>(smem + rfl_s(TB[15]));
    const uint16_t* dslot = reinterpret_cast<const uint16_t*>(smem + rfl_s(TB[16]));
    const uint16_t* cbase = reinterpret_cast<const uint16_t*>(smem + rfl_s(TB[17]));
    const uint8_t* rowof = smem + rfl_s(TB[18]);
    const uint8_t* lcol = smem + rfl_s(TB[19]);
    const uint16_t* lptr = reinterpret_cast<const uint16_t*>(smem + rfl_s(TB[20]));
    const uint32_t* uptr = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[21]));
    const uint32_t* eptr = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[22]));
    const uint32_t* UPD = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[23]));
    const uint32_t* ENT = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[24]));
    const uint32_t* PW = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[25]));
    const uint32_t* PE = reinterpret_cast<const uint32_t*>(smem + rfl_s(TB[26]));
    const uint8_t* colid = smem + rfl_s(TB[27]);

    unsigned char* const rows0 = smem + L.tab_bytes;
    unsigned char* base = rows0 + (uint32_t)myrow * L.stride;
    unsigned char* const wrows0 = rows0 + (uint32_t)(threadIdx.x >> 6) * 4u * L.stride;  // the blocks of this wavefront's four rows
    T* XS = reinterpret_cast<T*>(base);               // [nvt] working variables: the trial point on the free ones
    T* XC = reinterpret_cast<T*>(base + L.off_xc);    // [nfree] the current point
    T* D = reinterpret_cast<T*>(base + L.off_d);      // [nfree] the step (-Jt r -> y -> delta, in place)
    T* R = reinterpret_cast<T*>(base + L.off_r);      // [net]
    T* P = reinterpret_cast<T*>(base + L.off_p);      // [net] scaled parameters
    T* Lv = reinterpret_cast<T*>(base + L.off_l);     // the factor's slots (diagonal slots: 1 / L_kk)
    T* VOUT = Lv;                                     // [nvt] unscaled values, while the row finishes
    T* STASH = Lv + (nl > ((nvt + 1u) & ~1u) ? nl : ((nvt + 1u) & ~1u));  // [2] scale, SSE of the start point
    T* G = reinterpret_cast<T*>(base + L.off_g);      // compact Jacobian rows of the current point

    const fx_lm_opts o = prm.lm;
    auto gballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> gbase) & 0xFFFFu; };

    int phase = GP_NEXT;
    uint32_t s = 0;
    T sse = T(0);
    double lambda = 0.0;
    uint32_t accepted = 0, trials = 0, outer = 0, exit_code = FX_EXIT_MAX_OUTER;
    bool fresh = false;
    uint32_t held = 0;
    const int wrow = lane / RS;  // the row's place in its wavefront (the ladder's rows are a wavefront's)
    int lad_rank = 0, lad_width = 1, lad_lead = wrow;
    uint32_t lad_members = (uint32_t)wrow * 0x55u;
    bool qdone = false;
    uint32_t last_tk = 0;

    auto row_vars = [&](uint32_t row, const T* from, T (&v)[8]) {
        const uint2 gv = gvar[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = from[(gv.x >> (8 * e)) & 0xFFu];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 + e] = from[(gv.y >> (8 * e)) & 0xFFu];
    };
    // residuals (FULL: and Jacobian rows) of the point in XS; sum of squares: a lane's rows in ascending order, then the row of lanes
    auto eval_rows = [&](auto full_c) -> T {
        constexpr bool FULL = decltype(full_c)::value;
        T part = T(0);
        for (uint32_t row = hl; row < net; row += RS) {
            T v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            row_vars(row, XS, v);
            const int tag = (int)rtag[row];
            const T r = eval_expression<T, FULL, false>(tag, v, P[row], g);
            if constexpr (FULL) {
                R[row] = r;
                const uint32_t gb = gbaseT[row];
                const int kk = tag == FX_TAG_PPD ? 2 : tag_nvars(tag);  // (a distance row: its other two entries are the negatives)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < kk) G[gb + (uint32_t)e] = g[e];
            }
            part += r * r;
        }
        group_sync();
        return row_sum(part);
    };
    // -Jt r of the point whose rows are in G / R, into D (every trial anew: a copy of it kept per System would cost a wavefront
    // per CU for most shapes — LDS is what bounds this build)
    auto form_rhs = [&]() {
        for (uint32_t t0 = 0; t0 < n_pe; t0 += RS * 4u) {
            uint32_t w[4];
            T g1[4], rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = PE[t0 + (uint32_t)(u * RS + hl)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                g1[u] = G[ww & 0x3FFu];
                rr[u] = -R[(ww >> 10) & 0x3FFu];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T pr = g1[u] * rr[u];
                if (w[u] != 0xFFFFFFFFu) lds_add(&D[(w[u] >> 20) & 0xFFu], (w[u] >> 31) ? -pr : pr);  // (bit 31: the entry is kept negated)
            }
        }
    };
    // (Jt J + lam I) delta = -Jt r: assembled into the factor's slots, factored and solved level by level; D = delta.
    // Returns false when a pivot is not positive and finite (lm.rs:134-137).
    auto chol_step = [&](double lam) -> bool {
        {
            double2 z;
            z.x = z.y = 0.0;
            for (uint32_t i = hl; i < nl / 2u; i += RS) reinterpret_cast<double2*>(Lv)[i] = z;
        }
        for (uint32_t c = hl; c < nfree; c += RS) D[c] = T(0);
        group_sync();
        form_rhs();
        for (uint32_t t0 = 0; t0 < n_pw; t0 += RS * 4u) {
            uint32_t w[4];
            T g1[4], g2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = PW[t0 + (uint32_t)(u * RS + hl)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t ww = (w[u] == 0xFFFFFFFFu) ? 0u : w[u];
                g1[u] = G[ww & 0x3FFu];
                g2[u] = G[(ww >> 10) & 0x3FFu];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T pr = g1[u] * g2[u];
                if (w[u] != 0xFFFFFFFFu) lds_add(&Lv[(w[u] >> 20) & 0x3FFu], (w[u] >> 31) ? -pr : pr);
            }
        }
        group_sync();
        for (uint32_t c = hl; c < nfree; c += RS) Lv[dslot[c]] += lam;
        group_sync();
        bool bad = false;
        for (uint32_t lv = 0; lv < nlev; ++lv) {
            const uint32_t c0 = lptr[lv], c1 = lptr[lv + 1];
            for (uint32_t ci = c0 + (uint32_t)hl; ci < c1; ci += RS) {  // a lane, a column: scale it
                const uint32_t k = lcol[ci];
                const uint32_t s0 = cbase[k], s1 = cbase[k + 1];
                const T piv = Lv[s0];
                bad = bad || !(piv > T(0)) || !(piv < Lim<T>::huge());
                const T rs = rsqrt_refined(piv);
                Lv[s0] = rs;
                for (uint32_t e = s0 + 1u; e < s1; ++e) Lv[e] = Lv[e] * rs;
            }
            group_sync();
            const uint32_t u0 = uptr[lv], u1 = uptr[lv + 1];
            for (uint32_t t = u0 + (uint32_t)hl; t < u1; t += RS) {
                const uint32_t w = UPD[t];
                lds_add(&Lv[w & 0x3FFu], -(Lv[(w >> 10) & 0x3FFu] * Lv[w >> 20]));
            }
            group_sync();
        }
        if (gballot(bad) != 0u) return false;
        // L y = rhs, levels ascending
        for (uint32_t lv = 0; lv < nlev; ++lv) {
            const uint32_t c0 = lptr[lv], c1 = lptr[lv + 1];
            for (uint32_t ci = c0 + (uint32_t)hl; ci < c1; ci += RS) {
                const uint32_t k = lcol[ci], c = colid[k];
                D[c] = D[c] * Lv[cbase[k]];
            }
            group_sync();
            const uint32_t e0 = eptr[lv], e1 = eptr[lv + 1];
            for (uint32_t t = e0 + (uint32_t)hl; t < e1; t += RS) {
                const uint32_t w = ENT[t];
                lds_add(&D[w >> 18], -(Lv[w & 0x3FFu] * D[(w >> 10) & 0xFFu]));
            }
            group_sync();
        }
        // Lt x = y, levels descending: a lane gathers its column
        for (uint32_t lv = nlev; lv > 0; --lv) {
            const uint32_t c0 = lptr[lv - 1], c1 = lptr[lv];
            for (uint32_t ci = c0 + (uint32_t)hl; ci < c1; ci += RS) {
                const uint32_t k = lcol[ci], c = colid[k];
                const uint32_t s0 = cbase[k], s1 = cbase[k + 1];
                T acc = D[c];
                for (uint32_t e = s0 + 1u; e < s1; ++e) acc -= Lv[e] * D[rowof[e]];
                D[c] = acc * Lv[s0];
            }
            group_sync();
        }
        return true;
    };

    for (;;) {
        bool park = false;
        if (prm.ladder && prm.ladder_tail != 0u) {
            const bool straggler = __ballot(phase == GP_RUN && lad_rank == 0 && !fresh && trials >= prm.ladder_k) != 0ull;
            if (phase == GP_EXIT && !qdone && !straggler) phase = GP_NEXT;
            park = straggler && last_tk < b.n_systems && b.n_systems - last_tk <= prm.ladder_tail;
        }
        // ================= NEXT: take a System, scale and perturb it (assemble/mod.rs:32-44, 91-111) =================
        if (phase == GP_NEXT && park) phase = GP_EXIT;
        if (phase == GP_NEXT) {
            uint32_t tk = 0;
            if (hl == 0) {
                tk = atomicAdd(next_system, 1u);
                last_tk = tk;
            }
            const uint32_t nxt = (uint32_t)__shfl((int)tk, 0, RS);
            last_tk = (uint32_t)__shfl((int)last_tk, 0, RS);
            if (nxt >= b.n_systems) {
                phase = GP_EXIT;
                qdone = true;
            } else {
                s = nxt;
                const uint32_t v0 = s * nvt, e0 = s * net;
                // K0a: system scale, summed strictly in reference order (utils.rs:11-33): variables, then distance parameters
                double scale = 1.0, scale_recip = 1.0;
                if (prm.mode & 1u) {
                    double sum = 0.0;
                    uint32_t count = nvt;
                    for (uint32_t at = 0; at < nvt; at += RS) {
                        const uint32_t i = at + (uint32_t)hl;
                        const double v = i < nvt ? b.vars0[v0 + i] : 0.0;
                        seq_add(sum, v * v);
                    }
                    for (uint32_t at = 0; at < net; at += RS) {
                        const uint32_t i = at + (uint32_t)hl;
                        const int tag = i < net ? (int)rtag[i] : 0;
                        const bool isd = i < net && (tag == FX_TAG_PPD || tag == FX_TAG_PLD);
                        const double d = isd ? b.expr_param[e0 + i] : 0.0;
                        count += (uint32_t)__popc(gballot(isd));
                        seq_add(sum, d * d);
                    }
                    scale = ::sqrt(sum / (double)count);
                    scale_recip = 1.0 / scale;
                }
                for (uint32_t i = hl; i < nvt; i += RS) {
                    const double v = b.vars0[v0 + i];
                    double x = (prm.mode & 1u) ? v * scale_recip : v;
                    const int col = (int)vcol[i];
                    if (col >= 0 && (prm.mode & 2u)) {  // K0b: two draws of the LCG per free variable, in column order
                        uint32_t st = lcg_jump(42u, 2u * (uint32_t)col);
                        st = st * 1664525u + 1013904223u;
                        const double f1 = (1.0 / 4294967295.0) * (double)st;
                        st = st * 1664525u + 1013904223u;
                        const double f2 = (1.0 / 4294967295.0) * (double)st;
                        x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                    }
                    XS[i] = x;
                    if (col >= 0) XC[col] = x;
                    b.vars[v0 + i] = v;  // fixed variables stay bit-identical
                }
                for (uint32_t i = hl; i < net; i += RS) {
                    const int tag = (int)rtag[i];
                    double prm_e = b.expr_param[e0 + i];
                    if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                    P[i] = prm_e;
                }
                if (hl == 0) STASH[0] = scale;
                group_sync();
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
                    const uint32_t nl_ = (newlead >> (4 * wrow)) & 15u;
                    const bool joining = nl_ != 15u;
                    const int grp = joining ? (int)nl_ : lad_lead;
                    const int srcl = grp * RS + hl;
                    auto cp = [&](auto& v) {
                        const auto t = lane_get(v, srcl);
                        if (joining) v = t;
                    };
                    cp(trials); cp(accepted); cp(outer); cp(exit_code);
                    cp(sse); cp(lambda);
                    lad_width = (int)((wid >> (4 * grp)) & 15u);
                    lad_members = (mem >> (8 * grp)) & 255u;
                    if (joining) {
                        lad_lead = (int)nl_;
                        lad_rank = (int)((newrank >> (4 * wrow)) & 15u);
                        const uint4* lb = reinterpret_cast<const uint4*>(wrows0 + nl_ * L.stride);
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
            if (!fresh) {
                code = LC_REJECT;
                double lam_k = lambda;
                if (lad_rank > 0)
                    for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                if (trials + (uint32_t)lad_rank >= o.max_trials) {
                    code = LC_CAP;
                    go = false;
                }
                if (go && !chol_step(lam_k)) {  // lm.rs:134-137
                    code = LC_SINGULAR;
                    go = false;
                }
                if (go) {
                    T part = T(0);
                    for (uint32_t c = hl; c < nfree; c += RS) part += D[c] * D[c];
                    const T dn2 = row_sum(part);
                    if (!(dn2 == dn2)) {
                        code = LC_NAN;
                        go = false;
                    } else if (dn2 < (T)o.step_tol) {  // lm.rs:139-142
                        code = LC_STEP;
                        go = false;
                    }
                }
                if (go) {
                    for (uint32_t c = hl; c < nfree; c += RS) XS[fidx[c]] = XC[c] + D[c];
                    group_sync();
                }
            }
            T sse_t = T(0);
            if (go) {
                if (fresh) sse_t = eval_rows(std::true_type{});
                else sse_t = eval_rows(std::false_type{});
                if (!fresh) {
                    if (sse_t < sse) {
                        code = LC_ACCEPT;  // lm.rs:151-186
                    } else {               // lm.rs:187-190
                        double lam_k = lambda * o.reject_factor;
                        if (lad_rank > 0)
                            for (int k = 0; k < lad_rank; ++k) lam_k *= o.reject_factor;
                        if (!(sse_t == sse_t) && !(lam_k < 1.0e300)) code = LC_REJ_NAN;  // the reference would double lambda forever
                    }
                }
            }
            // --- the verdicts of a ladder group in rank order: the first that is not a plain reject decides
            int kw = (code != LC_REJECT) ? 0 : 1;
            int code_w = code;
            T sse_w = sse_t;
            int win_row = wrow;
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
                win_row = (int)((lad_members >> (2 * (kw < lad_width ? kw : 0))) & 3u);
                sse_w = lane_get(sse_t, win_row * RS + hl);
            }
            bool assemble = false, fin = false;
            if (fresh) {  // the start point: its rows are in G / R
                sse = sse_t;
                if (hl == 0) STASH[1] = sse_t;
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
                        // the accepted step is the winning row's: its D
                        const T* dw = reinterpret_cast<const T*>(wrows0 + (uint32_t)win_row * L.stride + L.off_d);
                        for (uint32_t c = hl; c < nfree; c += RS) {
                            const T x = XC[c] + dw[c];
                            XC[c] = x;
                            XS[fidx[c]] = x;
                        }
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
                        group_sync();
                        if (assemble) (void)eval_rows(std::true_type{});  // the accepted point once more, with its rows
                    } else {  // a reject that ends the solve
                        lambda *= o.reject_factor;
                        exit_code = FX_EXIT_NAN;
                        fin = true;
                    }
                }
            }
            if (assemble) {
                // top of the next outer iteration (lm.rs:108-112)
                if (fresh && (!(sse == sse) || !(sse < Lim<T>::huge()))) {
                    exit_code = FX_EXIT_NAN;
                    fin = true;
                } else if (outer >= o.max_outer) {
                    fin = true;
                } else if (sse < (T)o.sse_tol) {
                    exit_code = FX_EXIT_SSE;
                    fin = true;
                }
            }
            fresh = false;
            if (fin) {
                phase = GP_FINISH;
                if (lad_rank > 0) phase = GP_EXIT;
                lad_rank = 0;
                lad_width = 1;
                lad_lead = wrow;
                lad_members = (uint32_t)wrow * 0x55u;
            }
        }

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
            const uint32_t v0 = s * nvt, e0 = s * net;
            const double scale = STASH[0], sse0 = STASH[1];
            group_sync();  // (VOUT lies over the factor's slots)
            for (uint32_t i = hl; i < nvt; i += RS) {
                const int col = (int)vcol[i];
                double xo = b.vars0[v0 + i];
                if (col >= 0) {
                    const double x = XC[col];
                    xo = (prm.mode & 1u) ? scale * x : x;
                    b.vars[v0 + i] = xo;
                }
                VOUT[i] = xo;
            }
            group_sync();
            double part = 0.0;
            for (uint32_t i = hl; i < net; i += RS) {
                double v[8], g[8];
                row_vars(i, VOUT, v);
                const double r = eval_expression<double, false, false>((int)rtag[i], v, b.expr_param[e0 + i], g);
                part += r * r;
            }
            const double sse_u = row_sum(part);
            if (hl == 0) {
                fx_result res;
                res.accepted = accepted;
                res.trials = trials;
                res.exit = exit_code;
                res.ncomp = 1;
                res.scale = scale;
                res.sse0 = sse0;
                res.sse = (double)sse;
                res.sse_unscaled = sse_u;
                b.results[s] = res;
            }
            group_sync();
            phase = GP_NEXT;
        }

        if (__ballot(phase != GP_EXIT || (prm.ladder && !qdone)) == 0ull) break;
    }
}

// ------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------
// wavefronts of a workgroup: as many as fit a CU's LDS beside one copy of the program (at most 16)
static uint32_t gs_waves_per_group(const GsLayout& L) {
    const uint32_t room = 160u * 1024u - L.tab_bytes, per_wave = 4u * L.stride;
    uint32_t k = per_wave ? room / per_wave : 1u;
    return k < 1u ? 1u : k > 16u ? 16u : k;
}

bool grouped_s_applies(const DeviceBatch& b, const LmParams& p) {
    if (!p.grouped_one_structure || p.route_grouped == 0) return false;
    if (!b.gs_tab || !b.uniform || b.u_ncomp != 1u || !b.work_counter || b.has_pose) return false;
    if (p.route_grouped < 0 && b.n_systems < p.grouped_min_systems) return false;
    if (p.prof || p.lm.precision == 32 || p.lm.solver != FX_STEP_CHOLESKY || (p.mode & (MODE_UNITS | MODE_LBFGS))) return false;
    const GsLayout L = make_gs_layout(b);
    return (size_t)L.tab_bytes + 8u * (size_t)L.stride <= 160u * 1024u;  // two wavefronts per CU at least
}

hipError_t launch_solve_grouped_s(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    const GsLayout L = make_gs_layout(b);
    const uint32_t k = gs_waves_per_group(L);
    const uint32_t per_group = L.tab_bytes + k * 4u * L.stride;
    static const bool trace = getenv("FIKSI_AMD_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[fiksi_amd] grouped kernel, sparse one-structure build: workgroups of %u wavefronts, %u B of LDS each (program %u, 4 x %u per wavefront: %u slots of the factor, %u Jacobian entries)\n",
                k, per_group, L.tab_bytes, L.stride, b.gs_nl, b.gs_ng);
    static unsigned int raised = 0;
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&lm_solve_grouped_s_kernel), &raised);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(b.work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t waves = (b.n_systems + 3u) / 4u;
    uint32_t groups = (waves + k - 1u) / k;
    if (groups > (uint32_t)cus) groups = (uint32_t)cus;  // one workgroup per CU: every wavefront stays until the queue is empty
    LmParams pl = p;
    pl.spread = 0u;
    if (p.ladder && p.ladder_tail == 0xFFFFFFFFu) pl.ladder_tail = 32u * groups * k;
    hipLaunchKernelGGL(lm_solve_grouped_s_kernel, dim3(groups), dim3(64u * k), per_group, stream, b, pl, L, b.work_counter);
    return hipGetLastError();
}

}  // namespace fx
