// Grouped fused solve, the build for batches of one TINY structure (gfx950, wave64): at most eight variables and eight expressions
// per System — the reference's own bench sketch of one hinged triangle has six and three (fiksi/benches/fiksi_bench.rs:15-40, 46-73).
// Eight lanes per System, eight Systems per wavefront: a System lives in HALF a DPP row, the factor's broadcasts are two
// bank-masked DPP moves (lanes 0 ... 7 of every row read lane K, lanes 8 ... 15 lane K + 8) in front of a fused multiply-add, and a
// trial has eight pivots instead of sixteen: about the instructions the 16-column build of fx_grouped_c.hip issues for four Systems.
//
// Same algorithm, same arithmetic on the same operands in the same order as that build (reference: fiksi/src/assemble/mod.rs:46-167,
// fiksi/src/solve/lm.rs:21-193), hence the same bits:
//   * the program is the 16-column build's own (fx_programs.cpp: build_gc_program_t<1, 1>): row lists, free-variable map, the slots of
//     Jt J's pattern, the lane tables of slot numbers;
//   * Jt J and -Jt r are summed slot by slot and column by column IN LIST ORDER from the program's by-target lists (written for the
//     f32 build) — the order the 16-column build's LDS atomics arrive in: no atomics here;
//   * the sums over a System's lanes are the row butterfly without its last step, which in the 16-column build adds the other half's
//     +0.0; the sequential sum of the scale stops after lane 7 (+0.0 from there on);
//   * columns eight to fifteen of the 16-column build are identity padding nothing depends on: their pivots, solves and products
//     are left out.
// No device-side queue, no ladder: a wavefront takes eight consecutive Systems and runs until the last of them is done (Systems of
// one structure take similar numbers of trials; the ladder needs idle rows, which a wavefront of eight busy Systems does not have).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_grouped_rows.h"
#include "fx_wave.h"

namespace fx {
namespace {

constexpr int TS = 8;  // lanes per System

// lane K of the caller's half row (two v_mov_b64_dpp: banks 0-1 from lane K, banks 2-3 from lane K + 8; two wait states between
// the VALU write of the source and a DPP read of it — inline asm is opaque to the hazard recogniser)
template <int K>
__device__ __forceinline__ double hbcast(double v) {
    double r;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\t"
        "v_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc"
        : "=&v"(r)
        : "v"(v), "n"(K), "n"(K + 8));
    return r;
}
// acc = fma(-(m's lane K of the half row), w, acc): the broadcast through a register, then a plain fused multiply-add — the bits of
// the 16-column build's v_fmac_f64_dpp. (That instruction itself does not take a bank mask on gfx950: masked, it leaves the enabled
// lanes of banks 0-1 unchanged and drops the accumulator in banks 2-3 — tools/probes/dpp_bank_probe.hip; v_mov_b64_dpp does.)
template <int K>
__device__ __forceinline__ void fnma_h(double& acc, double m, double w) {
    acc = fma(-hbcast<K>(m), w, acc);
}
template <int K>
__device__ __forceinline__ void fnma_h_self(double& acc, double w) {
    acc = fma(-hbcast<K>(acc), w, acc);
}
// sum over the eight lanes of a System: row_sum without its last step (row_mirror: + the other half, +0.0 in the 16-column build)
__device__ __forceinline__ double half_sum(double v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    return v;
}
// sum += t(lane 0) + ... + t(lane 7) of the System, strictly in that order
template <int K>
__device__ __forceinline__ void seq_add8_from(double& sum, double t) {
    sum += hbcast<K>(t);
    if constexpr (K + 1 < TS) seq_add8_from<K + 1>(sum, t);
}
__device__ __forceinline__ void seq_add8(double& sum, double t) { seq_add8_from<0>(sum, t); }

// RStep<1, double, K>::factor / forward / backward of fx_grouped_rows.h on half rows (columns 0 ... 7)
template <int K>
__device__ __forceinline__ void tiny_factor_from(double (&a)[TS], double& invd, bool& bad, int hl) {
    const double piv = hbcast<K>(a[K]);
    bad = bad || !(piv > 0.0) || !(piv < Lim<double>::huge());
    const double rs = rsqrt_refined(piv);
    const double ip = rs * rs;  // 1/pivot
    const double ljk = a[K] * rs;
    const bool above = hl > K;
    const double mul = above ? a[K] * ip : 0.0;
    if (above || hl == K) a[K] = ljk;
    if (hl == K) invd = rs;
    if constexpr (K + 1 < TS) {
#pragma unroll
        for (int i = K + 1; i < TS; ++i) fnma_h_self<K>(a[i], mul);
        tiny_factor_from<K + 1>(a, invd, bad, hl);
    }
}
template <int K>
__device__ __forceinline__ void tiny_forward_from(const double (&a)[TS], double invd, double& acc, int hl) {
    if constexpr (K + 1 < TS) {  // (no column lies above the last one)
        const double t = acc * invd;
        const double w = hl > K ? a[K] : 0.0;
        fnma_h<K>(acc, t, w);
        tiny_forward_from<K + 1>(a, invd, acc, hl);
    }
}
template <int K>
__device__ __forceinline__ void tiny_backward_from(const double (&a)[TS], double invd2, double& acc, int hl) {
    if constexpr (K > 0) {  // (no column lies below the first one)
        const double t = acc * invd2;
        const double w = hl < K ? a[K] : 0.0;
        fnma_h<K>(acc, t, w);
        tiny_backward_from<K - 1>(a, invd2, acc, hl);
    }
}

struct TinyLayout {
    uint32_t tab_bytes, off_a, off_g, stride;  // the program; a System's block: 5 x 8 doubles, then Jt J's slots, then the compact Jacobian rows
};
TinyLayout make_tiny_layout(const DeviceBatch& b) {
    TinyLayout L;
    L.tab_bytes = (b.gc_words_all * 4u + 15u) & ~15u;
    L.off_a = 5u * 8u * TS;
    L.off_g = L.off_a + 8u * b.gc_nslots;
    L.stride = L.off_g + 8u * b.gc_ng;
    return L;
}

__global__ __launch_bounds__(64) void lm_solve_tiny_kernel(DeviceBatch b, LmParams prm, TinyLayout L, uint32_t pass_budget) {
    extern __shared__ __align__(16) unsigned char smem[];
    using TK = GcTable<1, 1>;
    const int lane = threadIdx.x;
    const int hl = lane & (TS - 1);
    const int gbase = lane & ~(TS - 1);
    const int grp = lane / TS;
    {
        const uint4* src = reinterpret_cast<const uint4*>(b.gc_tab);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = lane; i < b.gc_words_all / 4u; i += 64) dst[i] = src[i];
    }
    group_sync();
    const uint32_t* TB = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t nvt = TB[1], net = TB[2], nfree = TB[3], nslots = TB[6];
    const int8_t* vcol = reinterpret_cast<const int8_t*>(smem + TK::VCOL);
    const uint8_t* fidx = smem + TK::FIDX;
    const uint8_t* rtag = smem + TK::RTAG;
    const uint16_t* gbaseT = reinterpret_cast<const uint16_t*>(smem + TK::GBASE);
    const uint2* gvar = reinterpret_cast<const uint2*>(smem + TK::GVAR);
    const uint8_t* LT = smem + TK::LT + (uint32_t)hl * 16u;  // this lane's column: the slot of its row i at byte i
    const uint16_t* sptr = reinterpret_cast<const uint16_t*>(smem + TB[9]);   // products by slot, each slot's in list order
    const uint16_t* SPW = reinterpret_cast<const uint16_t*>(smem + TB[10]);
    const uint16_t* cptr = reinterpret_cast<const uint16_t*>(smem + TB[11]);  // right-hand-side entries by column
    const uint16_t* CPE = reinterpret_cast<const uint16_t*>(smem + TB[12]);

    unsigned char* base = smem + L.tab_bytes + (uint32_t)grp * L.stride;
    double* XS = reinterpret_cast<double*>(base);  // [8] working variables: trial point on the free ones
    double* rhsv = XS + TS;                        // [8] -Jt r
    double* R = XS + 2 * TS;                       // [8]
    double* P = XS + 3 * TS;                       // [8] scaled parameters
    double* VOUT = XS + 4 * TS;                    // [8] unscaled values as written back
    double* At = reinterpret_cast<double*>(base + L.off_a);
    double* G = reinterpret_cast<double*>(base + L.off_g);

    const fx_lm_opts o = prm.lm;
    auto gballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> gbase) & 0xFFu; };

    const uint32_t s = blockIdx.x * (64u / TS) + (uint32_t)grp;
    bool running = s < b.n_systems;
    const uint32_t my_vi = (uint32_t)hl < nfree ? (uint32_t)fidx[hl] : 0u;
    const uint32_t dslot = (uint32_t)LT[hl];
    const uint32_t v0 = s * nvt, e0 = s * net;  // (one structure: offsets are multiples)

    double xc = 0.0, diag = 1.0, rhs_l = 0.0, sse = 0.0, lambda = 0.0, scale = 1.0, sse0 = 0.0;
    double acol[TS];
#pragma unroll
    for (int r = 0; r < TS; ++r) acol[r] = 0.0;
    uint32_t accepted = 0, trials = 0, outer = 0, exit_code = FX_EXIT_MAX_OUTER;
    bool fresh = true;

    auto eval_rows = [&]() -> double {
        double part = 0.0;
        if ((uint32_t)hl < net) {
            double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint2 gv = gvar[hl];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = XS[(gv.x >> (8 * e)) & 0xFFu];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] = XS[(gv.y >> (8 * e)) & 0xFFu];
            const int tag = (int)rtag[hl];
            const double r = eval_expression<double, true, false>(tag, v, P[hl], g);
            R[hl] = r;
            const uint32_t gb = gbaseT[hl];
            const int kk = tag_nvars(tag);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < kk) G[gb + (uint32_t)e] = g[e];
            part = r * r;
        }
        group_sync();
        return half_sum(part);
    };
    // K3: Jt J into its slots and -Jt r, every target's products added in list order (fx_grouped_c.hip: form_normal)
    auto form_normal = [&]() {
        for (uint32_t sl = (uint32_t)hl; sl < nslots; sl += TS) {
            const uint32_t t0 = sptr[sl], t1 = sptr[sl + 1];
            double acc = 0.0;
            for (uint32_t t = t0; t < t1; ++t) {
                const uint32_t w = SPW[t];
                acc += G[w & 0xFFu] * G[w >> 8];
            }
            At[sl] = acc;
        }
        {
            const uint32_t t0 = cptr[hl], t1 = cptr[hl + 1];
            double acc = 0.0;
            for (uint32_t t = t0; t < t1; ++t) {
                const uint32_t w = CPE[t];
                acc += G[w & 0xFFu] * -R[w >> 8];
            }
            rhsv[hl] = acc;
        }
        group_sync();
        if ((uint32_t)hl >= nfree) At[dslot] = 1.0;  // identity padding
        group_sync();
        diag = At[dslot];
        rhs_l = rhsv[hl];
        const uint2 lt8 = *reinterpret_cast<const uint2*>(LT);
#pragma unroll
        for (int r = 0; r < TS; ++r) acol[r] = At[((r < 4 ? lt8.x : lt8.y) >> (8 * (r & 3))) & 0xFFu];
    };

    // ================= take the System, scale and perturb it (assemble/mod.rs:32-44, 91-111) =================
    double c_var = 0.0, c_param = 0.0;
    int tagk = 0, colk = -1;
    if (running) {
        c_var = (uint32_t)hl < nvt ? (b.vars_in ? b.vars_in : b.vars0)[v0 + (uint32_t)hl] : 0.0;
        colk = (uint32_t)hl < nvt ? (int)vcol[hl] : -1;
        c_param = (uint32_t)hl < net ? (b.param_in ? b.param_in : b.expr_param)[e0 + (uint32_t)hl] : 0.0;
        tagk = (uint32_t)hl < net ? (int)rtag[hl] : 0;
        if (b.param_in && (uint32_t)hl < net) b.expr_param[e0 + (uint32_t)hl] = c_param;
        if (b.vars_in && (uint32_t)hl < nvt) b.vars0[v0 + (uint32_t)hl] = c_var;
        double scale_recip = 1.0;
        if (prm.mode & 1u) {  // K0a: system scale, summed strictly in reference order (utils.rs:11-33)
            double sum = 0.0;
            uint32_t count = nvt;
            seq_add8(sum, c_var * c_var);
            const bool isd = (uint32_t)hl < net && (tagk == FX_TAG_PPD || tagk == FX_TAG_PLD);
            count += (uint32_t)__popc(gballot(isd));
            seq_add8(sum, isd ? c_param * c_param : 0.0);
            scale = ::sqrt(sum / (double)count);
            scale_recip = 1.0 / scale;
        }
        if ((uint32_t)hl < nvt) {
            double x = (prm.mode & 1u) ? c_var * scale_recip : c_var;
            if (colk >= 0 && (prm.mode & 2u)) {  // K0b: two draws of the LCG per free variable, in column order
                uint32_t st = lcg_jump(42u, 2u * (uint32_t)colk);
                st = st * 1664525u + 1013904223u;
                const double f1 = (1.0 / 4294967295.0) * (double)st;
                st = st * 1664525u + 1013904223u;
                const double f2 = (1.0 / 4294967295.0) * (double)st;
                x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
            }
            XS[hl] = x;
            VOUT[hl] = c_var;
            b.vars[v0 + (uint32_t)hl] = c_var;  // fixed variables stay bit-identical
        }
        if ((uint32_t)hl < net) {
            double prm_e = c_param;
            if ((prm.mode & 1u) && (tagk == FX_TAG_PPD || tagk == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
            P[hl] = prm_e;
        }
        group_sync();
        xc = (uint32_t)hl < nfree ? XS[my_vi] : 0.0;
        lambda = o.lambda0;
    }

    // ================= the trials (lm.rs:115-191), every running System of the wavefront side by side =================
    // pass_budget != 0: a System still running after that many passes of the loop (its start point and pass_budget - 1 trials) is a
    // straggler, and eight lanes in lock step with seven idle neighbours are the wrong place for it: it goes on the list of the
    // 16-column build, which starts it again — the same arithmetic from the same start values — with its lambda ladder.
    bool handed_over = false;
    uint32_t passes = 0;
    while (__ballot(running) != 0ull) {
        if (running && pass_budget && passes >= pass_budget) {
            running = false;
            handed_over = true;
        }
        passes += 1u;
        if (running) {
            int code = LC_FRESH;
            bool go = true;
            double delta = 0.0;
            if (!fresh) {
                code = LC_REJECT;
                if (trials >= o.max_trials) {
                    code = LC_CAP;
                    go = false;
                }
                if (go) {  // K4: factor (Jt J + lambda I) and solve for delta
                    double a[TS];
                    const double dl = diag + lambda;
#pragma unroll
                    for (int r = 0; r < TS; ++r) a[r] = (r == hl) ? dl : acol[r];
                    double invd = 1.0;
                    bool bad = false;
                    tiny_factor_from<0>(a, invd, bad, hl);
                    if (bad) {  // lm.rs:134-137
                        code = LC_SINGULAR;
                        go = false;
                    } else {
                        double acc = rhs_l;
                        const double invd2 = invd * invd;
                        tiny_forward_from<0>(a, invd, acc, hl);
                        tiny_backward_from<TS - 1>(a, invd2, acc, hl);
                        delta = (uint32_t)hl < nfree ? acc * invd2 : 0.0;
                    }
                }
                if (go) {
                    const double dn2 = half_sum(delta * delta);
                    if (!(dn2 == dn2)) {
                        code = LC_NAN;
                        go = false;
                    } else if (dn2 < o.step_tol) {  // lm.rs:139-142
                        code = LC_STEP;
                        go = false;
                    }
                }
                if (go) {
                    if ((uint32_t)hl < nfree) XS[my_vi] = xc + delta;
                    group_sync();
                }
            }
            double sse_t = 0.0;
            if (go) {
                sse_t = eval_rows();
                if (!fresh) {
                    if (sse_t < sse) {
                        code = LC_ACCEPT;  // lm.rs:151-186
                    } else {               // lm.rs:187-190
                        const double lam_k = lambda * o.reject_factor;
                        if (!(sse_t == sse_t) && !(lam_k < 1.0e300)) code = LC_REJ_NAN;  // the reference would double lambda forever
                    }
                }
            }
            bool assemble = false, fin = false;
            if (fresh) {  // the start point
                sse = sse_t;
                sse0 = sse_t;
                assemble = true;
            } else if (code == LC_REJECT) {  // lm.rs:189
                lambda *= o.reject_factor;
                trials += 1u;
            } else {
                trials += code != LC_CAP ? 1u : 0u;
                if (code == LC_CAP) {
                    exit_code = FX_EXIT_TRIAL_CAP;
                    fin = true;
                } else if (code == LC_SINGULAR) {  // lm.rs:134-137
                    lambda *= o.singular_factor;
                } else if (code == LC_NAN) {
                    exit_code = FX_EXIT_NAN;
                    fin = true;
                } else if (code == LC_STEP) {  // lm.rs:139-142
                    exit_code = FX_EXIT_STEP;
                    fin = true;
                } else if (code == LC_ACCEPT) {  // lm.rs:151-186
                    lambda *= o.accept_factor;
                    if (lambda < o.lambda_min) lambda = o.lambda_min;
                    if ((uint32_t)hl < nfree) xc = xc + delta;
                    accepted += 1;
                    const double rel = (sse - sse_t) / sse;
                    sse = sse_t;
                    if (rel <= o.ftol) {
                        exit_code = FX_EXIT_FTOL;
                        fin = true;
                    } else {
                        assemble = true;
                        outer += 1;
                    }
                } else {  // a reject that ends the solve
                    lambda *= o.reject_factor;
                    exit_code = (code == LC_REJ_NAN) ? FX_EXIT_NAN : FX_EXIT_FTOL;
                    fin = true;
                }
            }
            if (assemble) {
                form_normal();
                // top of the next outer iteration (lm.rs:108-112)
                if (fresh && (!(sse == sse) || !(sse < Lim<double>::huge()))) {
                    exit_code = FX_EXIT_NAN;
                    fin = true;
                } else if (outer >= o.max_outer) {
                    fin = true;  // exit_code is still FX_EXIT_MAX_OUTER
                } else if (sse < o.sse_tol) {
                    exit_code = FX_EXIT_SSE;
                    fin = true;
                }
            }
            fresh = false;
            if (fin) running = false;
        }
    }

    // ================= write back scale * x (assemble/mod.rs:161-166), the closing check (constraints/mod.rs:96-109), the record ====
    if (handed_over && hl == 0) b.order[atomicAdd(b.queue_len, 1u)] = s;
    if (s < b.n_systems && !handed_over) {
        const double c_par = (uint32_t)hl < net ? b.expr_param[e0 + (uint32_t)hl] : 0.0;  // the unscaled parameter of expression hl
        if ((uint32_t)hl < nfree) {
            const double xo = (prm.mode & 1u) ? scale * xc : xc;
            b.vars[v0 + my_vi] = xo;
            if (b.vars_out) b.vars_out[v0 + my_vi] = xo;
            VOUT[my_vi] = xo;
        }
        group_sync();
        double part = 0.0;
        if ((uint32_t)hl < net) {
            double v[8], g[8];
            const uint2 gv = gvar[hl];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = VOUT[(gv.x >> (8 * e)) & 0xFFu];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] = VOUT[(gv.y >> (8 * e)) & 0xFFu];
            const double r = eval_expression<double, false, false>((int)rtag[hl], v, c_par, g);
            part = r * r;
        }
        const double sse_u = half_sum(part);
        if (hl == 0) {
            fx_result res;
            res.accepted = accepted;
            res.trials = trials;
            res.exit = exit_code;
            res.ncomp = 1;
            res.scale = scale;
            res.sse0 = sse0;
            res.sse = sse;
            res.sse_unscaled = sse_u;
            b.results[s] = res;
            if (b.results_out) b.results_out[s] = res;
        }
    }
}

}  // namespace

// one structure, one component, at most eight variables and eight expressions, f64 Levenberg-Marquardt with the Cholesky step:
// what the 16-column build of fx_grouped_c.hip would take (the caller has checked grouped_c_applies)
bool grouped_tiny_applies(const DeviceBatch& b, const LmParams& p) {
    static const bool on = [] { const char* e = getenv("FIKSI_AMD_TINY"); return !e || atoi(e) != 0; }();
    if (!on || p.grouped_one_structure == 2) return false;
    if (!b.uniform || b.gc_nclasses || !b.gc_tab || b.gc_nc != 1u || b.gc_rc != 1u || p.lm.precision == 32) return false;
    if (b.u_nvars > (uint32_t)TS || b.u_nexprs > (uint32_t)TS || b.gc_words_all <= b.gc_words) return false;
    return (size_t)make_tiny_layout(b).tab_bytes + 8u * (size_t)make_tiny_layout(b).stride <= 64u * 1024u;
}

hipError_t launch_solve_tiny(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    const TinyLayout L = make_tiny_layout(b);
    const uint32_t per_wave = L.tab_bytes + 8u * L.stride;
    static const bool trace = getenv("FIKSI_AMD_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[fiksi_amd] grouped kernel, tiny one-structure build (8 lanes per System): %u B of LDS per wavefront (program %u, 8 x %u per System)\n",
                per_wave, L.tab_bytes, L.stride);
    // the hand-over of stragglers, when the caller has given the list a place (fx_solve.cpp: launch_solve_scheduled): 16 passes —
    // the reference's bench sketch takes 10; a batch of stragglers only pays those passes twice (tools/tiny_ab.py: 1.05 -> 1.3 ms at worst)
    static const uint32_t budget_env = [] { const char* e = getenv("FIKSI_AMD_TINY_BUDGET"); return e ? (uint32_t)atoi(e) : 16u; }();
    const bool hand_over = b.order && b.queue_len && budget_env != 0u && !b.gc_nclasses;
    if (hand_over) {
        hipError_t e = hipMemsetAsync(b.queue_len, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    DeviceBatch bt = b;
    if (!hand_over) bt.order = nullptr, bt.queue_len = nullptr;
    hipLaunchKernelGGL(lm_solve_tiny_kernel, dim3((b.n_systems + 7u) / 8u), dim3(64), per_wave, stream, bt, p, L, hand_over ? budget_env : 0u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !hand_over) return e;
    LmParams p2 = p;
    p2.grouped_one_structure = 2;  // (the 16-column build itself)
    p2.spread = 0u;                // (the list is no schedule: tickets in order)
    return launch_solve_grouped_c(b, p2, stream);
}

}  // namespace fx
