// Fused per-System solve for MEDIUM components: 65 .. 128 free variables (still <= 256 expressions per
// component and <= 512 variables per System, so everything stays in LDS).
//
// Same algorithm and control flow as lm_solve_kernel (fx_kernels.hip; reference
// fiksi/src/assemble/mod.rs:46-167 and fiksi/src/solve/lm.rs:21-193), still one wavefront per System,
// but the matrix no longer fits one column per lane: JtJ + lambda I lives in LDS as a packed lower
// triangle (66 KB for 128 columns) and is factored in place by a 2 x 2 blocked Cholesky whose diagonal
// blocks (64 columns each) run through the same register-resident chol_factor<64> as the fused kernel;
// the off-diagonal block is a register forward sweep per row, the Schur complement a row-in-registers
// times broadcast-row product. (A first version swept the triangle in LDS row by row: with one
// wavefront per CU every LDS round trip was exposed, ~1000 cycles per four rows.) JtJ is re-formed from
// the Jacobian rows of the current point at every lambda trial (a few hundred LDS atomics) instead of
// being kept next to the factor: two triangles would not fit.
//
// Without this kernel a System with 66 variables fell to the host-driven sparse path (~0.8 ms each, one
// after the other); here Systems run concurrently, one per CU (LDS-bound occupancy).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_device.h"
#include "fx_chol.h"
#include "fx_expr.h"
#include "fx_wave.h"

namespace fx {

struct WideLayout {
    uint32_t vt, mr, n;  // padded variables per System, rows per component, free variables per component
    uint32_t off_xs, off_vout, off_l, off_rhs, off_delta, off_aux, off_g, off_r, off_p, off_gvar, off_gcol, off_rtag, off_fidx,
        off_colof;
    uint32_t total;
};

static WideLayout make_wide_layout(uint32_t max_free, uint32_t max_vars, uint32_t max_rows, uint32_t qr_nx = 0) {
    WideLayout L;
    L.vt = (max_vars + 7u) & ~7u;
    L.mr = (max_rows + 7u) & ~7u;
    L.n = (max_free + 7u) & ~7u;
    if (L.vt == 0) L.vt = 8;
    if (L.mr == 0) L.mr = 8;
    if (L.n == 0) L.n = 8;
    uint32_t o = 0;
    auto take = [&](uint32_t bytes) { uint32_t at = o; o += (bytes + 15u) & ~15u; return at; };
    L.off_xs = take(2u * L.vt * 8u);
    L.off_vout = take(L.vt * 8u);
    // (QR build: the augmented matrix by its symbolic patterns, qr_nx doubles, in place of the triangle)
    L.off_l = take(qr_nx ? qr_nx * 8u : L.n * (L.n + 1u) / 2u * 8u);
    L.off_rhs = take(L.n * 8u);
    L.off_delta = take(L.n * 8u);
    L.off_aux = take(L.n * 8u);
    L.off_g = take(2u * L.mr * 8u * 8u);
    L.off_r = take(2u * L.mr * 8u);
    L.off_p = take(L.mr * 8u);
    L.off_gvar = take(L.mr * 8u * 2u);
    L.off_gcol = take(L.mr * 8u * 2u);
    L.off_rtag = take(L.mr);
    L.off_fidx = take(L.n * 2u);
    L.off_colof = take(L.vt * 2u);
    L.total = o;
    return L;
}

size_t wide_lds_bytes(const DeviceBatch& b) { return make_wide_layout(b.w_max_free, b.w_max_vars, b.w_max_rows).total; }

__device__ __forceinline__ uint32_t tri(uint32_t i, uint32_t j) { return i * (i + 1u) / 2u + j; }  // i >= j

// QR = true is FX_STEP_QR for these components: the reference's Householder QR of [J; sqrt(lambda) I]
// (solvi/src/decomposition/sparse/qr.rs:226-356) with the operations and the order of the one-wavefront QR kernel
// (fx_kernels.hip: qr_step), driven by the host's table program (fx_programs.cpp: build_qrg_program, wide form) — the matrix by
// its symbolic patterns in LDS, a lane per ACTIVE column of the Householder step at hand, the tables in global memory; the
// reference's sequential sums, the correctly rounded atan2. Every bit is the reference algorithm's (tests/test_gpu_qr_step.py).
template <bool QR>
__device__ __forceinline__ void wide_body(const DeviceBatch& b, const LmParams& prm, const WideLayout& L, unsigned char* smem) {
    const int lane = threadIdx.x;
    const uint32_t s = QR ? b.qr_none.qrw_list[blockIdx.x] : b.w_list[blockIdx.x];
    // diagnostic phase stamps (prm.prof != nullptr only from fx_debug_phase_cycles):
    // 0 setup, 1 eval, 2 form, 3 factor, 4 solve, 5 tail
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_last = prm.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int phase) {
        if (prm.prof) {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[phase] += t - t_last;
            t_last = t;
        }
    };
    double* XS = reinterpret_cast<double*>(smem + L.off_xs);      // [2][vt]
    double* VOUT = reinterpret_cast<double*>(smem + L.off_vout);  // [vt]
    double* Lm = reinterpret_cast<double*>(smem + L.off_l);       // packed lower triangle
    double* RHS = reinterpret_cast<double*>(smem + L.off_rhs);    // [n] -Jt r at the current point
    double* DEL = reinterpret_cast<double*>(smem + L.off_delta);  // [n] right-hand side -> step
    double* AUX = reinterpret_cast<double*>(smem + L.off_aux);    // [n] the unrefined step (FX_STEP_CHOLESKY_REFINED)
    double* G = reinterpret_cast<double*>(smem + L.off_g);        // [2][mr][8]
    double* R = reinterpret_cast<double*>(smem + L.off_r);        // [2][mr]
    double* P = reinterpret_cast<double*>(smem + L.off_p);        // [mr]
    uint16_t* gvar = reinterpret_cast<uint16_t*>(smem + L.off_gvar);
    int16_t* gcol = reinterpret_cast<int16_t*>(smem + L.off_gcol);
    uint8_t* rtag = reinterpret_cast<uint8_t*>(smem + L.off_rtag);
    uint16_t* fidx = reinterpret_cast<uint16_t*>(smem + L.off_fidx);
    int16_t* colof = reinterpret_cast<int16_t*>(smem + L.off_colof);
    const uint32_t vt = L.vt, mr = L.mr;

    const uint32_t v0 = b.var_off[s], nvt = b.var_off[s + 1] - v0;
    const uint32_t e0 = b.expr_off[s], net = b.expr_off[s + 1] - e0;
    const uint32_t ncomp = b.sys_ncomp[s];
    const fx_lm_opts o = prm.lm;

    // ---- K0a: system scale, summed strictly in reference order (assemble/mod.rs:32-44) ------------
    double scale = 1.0, scale_recip = 1.0;
    if (prm.mode & 1u) {
        scale = system_scale_wave(
            nvt, net, lane, [&](uint32_t i) { return b.vars0[v0 + i]; }, [&](uint32_t i) { return (int)(b.expr_tag[e0 + i] & 0x7F); },
            [&](uint32_t i) { return b.expr_param[e0 + i]; });
        scale_recip = 1.0 / scale;
    }
    for (uint32_t i = lane; i < nvt; i += 64) {
        double v = b.vars0[v0 + i];
        double xsv = (prm.mode & 1u) ? v * scale_recip : v;
        XS[i] = xsv;
        XS[vt + i] = xsv;
        VOUT[i] = v;
        b.vars[v0 + i] = v;
    }
    __syncthreads();

    uint32_t rng = 42u;
    uint32_t tot_accept = 0, tot_trials = 0, last_exit = FX_EXIT_SSE, comps_done = 0;
    double tot_sse0 = 0.0, tot_sse = 0.0;

    for (uint32_t c = 0; c < ncomp; ++c) {
        // ---- free variables of the component, ascending (:91-111); perturbation on the way (:113-124)
        const uint32_t rng_start = rng;  // the component's draws start here
        uint32_t nfree = 0;
        bool any_var = false;
        for (uint32_t base = 0; base < nvt; base += 64) {
            uint32_t i = base + lane;
            bool in = false, member = false;
            if (i < nvt) {
                uint16_t info = b.var_info[v0 + i];
                member = (info & VAR_COMP_MASK) == c;
                in = member && !(info & VAR_FIXED_BIT);
            }
            any_var = any_var || (__ballot(member) != 0ull);
            uint64_t mk = __ballot(in);
            uint32_t pos = nfree + (uint32_t)__popcll(mk & lanemask_lt(lane));
            if (i < nvt) colof[i] = in ? (int16_t)pos : (int16_t)-1;
            if (in) {
                fidx[pos] = (uint16_t)i;
                if (prm.mode & 2u) {
                    uint32_t st = lcg_jump(rng, 2u * pos);
                    st = st * 1664525u + 1013904223u;
                    double f1 = (1.0 / 4294967295.0) * (double)st;
                    st = st * 1664525u + 1013904223u;
                    double f2 = (1.0 / 4294967295.0) * (double)st;
                    double x = b.vars0[v0 + i];
                    if (prm.mode & 1u) x = x * scale_recip;
                    x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                    XS[i] = x;
                    XS[vt + i] = x;
                }
            }
            nfree += (uint32_t)__popcll(mk);
        }
        if (!uniform(any_var)) continue;
        if (prm.mode & 2u) rng = lcg_jump(rng, 2u * nfree);
        __syncthreads();

        // ---- rows of the component, ascending expression id (:139-145)
        uint32_t m_rows = 0;
        for (uint32_t base = 0; base < net; base += 64) {
            uint32_t i = base + lane;
            bool in = (i < net) && (b.expr_comp[e0 + i] == c);
            uint64_t mk = __ballot(in);
            uint32_t pos = m_rows + (uint32_t)__popcll(mk & lanemask_lt(lane));
            if (in) {
                int tag = b.expr_tag[e0 + i] & 0x7F;
                const uint16_t* f = b.expr_idx + 4 * (size_t)(e0 + i);
                uint16_t ff[4] = {f[0], f[1], f[2], f[3]};
                uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int k = expand_vars(tag, ff, vars8);
                double prm_e = b.expr_param[e0 + i];
                if ((prm.mode & 1u) && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) prm_e = scale_recip * prm_e;
                rtag[pos] = (uint8_t)tag;
                P[pos] = prm_e;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gvar[pos * 8 + e] = (uint16_t)vars8[e];
                    gcol[pos * 8 + e] = (e < k) ? colof[vars8[e]] : (int16_t)-1;
                }
            }
            m_rows += (uint32_t)__popcll(mk);
        }
        __syncthreads();

        auto eval_rows = [&](int buf) -> double {
            const double* xs = XS + buf * vt;
            double part = 0.0;
            for (uint32_t row = lane; row < m_rows; row += 64) {
                double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xs[gvar[row * 8 + e]];
                double r = eval_expression<double, true, QR>(rtag[row], v, P[row], g);
                R[buf * mr + row] = r;
#pragma unroll
                for (int e = 0; e < 8; ++e) G[(buf * mr + row) * 8 + e] = g[e];
                part += r * r;
            }
            if constexpr (QR) {  // the reference's sum of squares, in index order (lm.rs:195-197): every lane adds the same numbers
                __syncthreads();
                double acc = 0.0;
                const double* rb = R + buf * mr;
                for (uint32_t row0 = 0; row0 < m_rows; row0 += 16) {
                    double rr[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) rr[u] = (row0 + u < m_rows) ? rb[row0 + u] : 0.0;  // + 0.0: exact
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += rr[u] * rr[u];
                }
                return bcast(acc, 0);
            }
            return wave_sum(part);
        };
        // One LM trial's step by the reference's QR, table-driven. Returns false when R has an exactly zero diagonal entry
        // (sparse_col_mat.rs:800-810); the step lands in DEL, |delta|^2 (summed in index order) in dn2_out.
        auto qr_step = [&](double lam, int buf, double& dn2_out) -> bool {
            bool ok = true;
            if constexpr (QR) {
                const QrPlans& Q = b.qr_none;
                const uint32_t* TB = Q.qrw_words + Q.qrw_prog_off[Q.qrw_comp_first[blockIdx.x] + c];
                const uint32_t qn = TB[0], qm = TB[1], nx = TB[2], rhsbase = TB[12];  // (offsets into the matrix in elements)
                const uint16_t* scat = reinterpret_cast<const uint16_t*>(TB + TB[4]);
                const uint16_t* rhs_off = reinterpret_cast<const uint16_t*>(TB + TB[5]);
                const uint16_t* damp = reinterpret_cast<const uint16_t*>(TB + TB[6]);
                const uint16_t* cposT = reinterpret_cast<const uint16_t*>(TB + TB[7]);
                const uint32_t* stepT = TB + TB[8];
                const uint16_t* bptr = reinterpret_cast<const uint16_t*>(TB + TB[9]);
                const uint32_t* bent = TB + TB[10];
                double* X = Lm;
                const double sl = ::sqrt(lam);  // lm.rs:119
                __syncthreads();
                for (uint32_t i = lane; i < nx; i += 64) X[i] = 0.0;
                __syncthreads();
                // J (duplicates of a row summed in gradient order) and b = -r; the damping entry of every column
                for (uint32_t row = lane; row < qm; row += 64) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t off = scat[row * 8 + e];
                        if (off != 0xFFFFu) lds_add(&X[off], G[(buf * mr + row) * 8 + e]);
                    }
                    X[rhs_off[row]] = -R[buf * mr + row];
                }
                for (uint32_t cc = lane; cc < qn; cc += 64) X[damp[cc]] = sl;
                __syncthreads();
                for (uint32_t k = 0; k < qn; ++k) {
                    const uint32_t w0 = stepT[3 * k], ent = stepT[3 * k + 1], nact = stepT[3 * k + 2];
                    const uint32_t vdiag = w0 & 0xFFFFu, len = w0 >> 16;
                    const double v0 = X[vdiag];
                    double norm = 0.0, beta = 0.0, v0n = 1.0;
                    auto householder = [&](double sigma) {  // calculate_householder (qr.rs:244-275); every lane computes it
                        norm = ::fabs(v0);
                        beta = (v0 >= 0.0) ? 0.0 : 2.0;
                        v0n = 1.0;
                        if (sigma != 0.0) {
                            norm = ::sqrt(sigma + v0 * v0);
                            v0n = (v0 <= 0.0) ? v0 - norm : -sigma / (v0 + norm);
                            beta = -(1.0 / (norm * v0n));
                        }
                    };
                    for (uint32_t i0 = 0; i0 < nact; i0 += 64) {
                        const bool act = i0 + (uint32_t)lane < nact;
                        const uint32_t* e = TB + ent + (act ? i0 + (uint32_t)lane : 0u);  // (idle lanes follow column 0 and store nothing)
                        const uint32_t wk0 = e[0];
                        const double xk0 = X[wk0];
                        // the vector's entries in blocks of four (the host pads with the zero slot), three passes: the
                        // vector's own square sum, the column's inner product (qr.rs:226-240), the update
                        double sigma = 0.0;
                        for (uint32_t u = 0; u < len; u += 4) {
                            double vk[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) vk[t] = X[e[(1u + u + (uint32_t)t) * nact] >> 16];
#pragma unroll
                            for (int t = 0; t < 4; ++t) sigma = sigma + vk[t] * vk[t];
                        }
                        householder(sigma);
                        double tau = 0.0;
                        tau = tau + v0n * xk0;
                        for (uint32_t u = 0; u < len; u += 4) {
                            double vk[4], xj[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const uint32_t w = e[(1u + u + (uint32_t)t) * nact];
                                vk[t] = X[w >> 16];
                                xj[t] = X[w & 0xFFFFu];
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) tau = tau + vk[t] * xj[t];
                        }
                        tau = tau * beta;
                        if (act) {
                            X[wk0] = xk0 - v0n * tau;
                            for (uint32_t u = 0; u < len; u += 4) {
                                uint32_t w[4];
                                double vk[4], xj[4];
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    w[t] = e[(1u + u + (uint32_t)t) * nact];
                                    vk[t] = X[w[t] >> 16];
                                    xj[t] = X[w[t] & 0xFFFFu];
                                }
#pragma unroll
                                for (int t = 0; t < 4; ++t) X[w[t] & 0xFFFFu] = xj[t] - vk[t] * tau;  // (padding: 0 - 0 tau into the zero slot)
                            }
                        }
                        __syncthreads();
                    }
                    if (lane == 0) X[vdiag] = norm;  // R's diagonal (qr.rs:319)
                    __syncthreads();
                }
                // back substitution with R (sparse_col_mat.rs:788-826), in place on the right-hand side
                {
                    bool zero_diag = false;
                    for (uint32_t cc = lane; cc < qn; cc += 64) zero_diag = zero_diag || X[stepT[3 * cc] & 0xFFFFu] == 0.0;
                    ok = __ballot(zero_diag) == 0ull;
                }
                if (ok) {
                    for (uint32_t ii = qn; ii > 0; --ii) {
                        const uint32_t i = ii - 1u;
                        const uint32_t dgo = stepT[3 * i] & 0xFFFFu, eb = bptr[i], ee = bptr[i + 1];
                        const double coeff = X[rhsbase + i] / X[dgo];
                        for (uint32_t t = eb + (uint32_t)lane; t < ee; t += 64) {
                            const uint32_t wv = bent[t];
                            const uint32_t yo = rhsbase + (wv >> 16);
                            X[yo] = X[yo] - coeff * X[wv & 0xFFFFu];
                        }
                        __syncthreads();
                        if (lane == 0) X[rhsbase + i] = coeff;
                        __syncthreads();
                    }
                }
                // undo the column permutation (qr.rs:354): delta[colperm[j]] = x_j; |delta|^2 in index order
                for (uint32_t cc = lane; cc < qn; cc += 64) DEL[cc] = ok ? X[rhsbase + cposT[cc]] : 0.0;
                __syncthreads();
                double acc = 0.0;
                for (uint32_t c0 = 0; c0 < qn; c0 += 16) {
                    double dd[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) dd[u] = (c0 + u < qn) ? DEL[c0 + u] : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += dd[u] * dd[u];
                }
                dn2_out = bcast(acc, 0);
                __syncthreads();
            }
            return ok;
        };
        // rhs = -Jt r at `buf`, one row per wave instruction (row order)
        auto form_rhs = [&](int buf) {
            for (uint32_t i = lane; i < nfree; i += 64) RHS[i] = 0.0;
            __syncthreads();
            for (uint32_t row = 0; row < m_rows; ++row) {
                if (lane < 8) {
                    int col = gcol[row * 8 + lane];
                    if (col >= 0) lds_add(&RHS[col], G[(buf * mr + row) * 8 + lane] * -R[buf * mr + row]);
                }
            }
            __syncthreads();
        };
        // packed lower triangle of Jt J + lambda I from the rows at `buf`
        auto form_matrix = [&](int buf, double lambda) {
            const uint32_t len = nfree * (nfree + 1u) / 2u;
            for (uint32_t i = lane; i < len; i += 64) Lm[i] = 0.0;
            __syncthreads();
            const int e1 = lane >> 3, e2 = lane & 7;
            constexpr int RB = 4;
            for (uint32_t row0 = 0; row0 < m_rows; row0 += RB) {
                int c1[RB], c2[RB];
                double g1[RB], g2[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    uint32_t row = min(row0 + q, m_rows - 1);
                    c1[q] = gcol[row * 8 + e1];
                    c2[q] = gcol[row * 8 + e2];
                    g1[q] = G[(buf * mr + row) * 8 + e1];
                    g2[q] = G[(buf * mr + row) * 8 + e2];
                    if (row0 + q >= m_rows) c1[q] = -1;
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    // lower triangle; a column that repeats inside the row meets itself on the diagonal
                    // from both orders, as (g1 + g2)^2 requires
                    if (c1[q] >= 0 && c2[q] >= 0 && c1[q] >= c2[q]) lds_add(&Lm[tri((uint32_t)c1[q], (uint32_t)c2[q])], g1[q] * g2[q]);
                }
            }
            __syncthreads();
            for (uint32_t i = lane; i < nfree; i += 64) Lm[tri(i, i)] += lambda;
            __syncthreads();
        };
        // ---- blocked Cholesky of the packed triangle, in place: [A11 .; A21 A22] with a 64-column first
        // block. Both diagonal blocks go through the register-resident chol_factor<64> of the fused kernel
        // (one column per lane, v_readlane broadcasts); in between, L21 = A21 L11^-T row by row with the
        // register forward sweep, and the Schur complement A22 - L21 L21^T with each lane holding its own
        // row of L21 in registers against broadcast reads of the others. The factor is written back as
        // the plain lower triangle L; the solves reload a block into registers when they need it.
        const uint32_t n1 = min(nfree, 64u), n2 = nfree - n1;
        // block `base`: lane j gets column base + j of the (symmetric) block, identity where padded
        auto load_block = [&](double (&a)[64], uint32_t base, uint32_t nb) {
            const uint32_t c = base + (uint32_t)lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const uint32_t r = base + (uint32_t)i;
                const bool real = (uint32_t)lane < nb && (uint32_t)i < nb;
                const uint32_t hi = real ? max(r, c) : 0u, lo = real ? min(r, c) : 0u;
                const double v = Lm[tri(hi, lo)];
                a[i] = real ? v : ((i == lane) ? 1.0 : 0.0);
            }
        };
        // the combined storage of a factored block back from the plain triangle (see chol_factor)
        auto reload_factor = [&](double (&a)[64], double& invd, uint32_t base, uint32_t nb) {
            const uint32_t c = base + (uint32_t)lane;
            const bool lane_real = (uint32_t)lane < nb;
            const double d = lane_real ? Lm[tri(c, c)] : 1.0;
            invd = 1.0 / d;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const uint32_t r = base + (uint32_t)i;
                const bool real = lane_real && (uint32_t)i < nb;
                const uint32_t hi = real ? max(r, c) : 0u, lo = real ? min(r, c) : 0u;
                const double v = Lm[tri(hi, lo)];
                a[i] = real ? ((i > lane) ? v * d : v) : ((i == lane) ? 1.0 : 0.0);
            }
        };
        auto factor = [&]() -> bool {
            bool ok = true;
            double a[64], invd = 1.0;
            for (uint32_t blk = 0; blk < 2u; ++blk) {  // one call site of chol_factor<64>
                const uint32_t base = blk * 64u, nb = blk ? n2 : n1;
                if (nb == 0u) break;
                load_block(a, base, nb);
                __syncthreads();
                ok = chol_factor<64, double>(a, invd, lane, (int)nb) && ok;  // the padding of a short second block is skipped
                // row `lane` of L and d back into the triangle
                if ((uint32_t)lane < nb) {
                    const uint32_t c = base + (uint32_t)lane;
#pragma unroll
                    for (int p = 0; p < 64; ++p)
                        if (p <= lane) Lm[tri(c, base + (uint32_t)p)] = a[p];
                }
                __syncthreads();
                if (blk == 0u && n2 > 0u) {
                    // L21: row r of A21 against L11 (still in registers)
                    for (uint32_t r = 64u; r < nfree; ++r) {
                        const double v = Lm[tri(r, 0) + (uint32_t)lane];
                        const double y = chol_forward<64, double>(a, invd, v, lane);
                        Lm[tri(r, 0) + (uint32_t)lane] = y;
                    }
                    __syncthreads();
                    // Schur complement, column 64 + lane: own row of L21 in registers
                    const uint32_t c = 64u + (uint32_t)lane;
                    const bool mine = (uint32_t)lane < n2;
                    double rj[64];
#pragma unroll
                    for (int k = 0; k < 64; ++k) rj[k] = mine ? Lm[tri(c, 0) + (uint32_t)k] : 0.0;
                    for (uint32_t i = 64u; i < nfree; ++i) {
                        const uint32_t rb = tri(i, 0);
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < 64; ++k) acc = fma(Lm[rb + (uint32_t)k], rj[k], acc);  // broadcast reads
                        if (mine && i >= c) Lm[rb + c] -= acc;
                    }
                    __syncthreads();
                }
            }
            return ok;
        };
        // L L^T x = DEL in place through the blocks; returns |x|^2
        auto solve = [&]() -> double {
            double a[64], invd = 1.0;
            reload_factor(a, invd, 0u, n1);
            const double y1 = chol_forward<64, double>(a, invd, ((uint32_t)lane < n1) ? DEL[lane] : 0.0, lane);
            double u = y1;
            if (n2 > 0u) {
                __syncthreads();
                if ((uint32_t)lane < n1) DEL[lane] = y1;
                __syncthreads();
                // b2 - L21 y1, row 64 + lane
                double t = 0.0;
                if ((uint32_t)lane < n2) {
                    const uint32_t rb = tri(64u + (uint32_t)lane, 0);
                    t = DEL[64 + lane];
                    for (uint32_t k = 0; k < 64u; ++k) t = fma(-Lm[rb + k], DEL[k], t);
                }
                reload_factor(a, invd, 64u, n2);
                const double x2 = chol_solve<64, double>(a, invd, t, lane, (int)n2);
                __syncthreads();
                if ((uint32_t)lane < n2) DEL[64 + lane] = x2;
                __syncthreads();
                // y1 - L21^T x2, column lane
                for (uint32_t r = 0; r < n2; ++r) u = fma(-Lm[tri(64u + r, 0) + (uint32_t)lane], DEL[64u + r], u);
                reload_factor(a, invd, 0u, n1);
            }
            const double x1 = chol_backward<64, double>(a, invd, u, lane);
            __syncthreads();
            if ((uint32_t)lane < n1) DEL[lane] = x1;
            __syncthreads();
            double part = 0.0;
            for (uint32_t i = lane; i < nfree; i += 64) part += DEL[i] * DEL[i];
            return wave_sum(part);
        };

        int cur = 0;
        stamp(0);
        double sse = eval_rows(0);
        const double sse_start = sse;
        stamp(1);
        if constexpr (!QR) form_rhs(0);
        stamp(2);
        double lambda = o.lambda0;
        uint32_t accepted = 0, trials = 0, exit_code = FX_EXIT_MAX_OUTER;
        bool done = false;
        if (!(sse == sse) || !(sse < 1.0e300)) {
            exit_code = FX_EXIT_NAN;
            done = true;
        }
        for (uint32_t outer = 0; outer < o.max_outer && !done; ++outer) {
            if (sse < o.sse_tol) {  // lm.rs:110-112
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
                stamp(5);
                double dn2 = 0.0;
                if constexpr (QR) {
                    if (!uniform(qr_step(lambda, cur, dn2))) {  // lm.rs:134-137
                        lambda *= o.singular_factor;
                        continue;
                    }
                } else {
                form_matrix(cur, lambda);
                stamp(2);
                const bool factored = factor();
                stamp(3);
                if (!factored) {  // lm.rs:134-137
                    lambda *= o.singular_factor;
                    if (!(lambda < 1.0e300)) {
                        exit_code = FX_EXIT_NAN;
                        done = true;
                        break;
                    }
                    continue;
                }
                for (uint32_t i = lane; i < nfree; i += 64) DEL[i] = RHS[i];
                __syncthreads();
                dn2 = solve();
                stamp(4);
                }
                if (!QR && o.solver == FX_STEP_CHOLESKY_REFINED) {
                    // corrected semi-normal equations, as in lm_solve_kernel: t = -r - J delta from the rows,
                    // (JtJ + lambda I) e = Jt t - lambda delta with the factor at hand, delta += e
                    double* tr = R + (cur ^ 1) * mr;
                    for (uint32_t i = lane; i < nfree; i += 64) AUX[i] = DEL[i];
                    __syncthreads();
                    for (uint32_t row = lane; row < m_rows; row += 64) {
                        double acc = -R[cur * mr + row];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int cc = gcol[row * 8 + e];
                            if (cc >= 0) acc -= G[(cur * mr + row) * 8 + e] * AUX[cc];
                        }
                        tr[row] = acc;
                    }
                    for (uint32_t i = lane; i < nfree; i += 64) DEL[i] = 0.0;
                    __syncthreads();
                    for (uint32_t row = 0; row < m_rows; ++row) {
                        if (lane < 8) {
                            const int cc = gcol[row * 8 + lane];
                            if (cc >= 0) lds_add(&DEL[cc], G[(cur * mr + row) * 8 + lane] * tr[row]);
                        }
                    }
                    __syncthreads();
                    for (uint32_t i = lane; i < nfree; i += 64) DEL[i] -= lambda * AUX[i];
                    __syncthreads();
                    (void)solve();
                    double part = 0.0;
                    for (uint32_t i = lane; i < nfree; i += 64) {
                        const double d2 = AUX[i] + DEL[i];
                        DEL[i] = d2;
                        part += d2 * d2;
                    }
                    dn2 = wave_sum(part);
                    __syncthreads();
                }
                if (!(dn2 == dn2)) {
                    exit_code = FX_EXIT_NAN;
                    done = true;
                    break;
                }
                if (dn2 < o.step_tol) {  // lm.rs:139-142
                    exit_code = FX_EXIT_STEP;
                    done = true;
                    break;
                }
                const int trial = cur ^ 1;
                for (uint32_t i = lane; i < nfree; i += 64) {
                    uint32_t vi = fidx[i];
                    XS[trial * vt + vi] = XS[cur * vt + vi] + DEL[i];
                }
                __syncthreads();
                stamp(5);
                const double sse_t = eval_rows(trial);
                stamp(1);
                if (sse_t < sse) {  // accept, lm.rs:151-186
                    lambda *= o.accept_factor;
                    if (lambda < o.lambda_min) lambda = o.lambda_min;
                    cur = trial;
                    accepted += 1;
                    const double rel = (sse - sse_t) / sse;
                    sse = sse_t;
                    if (rel <= o.ftol) {
                        exit_code = FX_EXIT_FTOL;
                        done = true;
                        break;
                    }
                    __syncthreads();
                    if constexpr (!QR) form_rhs(cur);
                    break;
                } else {  // reject, lm.rs:187-190
                    lambda *= o.reject_factor;
                    if (!(sse_t == sse_t) && !(lambda < 1.0e300)) {
                        exit_code = FX_EXIT_NAN;
                        done = true;
                        break;
                    }
                }
            }
        }

        // ---- K6: write back scale * x (:161-166); later components see the pre-solve snapshot (quirk Q2)
        __syncthreads();
        for (uint32_t i = lane; i < nfree; i += 64) {
            uint32_t vi = fidx[i];
            double x = XS[cur * vt + vi];
            double xo = (prm.mode & 1u) ? scale * x : x;
            b.vars[v0 + vi] = xo;
            VOUT[vi] = xo;
        }
        __syncthreads();
        // restore the perturbed start values: recomputed exactly as above (same draws)
        {
            const uint32_t rng0 = rng_start;
            for (uint32_t i = lane; i < nfree; i += 64) {
                uint32_t vi = fidx[i];
                double x = b.vars0[v0 + vi];
                if (prm.mode & 1u) x = x * scale_recip;
                if (prm.mode & 2u) {
                    uint32_t st = lcg_jump(rng0, 2u * i);
                    st = st * 1664525u + 1013904223u;
                    double f1 = (1.0 / 4294967295.0) * (double)st;
                    st = st * 1664525u + 1013904223u;
                    double f2 = (1.0 / 4294967295.0) * (double)st;
                    x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                }
                XS[vi] = x;
                XS[vt + vi] = x;
            }
        }
        __syncthreads();
        tot_accept += accepted;
        tot_trials += trials;
        last_exit = exit_code;
        tot_sse0 += sse_start;
        tot_sse += sse;
        comps_done += 1;
    }

    // ---- post-solve check on unscaled variables (constraints/mod.rs:96-109)
    __syncthreads();
    double part = 0.0;
    for (uint32_t i = lane; i < net; i += 64) {
        int tag = b.expr_tag[e0 + i] & 0x7F;
        const uint16_t* f = b.expr_idx + 4 * (size_t)(e0 + i);
        uint16_t ff[4] = {f[0], f[1], f[2], f[3]};
        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        expand_vars(tag, ff, vars8);
        double v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = VOUT[vars8[e]];
        double r = eval_expression<double, false, QR>(tag, v, b.expr_param[e0 + i], g);
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
    if (prm.prof) {
        stamp(5);
        if (lane == 0)
            for (int i = 0; i < 6; ++i) atomicAdd(&prm.prof[i], ph[i]);
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_solve_wide_kernel(DeviceBatch b, LmParams prm, WideLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    wide_body<false>(b, prm, L, smem);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_solve_wide_qr_kernel(DeviceBatch b, LmParams prm, WideLayout L) {
    extern __shared__ __align__(16) unsigned char smem[];
    wide_body<true>(b, prm, L, smem);
}

hipError_t launch_solve_wide_qr(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    const QrPlans& Q = b.qr_none;
    if (Q.n_qrw == 0) return hipSuccess;
    if (p.lm.precision == 32 || (p.mode & (MODE_UNITS | MODE_LBFGS)) || p.prof) return hipErrorInvalidValue;
    WideLayout L = make_wide_layout(Q.qrw_free, Q.qrw_vars, Q.qrw_rows, Q.qrw_nx);
    if (L.total > 160u * 1024u) return hipErrorInvalidValue;
    static unsigned int raised = 0;
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&lm_solve_wide_qr_kernel), &raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lm_solve_wide_qr_kernel, dim3(Q.n_qrw), dim3(64), L.total, stream, b, p, L);
    return hipGetLastError();
}
// LDS bytes of that launch (the host checks them before it plans)
size_t wide_qr_lds_bytes(uint32_t max_free, uint32_t max_vars, uint32_t max_rows, uint32_t nx) { return make_wide_layout(max_free, max_vars, max_rows, nx).total; }

hipError_t launch_solve_wide(const DeviceBatch& b, const LmParams& p, hipStream_t stream) {
    if (b.n_wide == 0) return hipSuccess;
    WideLayout L = make_wide_layout(b.w_max_free, b.w_max_vars, b.w_max_rows);
    if (L.total > 160u * 1024u) return hipErrorInvalidValue;
    static unsigned int raised = 0;
    hipError_t e = raise_lds_limit_once(reinterpret_cast<const void*>(&lm_solve_wide_kernel), &raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lm_solve_wide_kernel, dim3(b.n_wide), dim3(64), L.total, stream, b, p, L);
    return hipGetLastError();
}

}  // namespace fx
