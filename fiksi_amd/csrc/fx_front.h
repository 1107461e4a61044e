// Multifrontal build of the large-component path — included by fx_sparse.hip inside its anonymous namespace, after
// fx_sparse_team.h (whose row evaluation, sums, LM state and value slabs it shares). Host side: fx_front_plan.h.
//
// The walkers of fx_sparse_team.h eliminate ONE column at a time, a wavefront per column: some 150 instructions issued in
// order plus three or four LDS round trips, 0.6 - 1.1 us per column, and cfg2's critical path is ~60 columns long (the
// round-4 counters: VALU active 7 % of a wavefront's lifetime). Here the unit of work is a FRONT — a connected piece of the
// elimination tree and the rows its columns reach, at most 15 columns — and a front lives in ONE ROW OF 16 LANES: lane c
// holds column c of the front's dense symmetric matrix in 16 registers, the right-hand side rides along as one more
// column, and the pivots are eliminated by the DPP-row Cholesky of the grouped kernels (fx_grouped_rows.h: one
// v_fmac_f64_dpp per row update, the pivot column broadcast inside the instruction), four fronts per wavefront, ~70 ns
// per pivot. After the last pivot the boundary's lanes hold the Schur complement — the front's contribution to its
// parent, passed on through LDS (through global memory from a part's root to the top) — and the right-hand side's lane
// holds y of the pivots (forward substitution came along) and the updated right-hand side of the boundary. The backward
// sweep walks the same fronts top-down with the same registers: boundary lanes carry the x already known.
// A front is assembled in a staging tile in LDS (16 x 17 doubles): the entries of A of its pivot columns by record, lambda
// on the pivots' diagonal, its children's contribution blocks one after the other in a fixed order (deterministic sums).
//   * mf_lm_solo_kernel: one workgroup per System runs the whole Levenberg-Marquardt loop (lm.rs:108-191) in ONE launch;
//   * mf_parts_up_kernel / mf_parts_down_kernel: a large System alone (cfg2) — the tree's parts side by side, one workgroup
//     each, the workgroup that finishes last goes on with the top; two launches per trial.
// What this replaces in the reference: Qr::factorize + Qr::solve_mut per trial (solvi/src/decomposition/sparse/qr.rs:281-356)
// in its normal-equation form (lm.rs:28-63), as every FX_STEP_CHOLESKY path does.

constexpr int MF_NW = 8;                       // wavefronts of a multifrontal workgroup
constexpr int MF_THREADS = 64 * MF_NW;
constexpr uint32_t MF_ROWS = 4u * MF_NW;       // fronts in flight per pass (a tile each)
using sparse_plan::MF_N;
using sparse_plan::MF_TILE;
using sparse_plan::MF_TS;
using sparse_plan::MF_LS;
using sparse_plan::MF_FRONT_WORDS;
using sparse_plan::MF_U_GLOBAL;
using sparse_plan::MF_FMAX;

struct MfSeg {  // a segment blob in LDS (fx_front_plan.h)
    const uint32_t *lev, *fr, *recs, *cols, *kids;
    uint32_t nlev, na, nc, a0, c0, l_doubles, u_doubles, widest;
};
__device__ __forceinline__ MfSeg mf_views(const uint32_t* w) {
    MfSeg s;
    s.nlev = w[1];
    s.na = w[2];
    s.nc = w[3];
    s.a0 = w[4];
    s.c0 = w[5];
    s.lev = w + w[6];
    s.fr = w + w[7];
    s.recs = w + w[8];
    s.cols = w + w[9];
    s.kids = w + w[10];
    s.l_doubles = w[11];
    s.u_doubles = w[12];
    s.widest = w[14];
    return s;
}
__device__ __forceinline__ void mf_load_blob(const uint32_t* __restrict__ g, uint32_t* s, uint32_t nwords) {
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* s4 = reinterpret_cast<uint4*>(s);
    for (uint32_t i = threadIdx.x; i < nwords / 4u; i += MF_THREADS) s4[i] = g4[i];
}

// The entries of A = Jt J of one segment and its share of -Jt r, into LDS (segment-local numbering): deterministic gathers,
// a thread per entry (fx_sparse_team.h: team_form_segment); long lists by a wavefront each. keep_a / keep_rhs (global, the
// factor's numbering; may be null): where a rejected trial finds them again when LDS does not outlive the launch.
__device__ __forceinline__ void mf_form(const SpBlock& B, const double* jc, const double* rc, const MfSeg& sg, double* s_a, double* s_rhs,
                                        double* keep_a, double* keep_rhs, bool need_form) {
    for (uint32_t k = threadIdx.x; k < sg.na; k += MF_THREADS) {
        const uint32_t ai = sg.a0 + k;
        double s = 0.0;
        if (need_form) {
            const uint32_t pb = B.apair_ptr[ai], pe = B.apair_ptr[ai + 1];
            if (pe - pb > FORM_LONG) continue;
            for (uint32_t p = pb; p < pe; ++p) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
            if (keep_a) keep_a[ai] = s;
        } else {
            s = keep_a[ai];
        }
        s_a[k] = s;
    }
    for (uint32_t c = threadIdx.x; c < sg.nc; c += MF_THREADS) {
        const uint32_t col = sg.c0 + c;
        double s = 0.0;
        if (need_form) {
            const uint32_t pb = B.cptr[col], pe = B.cptr[col + 1];
            if (pe - pb > FORM_LONG) continue;
            for (uint32_t p = pb; p < pe; ++p) s += jc[B.cidx[p]] * -rc[B.crow[p]];
            if (keep_rhs) keep_rhs[col] = s;
        } else {
            s = keep_rhs[col];
        }
        s_rhs[c] = s;
    }
    if (!need_form) return;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x >> 6; i < B.n_along + B.n_clong; i += MF_NW) {
        double s = 0.0;
        if (i < B.n_along) {
            const uint32_t k = B.along[i];
            if (k < sg.a0 || k >= sg.a0 + sg.na) continue;
            for (uint32_t p = B.apair_ptr[k] + lane; p < B.apair_ptr[k + 1]; p += 64) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
            s = wave_sum64(s);
            if (lane == 0) {
                if (keep_a) keep_a[k] = s;
                s_a[k - sg.a0] = s;
            }
        } else {
            const uint32_t c = B.along[i];
            if (c < sg.c0 || c >= sg.c0 + sg.nc) continue;
            for (uint32_t p = B.cptr[c] + lane; p < B.cptr[c + 1]; p += 64) s += jc[B.cidx[p]] * -rc[B.crow[p]];
            s = wave_sum64(s);
            if (lane == 0) {
                if (keep_rhs) keep_rhs[c] = s;
                s_rhs[c - sg.c0] = s;
            }
        }
    }
}

// one pivot step of a front's row (fx_grouped_rows.h: RStep<1, double, K>::factor, stopped per ROW: `act` = the row's
// front still has a pivot K — a row past its pivots multiplies by zero and keeps its registers)
template <int K>
__device__ __forceinline__ void mf_pivot(double (&a)[MF_N], double& invd, bool& bad, int hl, int npiv) {
    const bool act = K < npiv;
    const double pv = rbcast<K>(a[K]);
    const double piv = act ? pv : 1.0;
    bad = bad || !(piv > 0.0) || !(piv < 1.0e300);
    const double rs = rsqrt_refined(piv);
    const double ip = rs * rs;
    const double ljk = a[K] * rs;
    double mul = (act && hl > K) ? a[K] * ip : 0.0;
    if (act && hl >= K) a[K] = ljk;
    if (act && hl == K) invd = rs;
    if constexpr (K + 1 < (int)MF_N) {
        dpp_settle(mul);
#pragma unroll
        for (int i = K + 1; i < (int)MF_N; ++i) fnma_rbcast_self<K>(a[i], mul);
        asm volatile("s_nop 1");
    }
}
// the pivots in blocks of four, one test per block and the blocks nested (kmax: the most pivots of the wavefront's four fronts,
// wave-uniform): a test per pivot made the compiler copy all sixteen registers of a lane at every join
template <int K0>
__device__ __forceinline__ void mf_factor_block(double (&a)[MF_N], double& invd, bool& bad, int hl, int npiv) {
    mf_pivot<K0>(a, invd, bad, hl, npiv);
    if constexpr (K0 + 1 < (int)MF_N - 1) mf_pivot<K0 + 1>(a, invd, bad, hl, npiv);
    if constexpr (K0 + 2 < (int)MF_N - 1) mf_pivot<K0 + 2>(a, invd, bad, hl, npiv);
    if constexpr (K0 + 3 < (int)MF_N - 1) mf_pivot<K0 + 3>(a, invd, bad, hl, npiv);
}
template <int K0>
__device__ __forceinline__ void mf_factor_from(double (&a)[MF_N], double& invd, bool& bad, int hl, int npiv, int kmax) {
    if constexpr (K0 < (int)MF_N - 1) {
        if (K0 < kmax) {
            mf_factor_block<K0>(a, invd, bad, hl, npiv);
            mf_factor_from<K0 + 4>(a, invd, bad, hl, npiv, kmax);
        }
    }
}
// one step of the backward sweep: x_K (lane K: a pivot's acc / d^2, a boundary column's known x) off the lanes below it
template <int K>
__device__ __forceinline__ void mf_back(const double (&a)[MF_N], double& acc, double invd2, int hl) {
    double t = acc * invd2;
    dpp_settle(t);
    const double w = hl < K ? a[K] : 0.0;
    fnma_rbcast<K>(acc, t, w);
}
// the steps MF_N - 1 ... 1, skipping from the top what lies past the widest front of the wavefront (fmax, wave-uniform), four
// steps per test
template <int K>
__device__ __forceinline__ void mf_backward_from(const double (&a)[MF_N], double& acc, double invd2, int hl, int fmax) {
    if constexpr (K > 0) {
        if (K - 3 < fmax) {  // (some step of K, K - 1, K - 2, K - 3 is inside: the ones past fmax find zeros and change nothing)
            mf_back<K>(a, acc, invd2, hl);
            if constexpr (K - 1 > 0) mf_back<K - 1>(a, acc, invd2, hl);
            if constexpr (K - 2 > 0) mf_back<K - 2>(a, acc, invd2, hl);
            if constexpr (K - 3 > 0) mf_back<K - 3>(a, acc, invd2, hl);
        }
        mf_backward_from<K - 4>(a, acc, invd2, hl, fmax);
    }
}

// The children's contribution blocks into a front's tile: W lanes per child (lane c of the group: column c of the child's block;
// the block's last column is the right-hand side), 16 / W children per round, child after child in the order of the list. A
// column's W - 1 rows are in flight at once (a part's root hands its block over through global memory: one round trip per
// round of children, not one per value), then the additions (ds_add_f64: no read-back) — rows past the block go to the tile's
// spare row (fx_front_plan.h: the map's padding).
template <int W>
__device__ __forceinline__ void mf_add_children(const uint32_t* kids, uint32_t nch, const double* u_loc, const double* u_glob, double* tile, uint32_t ts, int hl, int F) {
    constexpr int G = (int)MF_N / W;  // children per round
    const int g = hl / W, lc = hl % W;
    for (uint32_t c0 = 0; c0 < nch; c0 += (uint32_t)G) {
        const uint32_t c = c0 + (uint32_t)g;
        if (c < nch) {
            const uint32_t* kp = kids + (size_t)c * sparse_plan::MF_CHILD_WORDS;
            const uint32_t uo = kp[0], nb = kp[1];
            const uint4 m4 = *reinterpret_cast<const uint4*>(kp + 2);
            if ((uint32_t)lc <= nb) {
                const uint32_t mw[4] = {m4.x, m4.y, m4.z, m4.w};
                double uv[W - 1];
                if (uo >> 31) {
                    const double* U = u_glob + (uo & 0x7FFFFFFFu) + (uint32_t)lc * MF_LS;
#pragma unroll
                    for (int r = 0; r < W - 1; ++r) uv[r] = U[r];
                } else {
                    const double* U = u_loc + uo + (uint32_t)lc * MF_LS;
#pragma unroll
                    for (int r = 0; r < W - 1; ++r) uv[r] = U[r];
                }
                const uint32_t wsel = lc < 4 ? mw[0] : lc < 8 ? mw[1] : lc < 12 ? mw[2] : mw[3];
                const uint32_t mc = (uint32_t)lc < nb ? (wsel >> (8 * (lc & 3))) & 0xFFu : (uint32_t)F;  // (the block's last column: the right-hand side)
                double* tcol = tile + mc;
#pragma unroll
                for (int r = 0; r < W - 1; ++r) lds_add(&tcol[((mw[r / 4] >> (8 * (r % 4))) & 0xFFu) * ts], uv[r]);
            }
        }
        group_sync();
    }
}

// The sweep up of one segment by the calling workgroup: its fronts level by level, a row of 16 lanes per front.
//   s_a / s_rhs: the segment's entries of A and of the right-hand side (LDS); l: the fronts' L blocks (LDS, or global memory
//   when the sweep down is another launch or needs the room); u_loc / u_glob: contribution blocks (fx_front_plan.h: layout);
//   tiles: `rows` staging tiles (LDS; rows: a multiple of 4, at most MF_ROWS). False when a pivot was not positive.
__device__ __forceinline__ bool mf_sweep_up(const MfSeg& sg, const double* s_a, const double* s_rhs, double lambda, double* l, double* u_loc,
                                            double* u_glob, double* tiles, uint32_t rows, unsigned long long* prof = nullptr) {
    // prof (diagnostics; else null): thread 0's 100 MHz ticks by phase: tile + entries of A, children, registers + pivots, stores, barrier
    const bool stamp = prof && threadIdx.x == 0;
    unsigned long long t_prev = stamp ? wall_clock64() : 0ull;
    auto mark = [&](int slot) {
        if (stamp) {
            const unsigned long long now = wall_clock64();
            prof[slot] += now - t_prev;
            t_prev = now;
        }
    };
    const int lane = threadIdx.x & 63, hl = lane & 15;
    const uint32_t rid = (threadIdx.x >> 4);  // this lane row's place among the workgroup's rows
    constexpr uint32_t ts = MF_TS;
    double* const tile = tiles + (size_t)(rid < rows ? rid : 0u) * MF_TILE;
    bool bad = false;
    for (uint32_t q = 0; q < sg.nlev; ++q) {
        const uint32_t f0 = sg.lev[q], f1 = sg.lev[q + 1];
        for (uint32_t base = f0; base < f1; base += rows) {
            if ((threadIdx.x >> 6) * 4u >= rows) break;  // (wave-uniform: a wavefront without tiles sits the level out)
            const uint32_t fi = base + rid;
            const bool on = rid < rows && fi < f1;
            const uint32_t* d = sg.fr + (size_t)(on ? fi : f0) * MF_FRONT_WORDS;
            const uint32_t w0 = on ? d[0] : 0u;
            const int npiv = (int)(w0 & 0xFFu), nbnd = (int)((w0 >> 8) & 0xFFu), F = npiv + nbnd;
            const uint32_t nch = (w0 >> 16) & 0xFFu, flags = w0 >> 24;
            const uint32_t* fc = sg.cols + d[3];
            // ---- the staging tile: zero, the entries of A (both triangles), lambda and the right-hand side of the pivots
            if (on) {
#pragma unroll
                for (int k = 0; k < (int)((MF_TILE + MF_N - 1) / MF_N); ++k)
                    if (k * (int)MF_N + hl < (int)MF_TILE) tile[hl + (int)MF_N * k] = 0.0;
            }
            group_sync();
            if (on) {
                const uint32_t nrec = d[2];
                const uint32_t* rp = sg.recs + d[1];
                for (uint32_t t0 = 0; t0 < nrec; t0 += 4u * MF_N) {  // (four records per lane and round: the two LDS trips of a record overlap)
                    uint32_t r[4];
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t t = t0 + (uint32_t)(u * (int)MF_N + hl);
                        r[u] = t < nrec ? rp[t] : 0xFFFFFFFFu;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = r[u] != 0xFFFFFFFFu ? s_a[r[u] >> 8] : 0.0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r[u] != 0xFFFFFFFFu) {
                            const uint32_t li = r[u] & 15u, lj = (r[u] >> 4) & 15u;
                            tile[li * ts + lj] = v[u];
                            tile[lj * ts + li] = v[u];
                        }
                }
            }
            group_sync();
            if (on && hl < npiv) {
                lds_add(&tile[(uint32_t)hl * ts + (uint32_t)hl], lambda);
                tile[(uint32_t)hl * ts + (uint32_t)F] = s_rhs[fc[hl] - sg.c0];
            }
            group_sync();
            mark(0);
            // ---- the children's contributions, one child after the other (a fixed order: LDS executes a wavefront's additions
            // in program order, so a tile entry gets the same sum every time); lane c adds column c of the child's block: its
            // fifteen rows in flight at once (a part's root hands its block over through global memory: one round trip per child,
            // not one per value), then the additions (ds_add_f64: no read-back) — rows past the block go to the tile's spare row
            // Children with small blocks go several at a time: a group of W lanes per child (W = 4 / 8 / 16 by the widest block), the
            // additions of one instruction into one entry arrive in lane order — the same order every time.
            if (on) {
                if (flags & sparse_plan::MF_KIDS_W4) mf_add_children<4>(sg.kids + d[4], nch, u_loc, u_glob, tile, ts, hl, F);
                else if (flags & sparse_plan::MF_KIDS_W8) mf_add_children<8>(sg.kids + d[4], nch, u_loc, u_glob, tile, ts, hl, F);
                else mf_add_children<16>(sg.kids + d[4], nch, u_loc, u_glob, tile, ts, hl, F);
            }
            group_sync();
            mark(1);
            // ---- registers: lane = column; the pivots; then every lane up to the right-hand side's stores its sixteen registers with one
            // address and immediate offsets (fx_front_plan.h: what no test keeps out lands in padding)
            double a[MF_N];
#pragma unroll
            for (int i = 0; i < (int)MF_N; ++i) a[i] = tile[i * (int)MF_TS + hl];
            double invd = 1.0;
            int kmax = __builtin_amdgcn_readlane(npiv, 0);
            kmax = max(kmax, __builtin_amdgcn_readlane(npiv, 16));
            kmax = max(kmax, __builtin_amdgcn_readlane(npiv, 32));
            kmax = max(kmax, __builtin_amdgcn_readlane(npiv, 48));
            mf_factor_from<0>(a, invd, bad, hl, npiv, kmax);
            mark(2);
            if (on) {
                if (hl < npiv || hl == F) {  // the pivots' columns of L, and behind them the right-hand side's lane (its rows above npiv: y)
                    double* lp = l + d[5] + (uint32_t)(hl < npiv ? hl : npiv) * MF_LS;
#pragma unroll
                    for (int i = 0; i < (int)MF_N; ++i) lp[i] = a[i];
                    if (hl < npiv) lp[hl] = invd;  // (the diagonal slot: 1 / d)
                }
                if (hl >= npiv && hl <= F && nbnd) {
                    const int off = (hl - npiv) * (int)MF_LS - npiv;
                    if (flags & MF_U_GLOBAL) {
                        double* U = u_glob + d[6] + off;
#pragma unroll
                        for (int i = 0; i < (int)MF_N; ++i) U[i] = a[i];
                    } else {
                        double* U = u_loc + d[6] + off;
#pragma unroll
                        for (int i = 0; i < (int)MF_N; ++i) U[i] = a[i];
                    }
                }
            }
            group_sync();
            mark(3);
        }
        __syncthreads();
        mark(4);
    }
    return !bad;
}

// ... and down, parents before children: x of the segment's columns into s_x. x_out: x of the columns of other segments a
// boundary reaches (global memory, the factor's numbering: a part's fronts end in the top's columns).
__device__ __forceinline__ void mf_sweep_down(const MfSeg& sg, const double* l, double* s_x, const double* x_out, uint32_t rows) {
    const int lane = threadIdx.x & 63, hl = lane & 15;
    const uint32_t rid = (threadIdx.x >> 4);
    for (uint32_t q = sg.nlev; q-- > 0;) {
        const uint32_t f0 = sg.lev[q], f1 = sg.lev[q + 1];
        for (uint32_t base = f0; base < f1; base += rows) {
            if ((threadIdx.x >> 6) * 4u >= rows) break;
            const uint32_t fi = base + rid;
            const bool on = rid < rows && fi < f1;
            const uint32_t* d = sg.fr + (size_t)(on ? fi : f0) * MF_FRONT_WORDS;
            const uint32_t w0 = on ? d[0] : 0u;
            const int npiv = (int)(w0 & 0xFFu), F = npiv + (int)((w0 >> 8) & 0xFFu);
            const uint32_t* fc = sg.cols + d[3];
            double a[MF_N];
#pragma unroll
            for (int i = 0; i < (int)MF_N; ++i) a[i] = 0.0;
            double acc = 0.0, invd2 = 1.0;
            uint32_t col = 0;
            if (on && hl < F) col = fc[hl];
            if (on && hl < npiv) {
                const double* lp = l + d[5] + (uint32_t)hl * MF_LS;
#pragma unroll
                for (int i = 0; i < (int)MF_N; ++i) a[i] = lp[i];  // (a[hl] is the diagonal slot: no step reads it)
                const double ad = lp[hl];                           // ... it holds 1 / d
                acc = l[d[5] + (uint32_t)npiv * MF_LS + (uint32_t)hl] / ad;  // y of this pivot: the right-hand side's lane, behind the columns
                invd2 = ad * ad;
            } else if (on && hl < F) {
                acc = (col >= sg.c0 && col < sg.c0 + sg.nc) ? s_x[col - sg.c0] : x_out[col];
            }
            int fmax = __builtin_amdgcn_readlane(F, 0);
            fmax = max(fmax, __builtin_amdgcn_readlane(F, 16));
            fmax = max(fmax, __builtin_amdgcn_readlane(F, 32));
            fmax = max(fmax, __builtin_amdgcn_readlane(F, 48));
            mf_backward_from<(int)MF_N - 1>(a, acc, invd2, hl, fmax);
            if (on && hl < npiv) s_x[col - sg.c0] = acc * invd2;
            group_sync();
        }
        __syncthreads();
    }
}

// tiles a launch needs: a multiple of four rows, at most MF_ROWS, enough for the widest level when that is less
__host__ __device__ inline uint32_t mf_tile_rows(uint32_t widest) {
    const uint32_t r = (widest + 3u) & ~3u;
    return r < 4u ? 4u : (r > MF_ROWS ? MF_ROWS : r);
}

// ---- the whole LM loop of one block, one workgroup per System (fx_sparse_team.h: sp_lm_team_kernel with the walkers
// replaced by the fronts; the entries of A stay in LDS from an accepted step to the next) --------------------------------
// dynamic LDS, doubles: [blob][s_a: na][s_rhs: nc][s_x: nc][contribution slots + 16][tiles][L][sums: red_n]
// MODE 1: the L blocks in global memory instead (a factor of some hundred columns: they would not fit beside the rest) — stores on
// the way up, one trip per level on the way down; MODE 2: the contribution blocks too (one more trip per level on the way up) —
// LDS then keeps enough staging tiles for a level's fronts to go side by side.
template <bool POSE, int MODE>
__global__ __launch_bounds__(MF_THREADS) void mf_lm_solo_kernel(SpRows rows, SpBlock B, SpVals V, SpAccum* __restrict__ accum, fx_lm_opts o,
                                                                uint32_t flags, double* __restrict__ vars_base, const uint64_t* __restrict__ out_off,
                                                                const uint32_t* __restrict__ blob, uint32_t blob_words, uint32_t red_n, uint32_t trows,
                                                                double* __restrict__ u_glob_base, size_t u_stride, unsigned long long* prof,
                                                                fx_result* __restrict__ results, uint32_t n_group) {
    constexpr bool LG = MODE >= 1, UG = MODE >= 2;
    // prof (diagnostics, FIKSI_AMD_TEAM_PROF=1; else null): workgroup 0 adds up the 100 MHz ticks of its phases
    const bool stamp = prof && blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long t_prev = stamp ? wall_clock64() : 0ull;
    auto mark = [&](int slot) {
        if (stamp) {
            const unsigned long long now = wall_clock64();
            prof[slot] += now - t_prev;
            t_prev = now;
        }
    };
    extern __shared__ double s_dyn[];
    __shared__ uint32_t s_bad;
    const uint32_t sys = blockIdx.x, tid = threadIdx.x;
    V.shift(sys);
    const double* sparam = rows.sparam + (size_t)sys * V.stride;
    const uint32_t m = B.m, nv = B.nv;
    uint32_t* s_blob = reinterpret_cast<uint32_t*>(s_dyn);
    mf_load_blob(blob, s_blob, blob_words);
    __syncthreads();
    const MfSeg sg = mf_views(s_blob);
    double* p = s_dyn + (blob_words + 1u) / 2u;
    double* const s_a = p;      p += (sg.na + 1u) & ~1u;
    double* const s_rhs = p;    p += (sg.nc + 1u) & ~1u;
    double* const s_x = p;      p += (sg.nc + 1u) & ~1u;
    double* const s_u = UG ? u_glob_base + sys * u_stride : p;
    if (!UG) p += ((sg.u_doubles + MF_LS + 1u) & ~1u);
    double* const tiles = p;    p += (size_t)trows * MF_TILE + (trows * MF_TILE & 1u);
    double* const s_l = LG ? V.l : p;
    if (!LG) p += sg.l_doubles;
    double* const s_red = p;

    for (uint32_t row = tid; row < m; row += MF_THREADS) team_eval_row<POSE>(rows, sparam, B.jac, row, V.xs0, V.r0, V.j0);
    __syncthreads();
    SpLm st;
    lm_state_init(st, team_sumsq<MF_NW>(V.r0, m, s_red, red_n), o);
    mark(0);
    while (!st.done) {
        const double* jc = st.cur ? V.j1 : V.j0;
        const double* rc = st.cur ? V.r1 : V.r0;
        if (st.need_form) mf_form(B, jc, rc, sg, s_a, s_rhs, nullptr, nullptr, true);  // (a rejected trial finds both in LDS)
        if (tid == 0) s_bad = 0;
        __syncthreads();
        mark(1);
        const bool ok = mf_sweep_up(sg, s_a, s_rhs, st.lambda, s_l, s_u, s_u, tiles, trows, stamp ? prof + 8 : nullptr);  // (one segment: no block goes through global memory)
        if (!ok && (tid & 63) == 0) atomicOr(&s_bad, 1u);
        __syncthreads();
        st.flag = s_bad;
        __syncthreads();
        mark(2);
        if (!st.flag) {
            if (LG) {  // (the L blocks have left for global memory; the same CU reads them back through its L1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            mf_sweep_down(sg, s_l, s_x, V.delta, trows);
            mark(3);
            st.dn2 = team_sumsq<MF_NW>(s_x, nv, s_red, red_n);
            const double* xc = st.cur ? V.xs1 : V.xs0;
            double* xt = st.cur ? V.xs0 : V.xs1;
            for (uint32_t k = tid; k < nv; k += MF_THREADS) {
                const uint32_t v = B.fvar[B.perm[k]];
                xt[v] = xc[v] + s_x[k];
            }
            __syncthreads();
            double* rt = st.cur ? V.r0 : V.r1;
            double* jt = st.cur ? V.j0 : V.j1;
            for (uint32_t row = tid; row < m; row += MF_THREADS) team_eval_row<POSE>(rows, sparam, B.jac, row, xt, rt, jt);
            __syncthreads();
            st.sse_t = team_sumsq<MF_NW>(rt, m, s_red, red_n);
            mark(5);
        }
        lm_state_control(st, o);
    }
    team_block_epilogue(B, V, st.cur, flags, vars_base + out_off[sys], tid, MF_THREADS);
    mark(6);
    if (stamp) prof[7] += st.trials;
    if (tid == 0) {
        SpAccum& ac = accum[sys];
        ac.accepted += st.accepted;
        ac.trials += st.trials;
        ac.exit_code = st.exit_code;
        ac.sse0 += st.sse_start;
        ac.sse += st.sse;
    }
    // The System's only block (results != null; every expression of the System is a row of it): the closing check on the unscaled
    // variables and the result record right here — spg_finish_kernel's statements, its 1 024 strided partial sums and its tree
    // (team_sumsq's narrow form), so the same bits — instead of one more launch (6 us of a lone 258-variable sketch's 154).
    if (results) {
        __syncthreads();  // (the epilogue's stores to the output slice)
        const double* x = vars_base + out_off[sys];
        const double* uparam = rows.param + (size_t)sys * V.stride;
        double* tmp = V.r0;
        for (uint32_t e = tid; e < rows.net; e += MF_THREADS) {
            const int tag = rows.tag[e] & 0x7F;
            const ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
            const uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
            uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            expand_vars<POSE>(tag, ff, vars8);
            double v[8], g[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = x[vars8[q]];
            tmp[e] = eval_expression<double, false, false, POSE>(tag, v, uparam[e], g);
        }
        __syncthreads();
        const double sse_u = team_sumsq<MF_NW>(tmp, rows.net, s_red, red_n);
        if (tid == 0) {
            const SpAccum ac = accum[sys];  // (this thread's own stores above)
            fx_result res{};
            res.accepted = ac.accepted;
            res.trials = ac.trials;
            res.exit = ac.exit_code;
            res.ncomp = ac.ncomp;
            res.scale = V.scal[0];
            res.sse0 = ac.sse0;
            res.sse = ac.sse;
            res.sse_unscaled = sse_u;
            results[out_off[2 * (size_t)n_group + sys]] = res;
        }
    }
}

// ---- a large System alone: the parts side by side, the last one to finish goes on with the top ---------------------------
// ... and the LAMBDA LADDER over whole launches. The trials that follow a rejected trial of lm.rs:114-190 read the same
// point, the same Jacobian and the same residuals and differ in lambda only (x reject_factor per reject, lm.rs:189): they are
// independent of each other, and a lone large System leaves most of the chip idle (cfg2: 64 parts on 256 CUs). So a launch
// runs `ranks` trials side by side — rank k with lambda * reject_factor^k, its own L blocks, step, trial point, residuals and
// Jacobian rows (grid z) —, and the decision reads the ranks' verdicts IN ORDER and applies the first one that is not a plain
// reject: the one the sequential loop would have met first (fx_grouped_rows.h: LadderCode; the grouped kernels' ladder across
// the rows of a wavefront, here across workgroups). Every counter, every lambda and every accepted point is the
// sequential loop's; cfg2's 89 trials (16 accepted) take 30-odd launches' worth of time.
constexpr uint32_t MF_MAX_RANKS = 8;
struct MfRank {  // what rank k's trial found
    double dn2, sse_t;
    uint32_t flag, pad;
};
struct MfLadder {
    uint32_t ranks;                 // trials a launch makes side by side (1: no ladder)
    double *xsx, *rx, *jx;          // the sets 2 ... ranks of (point, residuals, Jacobian rows) — sets 0 / 1 are SpVals' xs0 / xs1 ...; one
    size_t xs_step, r_step, j_step; // set is the current point's, rank k's trial goes to the k-th of the others
    double *lx, *dx, *gux;          // ranks 1 ... of (L blocks, step, the parts' roots' contribution blocks); rank 0's are SpVals' l / delta
    size_t l_step, d_step, gu_step; // and the launch's own contribution buffer
    MfRank* rk;                     // [Systems][MF_MAX_RANKS]
    uint32_t* tickets;              // [Systems][2 * MF_MAX_RANKS + 2]: per rank up / down, all ranks down
};
__device__ __forceinline__ uint32_t mf_set_of_rank(uint32_t cur, uint32_t k) { return k < cur ? k : k + 1u; }
struct MfSet {
    double *xs, *r, *j;
};
__device__ __forceinline__ MfSet mf_set(const SpVals& V, const MfLadder& Ld, size_t sys_off, uint32_t s) {
    MfSet o;
    if (s == 0u) o = {V.xs0, V.r0, V.j0};
    else if (s == 1u) o = {V.xs1, V.r1, V.j1};
    else o = {Ld.xsx + sys_off + (s - 2u) * Ld.xs_step, Ld.rx + sys_off + (s - 2u) * Ld.r_step, Ld.jx + sys_off + (s - 2u) * Ld.j_step};
    return o;
}

struct MfParts {
    const uint32_t* blobs;      // the segments' blobs
    const uint32_t* blob_off;   // [nseg + 1]
    const uint32_t* seg_l;      // [nseg + 1] first double of each segment's L storage (global memory: the sweep down is another launch)
    const uint32_t *erow_ptr, *erows;  // the block's rows by segment (fx_sparse_plan.h: PartsExtra)
    uint32_t nparts, top_rows, part_rows;  // tiles of the top's / a part's sweeps
    unsigned long long* prof;              // diagnostics (FIKSI_AMD_TEAM_PROF; else null): 100 MHz ticks of the phases, summed over the launches
};

// the ticket of a workgroup that is done with its share: true for the one that takes the last of `of` (it has acquired what the
// others released, and has put the counter back)
__device__ __forceinline__ bool mf_last_ticket(uint32_t* ticket, uint32_t of, uint32_t* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (t == of - 1u) ? 1u : 0u;
        if (*s_flag) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *ticket = 0;
        }
    }
    __syncthreads();
    return *s_flag != 0u;
}

// the top of the tree for one rank, by the workgroup that finished its part last
template <bool POSE>
__device__ __forceinline__ void mf_top_body(const SpRows& rows, const SpBlock& B, const MfParts& X, const SpVals& V, double lambda, bool need_form,
                                            const double* jc, const double* rc, const double* xc, const MfSet& trial, double* gl, double* delta,
                                            double* u_glob, uint32_t* flagp, double* s_dyn, uint32_t* s_badp, bool stamp) {
    if (*flagp) return;  // (a part met a bad pivot)
    const uint32_t top = X.nparts, tid = threadIdx.x;
    uint32_t* s_blob = reinterpret_cast<uint32_t*>(s_dyn);
    const uint32_t words = X.blob_off[top + 1] - X.blob_off[top];
    mf_load_blob(X.blobs + X.blob_off[top], s_blob, words);
    __syncthreads();
    const MfSeg sg = mf_views(s_blob);
    // up: [blob][s_a][s_rhs][contribution slots][tiles], the L blocks go to global memory; down: [blob][s_x][L], copied back in
    double* p = s_dyn + (words + 1u) / 2u;
    double* const s_a = p;      p += (sg.na + 1u) & ~1u;
    double* const s_rhs = p;    p += (sg.nc + 1u) & ~1u;
    double* const s_u = p;      p += ((sg.u_doubles + MF_LS + 1u) & ~1u);
    double* const tiles = p;
    double* const s_x = s_dyn + (words + 1u) / 2u;
    double* const s_l = s_x + ((sg.nc + 1u) & ~1u);
    unsigned long long t_prev = stamp ? wall_clock64() : 0ull;
    auto mark = [&](int slot) {
        if (stamp) {
            const unsigned long long now = wall_clock64();
            X.prof[slot] += now - t_prev;
            t_prev = now;
        }
    };
    mf_form(B, jc, rc, sg, s_a, s_rhs, V.a, V.rhs, need_form);
    if (tid == 0) *s_badp = 0;
    __syncthreads();
    mark(4);
    const bool ok = mf_sweep_up(sg, s_a, s_rhs, lambda, gl, s_u, u_glob, tiles, X.top_rows, stamp ? X.prof + 24 : nullptr);
    mark(5);
    if (!ok && (tid & 63) == 0) atomicOr(s_badp, 1u);
    __syncthreads();
    if (*s_badp) {
        if (tid == 0) *flagp = 1;
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the L blocks have left for global memory; the same CU reads them back through its L1)
    __syncthreads();
    for (uint32_t i = tid; i < sg.l_doubles; i += MF_THREADS) s_l[i] = gl[i];
    __syncthreads();
    mf_sweep_down(sg, s_l, s_x, delta, X.top_rows);
    mark(6);
    for (uint32_t c = tid; c < sg.nc; c += MF_THREADS) {  // the step of the top's columns, and their trial point
        const double dx = s_x[c];
        delta[sg.c0 + c] = dx;
        const uint32_t v = B.fvar[B.perm[sg.c0 + c]];
        trial.xs[v] = xc[v] + dx;
    }
    // (the top's own rows are evaluated by the parts' workgroups of the next launch, mf_parts_down_kernel: one workgroup going
    // through them here was 14 of an up launch's 50 us on cfg2)
}

// grid: (parts, Systems, ranks)
template <bool POSE>
__global__ __launch_bounds__(MF_THREADS) void mf_parts_up_kernel(SpRows rows, SpBlock B, MfParts X, SpVals V, MfLadder Ld, double* u_glob_base, size_t u_stride,
                                                                 SpLm* __restrict__ lm, fx_lm_opts o) {
    extern __shared__ double s_dyn[];
    __shared__ uint32_t s_last, s_bad;
    const SpLm* st = lm + blockIdx.y;
    if (st->done) return;
    const uint32_t part = blockIdx.x, tid = threadIdx.x, rank = blockIdx.z;
    if (st->trials + rank >= o.max_trials) return;  // (the sequential loop would have stopped at its cap before this trial: the decision says so)
    const size_t sys_off = (size_t)blockIdx.y * V.stride;
    V.shift(blockIdx.y);
    double lambda = st->lambda;
    for (uint32_t k = 0; k < rank; ++k) lambda *= o.reject_factor;  // (as the rejects in front of it would have multiplied it: lm.rs:189)
    MfRank* rk = Ld.rk + (size_t)blockIdx.y * MF_MAX_RANKS + rank;
    double* const gl_all = rank ? Ld.lx + sys_off + (rank - 1u) * Ld.l_step : V.l;
    double* const delta = rank ? Ld.dx + sys_off + (rank - 1u) * Ld.d_step : V.delta;
    double* const u_glob = rank ? Ld.gux + sys_off + (rank - 1u) * Ld.gu_step : u_glob_base + blockIdx.y * u_stride;
    const MfSet cur = mf_set(V, Ld, sys_off, st->cur);
    const MfSet trial = mf_set(V, Ld, sys_off, mf_set_of_rank(st->cur, rank));
    const bool stamp0 = X.prof && blockIdx.x == 0 && blockIdx.y == 0 && rank == 0 && tid == 0;
    unsigned long long t_prev = X.prof ? wall_clock64() : 0ull;
    auto mark = [&](bool who, int slot) {
        if (who) {
            const unsigned long long now = wall_clock64();
            X.prof[slot] += now - t_prev;
            t_prev = now;
        }
    };
    uint32_t* s_blob = reinterpret_cast<uint32_t*>(s_dyn);
    const uint32_t words = X.blob_off[part + 1] - X.blob_off[part];
    mf_load_blob(X.blobs + X.blob_off[part], s_blob, words);
    __syncthreads();
    const MfSeg sg = mf_views(s_blob);
    double* p = s_dyn + (words + 1u) / 2u;
    double* const s_a = p;      p += (sg.na + 1u) & ~1u;
    double* const s_rhs = p;    p += (sg.nc + 1u) & ~1u;
    double* const s_u = p;      p += ((sg.u_doubles + MF_LS + 1u) & ~1u);
    double* const tiles = p;
    mark(stamp0, 0);
    // (every rank forms the same entries of A from the same rows: the copies kept in global memory are written with the same bits)
    mf_form(B, cur.j, cur.r, sg, s_a, s_rhs, V.a, V.rhs, st->need_form != 0);
    __syncthreads();
    mark(stamp0, 1);
    // (the pivots' columns of L straight to global memory: the sweep down is the next launch)
    const bool ok = mf_sweep_up(sg, s_a, s_rhs, lambda, gl_all + X.seg_l[part], s_u, u_glob, tiles, X.part_rows, stamp0 ? X.prof + 16 : nullptr);
    mark(stamp0, 2);
    if (!ok && (tid & 63) == 0) atomicOr(&rk->flag, 1u);
    // the workgroup that finishes last goes on with the top (no launch boundary: every part publishes, the last one acquires)
    const bool last = mf_last_ticket(Ld.tickets + (size_t)blockIdx.y * (2 * MF_MAX_RANKS + 2) + rank, gridDim.x, &s_last);
    mark(stamp0, 3);
    if (!last) return;
    const bool stamp1 = X.prof && blockIdx.y == 0 && rank == 0 && tid == 0;
    if (stamp1) {
        X.prof[15] += 1;
        t_prev = wall_clock64();
    }
    mf_top_body<POSE>(rows, B, X, V, lambda, st->need_form != 0, cur.j, cur.r, cur.xs, trial, gl_all + X.seg_l[X.nparts], delta, u_glob, &rk->flag, s_dyn,
                      &s_bad, stamp1);
    mark(stamp1, 8);
}

// The ranks' verdicts in order (lm.rs:134-191 on every one of them; fx_sparse_team.h: lm_state_control is the one-rank case):
// the plain rejects in front of the first other verdict multiply lambda and count as trials, that verdict is applied, the ranks
// behind it never happened.
__device__ __forceinline__ void mf_ladder_control(SpLm& st, const MfRank* rk, uint32_t ranks, const fx_lm_opts& o) {
    st.need_form = 0;
    uint32_t kw = ranks;
    int code = LC_REJECT;
    double lam_k = st.lambda;
    for (uint32_t k = 0; k < ranks; ++k) {
        int c = LC_REJECT;
        if (st.trials + k >= o.max_trials) c = LC_CAP;
        else if (rk[k].flag) c = LC_SINGULAR;
        else if (!(rk[k].dn2 == rk[k].dn2)) c = LC_NAN;
        else if (rk[k].dn2 < o.step_tol) c = LC_STEP;
        else if (rk[k].sse_t < st.sse) c = LC_ACCEPT;
        else if (!(rk[k].sse_t == rk[k].sse_t) && !(lam_k * o.reject_factor < 1.0e300)) c = LC_REJ_NAN;
        if (c != LC_REJECT) {
            kw = k;
            code = c;
            break;
        }
        lam_k *= o.reject_factor;
    }
    for (uint32_t k = 0; k < kw; ++k) st.lambda *= o.reject_factor;  // the plain rejects in front (lm.rs:189)
    bool check_cap = true;
    if (kw == ranks) {
        st.trials += ranks;
    } else {
        st.trials += kw + (code != LC_CAP ? 1u : 0u);
        if (code == LC_CAP) {
            st.exit_code = FX_EXIT_TRIAL_CAP;
            st.done = 1;
            check_cap = false;
        } else if (code == LC_SINGULAR) {  // lm.rs:134-137
            st.lambda *= o.singular_factor;
            if (!(st.lambda < 1.0e300)) {
                st.exit_code = FX_EXIT_NAN;
                st.done = 1;
            }
        } else if (code == LC_NAN) {
            st.exit_code = FX_EXIT_NAN;
            st.done = 1;
            check_cap = false;
        } else if (code == LC_STEP) {  // lm.rs:139-142
            st.exit_code = FX_EXIT_STEP;
            st.done = 1;
            check_cap = false;
        } else if (code == LC_ACCEPT) {  // lm.rs:151-186
            const double sse_t = rk[kw].sse_t, sse = st.sse;
            double lam = st.lambda * o.accept_factor;
            if (lam < o.lambda_min) lam = o.lambda_min;
            st.lambda = lam;
            st.cur = mf_set_of_rank(st.cur, kw);
            st.accepted += 1;
            const double rel = (sse - sse_t) / sse;
            st.sse = sse_t;
            if (rel <= o.ftol) {
                st.exit_code = FX_EXIT_FTOL;
                st.done = 1;
                check_cap = false;
            } else {
                st.need_form = 1;
                st.outer += 1;
                if (st.outer >= o.max_outer) {
                    st.done = 1;  // exit_code is still FX_EXIT_MAX_OUTER
                    check_cap = false;
                } else if (sse_t < o.sse_tol) {
                    st.exit_code = FX_EXIT_SSE;
                    st.done = 1;
                    check_cap = false;
                }
            }
        } else {  // LC_REJ_NAN: rejected with a NaN trial point and lambda past 1e300 (lm_state_control's reject branch)
            st.lambda *= o.reject_factor;
            st.exit_code = FX_EXIT_NAN;
            st.done = 1;
            check_cap = false;
        }
    }
    if (check_cap && !st.done && st.trials >= o.max_trials) {
        st.exit_code = FX_EXIT_TRIAL_CAP;
        st.done = 1;
    }
    st.flag = 0;
}

// The parts' sweep down, their columns' trial point, the evaluation of their rows there — per rank; the block of a rank that
// finishes last sums that rank's step and residuals (fixed shapes: the same bits whoever it is), and the last of all the
// ranks takes the decision. grid: (parts, Systems, ranks)
template <bool POSE>
__global__ __launch_bounds__(MF_THREADS) void mf_parts_down_kernel(SpRows rows, SpBlock B, MfParts X, SpVals V, MfLadder Ld, SpLm* __restrict__ lm, fx_lm_opts o) {
    extern __shared__ double s_dyn[];
    __shared__ uint32_t s_last;
    SpLm* stg = lm + blockIdx.y;
    if (stg->done) return;
    const uint32_t part = blockIdx.x, tid = threadIdx.x, rank = blockIdx.z;
    const size_t sys_off = (size_t)blockIdx.y * V.stride;
    V.shift(blockIdx.y);
    MfRank* rk = Ld.rk + (size_t)blockIdx.y * MF_MAX_RANKS + rank;
    uint32_t* const tk = Ld.tickets + (size_t)blockIdx.y * (2 * MF_MAX_RANKS + 2);
    const bool ran = stg->trials + rank < o.max_trials;  // (mf_parts_up_kernel: a rank at the trial cap made no trial)
    const bool bad = !ran || rk->flag != 0;              // (a bad pivot: no step; the decision is still taken)
    const MfSet cur = mf_set(V, Ld, sys_off, stg->cur);
    const MfSet trial = mf_set(V, Ld, sys_off, mf_set_of_rank(stg->cur, rank));
    double* const delta = rank ? Ld.dx + sys_off + (rank - 1u) * Ld.d_step : V.delta;
    const double* const gl_all = rank ? Ld.lx + sys_off + (rank - 1u) * Ld.l_step : V.l;
    uint32_t* s_blob = reinterpret_cast<uint32_t*>(s_dyn);
    const uint32_t words = X.blob_off[part + 1] - X.blob_off[part];
    double* const s_red = s_dyn;  // (the last block's sums: after everything else in LDS is done with)
    if (!bad) {
        mf_load_blob(X.blobs + X.blob_off[part], s_blob, words);
        __syncthreads();
        const MfSeg sg = mf_views(s_blob);
        double* p = s_dyn + (words + 1u) / 2u;
        double* const s_x = p;      p += (sg.nc + 1u) & ~1u;
        double* const s_l = p;
        const double* gl = gl_all + X.seg_l[part];
        for (uint32_t i = tid; i < sg.l_doubles; i += MF_THREADS) s_l[i] = gl[i];
        __syncthreads();
        mf_sweep_down(sg, s_l, s_x, delta, X.part_rows);
        for (uint32_t c = tid; c < sg.nc; c += MF_THREADS) {
            const double dx = s_x[c];
            delta[sg.c0 + c] = dx;
            const uint32_t v = B.fvar[B.perm[sg.c0 + c]];
            trial.xs[v] = cur.xs[v] + dx;
        }
        __syncthreads();
        const double* sparam = rows.sparam + (size_t)blockIdx.y * V.stride;
        for (uint32_t q = X.erow_ptr[part] + tid; q < X.erow_ptr[part + 1]; q += MF_THREADS)
            team_eval_row<POSE>(rows, sparam, B.jac, X.erows[q], trial.xs, trial.r, trial.j);
        // ... and a share of the top's rows (they read the top's columns only, whose trial point the launch before has written)
        const uint32_t top = X.nparts;
        for (uint32_t q = X.erow_ptr[top] + part * MF_THREADS + tid; q < X.erow_ptr[top + 1]; q += gridDim.x * MF_THREADS)
            team_eval_row<POSE>(rows, sparam, B.jac, X.erows[q], trial.xs, trial.r, trial.j);
    }
    if (!mf_last_ticket(tk + MF_MAX_RANKS + rank, gridDim.x, &s_last)) return;
    if (!bad) {  // this rank's sums
        const double dn2 = team_sumsq<MF_NW>(delta, B.nv, s_red, 1024u);
        const double sse_t = team_sumsq<MF_NW>(trial.r, B.m, s_red, 1024u);
        if (tid == 0) {
            rk->dn2 = dn2;
            rk->sse_t = sse_t;
        }
    }
    if (!mf_last_ticket(tk + 2 * MF_MAX_RANKS, gridDim.z, &s_last)) return;
    if (tid == 0) {  // the last rank to finish: the decision, and the ranks' flags cleared for the next launch
        SpLm st = *stg;
        mf_ladder_control(st, Ld.rk + (size_t)blockIdx.y * MF_MAX_RANKS, gridDim.z, o);
        for (uint32_t k = 0; k < MF_MAX_RANKS; ++k) Ld.rk[(size_t)blockIdx.y * MF_MAX_RANKS + k].flag = 0;
        *stg = st;
    }
}

// after the last trial: the solved point into set 0, where the block's epilogue looks for it (fx_sparse_team.h: spt_block_end_kernel)
__global__ void mf_ladder_finish_kernel(SpVals V, MfLadder Ld, SpLm* __restrict__ lm, uint32_t nvt) {
    SpLm* st = lm + blockIdx.y;
    const uint32_t cur = st->cur;
    if (cur < 2u) return;
    const size_t sys_off = (size_t)blockIdx.y * V.stride;
    V.shift(blockIdx.y);
    const MfSet c = mf_set(V, Ld, sys_off, cur);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nvt; i += gridDim.x * blockDim.x) V.xs0[i] = c.xs[i];
}
__global__ void mf_ladder_finish_state_kernel(SpLm* __restrict__ lm, uint32_t n) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n && lm[s].cur >= 2u) lm[s].cur = 0u;
}
