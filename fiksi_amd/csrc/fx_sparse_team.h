// Workgroup ("team") kernels of the large-component path — included by fx_sparse.hip inside its anonymous namespace,
// after the structures and helpers they use (SpRows, SpJac, SpChol, SpRowsOfL, SpLm, lds_add_f64, wave_sum64, ...).
//
// Round 2 ran one host loop of tiny launches per large System: a level of the elimination tree per launch, ~30 launches
// per LM trial, each bound by its own launch latency (hinged triangles x 64: 1.6 k Systems/s, slower than one CPU core).
// Here the unit of work is a WORKGROUP of 16 wavefronts that owns a piece of the elimination tree (a "segment",
// fx_sparse_plan.h: TeamSchedule): its lists are walked by its wavefronts, a workgroup barrier ends a level, and what
// one wavefront stores the others read through the CU's own L1 — no launch boundary, no device-wide barrier.
//   * Systems whose factor is one segment (up to a few thousand columns): sp_lm_team_kernel — ONE launch runs the
//     whole Levenberg-Marquardt loop (lm.rs:108-191) of a block for every System of a batch that shares the structure,
//     one workgroup per System, control flow included; the host does not look at it until the end.
//   * larger Systems (cfg2): the tree is cut into parts (one workgroup each, side by side) and a top (one workgroup):
//     spt_* kernels, five launches per trial instead of thirty, LM control in the last block of the evaluation kernel.

constexpr int TEAM_THREADS = 1024;
constexpr int TEAM_NWAVES = TEAM_THREADS / 64;
static_assert(TEAM_NWAVES == (int)sparse_plan::TEAM_WAVES, "the schedules are built for this many wavefronts");

struct ColDesc {  // one column, as a wavefront meets it on its way through a list
    uint32_t j;             // the column
    uint32_t beg, end;      // its entries in L (the first one is the diagonal)
    uint32_t rbeg, rend;    // row j of L (strictly lower part)
    uint32_t pbeg, pend0;   // the products of its first 64 entries
    uint32_t pad;           // LDS build of the parts schedule: first entry whose row is in the top (the column's end elsewhere)
};

struct SpTeamSched {            // fx_sparse_plan.h: TeamSchedule on the device
    const uint32_t* seg_lev;    // [nseg + 1]
    const uint32_t* wptr;       // [nlev * 16 + 1]: wavefront w's columns of level q = cols[wptr[16 q + w] .. wptr[16 q + w + 1])
    const uint32_t* cols;       // [nv] walking order
    const ColDesc* cdesc;       // [nv] in the order of cols
    const uint32_t* cdesc_mid;  // (unused)
    uint32_t nparts;
};
constexpr uint32_t FORM_LONG = 32;  // gather lists of A / of the right-hand side beyond this are summed by a wavefront

struct SpBlock {  // structure of one block, shared by the Systems of a group
    SpJac jac;
    const uint32_t *fvar, *perm, *apair_ptr, *apairs, *cptr, *cidx, *crow, *jcol;
    SpChol chol;
    SpRowsOfL lrows;
    SpTeamSched sched;
    const uint32_t* a2l;     // [nnz_a] the entry of L an entry of A starts in
    const uint32_t* along;   // entries of A, then columns of the right-hand side, whose gather lists are long
    uint32_t n_along, n_clong;
    uint32_t m, nv, nnz_a, nnz_l;
};

// Value arrays of the group's first System; System s of the group lives `s * stride` doubles further in every one.
struct SpVals {
    double *xs0, *xs1, *snap, *r0, *r1, *j0, *j1, *a, *l, *rhs, *delta, *t, *e, *scal;
    double *hs, *hy;  // Optimizer::LBfgs: the five s / y history vectors (a, l, t, e are its gradient, direction, scratch)
    size_t stride;
    __device__ __forceinline__ void shift(uint32_t s) {
        const size_t o = (size_t)s * stride;
        xs0 += o; xs1 += o; snap += o; r0 += o; r1 += o; j0 += o; j1 += o; a += o; l += o;
        rhs += o; delta += o; t += o; e += o; scal += o; hs += o; hy += o;
    }
};

struct SpAccum {  // what a System's blocks add up to (fx_result)
    uint32_t accepted, trials, exit_code, ncomp;
    double sse0, sse;
};

constexpr uint32_t TEAM_REFINED = 1, TEAM_SINGLE_PASS = 2, TEAM_SCALE = 4;
constexpr uint32_t TEAM_PARTS_MAX_GROUP = 8;  // from this many Systems of a structure on, each one keeps to its own workgroup

// what one wavefront stored (global memory), its other lanes may load next: the stores have left (vmcnt) before any
// later load issues. Same CU, same L1 — nothing more is needed inside a workgroup.
__device__ __forceinline__ void wave_sync_mem() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// the same for LDS: a wavefront's LDS instructions execute in order, only the compiler has to keep them so — and the
// index loads of the next column stay in flight (no vmcnt wait)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool LDSV>
__device__ __forceinline__ void vals_sync() {
    if (LDSV) wave_sync_lds();
    else wave_sync_mem();
}

__device__ __forceinline__ double bcast_first(double s) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(s));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(s));
    return __hiloint2double(hi, lo);
}

// sum over the wavefront's lanes when only the first n hold something (n wave-uniform): one DPP row butterfly for n <= 16
__device__ __forceinline__ double wave_sum_first(double v, uint32_t n) {
    if (n <= 16u) {
        v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_move<0x141>(v);  // row_half_mirror
        v += dpp_move<0x140>(v);  // row_mirror
        return bcast(v, 0);
    }
    return wave_sum(v);
}

// ---- a wavefront's walk through a list ---------------------------------------------------------------------------
// A column is a chain of dependent steps (operands -> products -> pivot -> scaling -> the next column's operands), a
// few hundred cycles when its operands are in LDS — and 1.3 us when each column first fetches its own index words
// from HBM (measured: the first version of this file). So everything that does NOT depend on values is taken off the
// chain: the descriptions of up to 64 columns of the list are loaded at once, one 32-byte record per lane, and handed
// round by v_readlane; the index words of a column's first 64 products / row entries are loaded two columns ahead
// (three register sets in rotation); the entries of L start out as the entries of A (a parallel pass before the
// factorization), so a column never looks at A or at the A -> L map.
struct ColPre {  // what is prefetched for one column: products (two operands, target entry), row-j entry (value, column)
    uint2 pr;
    uint32_t pk, ri, rc;
};
__device__ __forceinline__ ColDesc desc_of_lane(const ColDesc& mine, int i) {
    ColDesc d;
    d.j = __builtin_amdgcn_readlane(mine.j, i);
    d.beg = __builtin_amdgcn_readlane(mine.beg, i);
    d.end = __builtin_amdgcn_readlane(mine.end, i);
    d.rbeg = __builtin_amdgcn_readlane(mine.rbeg, i);
    d.rend = __builtin_amdgcn_readlane(mine.rend, i);
    d.pbeg = __builtin_amdgcn_readlane(mine.pbeg, i);
    d.pend0 = __builtin_amdgcn_readlane(mine.pend0, i);
    d.pad = __builtin_amdgcn_readlane(mine.pad, i);
    return d;
}

// The columns [t0, t1) of a list, bottom-up: numeric Cholesky (left-looking by gather lists) with the forward sweep
// L y = b riding along — once column j is factored, row j of L is complete, so y_j follows at once. FACTOR = false: the
// forward sweep alone with the stored factor (the refined step needs it again). LDSV: l and b live in LDS.
// 64 entries of a column at a time (almost every column has fewer): all their products in one flat, lane-strided sweep
// — they are contiguous, lpair_ptr is a prefix over the entries — summed per entry with LDS atomics. `acc`: 64 doubles
// of LDS owned by the wavefront. On entry l holds the entries of A. Returns false when a pivot is not positive and finite.
// `first`: the lane's description of column t0 + lane, loaded by the caller (a level ahead: before the barrier).
// BLOB: the index data is a segment blob in LDS (fx_sparse_plan.h: SegmentBlobs) — segment-local 16-bit indices, packed.
template <bool LDSV, bool FACTOR, bool BLOB = false>
__device__ __forceinline__ bool team_walk_up(const SpChol& c, const SpRowsOfL& lr, const ColDesc* cd, uint32_t t0, uint32_t t1,
                                             const ColDesc& first, double lambda, double* l, double* b, double* acc, int lane, uint32_t lb = 0,
                                             uint32_t cb = 0) {
    // lb, cb: l and b hold a segment's run of entries / of columns, l[entry - lb], b[column - cb] (0: the whole factor)
    bool ok = true;
    auto ld_pair = [&](uint32_t p) -> uint2 {
        if (BLOB) {
            const uint32_t w = c.lpairs[p];
            return make_uint2(w & 0xFFFFu, w >> 16);
        }
        return reinterpret_cast<const uint2*>(c.lpairs)[p];
    };
    auto ld_pk = [&](uint32_t p) -> uint32_t { return BLOB ? (uint32_t)reinterpret_cast<const uint16_t*>(c.lpair_k)[p] : c.lpair_k[p]; };
    auto ld_row = [&](uint32_t r, uint32_t& ri, uint32_t& rc) {
        if (BLOB) {
            const uint32_t w = lr.ridx[r];
            ri = w & 0xFFFFu;
            rc = w >> 16;
        } else {
            ri = lr.ridx[r];
            rc = lr.rcol[r];
        }
    };
    auto fetch = [&](const ColDesc& d, ColPre& o) {
        if (FACTOR) {
            const uint32_t p = d.pbeg + lane;
            if (p < d.pend0) {
                o.pr = ld_pair(p);
                o.pk = ld_pk(p);
            }
        }
        const uint32_t r = d.rbeg + lane;
        if (r < d.rend) ld_row(r, o.ri, o.rc);
    };
    auto column = [&](const ColDesc& d, const ColPre& q) {
        double part = 0.0;  // forward-sweep gather for row j (all of its columns are final already)
        {
            uint32_t r = d.rbeg + lane;
            if (r < d.rend) part = l[q.ri - lb] * b[q.rc - cb];
            for (r += 64; r < d.rend; r += 64) {
                uint32_t ri, rc;
                ld_row(r, ri, rc);
                part = fma(l[ri - lb], b[rc - cb], part);
            }
        }
        double inv;  // 1 / L_jj
        if (FACTOR) {
            inv = 0.0;
            for (uint32_t base = d.beg; base < d.end; base += 64) {
                const bool first = base == d.beg;
                const uint32_t k = base + lane, cend = min(base + 64u, d.end);
                double s = 0.0;
                if (k < cend) {
                    s = l[k - lb];  // the entry of A (0 for fill-in)
                    if (k == d.beg) s += lambda;
                }
                const uint32_t pb = first ? d.pbeg : c.lpair_ptr[base], pe = first ? d.pend0 : c.lpair_ptr[cend];
                if (pe > pb) {  // (wave-uniform) leaves of the tree have no products at all
                    acc[lane] = 0.0;
                    // a pass whose 64 products all belong to one entry (the diagonal of a column every other column
                    // reaches): one butterfly and one add instead of 64 atomics on one address
                    auto add = [&](bool live, uint32_t tgt, double v) {
                        const uint32_t t0u = (uint32_t)__builtin_amdgcn_readfirstlane((int)tgt);
                        if (__ballot(!live || tgt != t0u) == 0ull) {
                            v = wave_sum(v);
                            if (lane == 0) acc[t0u] += v;
                        } else if (live) {
                            lds_add_f64(&acc[tgt], v);
                        }
                    };
                    uint32_t p = pb + lane;
                    if (first) {
                        add(p < pe, p < pe ? q.pk - base : 0u, p < pe ? -l[q.pr.x - lb] * l[q.pr.y - lb] : 0.0);
                        p += 64;
                    }
                    for (uint32_t p0 = pb + (first ? 64u : 0u); p0 < pe; p0 += 64, p += 64) {
                        uint2 q2 = make_uint2(0u, 0u);
                        uint32_t tg = 0;
                        if (p < pe) {
                            q2 = ld_pair(p);
                            tg = ld_pk(p) - base;
                        }
                        add(p < pe, tg, p < pe ? -l[q2.x - lb] * l[q2.y - lb] : 0.0);
                    }
                    wave_sync_lds();
                    if (k < cend) s += acc[lane];
                }
                if (first) {
                    const double piv = bcast_first(s);
                    ok = ok && (piv > 0.0) && (piv < 1.0e300);
                    inv = rsqrt_refined(piv);  // (v_rsq_f64 + two Newton steps: a quarter of sqrt + division on the chain)
                    if (lane == 0) s = piv * inv;
                }
                if (k < cend) l[k - lb] = (k == d.beg) ? s : s * inv;
            }
        } else {
            inv = 1.0 / l[d.beg - lb];  // (the stored diagonal is pivot / sqrt(pivot))
        }
        part = wave_sum(part);
        if (lane == 0) b[d.j - cb] = (b[d.j - cb] - part) * inv;
        vals_sync<LDSV>();  // the next column of this list may read what this one stored
    };
    // BLOB builds: everything a column touches is in LDS, and a lone wavefront's time is its instruction count (in-order
    // issue: ~250 instructions per column in the general code were 1.2 us) — so no prefetch rotation, and a short path
    // for the usual column: at most 64 entries, 64 products, 64 row entries, each one pass.
    auto lean_column = [&](const ColDesc& d, uint32_t pend_all) {
        const uint32_t k = d.beg + lane, p = d.pbeg + lane, r = d.rbeg + lane;
        const uint32_t np = d.pend0 - d.pbeg, nr = d.rend - d.rbeg;
        const bool has_k = k < d.end, has_p = FACTOR && p < d.pend0, has_r = r < d.rend;
        // loads without tests (an idle lane reads a word some busy lane reads anyway), so that they leave in two batches
        // with one wait each instead of seven waits one after the other
        const uint32_t rw = lr.ridx[has_r ? r : d.rbeg];
        const uint32_t pw = FACTOR ? c.lpairs[has_p ? p : d.pbeg] : 0u;
        const uint32_t pk = FACTOR ? (uint32_t)reinterpret_cast<const uint16_t*>(c.lpair_k)[has_p ? p : d.pbeg] : 0u;
        const double lk = l[has_k ? k : d.beg];
        const double lri = l[nr ? (rw & 0xFFFFu) : d.beg], brc = b[nr ? (rw >> 16) : d.j];
        double part = has_r ? lri * brc : 0.0;
        double inv;
        if (FACTOR) {
            double s = has_k ? lk : 0.0;
            if (lane == 0) s += lambda;
            if (np) {
                const double lx = l[pw & 0xFFFFu], ly = l[pw >> 16];
                acc[lane] = 0.0;
                if (has_p) lds_add_f64(&acc[pk - d.beg], -lx * ly);
                for (uint32_t p2 = p + 64u; p2 - lane < pend_all; p2 += 64u) {  // (a column of a few hundred products: more passes)
                    const bool live = p2 < pend_all;
                    const uint32_t w2 = c.lpairs[live ? p2 : d.pbeg];
                    const uint32_t k2 = (uint32_t)reinterpret_cast<const uint16_t*>(c.lpair_k)[live ? p2 : d.pbeg];
                    const double x2 = l[w2 & 0xFFFFu], y2 = l[w2 >> 16];
                    if (live) lds_add_f64(&acc[k2 - d.beg], -x2 * y2);
                }
                wave_sync_lds();
                s += acc[lane];
            }
            const double piv = bcast_first(s);
            ok = ok && (piv > 0.0) && (piv < 1.0e300);
            inv = rsqrt_refined(piv);
            if (has_k) l[k] = lane == 0 ? piv * inv : s * inv;
        } else {
            inv = 1.0 / bcast_first(lk);
        }
        if (nr) part = wave_sum_first(part, nr);
        if (lane == 0) b[d.j] = (b[d.j] - part) * inv;
        wave_sync_lds();
    };
    for (uint32_t tb = t0; tb < t1; tb += 64) {  // (more than 64 columns: 64 descriptions at a time)
        const uint32_t nb = min(64u, t1 - tb);
        ColDesc mine = first;
        if (tb != t0) {
            mine = ColDesc{};
            if ((uint32_t)lane < nb) mine = cd[tb + lane];
        }
        if (BLOB) {
            for (uint32_t i = 0; i < nb; ++i) {
                const ColDesc d = desc_of_lane(mine, (int)i);
                if (d.end - d.beg <= 64u && d.rend - d.rbeg <= 64u) {
                    lean_column(d, FACTOR ? c.lpair_ptr[d.end] : 0u);
                } else {
                    ColPre q{};
                    fetch(d, q);
                    column(d, q);
                }
            }
            continue;
        }
        ColDesc d0 = desc_of_lane(mine, 0), d1{}, d2{};
        ColPre q0{}, q1{}, q2{};
        fetch(d0, q0);
        if (nb > 1) {
            d1 = desc_of_lane(mine, 1);
            fetch(d1, q1);
        }
        for (uint32_t i = 0; i < nb; ++i) {  // column i + 2 is fetched while columns i and i + 1 compute
            if (i + 2 < nb) {
                d2 = desc_of_lane(mine, (int)(i + 2));
                fetch(d2, q2);
            }
            column(d0, q0);
            d0 = d1;  // (one copy of the column's code: the register sets rotate by moves)
            q0 = q1;
            d1 = d2;
            q1 = q2;
        }
    }
    return ok;
}

// Lt x = y, the columns of a list top-down: x_j = (y_j - sum_{i>j} L_ij x_i) / L_jj reads only ancestors of j
// `last`: the lane's description of column t1 - min(64, t1 - t0) + lane, loaded by the caller.
template <bool LDSV, bool USE_MID = false, bool BLOB = false>
__device__ __forceinline__ void team_walk_down(const SpChol& c, const ColDesc* cd, uint32_t t0, uint32_t t1, const ColDesc& last,
                                               const double* l, double* b, int lane, uint32_t lb = 0, uint32_t cb = 0) {
    auto ld_lrow = [&](uint32_t k) -> uint32_t { return BLOB ? (uint32_t)reinterpret_cast<const uint16_t*>(c.lrow)[k] : c.lrow[k]; };
    auto fetch = [&](const ColDesc& d, uint32_t& row) {
        const uint32_t k = d.beg + 1 + lane;
        if (k < (USE_MID ? d.pad : d.end)) row = ld_lrow(k);
    };
    auto column = [&](const ColDesc& d, uint32_t row) {
        double part = 0.0;
        uint32_t k = d.beg + 1 + lane;
        const uint32_t kend = USE_MID ? d.pad : d.end;  // (LDS build of a part: the entries whose rows are in the top were taken before)
        if (k < kend) part = l[k - lb] * b[row - cb];
        for (k += 64; k < kend; k += 64) part = fma(l[k - lb], b[ld_lrow(k) - cb], part);
        part = wave_sum(part);
        if (lane == 0) b[d.j - cb] = (b[d.j - cb] - part) / l[d.beg - lb];
        vals_sync<LDSV>();
    };
    for (uint32_t te = t1; te > t0;) {  // 64 descriptions at a time, from the end of the list
        const uint32_t nb = min(64u, te - t0), tb = te - nb;
        ColDesc mine = last;
        if (te != t1) {
            mine = ColDesc{};
            if ((uint32_t)lane < nb) mine = cd[tb + lane];
        }
        if (BLOB) {  // (LDS everywhere: no prefetch rotation, a short path for columns of at most 64 entries)
            for (uint32_t i = nb; i-- > 0;) {
                const ColDesc d = desc_of_lane(mine, (int)i);
                const uint32_t kend = USE_MID ? d.pad : d.end;
                if (kend - d.beg <= 65u) {
                    const uint32_t k = d.beg + 1 + lane, n = kend - d.beg - 1u;
                    double part = 0.0;
                    if (k < kend) part = l[k] * b[(uint32_t)reinterpret_cast<const uint16_t*>(c.lrow)[k]];
                    if (n) part = wave_sum_first(part, n);
                    if (lane == 0) b[d.j] = (b[d.j] - part) / l[d.beg];
                    wave_sync_lds();
                } else {
                    uint32_t row = 0;
                    fetch(d, row);
                    column(d, row);
                }
            }
            te = tb;
            continue;
        }
        ColDesc d0 = desc_of_lane(mine, (int)nb - 1), d1{}, d2{};
        uint32_t r0 = 0, r1 = 0, r2 = 0;
        fetch(d0, r0);
        if (nb > 1) {
            d1 = desc_of_lane(mine, (int)nb - 2);
            fetch(d1, r1);
        }
        for (uint32_t i = 0; i < nb; ++i) {  // i-th column from the end
            if (i + 2 < nb) {
                d2 = desc_of_lane(mine, (int)(nb - 3 - i));
                fetch(d2, r2);
            }
            column(d0, r0);
            d0 = d1;
            r0 = r1;
            d1 = d2;
            r1 = r2;
        }
        te = tb;
    }
}

// A segment's factorization + forward sweep by the calling workgroup: levels bottom-up, a barrier after each; a
// wavefront walks its run of columns of the level, and loads the descriptions of its next run before the barrier.
// Returns (to every thread of a wavefront) whether its pivots were fine.
// NW: the wavefronts of the calling workgroup. The schedules are built for TEAM_NWAVES = 16; a narrower team (the builds for
// many small Systems side by side, sp_lm_team_kernel<..., 4>) gives each of its wavefronts every NW-th list of a level.
// Which wavefront walks a column changes nothing in its arithmetic: same bits.
template <bool LDSV, bool FACTOR, bool BLOB = false, int NW = TEAM_NWAVES>
__device__ __forceinline__ bool team_factor_forward(const SpChol& c, const SpRowsOfL& lr, const SpTeamSched& sc, uint32_t seg,
                                                    double lambda, double* l, double* b, double* s_acc, unsigned long long* wprof = nullptr,
                                                    uint32_t lb = 0, uint32_t cb = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (NW != TEAM_NWAVES) {
        bool ok = true;
        for (uint32_t q = sc.seg_lev[seg]; q < sc.seg_lev[seg + 1]; ++q) {
            for (uint32_t vw = (uint32_t)wave; vw < (uint32_t)TEAM_NWAVES; vw += (uint32_t)NW) {
                const uint32_t t0 = sc.wptr[q * TEAM_NWAVES + vw], t1 = sc.wptr[q * TEAM_NWAVES + vw + 1];
                if (t0 >= t1) continue;
                ColDesc mine{};
                if (t0 + lane < t1) mine = sc.cdesc[t0 + lane];
                ok = team_walk_up<LDSV, FACTOR, BLOB>(c, lr, sc.cdesc, t0, t1, mine, lambda, l, b, s_acc + wave * 64, lane, lb, cb) && ok;
            }
            __syncthreads();
        }
        return ok;
    }
    const bool wstamp = wprof && threadIdx.x == 0;  // diagnostics: wavefront 0's walk / barrier time, level 0 and above
    bool ok = true;
    uint32_t q = sc.seg_lev[seg];
    const uint32_t qend = sc.seg_lev[seg + 1];
    if (q >= qend) return ok;
    uint32_t t0 = sc.wptr[q * TEAM_NWAVES + wave], t1 = sc.wptr[q * TEAM_NWAVES + wave + 1];
    ColDesc mine{};
    if (t0 + lane < t1) mine = sc.cdesc[t0 + lane];
    for (; q < qend; ++q) {
        uint32_t n0 = 0, n1 = 0;
        ColDesc next{};
        if (q + 1 < qend) {
            n0 = sc.wptr[(q + 1) * TEAM_NWAVES + wave];
            n1 = sc.wptr[(q + 1) * TEAM_NWAVES + wave + 1];
            if (n0 + lane < n1) next = sc.cdesc[n0 + lane];
        }
        const unsigned long long w0 = wstamp ? wall_clock64() : 0ull;
        if (t0 < t1) ok = team_walk_up<LDSV, FACTOR, BLOB>(c, lr, sc.cdesc, t0, t1, mine, lambda, l, b, s_acc + wave * 64, lane, lb, cb) && ok;
        const unsigned long long w1 = wstamp ? wall_clock64() : 0ull;
        __syncthreads();
        if (wstamp) {
            const unsigned long long w2 = wall_clock64();
            const int slot = q == sc.seg_lev[seg] ? 0 : 2;
            wprof[slot] += w1 - w0;
            wprof[slot + 1] += w2 - w1;
            if (slot == 0) wprof[4] += t1 - t0;
        }
        t0 = n0;
        t1 = n1;
        mine = next;
    }
    return ok;
}

template <bool LDSV, bool USE_MID = false, bool BLOB = false, int NW = TEAM_NWAVES>
__device__ __forceinline__ void team_backward(const SpChol& c, const SpTeamSched& sc, uint32_t seg, const double* l, double* b, uint32_t lb = 0,
                                              uint32_t cb = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qbeg = sc.seg_lev[seg];
    uint32_t q = sc.seg_lev[seg + 1];
    if (q <= qbeg) return;
    auto load_last = [&](uint32_t a0, uint32_t a1) {
        ColDesc d{};
        const uint32_t nb = min(64u, a1 - a0);
        if ((uint32_t)lane < nb) d = sc.cdesc[a1 - nb + lane];
        return d;
    };
    if constexpr (NW != TEAM_NWAVES) {
        for (; q-- > qbeg;) {
            for (uint32_t vw = (uint32_t)wave; vw < (uint32_t)TEAM_NWAVES; vw += (uint32_t)NW) {
                const uint32_t t0 = sc.wptr[q * TEAM_NWAVES + vw], t1 = sc.wptr[q * TEAM_NWAVES + vw + 1];
                if (t0 >= t1) continue;
                ColDesc mine = load_last(t0, t1);
                team_walk_down<LDSV, USE_MID, BLOB>(c, sc.cdesc, t0, t1, mine, l, b, lane, lb, cb);
            }
            __syncthreads();
        }
        return;
    }
    uint32_t t0 = sc.wptr[(q - 1) * TEAM_NWAVES + wave], t1 = sc.wptr[(q - 1) * TEAM_NWAVES + wave + 1];
    ColDesc mine = load_last(t0, t1);
    for (; q-- > qbeg;) {
        uint32_t n0 = 0, n1 = 0;
        ColDesc next{};
        if (q > qbeg) {
            n0 = sc.wptr[(q - 1) * TEAM_NWAVES + wave];
            n1 = sc.wptr[(q - 1) * TEAM_NWAVES + wave + 1];
            next = load_last(n0, n1);
        }
        if (t0 < t1) team_walk_down<LDSV, USE_MID, BLOB>(c, sc.cdesc, t0, t1, mine, l, b, lane, lb, cb);
        __syncthreads();
        t0 = n0;
        t1 = n1;
        mine = next;
    }
}

// A segment blob (fx_sparse_plan.h: SegmentBlobs), copied into LDS by the whole workgroup, and the views the walkers take
__device__ __forceinline__ void team_load_blob(const uint32_t* __restrict__ g, uint32_t* s, uint32_t nwords, uint32_t nthreads = TEAM_THREADS) {
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* s4 = reinterpret_cast<uint4*>(s);
    for (uint32_t i = threadIdx.x; i < nwords / 4u; i += nthreads) s4[i] = g4[i];
}
__device__ __forceinline__ void team_blob_views(const uint32_t* s, SpChol& c, SpRowsOfL& lr, SpTeamSched& sc) {
    c.lcolptr = nullptr;
    c.l2a = nullptr;
    c.lpair_ptr = s + s[7];
    c.lrow = s + s[8];
    c.lpairs = s + s[9];
    c.lpair_k = s + s[10];
    c.nv = s[1];
    lr.rptr = nullptr;
    lr.ridx = s + s[11];
    lr.rcol = nullptr;
    sc.seg_lev = s + 12;
    sc.wptr = s + s[5];
    sc.cols = nullptr;
    sc.cdesc = reinterpret_cast<const ColDesc*>(s + s[6]);
    sc.cdesc_mid = nullptr;
    sc.nparts = 0;
}

// K3 for the calling workgroup's share (first, step over the items): A = JtJ (lower triangle, permuted order) and
// -Jt r by deterministic gathers, one thread per entry; the few entries whose gather lists are long (the diagonal of a
// point every constraint hangs on: one product per constraint) are summed by a wavefront each, a fixed butterfly.
// The right-hand side goes to `rhs` (kept: a rejected trial solves with it again) and to `vec`, the vector the solves
// overwrite; an entry of A goes to `a` (kept likewise) and to its place in `lf`, where the factorization starts from
// (fill-in entries start as 0).
__device__ __forceinline__ void team_form(const SpBlock& B, const double* jc, const double* rc, double* a, double* rhs, double* vec, double* lf,
                                          uint32_t first, uint32_t step, uint32_t wave_first, uint32_t wave_step) {
    for (uint32_t k = first; k < B.nnz_l; k += step)
        if (B.chol.l2a[k] < 0) lf[k] = 0.0;
    for (uint32_t k = first; k < B.nnz_a; k += step) {
        const uint32_t pb = B.apair_ptr[k], pe = B.apair_ptr[k + 1];
        if (pe - pb > FORM_LONG) continue;
        double s = 0.0;
        for (uint32_t p = pb; p < pe; ++p) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
        a[k] = s;
        lf[B.a2l[k]] = s;
    }
    for (uint32_t c = first; c < B.nv; c += step) {
        const uint32_t pb = B.cptr[c], pe = B.cptr[c + 1];
        if (pe - pb > FORM_LONG) continue;
        double s = 0.0;
        for (uint32_t p = pb; p < pe; ++p) s += jc[B.cidx[p]] * -rc[B.crow[p]];
        rhs[c] = s;
        vec[c] = s;
    }
    const int lane = threadIdx.x & 63;
    for (uint32_t i = wave_first; i < B.n_along + B.n_clong; i += wave_step) {
        double s = 0.0;
        if (i < B.n_along) {
            const uint32_t k = B.along[i];
            for (uint32_t p = B.apair_ptr[k] + lane; p < B.apair_ptr[k + 1]; p += 64) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
            s = wave_sum64(s);
            if (lane == 0) {
                a[k] = s;
                lf[B.a2l[k]] = s;
            }
        } else {
            const uint32_t c = B.along[i];
            for (uint32_t p = B.cptr[c] + lane; p < B.cptr[c + 1]; p += 64) s += jc[B.cidx[p]] * -rc[B.crow[p]];
            s = wave_sum64(s);
            if (lane == 0) {
                rhs[c] = s;
                vec[c] = s;
            }
        }
    }
}

// K3 for one segment into the workgroup's LDS: the entries [eb, ee) of L start as the entries of A (0 for fill-in),
// the columns [cb, ce) of the vector as the right-hand side. need_form: A and -Jt r are formed from the current Jacobian
// rows (and kept in `a` / `rhs`); otherwise — a rejected trial — the kept values are taken again. Long gather lists as
// in team_form: a wavefront each.
__device__ __forceinline__ void team_form_segment(const SpBlock& B, const double* jc, const double* rc, double* a, double* rhs, double* s_l,
                                                  double* s_b, uint32_t eb, uint32_t ee, uint32_t cb, uint32_t ce, bool need_form) {
    for (uint32_t k = eb + threadIdx.x; k < ee; k += TEAM_THREADS) {
        const int32_t ai = B.chol.l2a[k];
        double s = 0.0;
        if (ai >= 0) {
            if (need_form) {
                const uint32_t pb = B.apair_ptr[ai], pe = B.apair_ptr[ai + 1];
                if (pe - pb > FORM_LONG) continue;
                for (uint32_t p = pb; p < pe; ++p) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
                a[ai] = s;
            } else {
                s = a[ai];
            }
        }
        s_l[k - eb] = s;
    }
    for (uint32_t c = cb + threadIdx.x; c < ce; c += TEAM_THREADS) {
        double s = 0.0;
        if (need_form) {
            const uint32_t pb = B.cptr[c], pe = B.cptr[c + 1];
            if (pe - pb > FORM_LONG) continue;
            for (uint32_t p = pb; p < pe; ++p) s += jc[B.cidx[p]] * -rc[B.crow[p]];
            rhs[c] = s;
        } else {
            s = rhs[c];
        }
        s_b[c - cb] = s;
    }
    if (!need_form) return;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x >> 6; i < B.n_along + B.n_clong; i += TEAM_NWAVES) {
        double s = 0.0;
        if (i < B.n_along) {
            const uint32_t k = B.along[i], le = B.a2l[k];
            if (le < eb || le >= ee) continue;
            for (uint32_t p = B.apair_ptr[k] + lane; p < B.apair_ptr[k + 1]; p += 64) s += jc[B.apairs[2 * p]] * jc[B.apairs[2 * p + 1]];
            s = wave_sum64(s);
            if (lane == 0) {
                a[k] = s;
                s_l[le - eb] = s;
            }
        } else {
            const uint32_t c = B.along[i];
            if (c < cb || c >= ce) continue;
            for (uint32_t p = B.cptr[c] + lane; p < B.cptr[c + 1]; p += 64) s += jc[B.cidx[p]] * -rc[B.crow[p]];
            s = wave_sum64(s);
            if (lane == 0) {
                rhs[c] = s;
                s_b[c - cb] = s;
            }
        }
    }
}

// sum v[i]^2 with the shape of sp_sumsq_kernel (1024 strided partial sums, then a binary tree): same bits
template <int NW = TEAM_NWAVES>
__device__ __forceinline__ double team_sumsq(const double* v, uint32_t n, double* s_red, uint32_t red_n = TEAM_THREADS) {
    if constexpr (NW == TEAM_NWAVES) {
        double s = 0.0;
        for (uint32_t i = threadIdx.x; i < n; i += TEAM_THREADS) s += v[i] * v[i];
        return block_sum_1024(s, s_red);
    } else {
        // a narrower team: the same 1024 strided partial sums and the same tree, several per thread. Of a vector shorter
        // than red_n (a power of two, at most 1024) only the first red_n partial sums are not zero, and the tree's steps
        // above red_n would add + 0.0 to sums of squares — exact — so the tree starts at red_n: red_n doubles of LDS, not 1024
        constexpr uint32_t NT = 64u * (uint32_t)NW;
        for (uint32_t p = threadIdx.x; p < red_n; p += NT) {
            double s = 0.0;
            for (uint32_t i = p; i < n; i += TEAM_THREADS) s += v[i] * v[i];
            s_red[p] = s;
        }
        __syncthreads();
        for (uint32_t w = red_n >> 1; w > 0; w >>= 1) {
            for (uint32_t p = threadIdx.x; p < w; p += NT) s_red[p] += s_red[p + w];
            __syncthreads();
        }
        const double out = s_red[0];
        __syncthreads();
        return out;
    }
}

// K1 / K2 for one row of a block (subsystem.rs:93-166), the body of sp_eval_kernel
// OVERWRITE: entries of a row that share a column — 0: summed (sparse J, sparse_col_mat.rs:710-711); 1: the last one wins
// (the dense J of L-BFGS, expressions.rs:993-1008) — quirk Q4
template <bool POSE, bool OVERWRITE = false>
__device__ __forceinline__ void team_eval_row(const SpRows& rows, const double* sparam, const SpJac& jac, uint32_t row, const double* xs,
                                              double* r, double* jvals) {
    const uint32_t e = jac.rows[row];
    const int tag = rows.tag[e] & 0x7F;
    const ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
    const uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    expand_vars<POSE>(tag, ff, vars8);
    double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = xs[vars8[q]];
    r[row] = eval_expression<double, true, false, POSE>(tag, v, sparam[e], g);
    const uint32_t slots = jac.jslot[row];
    const uint32_t base = jac.jrow_ptr[row];
    const uint32_t cnt = jac.jrow_ptr[row + 1] - base;
    double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t sl = (slots >> (4 * q)) & 0xFu;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OVERWRITE) out[u] = (sl == (uint32_t)u) ? g[q] : out[u];
            else out[u] += (sl == (uint32_t)u) ? g[q] : 0.0;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if ((uint32_t)u < cnt) jvals[base + u] = out[u];
}

// lm.rs:108-112 on a state in registers
__device__ __forceinline__ void lm_state_init(SpLm& st, double sse, const fx_lm_opts& o) {
    st.sse = st.sse_start = sse;
    st.sse_t = st.dn2 = 0.0;
    st.lambda = o.lambda0;
    st.cur = 0;
    st.accepted = st.trials = st.outer = 0;
    st.exit_code = FX_EXIT_MAX_OUTER;
    st.need_form = 1;
    st.flag = 0;
    st.done = 0;
    if (!(sse == sse) || !(sse < 1.0e300)) {
        st.exit_code = FX_EXIT_NAN;
        st.done = 1;
    } else if (o.max_outer == 0) {
        st.done = 1;
    } else if (sse < o.sse_tol) {  // lm.rs:110-112
        st.exit_code = FX_EXIT_SSE;
        st.done = 1;
    } else if (o.max_trials == 0) {
        st.exit_code = FX_EXIT_TRIAL_CAP;
        st.done = 1;
    }
}

// after a trial: accept / reject / stop (lm.rs:134-191) — sp_lm_control_kernel's decisions; need_form tells whether
// the next trial starts from a new point
__device__ __forceinline__ void lm_state_control(SpLm& st, const fx_lm_opts& o) {
    st.trials += 1;
    st.need_form = 0;
    bool check_cap = true;
    if (st.flag) {  // lm.rs:134-137
        st.lambda *= o.singular_factor;
        if (!(st.lambda < 1.0e300)) {
            st.exit_code = FX_EXIT_NAN;
            st.done = 1;
        }
    } else {
        const double dn2 = st.dn2, sse_t = st.sse_t, sse = st.sse;
        if (!(dn2 == dn2)) {
            st.exit_code = FX_EXIT_NAN;
            st.done = 1;
            check_cap = false;
        } else if (dn2 < o.step_tol) {  // lm.rs:139-142
            st.exit_code = FX_EXIT_STEP;
            st.done = 1;
            check_cap = false;
        } else if (sse_t < sse) {  // accept, lm.rs:151-186
            double lam = st.lambda * o.accept_factor;
            if (lam < o.lambda_min) lam = o.lambda_min;
            st.lambda = lam;
            st.cur ^= 1u;
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
        } else {  // reject, lm.rs:187-190
            st.lambda *= o.reject_factor;
            if (!(sse_t == sse_t) && !(st.lambda < 1.0e300)) {
                st.exit_code = FX_EXIT_NAN;
                st.done = 1;
                check_cap = false;
            }
        }
    }
    if (check_cap && !st.done && st.trials >= o.max_trials) {
        st.exit_code = FX_EXIT_TRIAL_CAP;
        st.done = 1;
    }
    st.flag = 0;
}

// what ends a block (assemble/mod.rs:161-166, :201-207): solved values out, working vectors ready for the next block
__device__ __forceinline__ void team_block_epilogue(const SpBlock& B, const SpVals& V, uint32_t cur, uint32_t flags, double* vars_out,
                                                    uint32_t first, uint32_t step) {
    const double* xc = cur ? V.xs1 : V.xs0;
    double* xo = cur ? V.xs0 : V.xs1;
    const double scale = V.scal[0];
    for (uint32_t k = first; k < B.nv; k += step) {
        const uint32_t vi = B.fvar[k];
        const double x = xc[vi];
        vars_out[vi] = (flags & TEAM_SCALE) ? scale * x : x;
        if (flags & TEAM_SINGLE_PASS) {
            xo[vi] = x;  // later blocks see this one (:201-207)
        } else {         // later components see the pre-solve snapshot (quirk Q2)
            const double sn = V.snap[vi];
            V.xs0[vi] = sn;
            V.xs1[vi] = sn;
        }
    }
}

// a rejected trial factors the same A again with a larger lambda: the factor array back to the entries of A
__device__ __forceinline__ void team_refill(const SpBlock& B, const double* a, const double* rhs, double* vec, double* lf, uint32_t first,
                                            uint32_t step) {
    for (uint32_t k = first; k < B.nnz_l; k += step) {
        const int32_t ai = B.chol.l2a[k];
        lf[k] = ai >= 0 ? a[ai] : 0.0;
    }
    for (uint32_t c = first; c < B.nv; c += step) vec[c] = rhs[c];
}

// ---- the whole LM loop of one block, one workgroup per System ------------------------------------------------------
// LDSV: the factor and the solves' vectors live in LDS (dynamic: lds_l doubles of L, then two vectors of lds_v) — every
// step of a column's dependent chain is then an LDS round trip instead of an L2 one. The host picks it when they fit.
// BLOB (with LDSV): the factor's index data sits in LDS as well, copied once for the whole solve (fx_sparse_plan.h:
// SegmentBlobs) — a column's chain then touches HBM not at all.
// NW: wavefronts per System — 16 for one System (or few) at the lowest latency, 4 for batches of small Systems: eight
// workgroups to a CU instead of two, and barriers a quarter as wide. Same bits either way.
template <bool POSE, bool LDSV, bool BLOB, int NW = TEAM_NWAVES>
__global__ __launch_bounds__(64 * NW) void sp_lm_team_kernel(SpRows rows, SpBlock B, SpVals V, SpAccum* __restrict__ accum,
                                                                   fx_lm_opts o, uint32_t flags, double* __restrict__ vars_base,
                                                                   const uint64_t* __restrict__ out_off, uint32_t lds_l, uint32_t lds_v,
                                                                   const uint32_t* __restrict__ blob, uint32_t blob_words, unsigned long long* prof,
                                                                   uint32_t red_off, uint32_t red_n) {
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
    constexpr uint32_t NT = 64u * (uint32_t)NW;
    __shared__ double s_acc[NW * 64];
    __shared__ double s_red_all[NW == TEAM_NWAVES ? TEAM_THREADS : 1];
    // (a narrow team keeps team_sumsq's partial sums in dynamic LDS, red_n of them at red_off doubles: see team_sumsq)
    double* const s_red = NW == TEAM_NWAVES ? s_red_all : s_dyn + red_off;
    __shared__ uint32_t s_bad;
    const uint32_t sys = blockIdx.x, tid = threadIdx.x;
    V.shift(sys);
    const double* sparam = rows.sparam + (size_t)sys * V.stride;
    const uint32_t m = B.m, nv = B.nv;
    double* const fl = LDSV ? s_dyn : V.l;                      // the factor
    double* const dvec = LDSV ? s_dyn + lds_l : V.delta;        // right-hand side -> step
    double* const evec = LDSV ? s_dyn + lds_l + lds_v : V.e;    // the refinement's
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SpChol chol = B.chol;
    SpRowsOfL lrows = B.lrows;
    SpTeamSched sched = B.sched;
    if (BLOB) {
        uint32_t* s_blob = reinterpret_cast<uint32_t*>(s_dyn + lds_l + 2 * lds_v);
        team_load_blob(blob, s_blob, blob_words, NT);
        __syncthreads();
        team_blob_views(s_blob, chol, lrows, sched);
    }

    for (uint32_t row = tid; row < m; row += NT) team_eval_row<POSE>(rows, sparam, B.jac, row, V.xs0, V.r0, V.j0);
    __syncthreads();
    SpLm st;
    lm_state_init(st, team_sumsq<NW>(V.r0, m, s_red, red_n), o);
    mark(0);
    while (!st.done) {
        const double* jc = st.cur ? V.j1 : V.j0;
        const double* rc = st.cur ? V.r1 : V.r0;
        if (st.need_form) team_form(B, jc, rc, V.a, V.rhs, dvec, fl, tid, NT, wave, NW);
        else team_refill(B, V.a, V.rhs, dvec, fl, tid, NT);
        if (tid == 0) s_bad = 0;
        __syncthreads();
        mark(1);
        const bool ok = team_factor_forward<LDSV, true, BLOB, NW>(chol, lrows, sched, 0, st.lambda, fl, dvec, s_acc, stamp ? prof + 8 : nullptr);
        if (!ok && (tid & 63) == 0) atomicOr(&s_bad, 1u);
        __syncthreads();
        st.flag = s_bad;
        __syncthreads();  // (s_bad is cleared again at the top of the next trial)
        mark(2);
        if (!st.flag) {
            team_backward<LDSV, false, BLOB, NW>(chol, sched, 0, fl, dvec);
            mark(3);
            if (flags & TEAM_REFINED) {  // corrected semi-normal equations, as sp_refine_*: t = -r - J delta, (A + lambda I) e = Jt t - lambda delta
                for (uint32_t row = tid; row < m; row += NT) {
                    double acc = -rc[row];
                    for (uint32_t p = B.jac.jrow_ptr[row]; p < B.jac.jrow_ptr[row + 1]; ++p) acc -= jc[p] * dvec[B.jcol[p]];
                    V.t[row] = acc;
                }
                __syncthreads();
                for (uint32_t c = tid; c < nv; c += NT) {
                    double s = 0.0;
                    for (uint32_t p = B.cptr[c]; p < B.cptr[c + 1]; ++p) s += jc[B.cidx[p]] * V.t[B.crow[p]];
                    evec[c] = s - st.lambda * dvec[c];
                }
                __syncthreads();
                team_factor_forward<LDSV, false, BLOB, NW>(chol, lrows, sched, 0, 0.0, fl, evec, s_acc);
                team_backward<LDSV, false, BLOB, NW>(chol, sched, 0, fl, evec);
                for (uint32_t c = tid; c < nv; c += NT) dvec[c] += evec[c];
                __syncthreads();
                mark(4);
            }
            st.dn2 = team_sumsq<NW>(dvec, nv, s_red, red_n);
            const double* xc = st.cur ? V.xs1 : V.xs0;
            double* xt = st.cur ? V.xs0 : V.xs1;
            for (uint32_t k = tid; k < nv; k += NT) {
                const uint32_t v = B.fvar[B.perm[k]];
                xt[v] = xc[v] + dvec[k];
            }
            __syncthreads();
            double* rt = st.cur ? V.r0 : V.r1;
            double* jt = st.cur ? V.j0 : V.j1;
            for (uint32_t row = tid; row < m; row += NT) team_eval_row<POSE>(rows, sparam, B.jac, row, xt, rt, jt);
            __syncthreads();
            st.sse_t = team_sumsq<NW>(rt, m, s_red, red_n);
            mark(5);
        }
        lm_state_control(st, o);
    }
    team_block_epilogue(B, V, st.cur, flags, vars_base + out_off[sys], tid, 64u * (uint32_t)NW);
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
}

// ---- Optimizer::LBfgs (solve/lbfgs.rs:20-193) of one block, one workgroup per System -------------------------------
// Round 2 ran the line search on the host, two scalars read back per evaluation. Here the whole optimizer is one launch:
// every thread carries the Hager-Zhang machine (fx_lbfgs.h: the reference's nested line search turned inside out around
// ONE evaluation site) and feeds it the same (p, phi, phi') — uniform decisions, no host. Vectors live in the block's
// free-variable order; every sum is taken in the reference's order — products written side by side, then added up by
// one thread from first to last (dot_product :213-216, sum_squares utils.rs:11-19, compute_gradient :199-210 row by
// row) — so distance-only sketches reproduce the reference algorithm bit for bit at any size, as the one-wavefront build does.
constexpr uint32_t SEQ_CHUNK = 4096;  // doubles of LDS the products of a sequential sum are staged in

// sum of f(i), i = 0 .. n-1, added in index order; every thread gets the result
template <typename F>
__device__ __forceinline__ double team_seq_sum(uint32_t n, F f, double* s_seq, double* s_out) {
    double sum = 0.0;
    for (uint32_t base = 0; base < n; base += SEQ_CHUNK) {
        const uint32_t cnt = min(SEQ_CHUNK, n - base);
        for (uint32_t i = threadIdx.x; i < cnt; i += TEAM_THREADS) s_seq[i] = f(base + i);
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t i = 0; i < cnt; i += 16u) {
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = (i + u < cnt) ? s_seq[i + u] : 0.0;  // (+0.0 changes no sum that has begun)
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (i + u < cnt) sum += t[u];
            }
            s_out[0] = sum;
        }
        __syncthreads();
        sum = s_out[0];
        __syncthreads();
    }
    return sum;
}

template <bool POSE>
__global__ __launch_bounds__(TEAM_THREADS) void sp_lbfgs_team_kernel(SpRows rows, SpBlock B, SpVals V, SpAccum* __restrict__ accum, uint32_t flags,
                                                                      double* __restrict__ vars_base, const uint64_t* __restrict__ out_off) {
    __shared__ double s_seq[SEQ_CHUNK];
    __shared__ double s_out[2];
    const uint32_t sys = blockIdx.x, tid = threadIdx.x;
    V.shift(sys);
    const double* sparam = rows.sparam + (size_t)sys * V.stride;
    const uint32_t m = B.m, nv = B.nv;
    double* const grad = V.a;   // (the LM arrays of the same slab, sized for at least nv)
    double* const dir = V.rhs;
    double* const hs = V.hs;
    double* const hy = V.hy;
    double rho[5] = {0., 0., 0., 0., 0.};

    // residuals + dense-Jacobian rows at x (generation g), then the gradient Jt r: per variable over the rows, ascending
    auto evaluate = [&](const double* xs, double* r, double* jv) {
        for (uint32_t row = tid; row < m; row += TEAM_THREADS) team_eval_row<POSE, true>(rows, sparam, B.jac, row, xs, r, jv);
        __syncthreads();
        for (uint32_t c = tid; c < nv; c += TEAM_THREADS) {
            double g = 0.0;
            for (uint32_t p = B.cptr[c]; p < B.cptr[c + 1]; ++p) g += jv[B.cidx[p]] * r[B.crow[p]];
            grad[B.perm[c]] = g;
        }
        __syncthreads();
    };
    evaluate(V.xs0, V.r0, V.j0);
    uint32_t accepted = 0, trials = 1, exit_code = FX_EXIT_MAX_OUTER;
    double prev = team_seq_sum(m, [&](uint32_t i) { return V.r0[i] * V.r0[i]; }, s_seq, s_out);
    const double sse_start = prev;
    double sse = prev;
    if (!(prev == prev)) {
        exit_code = FX_EXIT_NAN;
    } else if (prev < LbfgsConst::START_THRESHOLD) {
        exit_code = FX_EXIT_SSE;
    } else {
        for (uint32_t c = tid; c < 5u * nv; c += TEAM_THREADS) {
            hs[c] = 0.0;
            hy[c] = 0.0;
        }
        __syncthreads();
        for (uint32_t k = 0; k < LbfgsConst::MAX_ITERATIONS; ++k) {
            // ---- the two-loop recursion (:86-139), ring indexing (k + i) % 5 as the reference has it
            const uint32_t hl = k < 5u ? k : 5u;
            double alpha[5] = {0., 0., 0., 0., 0.};
            for (uint32_t j = tid; j < nv; j += TEAM_THREADS) dir[j] = grad[j];
            __syncthreads();
            for (int i = 4; i >= 0; --i) {
                if ((uint32_t)i >= hl) continue;
                const uint32_t h = (k + (uint32_t)i) % 5u;
                const double* s_i = hs + (size_t)h * nv;
                const double* y_i = hy + (size_t)h * nv;
                const double dp = team_seq_sum(nv, [&](uint32_t j) { return s_i[j] * dir[j]; }, s_seq, s_out);
                alpha[i] = rho[h] * dp;
                for (uint32_t j = tid; j < nv; j += TEAM_THREADS) dir[j] -= alpha[i] * y_i[j];
                __syncthreads();
            }
            if (k > 0) {
                const uint32_t h = (k - 1u) % 5u;
                const double* s_h = hs + (size_t)h * nv;
                const double* y_h = hy + (size_t)h * nv;
                const double s_dot_y = team_seq_sum(nv, [&](uint32_t j) { return s_h[j] * y_h[j]; }, s_seq, s_out);
                const double y_dot_y = team_seq_sum(nv, [&](uint32_t j) { return y_h[j] * y_h[j]; }, s_seq, s_out);
                if (y_dot_y > 0.) {
                    const double scale = s_dot_y / y_dot_y;
                    for (uint32_t j = tid; j < nv; j += TEAM_THREADS) dir[j] *= scale;
                    __syncthreads();
                }
            }
            for (int i = 0; i < 5; ++i) {
                if ((uint32_t)i >= hl) continue;
                const uint32_t h = (k + (uint32_t)i) % 5u;
                const double* s_i = hs + (size_t)h * nv;
                const double* y_i = hy + (size_t)h * nv;
                const double dp = team_seq_sum(nv, [&](uint32_t j) { return y_i[j] * dir[j]; }, s_seq, s_out);
                const double beta = rho[h] * dp;
                for (uint32_t j = tid; j < nv; j += TEAM_THREADS) dir[j] += s_i[j] * (alpha[i] - beta);
                __syncthreads();
            }
            const uint32_t h = k % 5u;
            double* y_k = hy + (size_t)h * nv;
            for (uint32_t j = tid; j < nv; j += TEAM_THREADS) {
                dir[j] *= -1.;
                y_k[j] = grad[j];  // the old gradient, parked until the update (:141-143)
            }
            __syncthreads();
            // ---- the line search (:218-506): one evaluation site, fed to the machine
            HzMachine hz;
            double step = hz.start(prev, team_seq_sum(nv, [&](uint32_t j) { return grad[j] * dir[j]; }, s_seq, s_out));
            HzParam acc_pt{0., 0., 0.};
            for (;;) {  // calculate_phi (:270-284): xs1 = xs0 + step * dir
                for (uint32_t j = tid; j < nv; j += TEAM_THREADS) {
                    const uint32_t v = B.fvar[j];
                    V.xs1[v] = V.xs0[v] + step * dir[j];
                }
                __syncthreads();
                evaluate(V.xs1, V.r1, V.j1);
                const double phi = team_seq_sum(m, [&](uint32_t i) { return V.r1[i] * V.r1[i]; }, s_seq, s_out);
                const double dphi = team_seq_sum(nv, [&](uint32_t j) { return grad[j] * dir[j]; }, s_seq, s_out);
                trials += 1;
                if (hz.feed(HzParam{step, phi, dphi}, step, acc_pt)) break;
            }
            // ---- variables = scratch; s_k = step * dir, y_k = grad - old grad, rho_k = 1 / s_k.y_k (:168-180)
            for (uint32_t j = tid; j < nv; j += TEAM_THREADS) {
                const uint32_t v = B.fvar[j];
                V.xs0[v] = V.xs1[v];
                hs[(size_t)h * nv + j] = acc_pt.p * dir[j];
                y_k[j] = grad[j] - y_k[j];
            }
            __syncthreads();
            const double* s_k = hs + (size_t)h * nv;
            rho[h] = 1.0 / team_seq_sum(nv, [&](uint32_t j) { return s_k[j] * y_k[j]; }, s_seq, s_out);
            accepted += 1;
            sse = acc_pt.phi;
            if (hz.capped) {
                exit_code = FX_EXIT_TRIAL_CAP;
                break;
            }
            if (!(acc_pt.phi == acc_pt.phi)) {
                exit_code = FX_EXIT_NAN;
                break;
            }
            if (::fabs(prev - acc_pt.phi) < LbfgsConst::CONVERGENCE_THRESHOLD) {
                exit_code = FX_EXIT_FTOL;
                break;
            }
            if (acc_pt.phi < LbfgsConst::RESIDUAL_THRESHOLD) {
                exit_code = FX_EXIT_SSE;
                break;
            }
            prev = acc_pt.phi;
        }
    }
    __syncthreads();
    team_block_epilogue(B, V, 0u, flags, vars_base + out_off[sys], tid, TEAM_THREADS);
    if (tid == 0) {
        SpAccum& ac = accum[sys];
        ac.accepted += accepted;
        ac.trials += trials;
        ac.exit_code = exit_code;
        ac.sse0 += sse_start;
        ac.sse += sse;
    }
}

// ---- larger Systems: parts side by side, the top in one workgroup; five launches per trial --------------------------
// (grid.y = System of the group; st = the System's SpLm in HBM; every kernel returns at once when its solve is over)
__global__ __launch_bounds__(256) void spt_form_kernel(SpBlock B, SpVals V, const SpLm* __restrict__ lm) {
    const SpLm* st = lm + blockIdx.y;
    if (st->done) return;
    V.shift(blockIdx.y);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    if (st->need_form) team_form(B, st->cur ? V.j1 : V.j0, st->cur ? V.r1 : V.r0, V.a, V.rhs, V.delta, V.l, i, n, i >> 6, n >> 6);
    else team_refill(B, V.a, V.rhs, V.delta, V.l, i, n);
}

// which: 0 = factor + forward sweep on delta; 1 = forward sweep of the refinement on e
__global__ __launch_bounds__(TEAM_THREADS) void spt_parts_up_kernel(SpBlock B, SpVals V, SpLm* __restrict__ lm, uint32_t which) {
    __shared__ double s_acc[TEAM_NWAVES * 64];
    SpLm* st = lm + blockIdx.y;
    if (st->done || (which && st->flag)) return;
    V.shift(blockIdx.y);
    if (which == 0) {
        const bool ok = team_factor_forward<false, true>(B.chol, B.lrows, B.sched, blockIdx.x, st->lambda, V.l, V.delta, s_acc);
        if (!ok && (threadIdx.x & 63) == 0) atomicOr(&st->flag, 1u);
    } else {
        team_factor_forward<false, false>(B.chol, B.lrows, B.sched, blockIdx.x, 0.0, V.l, V.e, s_acc);
    }
}

// the trial point of the columns [first, last) of the walking order: xs[cur ^ 1] = xs[cur] + delta
__device__ __forceinline__ void team_trial_point(const SpBlock& B, const SpVals& V, uint32_t cur, uint32_t first, uint32_t last) {
    const double* xc = cur ? V.xs1 : V.xs0;
    double* xt = cur ? V.xs0 : V.xs1;
    for (uint32_t t = first + threadIdx.x; t < last; t += TEAM_THREADS) {
        const uint32_t k = B.sched.cols[t];
        const uint32_t v = B.fvar[B.perm[k]];
        xt[v] = xc[v] + V.delta[k];
    }
}
__device__ __forceinline__ void team_segment_columns(const SpTeamSched& sc, uint32_t seg, uint32_t& first, uint32_t& last) {
    first = sc.wptr[sc.seg_lev[seg] * TEAM_NWAVES];
    last = sc.wptr[sc.seg_lev[seg + 1] * TEAM_NWAVES];
}

// the top of the tree: its factorization + forward sweep, then its backward sweep. which as above; `last` = nothing
// follows the backward sweep of this vector but the trial point (plain step: which 0; refined step: which 1)
__global__ __launch_bounds__(TEAM_THREADS) void spt_top_kernel(SpBlock B, SpVals V, SpLm* __restrict__ lm, uint32_t which, uint32_t last) {
    __shared__ double s_acc[TEAM_NWAVES * 64];
    __shared__ uint32_t s_bad;
    SpLm* st = lm + blockIdx.y;
    if (st->done || (which && st->flag)) return;
    V.shift(blockIdx.y);
    const uint32_t top = B.sched.nparts;
    double* vec = which ? V.e : V.delta;
    if (which == 0) {
        if (st->flag) return;  // a part met a bad pivot
        if (threadIdx.x == 0) s_bad = 0;
        __syncthreads();
        const bool ok = team_factor_forward<false, true>(B.chol, B.lrows, B.sched, top, st->lambda, V.l, V.delta, s_acc);
        if (!ok && (threadIdx.x & 63) == 0) atomicOr(&s_bad, 1u);
        __syncthreads();
        if (s_bad) {
            if (threadIdx.x == 0) st->flag = 1;
            return;
        }
    } else {
        team_factor_forward<false, false>(B.chol, B.lrows, B.sched, top, 0.0, V.l, V.e, s_acc);
    }
    team_backward<false>(B.chol, B.sched, top, V.l, vec);
    uint32_t first, end;
    team_segment_columns(B.sched, top, first, end);
    if (which) {
        for (uint32_t t = first + threadIdx.x; t < end; t += TEAM_THREADS) {
            const uint32_t k = B.sched.cols[t];
            V.delta[k] += V.e[k];
        }
        __syncthreads();
    }
    if (last) team_trial_point(B, V, st->cur, first, end);
}

__global__ __launch_bounds__(TEAM_THREADS) void spt_parts_down_kernel(SpBlock B, SpVals V, const SpLm* __restrict__ lm, uint32_t which, uint32_t last) {
    const SpLm* st = lm + blockIdx.y;
    if (st->done || st->flag) return;
    V.shift(blockIdx.y);
    double* vec = which ? V.e : V.delta;
    team_backward<false>(B.chol, B.sched, blockIdx.x, V.l, vec);
    uint32_t first, end;
    team_segment_columns(B.sched, blockIdx.x, first, end);
    if (which) {
        for (uint32_t t = first + threadIdx.x; t < end; t += TEAM_THREADS) {
            const uint32_t k = B.sched.cols[t];
            V.delta[k] += V.e[k];
        }
        __syncthreads();
    }
    if (last) team_trial_point(B, V, st->cur, first, end);
}

// ---- the same three steps with every segment's values in LDS (the plain step; fx_sparse_plan.h: PartsExtra) ----------
// A column's chain then runs on LDS round trips instead of L2 ones (measured on cfg2: the top alone took 100 us per
// trial from HBM). Segments are runs of columns and of entries, so LDS slot = index - first index of the segment. What
// the top needs from the parts — the products of its entries and of its right-hand side that belong to part columns —
// each part sums from its own LDS into a slot of a contribution buffer; the top subtracts its slots and then only sees
// its own columns.
struct SpPartsX {
    const uint32_t *seg_col, *seg_ent;
    const uint32_t *frun_ptr, *frun, *fslot_ptr, *brun_ptr, *brun, *bslot_ptr;
    const uint32_t* blobs;     // the segments' index data for LDS (fx_sparse_plan.h: SegmentBlobs; the top's: its own products)
    const uint32_t* blob_off;  // [nseg + 1]
    const uint32_t* cmid;      // [nv] first entry of a column whose row is in the top
    const uint32_t *erow_ptr, *erows;  // [nseg + 1], [m]: the block's rows by segment — a row reads variables of ONE part and of the top
    uint32_t nparts;
};
struct SpContrib {
    double *f, *b;         // [slots] per System, at System * stride
    size_t stride;
};

template <bool POSE>
__device__ __forceinline__ void sptl_top_body(const SpRows& rows, const SpBlock& B, const SpPartsX& X, const SpVals& V, const SpContrib& C, SpLm* st,
                                              double* s_dyn, double* s_acc, uint32_t* s_badp);

template <bool POSE>
__global__ __launch_bounds__(TEAM_THREADS) void sptl_parts_up_kernel(SpRows rows, SpBlock B, SpPartsX X, SpVals V, SpContrib C, SpLm* __restrict__ lm,
                                                                      uint32_t* __restrict__ tickets, unsigned long long* prof) {
    extern __shared__ double s_dyn[];
    __shared__ double s_acc[TEAM_NWAVES * 64];
    __shared__ uint32_t s_last, s_bad;
    SpLm* st = lm + blockIdx.y;
    if (st->done) return;
    const bool stamp = prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;  // diagnostics (FIKSI_AMD_TEAM_PROF)
    unsigned long long t_prev = stamp ? wall_clock64() : 0ull;
    auto mark = [&](int slot) {
        if (stamp) {
            const unsigned long long now = wall_clock64();
            prof[slot] += now - t_prev;
            t_prev = now;
        }
    };
    V.shift(blockIdx.y);
    const uint32_t part = blockIdx.x, tid = threadIdx.x;
    const uint32_t eb = X.seg_ent[part], ne = X.seg_ent[part + 1] - eb, cb = X.seg_col[part], nc = X.seg_col[part + 1] - cb;
    double* const s_l = s_dyn;
    double* const s_b = s_dyn + ((ne + 1u) & ~1u);
    uint32_t* const s_blob = reinterpret_cast<uint32_t*>(s_b + ((nc + 1u) & ~1u));
    team_load_blob(X.blobs + X.blob_off[part], s_blob, X.blob_off[part + 1] - X.blob_off[part]);
    team_form_segment(B, st->cur ? V.j1 : V.j0, st->cur ? V.r1 : V.r0, V.a, V.rhs, s_l, s_b, eb, eb + ne, cb, cb + nc, st->need_form != 0);
    __syncthreads();
    SpChol chol;
    SpRowsOfL lrows;
    SpTeamSched sched;
    team_blob_views(s_blob, chol, lrows, sched);
    mark(0);
    const bool ok = team_factor_forward<true, true, true>(chol, lrows, sched, 0, st->lambda, s_l, s_b, s_acc, stamp ? prof + 8 : nullptr);
    if (!ok && (tid & 63) == 0) atomicOr(&st->flag, 1u);
    mark(1);
    // the part's columns of L and of y for the backward sweep (another launch); its share of the top's sums
    for (uint32_t i = tid; i < ne; i += TEAM_THREADS) V.l[eb + i] = s_l[i];
    for (uint32_t c = tid; c < nc; c += TEAM_THREADS) V.delta[cb + c] = s_b[c];
    double* const cf = C.f + blockIdx.y * C.stride;
    double* const cbv = C.b + blockIdx.y * C.stride;
    for (uint32_t r = X.frun_ptr[part] + tid; r < X.frun_ptr[part + 1]; r += TEAM_THREADS) {
        const uint32_t slot = X.frun[3 * r], lo = X.frun[3 * r + 1], hi = X.frun[3 * r + 2];
        double s = 0.0;
        for (uint32_t t = lo; t < hi; ++t) s += s_l[B.chol.lpairs[2 * t] - eb] * s_l[B.chol.lpairs[2 * t + 1] - eb];
        cf[slot] = s;
    }
    for (uint32_t r = X.brun_ptr[part] + tid; r < X.brun_ptr[part + 1]; r += TEAM_THREADS) {
        const uint32_t slot = X.brun[3 * r], lo = X.brun[3 * r + 1], hi = X.brun[3 * r + 2];
        double s = 0.0;
        for (uint32_t t = lo; t < hi; ++t) s += s_l[B.lrows.ridx[t] - eb] * s_b[B.lrows.rcol[t] - cb];
        cbv[slot] = s;
    }
    mark(2);
    if (stamp) prof[7] += 1;
    // the workgroup that finishes last goes on with the top (no launch boundary: every part publishes, the last one acquires)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tickets[blockIdx.y] = 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    sptl_top_body<POSE>(rows, B, X, V, C, st, s_dyn, s_acc, &s_bad);
}

// The top of the tree, by the workgroup that finished its part last (V is that System's already): the parts' sums off
// its entries and its right-hand side, its factorization + forward sweep, its backward sweep, its columns' trial point,
// the rows that read no part's variables.
template <bool POSE>
__device__ __forceinline__ void sptl_top_body(const SpRows& rows, const SpBlock& B, const SpPartsX& X, const SpVals& V, const SpContrib& C, SpLm* st,
                                              double* s_dyn, double* s_acc, uint32_t* s_badp) {
    uint32_t& s_bad = *s_badp;
    if (st->flag) return;  // (a part met a bad pivot)
    const uint32_t top = X.nparts, tid = threadIdx.x;
    const uint32_t eb = X.seg_ent[top], ne = X.seg_ent[top + 1] - eb, cb = X.seg_col[top], nc = X.seg_col[top + 1] - cb;
    double* const s_l = s_dyn;
    double* const s_b = s_dyn + ((ne + 1u) & ~1u);
    uint32_t* const s_blob = reinterpret_cast<uint32_t*>(s_b + ((nc + 1u) & ~1u));
    team_load_blob(X.blobs + X.blob_off[top], s_blob, X.blob_off[top + 1] - X.blob_off[top]);
    const double* const cf = C.f + blockIdx.y * C.stride;
    const double* const cbv = C.b + blockIdx.y * C.stride;
    team_form_segment(B, st->cur ? V.j1 : V.j0, st->cur ? V.r1 : V.r0, V.a, V.rhs, s_l, s_b, eb, eb + ne, cb, cb + nc, st->need_form != 0);
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += TEAM_THREADS) {  // ... minus what the parts summed for it
        double s = s_l[i];
        for (uint32_t q = X.fslot_ptr[i]; q < X.fslot_ptr[i + 1]; ++q) s -= cf[q];
        s_l[i] = s;
    }
    for (uint32_t c = tid; c < nc; c += TEAM_THREADS) {
        double s = s_b[c];
        for (uint32_t q = X.bslot_ptr[c]; q < X.bslot_ptr[c + 1]; ++q) s -= cbv[q];
        s_b[c] = s;
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    SpChol chol;
    SpRowsOfL lrows;
    SpTeamSched sched;
    team_blob_views(s_blob, chol, lrows, sched);
    const bool ok = team_factor_forward<true, true, true>(chol, lrows, sched, 0, st->lambda, s_l, s_b, s_acc);
    if (!ok && (tid & 63) == 0) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (s_bad) {
        if (tid == 0) st->flag = 1;
        return;
    }
    team_backward<true, false, true>(chol, sched, 0, s_l, s_b);
    const double* xc = st->cur ? V.xs1 : V.xs0;
    double* xt = st->cur ? V.xs0 : V.xs1;
    for (uint32_t c = tid; c < nc; c += TEAM_THREADS) {  // the step of the top's columns, and their trial point
        const double dx = s_b[c];
        V.delta[cb + c] = dx;
        const uint32_t v = B.fvar[B.perm[cb + c]];
        xt[v] = xc[v] + dx;
    }
    __syncthreads();
    // K2 / K1 at the trial point for the rows that read no part's variables (the parts evaluate theirs after their sweep)
    const double* sparam = rows.sparam + (size_t)blockIdx.y * V.stride;
    for (uint32_t q = X.erow_ptr[top] + tid; q < X.erow_ptr[top + 1]; q += TEAM_THREADS)
        team_eval_row<POSE>(rows, sparam, B.jac, X.erows[q], xt, st->cur ? V.r0 : V.r1, st->cur ? V.j0 : V.j1);
}

// The parts' backward sweep, their columns' trial point, the evaluation of their rows there — and, in the block that
// finishes last, the sums over all rows / all columns (fixed shapes: the same bits whoever it is) and the trial's decision
// (lm.rs:134-191): what spt_eval_kernel did in a launch of its own.
template <bool POSE>
__global__ __launch_bounds__(TEAM_THREADS) void sptl_parts_down_kernel(SpRows rows, SpBlock B, SpPartsX X, SpVals V, SpLm* __restrict__ lm,
                                                                        uint32_t* __restrict__ tickets, fx_lm_opts o) {
    extern __shared__ double s_dyn[];
    __shared__ double s_red[TEAM_THREADS];
    __shared__ uint32_t s_last;
    SpLm* stg = lm + blockIdx.y;
    if (stg->done) return;
    const bool bad = stg->flag != 0;  // (a bad pivot: no step; the decision is still taken)
    V.shift(blockIdx.y);
    const uint32_t part = blockIdx.x, tid = threadIdx.x, cur = stg->cur;
    double* rt = cur ? V.r0 : V.r1;
    if (!bad) {
        const uint32_t eb = X.seg_ent[part], ne = X.seg_ent[part + 1] - eb, cb = X.seg_col[part], nc = X.seg_col[part + 1] - cb;
        double* const s_l = s_dyn;
        double* const s_b = s_dyn + ((ne + 1u) & ~1u);
        uint32_t* const s_blob = reinterpret_cast<uint32_t*>(s_b + ((nc + 1u) & ~1u));
        team_load_blob(X.blobs + X.blob_off[part], s_blob, X.blob_off[part + 1] - X.blob_off[part]);
        for (uint32_t i = tid; i < ne; i += TEAM_THREADS) s_l[i] = V.l[eb + i];
        // y_j minus what the top's columns of x take from it: the entries of column j whose rows are in the top come last
        for (uint32_t c = tid; c < nc; c += TEAM_THREADS) {
            const uint32_t j = cb + c;
            double s = V.delta[j];
            for (uint32_t k = X.cmid[j]; k < B.chol.lcolptr[j + 1]; ++k) s -= V.l[k] * V.delta[B.chol.lrow[k]];
            s_b[c] = s;
        }
        __syncthreads();
        SpChol chol;
        SpRowsOfL lrows;
        SpTeamSched sched;
        team_blob_views(s_blob, chol, lrows, sched);
        team_backward<true, true, true>(chol, sched, 0, s_l, s_b);
        const double* xc = cur ? V.xs1 : V.xs0;
        double* xt = cur ? V.xs0 : V.xs1;
        for (uint32_t c = tid; c < nc; c += TEAM_THREADS) {
            const double dx = s_b[c];
            V.delta[cb + c] = dx;
            const uint32_t v = B.fvar[B.perm[cb + c]];
            xt[v] = xc[v] + dx;
        }
        __syncthreads();
        const double* sparam = rows.sparam + (size_t)blockIdx.y * V.stride;
        for (uint32_t q = X.erow_ptr[part] + tid; q < X.erow_ptr[part + 1]; q += TEAM_THREADS)
            team_eval_row<POSE>(rows, sparam, B.jac, X.erows[q], xt, rt, cur ? V.j0 : V.j1);
    }
    // every wavefront's stores have left, then one lane publishes for the workgroup (MI355X_MICROARCH: visibility, valid forms)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tickets[blockIdx.y] = 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    SpLm st = *stg;
    if (!bad) {
        st.dn2 = team_sumsq(V.delta, B.nv, s_red);
        st.sse_t = team_sumsq(rt, B.m, s_red);
    }
    lm_state_control(st, o);
    if (tid == 0) *stg = st;
}

// refined step, between the two solves: t = -r - J delta (rows), then e = Jt t - lambda delta (columns)
__global__ __launch_bounds__(256) void spt_refine_t_kernel(SpBlock B, SpVals V, const SpLm* __restrict__ lm) {
    const SpLm* st = lm + blockIdx.y;
    if (st->done || st->flag) return;
    V.shift(blockIdx.y);
    const double* jc = st->cur ? V.j1 : V.j0;
    const double* rc = st->cur ? V.r1 : V.r0;
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B.m) return;
    double acc = -rc[row];
    for (uint32_t p = B.jac.jrow_ptr[row]; p < B.jac.jrow_ptr[row + 1]; ++p) acc -= jc[p] * V.delta[B.jcol[p]];
    V.t[row] = acc;
}
__global__ __launch_bounds__(256) void spt_refine_rhs_kernel(SpBlock B, SpVals V, const SpLm* __restrict__ lm) {
    const SpLm* st = lm + blockIdx.y;
    if (st->done || st->flag) return;
    V.shift(blockIdx.y);
    const double* jc = st->cur ? V.j1 : V.j0;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B.nv) return;
    double s = 0.0;
    for (uint32_t p = B.cptr[c]; p < B.cptr[c + 1]; ++p) s += jc[B.cidx[p]] * V.t[B.crow[p]];
    V.e[c] = s - st->lambda * V.delta[c];
}

// K2/K1 at the trial point; the block that finishes last adds up the squares (fixed shapes: same bits whoever it is)
// and takes the trial's decision. start = 1: the start point of the block instead (generation 0, lm.rs:81-112).
template <bool POSE>
__global__ __launch_bounds__(TEAM_THREADS) void spt_eval_kernel(SpRows rows, SpBlock B, SpVals V, SpLm* __restrict__ lm,
                                                                 uint32_t* __restrict__ tickets, fx_lm_opts o, uint32_t start) {
    __shared__ double s_red[TEAM_THREADS];
    __shared__ uint32_t s_last;
    SpLm* stg = lm + blockIdx.y;
    if (!start && stg->done) return;
    V.shift(blockIdx.y);
    const double* sparam = rows.sparam + (size_t)blockIdx.y * V.stride;
    const uint32_t gen = start ? 0u : (stg->cur ^ 1u);
    const bool bad = !start && stg->flag;
    double* rt = gen ? V.r1 : V.r0;
    if (!bad) {
        const uint32_t row = blockIdx.x * TEAM_THREADS + threadIdx.x;
        if (row < B.m) team_eval_row<POSE>(rows, sparam, B.jac, row, gen ? V.xs1 : V.xs0, rt, gen ? V.j1 : V.j0);
    }
    // every wavefront's stores have left, then one lane publishes for the workgroup (MI355X_MICROARCH: visibility, valid forms)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tickets[blockIdx.y] = 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    SpLm st = *stg;
    if (start) {
        lm_state_init(st, team_sumsq(rt, B.m, s_red), o);
    } else {
        if (!bad) {
            st.dn2 = team_sumsq(V.delta, B.nv, s_red);
            st.sse_t = team_sumsq(rt, B.m, s_red);
        }
        lm_state_control(st, o);
    }
    if (threadIdx.x == 0) *stg = st;
}

// end of a block on the two-tier path: values out, the System's sums
__global__ __launch_bounds__(256) void spt_block_end_kernel(SpBlock B, SpVals V, const SpLm* __restrict__ lm, SpAccum* __restrict__ accum,
                                                           uint32_t flags, double* __restrict__ vars_base, const uint64_t* __restrict__ out_off) {
    const uint32_t sys = blockIdx.y;
    const SpLm st = lm[sys];
    V.shift(sys);
    team_block_epilogue(B, V, st.cur, flags, vars_base + out_off[sys], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SpAccum& ac = accum[sys];
        ac.accepted += st.accepted;
        ac.trials += st.trials;
        ac.exit_code = st.exit_code;
        ac.sse0 += st.sse_start;
        ac.sse += st.sse;
    }
}

// ---- around the blocks: start values, the closing check — one launch for the whole group -----------------------------
// off[0..n) = first variable of the group's System k in the batch's arrays, off[n..2n) = its first expression,
// off[2n..3n) = its index in the batch
// start of a group solve: the Systems' start values and parameters into their slabs, the batch's output slice starts
// as a copy of the start values, sums and tickets cleared
__global__ void spg_begin_kernel(const double* __restrict__ vars0_base, const double* __restrict__ param_base, const uint64_t* __restrict__ off,
                                 uint32_t n, uint32_t nvt, uint32_t net, double* __restrict__ vars0, double* __restrict__ param, size_t stride,
                                 double* __restrict__ vars_base, SpAccum* __restrict__ accum, uint32_t* __restrict__ tickets) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, sys = blockIdx.y;
    if (i < nvt) {
        const double v = vars0_base[off[sys] + i];
        vars0[(size_t)sys * stride + i] = v;
        vars_base[off[sys] + i] = v;
    }
    if (i < net) param[(size_t)sys * stride + i] = param_base[off[n + sys] + i];
    if (i == 0) {
        accum[sys] = SpAccum{0, 0, FX_EXIT_SSE, 0, 0.0, 0.0};
        tickets[sys] = 0;
    }
}
// start of a component: the pre-solve snapshot of the working vector (quirk Q2); the component counts
__global__ void spg_component_kernel(const double* __restrict__ xs, double* __restrict__ snap, uint32_t nvt, size_t stride,
                                     SpAccum* __restrict__ accum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, sys = blockIdx.y;
    if (i < nvt) snap[(size_t)sys * stride + i] = xs[(size_t)sys * stride + i];
    if (i == 0) {
        accum[sys].ncomp += 1;
        accum[sys].exit_code = FX_EXIT_SSE;
    }
}

// post-solve check on the unscaled variables (constraints/mod.rs:96-109) and the System's result record: the sums of
// sp_identity_residual_kernel + sp_sumsq_kernel in one workgroup per System (same order: same bits)
template <bool POSE>
__global__ __launch_bounds__(TEAM_THREADS) void spg_finish_kernel(SpRows rows, size_t stride, const double* __restrict__ scal,
                                                                   const SpAccum* __restrict__ accum, const double* __restrict__ vars_base,
                                                                   const uint64_t* __restrict__ off, uint32_t n,
                                                                   fx_result* __restrict__ results) {
    __shared__ double s_red[TEAM_THREADS];
    const uint32_t sys = blockIdx.x;
    const double* x = vars_base + off[sys];
    const double* param = rows.param + (size_t)sys * stride;
    double s = 0.0;
    for (uint32_t e = threadIdx.x; e < rows.net; e += TEAM_THREADS) {
        const int tag = rows.tag[e] & 0x7F;
        const ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
        const uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        expand_vars<POSE>(tag, ff, vars8);
        double v[8], g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[vars8[q]];
        const double r = eval_expression<double, false, false, POSE>(tag, v, param[e], g);
        s += r * r;
    }
    s = block_sum_1024(s, s_red);
    if (threadIdx.x == 0) {
        const SpAccum ac = accum[sys];
        fx_result res{};
        res.accepted = ac.accepted;
        res.trials = ac.trials;
        res.exit = ac.exit_code;
        res.ncomp = ac.ncomp;
        res.scale = scal[(size_t)sys * stride];
        res.sse0 = ac.sse0;
        res.sse = ac.sse;
        res.sse_unscaled = s;
        results[off[2 * (size_t)n + sys]] = res;
    }
}
