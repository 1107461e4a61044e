// Large-component path: Systems whose connected components exceed the one-wavefront limits of the
// fused kernel (more than 64 free variables or 256 expressions) — e.g. BASELINE cfg2, one sketch of
// 5 000 points / 10 000 constraints, or the reference's 64-triangle bench sketch (258 variables).
//
// Same algorithm as the fused kernel (fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193,
// LM step in normal-equation form), organised for sparse problems:
//   host   : structure only (fx_sparse_plan.h) — free-column list, CSR pattern of J, pattern of A = JtJ, a
//            nested-dissection order (the reference runs COLAMD on the host too, qr.rs:118-206), symbolic
//            Cholesky and, for every non-zero of A and of L, the list of products that define it ("gather
//            lists"), and the elimination-tree schedules. One plan per STRUCTURE: the Systems of a batch that
//            share it are solved together (sparse_solve_group).
//   device : all arithmetic and, for Levenberg-Marquardt, all control flow (fx_sparse_team.h): scale/perturb
//            (K0), residual + Jacobian rows (K1/K2), A = JtJ and g = -Jt r by deterministic gathers (K3),
//            numeric sparse Cholesky and the triangular solves by workgroups over the elimination tree (K4),
//            accept / reject (K5), write-back (K6).
// No floating-point work happens on the host. Optimizer::LBfgs keeps its host-side line search
// (sparse_solve_system: two scalars per evaluation).
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <queue>
#include <vector>

#include "fx_decompose.h"
#include "fx_device.h"
#include "fx_expr.h"
#include "fx_lbfgs.h"
#include "fx_sparse.h"
#include "fx_grouped_rows.h"
#include "fx_sparse_plan.h"
#include "fx_wave.h"

namespace fx {

namespace {

// ------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------
struct SpRows {                 // expression data of the whole System (device)
    const uint8_t* tag;         // [net]
    const uint16_t* idx;        // [4*net]
    const double* param;        // [net] unscaled
    double* sparam;             // [net] scaled (distance parameters * 1/scale)
    uint32_t net;
    uint32_t has_pose;          // the System holds pose rows (cluster problems): the POSE builds of the row kernels run it
};

// state of a device-controlled LM loop (the kernels that use it come further down)
struct SpLm {
    double lambda, sse, sse_t, dn2, sse_start;
    uint32_t cur, accepted, trials, outer, exit_code, done, need_form, flag;
};
// scal[0] = scale, scal[1] = 1/scale. Sequential summation in the reference's order (assemble/mod.rs:32-44: all
// variables, then the distance parameters, each ascending) so the scale is bit-identical — the sum itself cannot be
// split, but everything around it can: 256 threads square 4096 values at a time into LDS (+0.0 for expressions without
// a distance: exact), then ONE thread adds them up in order, sixteen LDS values in flight per step. (The first
// version added 64 values per step through v_readlane pairs: 0.83 ms for cfg2's 20 000 values; this one ~0.1 ms.)
// (the body: a workgroup of 256 threads, pointers of ITS System — shared by the kernel below and by spg_prologue_kernel)
__device__ __forceinline__ void sp_scale_body(const double* __restrict__ vars0, uint32_t nvt, const SpRows& rows, double* __restrict__ scal,
                                              int do_scale) {
    constexpr uint32_t CH = 4096;
    __shared__ double sq[CH + 64];
    __shared__ uint32_t cnt_s[256];
    double sum = 0.0;
    uint32_t ndist = 0;
    const uint32_t total = do_scale ? nvt + rows.net : 0u;
    for (uint32_t base = 0; base < total; base += CH) {
        const uint32_t n = min(CH, total - base), n_pad = (n + 31u) & ~31u;
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < n_pad; i += 256u) {  // (+0.0 past the end and for rows without a distance: exact)
            const uint32_t g = base + i;
            double v = 0.0;
            if (i >= n) {
            } else if (g < nvt) {
                v = vars0[g];
            } else {
                const int tag = rows.tag[g - nvt] & 0x7F;
                if (tag == 1 || tag == 4) {  // FX_TAG_PPD, FX_TAG_PLD
                    v = rows.param[g - nvt];
                    mine += 1;
                }
            }
            sq[i] = v * v;
        }
        cnt_s[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            // one chain of dependent adds (the order IS the result): 32 values per step, no test inside, and the next 32
            // are already on their way from LDS while these are added (the last step loads 32 it never adds)
            const double2* sq2 = reinterpret_cast<const double2*>(sq);
            double2 t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = sq2[u];
            for (uint32_t i = 0; i < n_pad; i += 32u) {
                double2 nx[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) nx[u] = sq2[(i + 32u) / 2u + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    sum += t[u].x;
                    sum += t[u].y;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = nx[u];
            }
            for (uint32_t i = 0; i < 256u; ++i) ndist += cnt_s[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double scale = do_scale ? ::sqrt(sum / (double)(nvt + ndist)) : 1.0;
        scal[0] = scale;
        scal[1] = 1.0 / scale;
    }
}
__global__ __launch_bounds__(256) void sp_scale_kernel(const double* __restrict__ vars0, uint32_t nvt, SpRows rows,
                                                      double* __restrict__ scal, int do_scale, size_t stride) {
    // (stride: value arrays of the group's System blockIdx.y lie blockIdx.y * stride doubles further; 0 for one System)
    rows.param += blockIdx.y * stride;
    sp_scale_body(vars0 + blockIdx.y * stride, nvt, rows, scal + blockIdx.y * stride, do_scale);
}

__global__ void sp_init_kernel(const double* __restrict__ vars0, uint32_t nvt, SpRows rows, const double* __restrict__ scal,
                               double* __restrict__ xs_a, double* __restrict__ xs_b, int do_scale, size_t stride) {
    const size_t so = blockIdx.y * stride;
    vars0 += so;
    rows.param += so;
    rows.sparam += so;
    scal += so;
    xs_a += so;
    xs_b += so;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const double recip = scal[1];
    if (i < nvt) {
        double v = vars0[i];
        double x = do_scale ? v * recip : v;
        xs_a[i] = x;
        xs_b[i] = x;
    }
    if (i < rows.net) {
        int tag = rows.tag[i] & 0x7F;
        double p = rows.param[i];
        if (do_scale && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) p = recip * p;
        rows.sparam[i] = p;
    }
}

// LCG skip-ahead: state after n steps of s <- s*A + C is s*A^n + C*(A^n-1)/(A-1) (mod 2^32),
// computed by squaring the affine map.
__device__ __forceinline__ uint32_t lcg_skip(uint32_t s, uint32_t n) {
    uint32_t a = 1664525u, c = 1013904223u;  // fiksi/src/rand.rs:24-30
    uint32_t acc_a = 1u, acc_c = 0u;
    while (n) {
        if (n & 1u) {
            acc_c = acc_c * a + c;
            acc_a = acc_a * a;
        }
        c = c * a + c;
        a = a * a;
        n >>= 1;
    }
    return s * acc_a + acc_c;
}

// assemble/mod.rs:113-124: free variable k (ascending) takes draws 2k and 2k+1 of the shared Rng.
__global__ void sp_perturb_kernel(const uint32_t* __restrict__ fvar, uint32_t nv, uint32_t rng_state,
                                  double* __restrict__ xs_a, double* __restrict__ xs_b, size_t stride) {
    xs_a += blockIdx.y * stride;
    xs_b += blockIdx.y * stride;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    uint32_t st = lcg_skip(rng_state, 2u * k);
    st = st * 1664525u + 1013904223u;
    double f1 = (1.0 / 4294967295.0) * (double)st;
    st = st * 1664525u + 1013904223u;
    double f2 = (1.0 / 4294967295.0) * (double)st;
    uint32_t vi = fvar[k];
    double x = xs_a[vi];
    x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
    xs_a[vi] = x;
    xs_b[vi] = x;
}

struct SpJac {                   // J of one component (device)
    const uint32_t* rows;        // [m] expression id of row
    const uint32_t* jrow_ptr;    // [m+1]
    const uint32_t* jslot;       // [m] 8 x 4-bit slot of each gradient entry (0xF = not a free column)
    uint32_t m;
    int overwrite;               // entries sharing a column: 0 = summed (sparse J, sparse_col_mat.rs:710-711),
                                 // 1 = the last one wins (dense J of L-BFGS, expressions.rs:993-1008) — quirk Q4
};

// sum over the 1024 threads of a workgroup: a fixed binary tree (deterministic); every thread gets the result
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    double out = sh[0];
    __syncthreads();
    return out;
}

struct SpChol {                  // symbolic factor (device)
    const uint32_t* lcolptr;     // [nv+1]; first entry of every column is its diagonal
    const uint32_t* lrow;        // [nnzL]
    const int32_t* l2a;          // [nnzL] index into A or -1 (fill-in)
    const uint32_t* lpair_ptr;   // [nnzL+1]
    const uint32_t* lpairs;      // [2*npairs] indices into L
    const uint32_t* lpair_k;     // [npairs] the entry of L a product belongs to (the inverse of lpair_ptr)
    uint32_t nv;
};

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct SpRowsOfL {               // L by rows (strictly lower part), for the forward sweep
    const uint32_t* rptr;        // [nv+1]
    const uint32_t* ridx;        // index of the entry in L's value array
    const uint32_t* rcol;        // its column
};

#include "fx_sparse_team.h"
#include "fx_front.h"

// Everything before the first block of a System of ONE component, in one launch (a workgroup per System of the group): what
// spg_begin_kernel, sp_scale_kernel, sp_init_kernel, sp_perturb_kernel and spg_component_kernel do one after the other — each of
// those launches costs the host ~7 us to issue, which a lone 258-variable sketch (0.18 ms a solve) sees as a fifth of its time.
// Same statements per value in the same order: same bits.
__global__ __launch_bounds__(256) void spg_prologue_kernel(const double* __restrict__ vars0_base, const double* __restrict__ param_base,
                                                           const uint64_t* __restrict__ off, uint32_t n, uint32_t nvt, double* __restrict__ vars0,
                                                           SpRows rows, size_t stride, double* __restrict__ vars_base, SpAccum* __restrict__ accum,
                                                           uint32_t* __restrict__ tickets, double* __restrict__ scal, double* __restrict__ xs_a,
                                                           double* __restrict__ xs_b, double* __restrict__ snap, int do_scale,
                                                           const uint32_t* __restrict__ fvar, uint32_t nfv, uint32_t rng_state, int perturb) {
    const uint32_t sys = blockIdx.y, net = rows.net;
    const size_t so = (size_t)sys * stride;
    vars0 += so;
    scal += so;
    xs_a += so;
    xs_b += so;
    snap += so;
    double* param = const_cast<double*>(rows.param) + so;
    double* sparam = const_cast<double*>(rows.sparam) + so;
    rows.param = param;
    rows.sparam = sparam;
    // spg_begin_kernel
    const uint64_t v_at = off[sys], e_at = off[n + sys];
    for (uint32_t i = threadIdx.x; i < nvt; i += 256u) {
        const double v = vars0_base[v_at + i];
        vars0[i] = v;
        vars_base[v_at + i] = v;
    }
    for (uint32_t i = threadIdx.x; i < net; i += 256u) param[i] = param_base[e_at + i];
    if (threadIdx.x == 0) {
        accum[sys] = SpAccum{0, 0, FX_EXIT_SSE, 1, 0.0, 0.0};  // (spg_component_kernel: the one component counts)
        tickets[sys] = 0;
    }
    __syncthreads();
    // sp_scale_kernel
    sp_scale_body(vars0, nvt, rows, scal, do_scale);
    __syncthreads();
    // sp_init_kernel
    const double recip = scal[1];
    for (uint32_t i = threadIdx.x; i < nvt; i += 256u) {
        const double v = vars0[i];
        const double x = do_scale ? v * recip : v;
        xs_a[i] = x;
        xs_b[i] = x;
    }
    for (uint32_t i = threadIdx.x; i < net; i += 256u) {
        const int tag = rows.tag[i] & 0x7F;
        double p = param[i];
        if (do_scale && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) p = recip * p;
        sparam[i] = p;
    }
    __syncthreads();
    // sp_perturb_kernel
    if (perturb) {
        for (uint32_t k = threadIdx.x; k < nfv; k += 256u) {
            uint32_t st = lcg_skip(rng_state, 2u * k);
            st = st * 1664525u + 1013904223u;
            const double f1 = (1.0 / 4294967295.0) * (double)st;
            st = st * 1664525u + 1013904223u;
            const double f2 = (1.0 / 4294967295.0) * (double)st;
            const uint32_t vi = fvar[k];
            double x = xs_a[vi];
            x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
            xs_a[vi] = x;
            xs_b[vi] = x;
        }
        __syncthreads();
    }
    // spg_component_kernel
    for (uint32_t i = threadIdx.x; i < nvt; i += 256u) snap[i] = xs_a[i];
}

// ------------------------------------------------------------------------------------------------
// host: structure
// ------------------------------------------------------------------------------------------------
using sparse_plan::ComponentPlan;
using sparse_plan::TeamSchedule;
using sparse_plan::plan_component;

template <typename T>
struct DevArr {
    T* p = nullptr;
    size_t n = 0;
};

// Device memory of one System's solve: a few large hipMalloc chunks handed out by bumping a pointer.
// hipMalloc / hipFree synchronise the whole device, and a SinglePass solve creates dozens of small arrays
// for each of thousands of blocks — and several Systems may be solved at once from different host
// threads — so the per-array calls are what must go.
struct Arena {
    struct Chunk { char* base; size_t size; };
    struct Mark { size_t cur, used; };
    std::vector<Chunk> chunks;
    size_t cur = 0, used = 0;
    void* take(size_t bytes, hipError_t& err) {
        bytes = (std::max<size_t>(bytes, 1) + 255u) & ~size_t(255);
        for (; cur < chunks.size(); ++cur, used = 0)
            if (used + bytes <= chunks[cur].size) {
                void* p = chunks[cur].base + used;
                used += bytes;
                return p;
            }
        void* d = nullptr;
        const size_t size = std::max<size_t>(bytes, size_t(4) << 20);
        chunks.reserve(chunks.size() + 1);  // (so that the chunk cannot be lost between hipMalloc and the list)
        err = hipMalloc(&d, size);
        if (err != hipSuccess) return nullptr;
        chunks.push_back({static_cast<char*>(d), size});
        cur = chunks.size() - 1;
        used = bytes;
        return d;
    }
    Mark mark() const { return {cur, used}; }
    void reset(Mark m) { cur = m.cur; used = m.used; }
    size_t bytes() const {
        size_t t = 0;
        for (const Chunk& c : chunks) t += c.size;
        return t;
    }
    // nothing of the arena is in use (the stream is synced, every Pool has ended): memory beyond `keep` goes back to the driver
    void trim(size_t keep) {
        if (bytes() <= keep) return;
        for (auto& c : chunks) (void)hipFree(c.base);
        chunks.clear();
        cur = used = 0;
    }
    ~Arena() {
        for (auto& c : chunks) (void)hipFree(c.base);
    }
};

struct Pool {  // the arrays of one scope (System or block); released together when it ends (stream synced by then)
    Arena* arena;
    Arena::Mark start;
    hipStream_t stream = nullptr;
    hipError_t err = hipSuccess;
    explicit Pool(Arena* a) : arena(a), start(a->mark()) {}
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    template <typename T>
    T* up(const std::vector<T>& h) {
        T* d = alloc<T>(h.size());
        if (d && !h.empty() && err == hipSuccess)
            err = hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream);
        return d;
    }
    template <typename T>
    T* alloc(size_t n) {
        if (err != hipSuccess) return nullptr;
        return static_cast<T*>(arena->take(n * sizeof(T), err));
    }
    ~Pool() { arena->reset(start); }
};



// Structure of one block on the device (index arrays only; what plan_component produced, uploaded).
struct BlockOnDevice {
    ComponentPlan P;
    SpBlock dev{};             // device view; dev.sched = the whole factor as one segment
    SpTeamSched parts{};       // parts + top (large factors only)
    bool has_parts = false;
    SpPartsX px{};             // ... and its LDS build (fx_sparse_plan.h: PartsExtra)
    const uint32_t* solo_blob = nullptr;  // the whole factor's index data for LDS (small factors)
    uint32_t solo_blob_words = 0;
    uint32_t n_fslots = 0, n_bslots = 0;
    size_t lds_part_bytes = 0, lds_top_bytes = 0;
    double plan_ms = 0.0;
    // the multifrontal build (fx_front.h; 0 bytes: the structure has a front beyond a row of lanes, or its data no room in LDS)
    const uint32_t* mf_solo_blob = nullptr;
    uint32_t mf_solo_words = 0, mf_solo_red = 0, mf_solo_rows = 0;
    int mf_solo_mode = 0;     // 1: its L blocks in global memory (they do not fit LDS beside the rest); 2: the contribution blocks too
    size_t mf_solo_lds = 0;
    MfParts mfp{};
    size_t mf_up_lds = 0, mf_down_lds = 0;
    uint32_t mf_l_doubles = 0, mf_gu_doubles = 0;  // global memory per System: the parts' columns of L, their roots' contributions
};

struct CompOnDevice {          // one connected component that holds variables (assemble/mod.rs:81-111)
    uint32_t nfv = 0;          // its free variables, ascending: the perturbation's order
    uint32_t* d_fvar = nullptr;
    uint32_t first_block = 0, n_blocks = 0;
};

}  // namespace

// Plan of one STRUCTURE (the fixed flags, tags, fields and components of a System) in one decomposer mode: the
// components in visiting order, their blocks, every index array on the device. Values never enter it, so every System
// of the structure — of this call or a later one — is solved with it; ordering, symbolic factorisation, gather lists and
// their upload (12 ms for cfg2) are paid once.
struct SparsePlanCache {
    Arena arena;                                     // holds the blocks' index arrays
    std::unique_ptr<Pool> pool;
    std::vector<CompOnDevice> comps;
    std::vector<std::unique_ptr<BlockOnDevice>> blocks;  // in visiting order
    uint32_t max_m = 0, max_nv = 0, max_nnz_j = 0, max_nnz_a = 0, max_nnz_l = 0, max_fslots = 0, max_bslots = 0;
    uint32_t max_mf_l = 0, max_mf_gu = 0;            // the multifrontal build's global storage (fx_front.h)
    // a resident batch's group (sparse_cache_keep_slab(SIZE_MAX)): where its Systems sit in the batch's arrays, on the device — the
    // same every solve, so uploaded once (the host copy stays for as long as the copy may be in flight)
    std::vector<uint32_t> off_systems;
    std::vector<uint64_t> off_host;
    uint64_t* d_off = nullptr;
    Arena values;                                    // the group solves' value slabs, kept between calls (one solve at a time)
    size_t keep_values = size_t(256) << 20;          // ... up to this many bytes (sparse_cache_keep_slab)
    bool ready = false;
};
SparsePlanCache* sparse_cache_new() { return new SparsePlanCache(); }
void sparse_cache_keep_slab(SparsePlanCache* c, size_t bytes) {
    if (c) c->keep_values = bytes;
}
void sparse_cache_free(SparsePlanCache* c) { delete c; }
bool sparse_cache_ready(const SparsePlanCache* c) { return c && c->ready; }

namespace {

inline dim3 grid_for(uint32_t n, uint32_t block = 256) { return dim3((n + block - 1) / block ? (n + block - 1) / block : 1); }
inline dim3 grid_for2(uint32_t n, uint32_t ny, uint32_t block = 256) { return dim3((n + block - 1) / block ? (n + block - 1) / block : 1, ny); }

SpTeamSched upload_schedule(Pool& sp, const ComponentPlan& P, const TeamSchedule& t) {
    SpTeamSched d{};
    d.seg_lev = sp.up(t.seg_lev);
    d.wptr = sp.up(t.wptr);
    d.cols = sp.up(t.cols);
    std::vector<ColDesc> cd(t.cols.size());
    for (size_t q = 0; q < cd.size(); ++q) {
        const uint32_t j = t.cols[q];
        ColDesc& c = cd[q];
        c.j = j;
        c.beg = P.lcolptr[j];
        c.end = P.lcolptr[j + 1];
        c.rbeg = P.rptr[j];
        c.rend = P.rptr[j + 1];
        c.pbeg = P.lpair_ptr[c.beg];
        c.pend0 = P.lpair_ptr[std::min(c.beg + 64u, c.end)];
        c.pad = 0;
    }
    d.cdesc = sp.up(cd);
    d.cdesc_mid = nullptr;
    d.nparts = t.nparts;
    return d;
}

// Fills `cache` with the plan of System s (structure only): components in order, the blocks of each — the whole
// component, or its SinglePass decomposition (analyze/graph/equations.rs) — planned and uploaded on `stream`.
hipError_t ensure_plan(const fx_batch* b, uint32_t s, bool single_pass, hipStream_t stream, SparsePlanCache* cache) {
    if (cache->ready) return hipSuccess;
    cache->comps.clear();  // (an earlier attempt may have failed half-way)
    cache->blocks.clear();
    cache->max_m = cache->max_nv = cache->max_nnz_j = cache->max_nnz_a = cache->max_nnz_l = cache->max_fslots = cache->max_bslots = 0;
    cache->max_mf_l = cache->max_mf_gu = 0;
    if (!cache->pool) cache->pool.reset(new Pool(&cache->arena));
    Pool& sp = *cache->pool;
    sp.stream = stream;
    const uint32_t v0 = b->var_off[s], nvt = b->var_off[s + 1] - v0;
    const uint32_t e0 = b->expr_off[s], net = b->expr_off[s + 1] - e0;
    uint32_t ncomp = 0;
    for (uint32_t i = 0; i < nvt; ++i) {
        const uint16_t c = b->var_comp ? b->var_comp[v0 + i] : 0;
        if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
    }
    for (uint32_t i = 0; i < net; ++i) {
        const uint16_t c = b->expr_comp ? b->expr_comp[e0 + i] : 0;
        if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
    }
    Incidence inc;
    std::unique_ptr<SinglePassDecomposer> decomposer;
    for (uint32_t c = 0; c < ncomp; ++c) {
        std::vector<uint32_t> crow_ids, fvar;
        bool any_var = false;
        for (uint32_t i = 0; i < nvt; ++i) {
            const uint16_t vc = b->var_comp ? b->var_comp[v0 + i] : 0;
            if (vc != c) continue;
            any_var = true;
            if (!b->var_fixed[v0 + i]) fvar.push_back(i);
        }
        if (!any_var) continue;
        for (uint32_t i = 0; i < net; ++i)
            if ((b->expr_comp ? b->expr_comp[e0 + i] : 0) == c) crow_ids.push_back(i);
        UnitList units;
        if (single_pass) {
            if (!decomposer) {
                inc.build(nvt, net, b->expr_tag + e0, b->expr_idx + 4 * (size_t)e0);
                decomposer.reset(new SinglePassDecomposer(inc));
            }
            decomposer->run(fvar, units);
        } else {
            units.rows = crow_ids;
            units.vars = fvar;
            units.row_off.push_back((uint32_t)crow_ids.size());
            units.var_off.push_back((uint32_t)fvar.size());
        }
        CompOnDevice comp;
        comp.nfv = (uint32_t)fvar.size();
        comp.d_fvar = sp.up(fvar);
        comp.first_block = (uint32_t)cache->blocks.size();
        comp.n_blocks = units.count();
        cache->comps.push_back(comp);
        for (uint32_t u = 0; u < units.count(); ++u) {
            cache->blocks.emplace_back(new BlockOnDevice());
            BlockOnDevice* blk = cache->blocks.back().get();
            const auto t_plan0 = std::chrono::steady_clock::now();
            plan_component(b, s,
                           std::vector<uint32_t>(units.rows.begin() + units.row_off[u], units.rows.begin() + units.row_off[u + 1]),
                           std::vector<uint32_t>(units.vars.begin() + units.var_off[u], units.vars.begin() + units.var_off[u + 1]),
                           blk->P);
            blk->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
            const ComponentPlan& Q = blk->P;
            SpBlock& d = blk->dev;
            d.fvar = sp.up(Q.fvar);
            d.perm = sp.up(Q.perm);
            d.jac.rows = sp.up(Q.rows);
            d.jac.jrow_ptr = sp.up(Q.jrow_ptr);
            d.jac.jslot = sp.up(Q.jslot);
            d.jac.m = Q.m;
            d.jac.overwrite = 0;
            d.apair_ptr = sp.up(Q.apair_ptr);
            d.apairs = sp.up(Q.apairs);
            d.cptr = sp.up(Q.cptr);
            d.cidx = sp.up(Q.cidx);
            d.crow = sp.up(Q.crow);
            d.jcol = sp.up(Q.jcol);
            d.chol.lcolptr = sp.up(Q.lcolptr);
            d.chol.lrow = sp.up(Q.lrow);
            d.chol.l2a = sp.up(Q.l2a);
            d.chol.lpair_ptr = sp.up(Q.lpair_ptr);
            d.chol.lpairs = sp.up(Q.lpairs);
            d.chol.lpair_k = sp.up(Q.lpair_k);
            d.chol.nv = Q.nv;
            d.lrows.rptr = sp.up(Q.rptr);
            d.lrows.ridx = sp.up(Q.ridx);
            d.lrows.rcol = sp.up(Q.rcol);
            d.sched = upload_schedule(sp, Q, Q.solo);
            {
                std::vector<uint32_t> a2l(Q.nnz_a, 0);
                for (uint32_t k = 0; k < Q.nnz_l; ++k)
                    if (Q.l2a[k] >= 0) a2l[(uint32_t)Q.l2a[k]] = k;
                d.a2l = sp.up(a2l);
            }
            {
                std::vector<uint32_t> along;  // long gather lists: entries of A first, then columns of the right-hand side
                for (uint32_t k = 0; k < Q.nnz_a; ++k)
                    if (Q.apair_ptr[k + 1] - Q.apair_ptr[k] > FORM_LONG) along.push_back(k);
                d.n_along = (uint32_t)along.size();
                for (uint32_t c2 = 0; c2 < Q.nv; ++c2)
                    if (Q.cptr[c2 + 1] - Q.cptr[c2] > FORM_LONG) along.push_back(c2);
                d.n_clong = (uint32_t)along.size() - d.n_along;
                d.along = sp.up(along);
            }
            d.m = Q.m;
            d.nv = Q.nv;
            d.nnz_a = Q.nnz_a;
            d.nnz_l = Q.nnz_l;
            if (!Q.solo_blob.empty()) {
                blk->solo_blob = sp.up(Q.solo_blob.words);
                blk->solo_blob_words = (uint32_t)Q.solo_blob.words.size();
            }
            blk->has_parts = !Q.parts.empty();
            if (blk->has_parts) {
                blk->parts = upload_schedule(sp, Q, Q.parts);
                blk->parts.cdesc_mid = nullptr;
                // the LDS build: the same lists, the top's columns described by their own products and row entries only
                const sparse_plan::PartsExtra& E = Q.px;
                const uint32_t np = Q.parts.nparts, ctop = E.seg_col[np], etop = E.seg_ent[np];
                SpPartsX& X = blk->px;
                X.nparts = np;
                X.seg_col = sp.up(E.seg_col);
                X.seg_ent = sp.up(E.seg_ent);
                X.frun_ptr = sp.up(E.frun_ptr);
                X.frun = sp.up(E.frun);
                X.fslot_ptr = sp.up(E.fslot_ptr);
                X.brun_ptr = sp.up(E.brun_ptr);
                X.brun = sp.up(E.brun);
                X.bslot_ptr = sp.up(E.bslot_ptr);
                X.blobs = sp.up(Q.parts_blobs.words);
                X.blob_off = sp.up(Q.parts_blobs.seg_off);
                X.cmid = sp.up(E.cmid);
                X.erow_ptr = sp.up(E.erow_ptr);
                X.erows = sp.up(E.erows);
                blk->n_fslots = E.fslot_ptr.back();
                blk->n_bslots = E.bslot_ptr.back();
                // values (entries of L, the vector) + the segment's blob; 0: no LDS build (a segment beyond 16-bit local indices)
                blk->lds_part_bytes = Q.parts_blobs.empty() ? 0 : (((size_t)E.max_part_ent + 1u) & ~size_t(1)) * 8 + (((size_t)E.max_part_cols + 1u) & ~size_t(1)) * 8 + (size_t)Q.parts_blobs.max_words * 4;
                blk->lds_top_bytes = Q.parts_blobs.empty() ? 0 : (((size_t)(Q.nnz_l - etop) + 1u) & ~size_t(1)) * 8 + (((size_t)(Q.nv - ctop) + 1u) & ~size_t(1)) * 8 + (size_t)Q.parts_blobs.top_words * 4;
                cache->max_fslots = std::max(cache->max_fslots, blk->n_fslots);
                cache->max_bslots = std::max(cache->max_bslots, blk->n_bslots);
            }
            // the multifrontal build: one workgroup per System when everything of the factor fits its LDS ...
            constexpr size_t MF_LDS_MAX = size_t(156) << 10;
            const size_t even = ~size_t(1);
            auto tiles_bytes = [](uint32_t rows2, uint32_t) { return (((size_t)rows2 * MF_TILE + 1) & ~size_t(1)) * 8; };
            if (Q.fronts_solo.ok) {
                const sparse_plan::FrontPlan& fp = Q.fronts_solo;
                uint32_t red = 64u;
                while (red < std::max(Q.m, Q.nv) && red < 1024u) red <<= 1;
                const size_t base = (size_t)fp.words.size() * 4 + (((size_t)Q.nnz_a + 1) & even) * 8 + 2 * (((size_t)Q.nv + 1) & even) * 8 + (size_t)red * 8;
                const size_t l_bytes = (size_t)fp.max_l_doubles * 8, u_bytes = ((size_t)fp.max_u_doubles + MF_LS + 1) * 8;
                // everything in LDS when a fair number of staging tiles still fits beside it; else the L blocks in global memory; else
                // the contribution blocks too
                const uint32_t want = std::min(16u, mf_tile_rows(fp.max_level_fronts));
                for (int mode = 0; mode < 3 && !blk->mf_solo_lds; ++mode) {
                    const size_t fixed = base + (mode < 1 ? l_bytes : 0) + (mode < 2 ? u_bytes : 0);
                    uint32_t rows2 = mf_tile_rows(fp.max_level_fronts);  // as many tiles as the widest level takes, or as fit
                    while (rows2 > 4u && fixed + tiles_bytes(rows2, fp.max_ts) > MF_LDS_MAX) rows2 -= 4u;
                    if (fixed + tiles_bytes(rows2, fp.max_ts) > MF_LDS_MAX || (mode < 2 && rows2 < want)) continue;
                    blk->mf_solo_blob = sp.up(fp.words);
                    blk->mf_solo_words = (uint32_t)fp.words.size();
                    blk->mf_solo_red = red;
                    blk->mf_solo_rows = rows2;
                    blk->mf_solo_mode = mode;
                    blk->mf_solo_lds = fixed + tiles_bytes(rows2, fp.max_ts);
                    if (mode >= 1) cache->max_mf_l = std::max(cache->max_mf_l, fp.max_l_doubles);
                    if (mode >= 2) cache->max_mf_gu = std::max(cache->max_mf_gu, fp.max_u_doubles + 2 * MF_LS);
                }
            }
            // ... a large System alone: its parts side by side and the top (the same segments as the walkers' parts schedule)
            if (blk->has_parts && Q.fronts_parts.ok) {
                const sparse_plan::FrontPlan& fp = Q.fronts_parts;
                const uint32_t np = Q.parts.nparts;
                std::vector<uint32_t> seg_l((size_t)fp.nseg + 1, 0);
                uint32_t widest_part = 0, widest_top = 0, ts_part = 3, ts_top = 3;
                size_t fixed_part = 0, fixed_top = 0, down = 0;
                for (uint32_t sgm = 0; sgm < fp.nseg; ++sgm) {
                    const uint32_t* w = fp.words.data() + fp.seg_off[sgm];
                    seg_l[sgm + 1] = seg_l[sgm] + w[11];
                    (sgm == np ? widest_top : widest_part) = std::max(sgm == np ? widest_top : widest_part, w[14]);
                    (sgm == np ? ts_top : ts_part) = std::max(sgm == np ? ts_top : ts_part, w[16]);
                    // up: the blob, the entries of A, the right-hand side, the contribution slots; down: the blob, x, the L blocks
                    const size_t up = (size_t)w[13] * 4 + (((size_t)w[2] + 1) & even) * 8 + (((size_t)w[3] + 1) & even) * 8 + ((size_t)w[12] + MF_LS + 1) * 8;
                    const size_t dn = (size_t)w[13] * 4 + (((size_t)w[3] + 1) & even) * 8 + (size_t)w[11] * 8;
                    if (sgm == np) {
                        fixed_top = up;  // (the top sweeps down inside the same launch: top_dn below)
                    } else {
                        fixed_part = std::max(fixed_part, up);
                        down = std::max(down, dn);
                    }
                }
                uint32_t rows_part = mf_tile_rows(widest_part), rows_top = mf_tile_rows(widest_top);
                while (rows_part > 4u && fixed_part + tiles_bytes(rows_part, ts_part) > MF_LDS_MAX) rows_part -= 4u;
                while (rows_top > 4u && fixed_top + tiles_bytes(rows_top, ts_top) > MF_LDS_MAX) rows_top -= 4u;
                size_t top_dn = 0;
                {
                    const uint32_t* w = fp.words.data() + fp.seg_off[np];
                    top_dn = (size_t)w[13] * 4 + (((size_t)w[3] + 1) & even) * 8 + (size_t)w[11] * 8;
                }
                const size_t up = std::max(std::max(fixed_part + tiles_bytes(rows_part, ts_part), fixed_top + tiles_bytes(rows_top, ts_top)), top_dn);
                down = std::max<size_t>(down, 8192);  // (the last block's sums)
                if (up <= MF_LDS_MAX && down <= MF_LDS_MAX) {
                    MfParts& X = blk->mfp;
                    X.blobs = sp.up(fp.words);
                    X.blob_off = sp.up(fp.seg_off);
                    X.seg_l = sp.up(seg_l);
                    X.erow_ptr = blk->px.erow_ptr;
                    X.erows = blk->px.erows;
                    X.nparts = np;
                    X.top_rows = rows_top;
                    X.part_rows = rows_part;
                    X.prof = nullptr;
                    blk->mf_up_lds = up;
                    blk->mf_down_lds = down;
                    blk->mf_l_doubles = seg_l[fp.nseg];
                    blk->mf_gu_doubles = fp.global_u_doubles + 2 * MF_LS;
                    cache->max_mf_l = std::max(cache->max_mf_l, blk->mf_l_doubles);
                    cache->max_mf_gu = std::max(cache->max_mf_gu, blk->mf_gu_doubles);
                }
            }
            if (sp.err != hipSuccess) return sp.err;
            cache->max_m = std::max(cache->max_m, Q.m);
            cache->max_nv = std::max(cache->max_nv, Q.nv);
            cache->max_nnz_j = std::max(cache->max_nnz_j, Q.nnz_j);
            cache->max_nnz_a = std::max(cache->max_nnz_a, Q.nnz_a);
            cache->max_nnz_l = std::max(cache->max_nnz_l, Q.nnz_l);
        }
    }
    if (sp.err != hipSuccess) return sp.err;
    // plan_component's vectors leave this frame with the cache, but a failed call must not leave copies in flight
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    cache->ready = true;
    return hipSuccess;
}

constexpr size_t TEAM_LDS_VALUES_MAX = size_t(140) << 10;  // of the CU's 160 KB

template <bool POSE, bool LDSV, bool BLOB, int NW>
hipError_t launch_team_t(uint32_t n, size_t lds_bytes, hipStream_t stream, const SpRows& rows, const SpBlock& B, const SpVals& V, SpAccum* accum,
                         const fx_lm_opts& o, uint32_t flags, double* vars_base, const uint64_t* off, uint32_t lds_l, uint32_t lds_v,
                         const uint32_t* blob, uint32_t blob_words, unsigned long long* prof) {
    // (per instantiation AND device: the attribute belongs to the function on the current device; groups of different
    // structures launch from several host threads)
    static std::atomic<uint32_t> raised_on{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    // a narrow team's partial sums of team_sumsq: a power of two that covers the block's rows and columns, behind the rest
    // (up to 8 KB — the sixteen-wavefront team keeps them in a static buffer of that size)
    constexpr size_t RED_MAX = NW != TEAM_NWAVES ? size_t(TEAM_THREADS) * 8u : 0u;
    uint32_t red_n = 0, red_off = 0;
    if (NW != TEAM_NWAVES) {
        red_n = 64u;
        while (red_n < std::max(B.m, B.nv) && red_n < (uint32_t)TEAM_THREADS) red_n <<= 1;
        red_off = (uint32_t)((lds_bytes + 7u) / 8u);
        lds_bytes = (size_t)red_off * 8u + (size_t)red_n * 8u;
        if (lds_bytes > TEAM_LDS_VALUES_MAX + RED_MAX) return hipErrorInvalidValue;
    }
    // the function's dynamic-LDS limit covers everything a launch of this instantiation may ask for: the values (the caller
    // takes the LDSV build up to TEAM_LDS_VALUES_MAX) AND a narrow team's sums behind them
    if (LDSV && !(raised_on.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_lm_team_kernel<POSE, LDSV, BLOB, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(TEAM_LDS_VALUES_MAX + RED_MAX));
        if (e != hipSuccess) return e;
        raised_on.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((sp_lm_team_kernel<POSE, LDSV, BLOB, NW>), dim3(n), dim3(64 * NW), lds_bytes, stream, rows, B, V, accum, o, flags, vars_base, off,
                       lds_l, lds_v, blob, blob_words, prof, red_off, red_n);
    return hipGetLastError();
}
// Wavefronts per System: 16 run one System (or a few hundred) at the lowest latency; a batch of many Systems takes fewer —
// more workgroups to a CU (two of 16 wavefronts fit), narrower barriers. Measured on the reference's hinged-triangle sketches
// (tools/hinged_batch.py, ms per batch at 16 / 8 / 4 / 2 / 1 wavefronts): 66 variables x 20 000: 10.4 / 6.6 / 4.8 / 3.8 / 4.5
// (x 256: 0.19 / - / 0.26 / 0.33 / 0.47); 126 variables x 8 192: 4.6 / 3.1 / 2.2 / 2.9 / 4.8; 258 variables x 2 048: 1.8 / 1.4 /
// 1.8 / 2.6 / -. FIKSI_AMD_TEAM_WAVES=1|2|4|8|16 pins one (measurements). Which wavefront walks a column changes nothing in
// its arithmetic: same bits at any width (tests/test_gpu_team.py).
static int team_waves_for(uint32_t n_systems, uint32_t nv) {
    static const int forced = [] { const char* e = std::getenv("FIKSI_AMD_TEAM_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
    if (n_systems < 768u) return 16;
    return nv <= 100u ? 2 : nv <= 200u ? 4 : nv <= 640u ? 8 : 16;
}
hipError_t launch_team(bool pose, bool ldsv, bool blob_in_lds, uint32_t n, size_t lds_bytes, hipStream_t stream, const SpRows& rows, const SpBlock& B,
                       const SpVals& V, SpAccum* accum, const fx_lm_opts& o, uint32_t flags, double* vars_base, const uint64_t* off, uint32_t lds_l,
                       uint32_t lds_v, const uint32_t* blob, uint32_t blob_words, unsigned long long* prof) {
#define FX_TEAM(P, L, K, W) launch_team_t<P, L, K, W>(n, lds_bytes, stream, rows, B, V, accum, o, flags, vars_base, off, lds_l, lds_v, blob, blob_words, prof)
    const int tw = pose ? 16 : team_waves_for(n, B.nv);
    if (tw == 4) return !ldsv ? FX_TEAM(false, false, false, 4) : blob_in_lds ? FX_TEAM(false, true, true, 4) : FX_TEAM(false, true, false, 4);
    if (tw == 1) return !ldsv ? FX_TEAM(false, false, false, 1) : blob_in_lds ? FX_TEAM(false, true, true, 1) : FX_TEAM(false, true, false, 1);
    if (tw == 2) return !ldsv ? FX_TEAM(false, false, false, 2) : blob_in_lds ? FX_TEAM(false, true, true, 2) : FX_TEAM(false, true, false, 2);
    if (tw == 8) return !ldsv ? FX_TEAM(false, false, false, 8) : blob_in_lds ? FX_TEAM(false, true, true, 8) : FX_TEAM(false, true, false, 8);
    if (pose) return !ldsv ? FX_TEAM(true, false, false, 16) : blob_in_lds ? FX_TEAM(true, true, true, 16) : FX_TEAM(true, true, false, 16);
    return !ldsv ? FX_TEAM(false, false, false, 16) : blob_in_lds ? FX_TEAM(false, true, true, 16) : FX_TEAM(false, true, false, 16);
#undef FX_TEAM
}

// the LDS builds of the parts kernels may ask for more than the default 64 KB of dynamic LDS
hipError_t raise_lds_limits() {
    static std::atomic<uint32_t> raised_on{0};  // bit d: done on device d
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    if (raised_on.load(std::memory_order_relaxed) & bit) return hipSuccess;
    hipError_t e = hipSuccess;
    for (const void* f : {reinterpret_cast<const void*>(&sptl_parts_up_kernel<false>), reinterpret_cast<const void*>(&sptl_parts_up_kernel<true>),
                          reinterpret_cast<const void*>(&sptl_parts_down_kernel<false>), reinterpret_cast<const void*>(&sptl_parts_down_kernel<true>)})
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAM_LDS_VALUES_MAX);
    if (e == hipSuccess) raised_on.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

// ... and so may the multifrontal kernels (fx_front.h)
hipError_t raise_mf_lds_limits() {
    static std::atomic<uint32_t> raised_on{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    if (raised_on.load(std::memory_order_relaxed) & bit) return hipSuccess;
    hipError_t e = hipSuccess;
    for (const void* f : {reinterpret_cast<const void*>(&mf_parts_up_kernel<false>), reinterpret_cast<const void*>(&mf_parts_up_kernel<true>),
                          reinterpret_cast<const void*>(&mf_parts_down_kernel<false>), reinterpret_cast<const void*>(&mf_parts_down_kernel<true>),
                          reinterpret_cast<const void*>(&mf_lm_solo_kernel<false, 0>), reinterpret_cast<const void*>(&mf_lm_solo_kernel<true, 0>),
                          reinterpret_cast<const void*>(&mf_lm_solo_kernel<false, 1>), reinterpret_cast<const void*>(&mf_lm_solo_kernel<true, 1>),
                          reinterpret_cast<const void*>(&mf_lm_solo_kernel<false, 2>), reinterpret_cast<const void*>(&mf_lm_solo_kernel<true, 2>)})
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);  // (beside a few bytes of static LDS: 160 KB in all)
    if (e == hipSuccess) raised_on.fetch_or(bit, std::memory_order_relaxed);
    return e;
}
hipError_t launch_mf_solo(bool pose, uint32_t n, size_t lds_bytes, hipStream_t stream, const SpRows& rows, const SpBlock& B, const SpVals& V, SpAccum* accum,
                          const fx_lm_opts& o, uint32_t flags, double* vars_base, const uint64_t* off, const uint32_t* blob, uint32_t blob_words, uint32_t red_n, uint32_t trows,
                          int mode, double* u_glob, size_t u_stride, unsigned long long* prof, fx_result* results) {
    hipError_t e = raise_mf_lds_limits();
    if (e != hipSuccess) return e;
#define FX_MF_SOLO(P, G) hipLaunchKernelGGL((mf_lm_solo_kernel<P, G>), dim3(n), dim3(MF_THREADS), lds_bytes, stream, rows, B, V, accum, o, flags, vars_base, off, blob, blob_words, red_n, trows, u_glob, u_stride, prof, results, n)
    if (pose) {
        if (mode == 2) FX_MF_SOLO(true, 2);
        else if (mode == 1) FX_MF_SOLO(true, 1);
        else FX_MF_SOLO(true, 0);
    } else {
        if (mode == 2) FX_MF_SOLO(false, 2);
        else if (mode == 1) FX_MF_SOLO(false, 1);
        else FX_MF_SOLO(false, 0);
    }
#undef FX_MF_SOLO
    return hipGetLastError();
}

void trace_block(const BlockOnDevice& blk, uint32_t trials, double ms) {
    const ComponentPlan& P = blk.P;
    for (const TeamSchedule* t : {&P.solo, &P.parts}) {
        if (t->empty()) continue;
        fprintf(stderr, "[fiksi_amd]   schedule with %u parts:", t->nparts);
        for (uint32_t sg = 0; sg < t->nseg(); ++sg) {
            if (sg > 1 && sg + 1 < t->nseg()) continue;  // (the first parts and the top)
            fprintf(stderr, " {");
            for (uint32_t q = t->seg_lev[sg]; q < t->seg_lev[sg + 1]; ++q) {
                uint32_t maxc = 0;
                for (uint32_t w = 0; w < sparse_plan::TEAM_WAVES; ++w)
                    maxc = std::max(maxc, t->wptr[q * sparse_plan::TEAM_WAVES + w + 1] - t->wptr[q * sparse_plan::TEAM_WAVES + w]);
                fprintf(stderr, " %u columns / longest run %u;", t->wptr[(q + 1) * sparse_plan::TEAM_WAVES] - t->wptr[q * sparse_plan::TEAM_WAVES], maxc);
            }
            fprintf(stderr, " }");
        }
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "[fiksi_amd] sparse block: %u rows, %u cols, nnz J %u A %u L %u (%zu products); plan %.2f ms, solve %.2f ms (%u trials)\n",
            P.m, P.nv, P.nnz_j, P.nnz_a, P.nnz_l, P.lpairs.size() / 2, blk.plan_ms, ms, trials);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host: a group of Systems of ONE structure — Levenberg-Marquardt or L-BFGS, everything on the device
// ------------------------------------------------------------------------------------------------
hipError_t sparse_solve_group(const fx_batch* b, const DeviceBatch& d, const uint32_t* systems, uint32_t n_sys, const LmParams& prm,
                              hipStream_t stream, SparsePlanCache* cache, bool stay_async) {
    if (!n_sys) return hipSuccess;
    const bool trace = std::getenv("FIKSI_AMD_TRACE") != nullptr;  // diagnostics on stderr
    const bool team_prof = std::getenv("FIKSI_AMD_TEAM_PROF") != nullptr;
    const bool single_pass = (prm.mode & MODE_UNITS) != 0;
    std::unique_ptr<SparsePlanCache> local;
    if (!cache) {
        local.reset(new SparsePlanCache());
        cache = local.get();
    }
    const uint32_t s0 = systems[0];
    hipError_t e = ensure_plan(b, s0, single_pass, stream, cache);
    if (e != hipSuccess) return e;
    const uint32_t nvt = b->var_off[s0 + 1] - b->var_off[s0], net = b->expr_off[s0 + 1] - b->expr_off[s0];
    const fx_lm_opts o = prm.lm;
    const int do_scale = (prm.mode & 1u) ? 1 : 0;
    const bool lbfgs = (prm.mode & MODE_LBFGS) != 0;
    const bool refined = !lbfgs && o.solver != FX_STEP_CHOLESKY;  // (FX_STEP_QR beyond one wavefront: the refined step)
    // FIKSI_AMD_FRONTS=0 keeps the column walkers of fx_sparse_team.h (A / B, and the tests that compare the two)
    static const bool fronts_env = [] { const char* s = std::getenv("FIKSI_AMD_FRONTS"); return !(s && s[0] == '0'); }();
    const bool fronts = fronts_env && prm.sparse_fronts != 0;

    // ---- the value slab of one System, in doubles; System k of the group starts k * stride further
    auto pad = [](size_t n) { return (n + 15) & ~size_t(15); };
    size_t at = 0;
    auto take = [&](size_t n) {
        const size_t o2 = at;
        at += pad(std::max<size_t>(n, 1));
        return o2;
    };
    const size_t o_vars0 = take(nvt), o_xs0 = take(nvt), o_xs1 = take(nvt), o_snap = take(nvt), o_param = take(net), o_sparam = take(net),
                 o_scal = take(16), o_r0 = take(cache->max_m), o_r1 = take(cache->max_m), o_j0 = take(cache->max_nnz_j),
                 o_j1 = take(cache->max_nnz_j), o_a = take(std::max(cache->max_nnz_a, cache->max_nv)), o_l = take(lbfgs ? 0 : std::max(cache->max_nnz_l, cache->max_mf_l)),
                 o_rhs = take(cache->max_nv), o_delta = take(cache->max_nv), o_t = take(refined ? cache->max_m : 0),
                 o_e = take(refined ? cache->max_nv : 0), o_hs = take(lbfgs ? 5 * (size_t)cache->max_nv : 0),
                 o_hy = take(lbfgs ? 5 * (size_t)cache->max_nv : 0), o_cf = take(std::max(cache->max_fslots, cache->max_mf_gu)), o_cb = take(cache->max_bslots);
    // the multifrontal build's lambda ladder (fx_front.h): ranks - 1 more sets of (point, residuals, Jacobian rows) and of (L blocks,
    // step, contribution blocks). One block, one component only: its trial sets then need no other variable kept in step.
    uint32_t mf_ranks = 1;
    if (fronts && !lbfgs && !refined && cache->blocks.size() == 1 && cache->blocks[0]->mf_up_lds && n_sys < TEAM_PARTS_MAX_GROUP) {
        static const int forced = [] { const char* s = std::getenv("FIKSI_AMD_FRONT_RANKS"); return s ? atoi(s) : 0; }();
        const uint32_t np = cache->blocks[0]->mfp.nparts;
        // (a workgroup per CU and half as many again: the parts differ in size, so the second round fills the gaps of the first —
        // cfg2, 64 parts: 4 ranks 3.52 ms, 6 ranks 2.89, 8 ranks 3.29; large_sketch(1500): 4 ranks 1.89 ms, 8 ranks 1.37)
        mf_ranks = std::max(1u, std::min<uint32_t>(MF_MAX_RANKS, 384u / std::max(1u, np * n_sys)));
        if (forced >= 1 && forced <= (int)MF_MAX_RANKS) mf_ranks = (uint32_t)forced;
        if (prm.sparse_front_ranks >= 1u && prm.sparse_front_ranks <= MF_MAX_RANKS) mf_ranks = prm.sparse_front_ranks;
    }
    const size_t xr = mf_ranks - 1u;
    const size_t s_xs = pad(nvt), s_r = pad(cache->max_m), s_j = pad(cache->max_nnz_j), s_l = pad(cache->max_mf_l), s_d = pad(cache->max_nv),
                 s_gu = pad(cache->max_mf_gu);
    const size_t o_xsx = take(xr * s_xs), o_rx = take(xr * s_r), o_jx = take(xr * s_j), o_lx = take(xr * s_l), o_dx = take(xr * s_d), o_gux = take(xr * s_gu);
    const size_t stride = at;

    // the group in slices that fit a bounded slab (1 GiB)
    const uint32_t per_slice = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_sys, (size_t(1) << 27) / stride));
    for (uint32_t g0 = 0; g0 < n_sys; g0 += per_slice) {
        const uint32_t n = std::min(per_slice, n_sys - g0);
        Pool pool(&cache->values);
        pool.stream = stream;
        double* slab = pool.alloc<double>((size_t)n * stride);
        SpLm* d_lm = pool.alloc<SpLm>(n);
        SpAccum* d_accum = pool.alloc<SpAccum>(n);
        uint32_t* d_tickets = pool.alloc<uint32_t>(n);
        MfLadder Ld{};
        Ld.ranks = mf_ranks;
        Ld.xsx = slab + o_xsx; Ld.rx = slab + o_rx; Ld.jx = slab + o_jx; Ld.lx = slab + o_lx; Ld.dx = slab + o_dx; Ld.gux = slab + o_gux;
        Ld.xs_step = s_xs; Ld.r_step = s_r; Ld.j_step = s_j; Ld.l_step = s_l; Ld.d_step = s_d; Ld.gu_step = s_gu;
        bool any_parts_build = false;
        for (const auto& bp : cache->blocks) any_parts_build = any_parts_build || (bp->mf_up_lds != 0 && n < TEAM_PARTS_MAX_GROUP);
        if (fronts && any_parts_build) {  // (what only the parts + top kernels look at)
            Ld.rk = pool.alloc<MfRank>((size_t)n * MF_MAX_RANKS);
            Ld.tickets = pool.alloc<uint32_t>((size_t)n * (2 * MF_MAX_RANKS + 2));
            if (pool.err != hipSuccess) return pool.err;
            (void)hipMemsetAsync(Ld.rk, 0, (size_t)n * MF_MAX_RANKS * sizeof(MfRank), stream);
            (void)hipMemsetAsync(Ld.tickets, 0, (size_t)n * (2 * MF_MAX_RANKS + 2) * sizeof(uint32_t), stream);
        }
        // [out / vars0 offset | parameter offset | system id] per System. A resident batch's group, solved whole: uploaded once and kept
        // with the plan — such a solve then has nothing of the host's in flight and nothing to give back, and ends without a wait
        // (stay_async: the caller's next work is on this stream too)
        // (a context's own plan — one-shot calls — as well, as long as its slab stays under the bound it may keep: trimming needs the wait)
        const bool keep_off = stay_async && n == n_sys && cache->values.bytes() <= cache->keep_values && !team_prof && !trace;
        bool off_cached = keep_off && cache->d_off && cache->off_systems.size() == n && std::equal(systems, systems + n, cache->off_systems.begin());
        std::vector<uint64_t> h_off;
        uint64_t* d_off = nullptr;
        if (off_cached) {
            d_off = cache->d_off;
        } else {
            h_off.resize(3 * (size_t)n);
            for (uint32_t k = 0; k < n; ++k) {
                h_off[k] = b->var_off[systems[g0 + k]];
                h_off[n + k] = b->expr_off[systems[g0 + k]];
                h_off[2 * (size_t)n + k] = systems[g0 + k];
            }
            if (keep_off) {  // (this solve still waits at its end: the next one finds the copy done)
                hipError_t e2 = hipSuccess;
                cache->off_systems.clear();
                cache->d_off = static_cast<uint64_t*>(cache->arena.take(h_off.size() * sizeof(uint64_t), e2));
                if (!cache->d_off) return e2;
                cache->off_host = h_off;
                e2 = hipMemcpyAsync(cache->d_off, cache->off_host.data(), h_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
                if (e2 != hipSuccess) {
                    cache->d_off = nullptr;
                    return e2;
                }
                cache->off_systems.assign(systems, systems + n);
                d_off = cache->d_off;
            } else {
                d_off = pool.up(h_off);
                if (pool.err != hipSuccess) return pool.err;
            }
        }

        SpRows rows;
        rows.tag = d.expr_tag + b->expr_off[s0];
        rows.idx = d.expr_idx + 4 * (size_t)b->expr_off[s0];
        rows.param = slab + o_param;
        rows.sparam = slab + o_sparam;
        rows.net = net;
        rows.has_pose = d.has_pose;
        SpVals V{};
        V.xs0 = slab + o_xs0; V.xs1 = slab + o_xs1; V.snap = slab + o_snap; V.r0 = slab + o_r0; V.r1 = slab + o_r1;
        V.j0 = slab + o_j0; V.j1 = slab + o_j1; V.a = slab + o_a; V.l = slab + o_l; V.rhs = slab + o_rhs; V.delta = slab + o_delta;
        V.t = slab + o_t; V.e = slab + o_e; V.scal = slab + o_scal; V.hs = slab + o_hs; V.hy = slab + o_hy;
        V.stride = stride;
        double* d_vars0 = slab + o_vars0;

        // a System of one component: everything before its first block in one launch (spg_prologue_kernel)
        static const bool fuse_env = [] { const char* e = getenv("FIKSI_AMD_FUSED_PROLOGUE"); return !e || atoi(e) != 0; }();
        // (one workgroup per System: up to a few thousand values; cfg2's 20 000 take the five launches' many workgroups — 0.21 ms fused, 0.13 apart)
        const bool fused_prologue = fuse_env && cache->comps.size() == 1 && std::max(nvt, net) <= 4096u;
        if (fused_prologue) {
            const CompOnDevice& c0 = cache->comps[0];
            hipLaunchKernelGGL(spg_prologue_kernel, dim3(1, n), dim3(256), 0, stream, d.vars0, d.expr_param, d_off, n, nvt, d_vars0, rows, stride,
                               d.vars, d_accum, d_tickets, V.scal, V.xs0, V.xs1, V.snap, do_scale ? 1 : 0, c0.d_fvar, c0.nfv, 42u,
                               ((prm.mode & 2u) && c0.nfv) ? 1 : 0);
        } else {
            hipLaunchKernelGGL(spg_begin_kernel, grid_for2(std::max(nvt, net), n), dim3(256), 0, stream, d.vars0, d.expr_param, d_off, n, nvt, net,
                               d_vars0, slab + o_param, stride, d.vars, d_accum, d_tickets);
            hipLaunchKernelGGL(sp_scale_kernel, dim3(1, n), dim3(256), 0, stream, d_vars0, nvt, rows, V.scal, do_scale, stride);
            hipLaunchKernelGGL(sp_init_kernel, grid_for2(std::max(nvt, net), n), dim3(256), 0, stream, d_vars0, nvt, rows, V.scal, V.xs0, V.xs1,
                               do_scale, stride);
        }
        bool finish_done = false;  // (mf_lm_solo_kernel has written the result records itself)
        uint32_t rng = 42u;  // Rng::from_seed(42), shared by the components (:47)
        for (const CompOnDevice& comp : cache->comps) {
            // ---- the component's perturbation (:91-124), before any of its blocks
            if (!fused_prologue && (prm.mode & 2u) && comp.nfv) {
                hipLaunchKernelGGL(sp_perturb_kernel, grid_for2(comp.nfv, n), dim3(256), 0, stream, comp.d_fvar, comp.nfv, rng, V.xs0, V.xs1, stride);
                for (uint32_t k = 0; k < 2 * comp.nfv; ++k) rng = rng * 1664525u + 1013904223u;  // integer bookkeeping only
            }
            // the pre-solve snapshot (quirk Q2); the component counts, its exit code starts as "nothing to do"
            if (!fused_prologue) hipLaunchKernelGGL(spg_component_kernel, grid_for2(nvt, n), dim3(256), 0, stream, V.xs0, V.snap, nvt, stride, d_accum);
            for (uint32_t u = 0; u < comp.n_blocks; ++u) {
                const BlockOnDevice& blk = *cache->blocks[comp.first_block + u];
                const uint32_t flags = (refined ? TEAM_REFINED : 0u) | (single_pass ? TEAM_SINGLE_PASS : 0u) | (do_scale ? TEAM_SCALE : 0u);
                const auto t_lm0 = std::chrono::steady_clock::now();
                // one workgroup per System runs the whole loop; a lone large System is spread over the chip instead
                const bool two_tier = !lbfgs && blk.has_parts && n < TEAM_PARTS_MAX_GROUP;
                if (lbfgs) {  // Optimizer::LBfgs: the whole optimizer in one launch as well, line search included
                    if (rows.has_pose)
                        hipLaunchKernelGGL(sp_lbfgs_team_kernel<true>, dim3(n), dim3(TEAM_THREADS), 0, stream, rows, blk.dev, V, d_accum, flags, d.vars, d_off);
                    else
                        hipLaunchKernelGGL(sp_lbfgs_team_kernel<false>, dim3(n), dim3(TEAM_THREADS), 0, stream, rows, blk.dev, V, d_accum, flags, d.vars, d_off);
                } else if (!two_tier && fronts && !refined && blk.mf_solo_lds) {
                    // the multifrontal build: the whole loop in one launch, a front per row of 16 lanes (fx_front.h)
                    unsigned long long* d_prof = nullptr;
                    if (team_prof) {
                        d_prof = pool.alloc<unsigned long long>(16);
                        if (pool.err != hipSuccess) return pool.err;
                        (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), stream);
                    }
                    // (the System's only block, every expression a row of it: the closing check and the record in the same launch)
                    const bool fuse_finish = fused_prologue && comp.n_blocks == 1u && blk.dev.m == net && !single_pass;
                    e = launch_mf_solo(rows.has_pose != 0, n, blk.mf_solo_lds, stream, rows, blk.dev, V, d_accum, o, flags, d.vars, d_off, blk.mf_solo_blob,
                                       blk.mf_solo_words, blk.mf_solo_red, blk.mf_solo_rows, blk.mf_solo_mode, slab + o_cf, stride, d_prof,
                                       fuse_finish ? d.results : nullptr);
                    if (e != hipSuccess) return e;
                    finish_done = fuse_finish;
                    if (team_prof) {
                        unsigned long long h[16];
                        (void)hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, stream);
                        (void)hipStreamSynchronize(stream);
                        fprintf(stderr, "[fiksi_amd] mf_lm_solo, %u Systems, workgroup 0 (us over %llu trials): start %.1f | form %.1f fronts up %.1f (tile + A %.1f children %.1f "
                                        "registers + pivots %.1f stores %.1f barrier %.1f) down %.1f trial + eval + sums %.1f | epilogue %.1f\n", n, h[7], h[0] * 0.01, h[1] * 0.01,
                                h[2] * 0.01, h[8] * 0.01, h[9] * 0.01, h[10] * 0.01, h[11] * 0.01, h[12] * 0.01, h[3] * 0.01, h[5] * 0.01, h[6] * 0.01);
                    }
                } else if (!two_tier) {
                    unsigned long long* d_prof = nullptr;
                    if (team_prof) {
                        d_prof = pool.alloc<unsigned long long>(16);
                        if (pool.err != hipSuccess) return pool.err;
                        (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), stream);
                    }
                    // the factor and the solves' vectors in LDS when they fit beside the kernel's own 17 KB
                    const uint32_t lds_l = (blk.dev.nnz_l + 15u) & ~15u, lds_v = (blk.dev.nv + 15u) & ~15u;
                    const size_t lds_vals = ((size_t)lds_l + 2 * (size_t)lds_v) * sizeof(double);
                    const bool ldsv = lds_vals <= TEAM_LDS_VALUES_MAX;
                    const bool with_blob = ldsv && blk.solo_blob && lds_vals + (size_t)blk.solo_blob_words * 4 <= TEAM_LDS_VALUES_MAX;  // the index data too
                    const size_t lds_bytes = ldsv ? lds_vals + (with_blob ? (size_t)blk.solo_blob_words * 4 : 0) : 0;
                    e = launch_team(rows.has_pose != 0, ldsv, with_blob, n, lds_bytes, stream, rows, blk.dev, V, d_accum, o, flags, d.vars, d_off, lds_l, lds_v,
                                    blk.solo_blob, blk.solo_blob_words, d_prof);
                    if (e != hipSuccess) return e;
                    if (team_prof) {
                        unsigned long long h[16];
                        (void)hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, stream);
                        (void)hipStreamSynchronize(stream);
                        fprintf(stderr, "[fiksi_amd]   factorization, wavefront 0 (us): level 0 walk %.1f (%llu columns) + wait %.1f | levels above: walk %.1f + wait %.1f\n",
                                h[8] * 0.01, h[12], h[9] * 0.01, h[10] * 0.01, h[11] * 0.01);
                        fprintf(stderr, "[fiksi_amd] team kernel, %u Systems, workgroup 0 (us): start %.1f | form %.1f factor+forward %.1f backward %.1f refine %.1f "
                                        "trial+eval %.1f | epilogue %.1f; %llu trials\n", n, h[0] * 0.01, h[1] * 0.01, h[2] * 0.01, h[3] * 0.01, h[4] * 0.01,
                                h[5] * 0.01, h[6] * 0.01, h[7]);
                    }
                } else {
                    SpBlock B = blk.dev;
                    B.sched = blk.parts;
                    const uint32_t np = B.sched.nparts;
                    const dim3 g_rows((B.m + TEAM_THREADS - 1) / TEAM_THREADS ? (B.m + TEAM_THREADS - 1) / TEAM_THREADS : 1, n);
                    auto eval = [&](uint32_t start) {
                        if (rows.has_pose) hipLaunchKernelGGL(spt_eval_kernel<true>, g_rows, dim3(TEAM_THREADS), 0, stream, rows, B, V, d_lm, d_tickets, o, start);
                        else hipLaunchKernelGGL(spt_eval_kernel<false>, g_rows, dim3(TEAM_THREADS), 0, stream, rows, B, V, d_lm, d_tickets, o, start);
                    };
                    eval(1u);
                    if (mf_ranks > 2u)  // the ladder's further sets: whole copies of the start point (a trial overwrites the block's own variables)
                        for (uint32_t k = 0; k < n; ++k)
                            for (uint32_t s2 = 2; s2 <= mf_ranks; ++s2)
                                (void)hipMemcpyAsync(slab + (size_t)k * stride + o_xsx + (s2 - 2u) * s_xs, slab + (size_t)k * stride + o_xs0, (size_t)nvt * sizeof(double),
                                                     hipMemcpyDeviceToDevice, stream);
                    // the plain step with every segment's values in LDS when they fit (they do unless a part is enormous)
                    const bool mf_build = fronts && !refined && blk.mf_up_lds != 0;
                    const bool lds_build = !mf_build && !refined && blk.lds_part_bytes && blk.lds_part_bytes <= TEAM_LDS_VALUES_MAX && blk.lds_top_bytes <= TEAM_LDS_VALUES_MAX;
                    SpContrib Cn{slab + o_cf, slab + o_cb, stride};
                    unsigned long long* d_prof = nullptr;
                    if (team_prof && lds_build) {
                        d_prof = pool.alloc<unsigned long long>(16);
                        if (pool.err != hipSuccess) return pool.err;
                        (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), stream);
                    }
                    if (lds_build) {
                        e = raise_lds_limits();
                        if (e != hipSuccess) return e;
                    }
                    MfParts mfp = blk.mfp;
                    if (mf_build) {
                        e = raise_mf_lds_limits();
                        if (e != hipSuccess) return e;
                        if (team_prof) {
                            d_prof = pool.alloc<unsigned long long>(32);
                            if (pool.err != hipSuccess) return pool.err;
                            (void)hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), stream);
                        }
                        mfp.prof = d_prof;
                    }
                    std::vector<SpLm> h_lm(n);
                    for (uint32_t chunk = 4;; chunk = std::min<uint32_t>(2 * chunk, 16)) {
                        e = hipMemcpyAsync(h_lm.data(), d_lm, n * sizeof(SpLm), hipMemcpyDeviceToHost, stream);
                        if (e == hipSuccess) e = hipStreamSynchronize(stream);
                        if (e != hipSuccess) return e;
                        bool all_done = true;
                        for (const SpLm& st : h_lm) all_done = all_done && st.done;
                        if (all_done) break;
                        for (uint32_t t = 0; t < chunk; ++t) {
                            if (mf_build) {  // (two launches per ROUND of mf_ranks trials: the parts' fronts up — the last part of a rank goes on with
                                             // its top —, then down + evaluation + the ranks' verdicts in order)
                                if (rows.has_pose) {
                                    hipLaunchKernelGGL(mf_parts_up_kernel<true>, dim3(np, n, mf_ranks), dim3(MF_THREADS), blk.mf_up_lds, stream, rows, B, mfp, V, Ld, slab + o_cf, stride, d_lm, o);
                                    hipLaunchKernelGGL(mf_parts_down_kernel<true>, dim3(np, n, mf_ranks), dim3(MF_THREADS), blk.mf_down_lds, stream, rows, B, mfp, V, Ld, d_lm, o);
                                } else {
                                    hipLaunchKernelGGL(mf_parts_up_kernel<false>, dim3(np, n, mf_ranks), dim3(MF_THREADS), blk.mf_up_lds, stream, rows, B, mfp, V, Ld, slab + o_cf, stride, d_lm, o);
                                    hipLaunchKernelGGL(mf_parts_down_kernel<false>, dim3(np, n, mf_ranks), dim3(MF_THREADS), blk.mf_down_lds, stream, rows, B, mfp, V, Ld, d_lm, o);
                                }
                                continue;
                            }
                            if (!lds_build) hipLaunchKernelGGL(spt_form_kernel, grid_for2(std::max(B.nnz_a, B.nv), n), dim3(256), 0, stream, B, V, d_lm);
                            if (lds_build) {  // (each segment forms its own entries of A and of the right-hand side)
                                const size_t lds_up = std::max(blk.lds_part_bytes, blk.lds_top_bytes);  // (the last part's workgroup goes on with the top)
                                if (rows.has_pose) {
                                    hipLaunchKernelGGL(sptl_parts_up_kernel<true>, dim3(np, n), dim3(TEAM_THREADS), lds_up, stream, rows, B, blk.px, V, Cn, d_lm, d_tickets, d_prof);
                                    hipLaunchKernelGGL(sptl_parts_down_kernel<true>, dim3(np, n), dim3(TEAM_THREADS), blk.lds_part_bytes, stream, rows, B, blk.px, V, d_lm, d_tickets, o);
                                } else {
                                    hipLaunchKernelGGL(sptl_parts_up_kernel<false>, dim3(np, n), dim3(TEAM_THREADS), lds_up, stream, rows, B, blk.px, V, Cn, d_lm, d_tickets, d_prof);
                                    hipLaunchKernelGGL(sptl_parts_down_kernel<false>, dim3(np, n), dim3(TEAM_THREADS), blk.lds_part_bytes, stream, rows, B, blk.px, V, d_lm, d_tickets, o);
                                }
                                continue;  // (two launches per trial: up — the last part goes on with the top —, down + evaluation + decision)
                            }
                            hipLaunchKernelGGL(spt_parts_up_kernel, dim3(np, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 0u);
                            hipLaunchKernelGGL(spt_top_kernel, dim3(1, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 0u, refined ? 0u : 1u);
                            hipLaunchKernelGGL(spt_parts_down_kernel, dim3(np, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 0u, refined ? 0u : 1u);
                            if (refined) {
                                hipLaunchKernelGGL(spt_refine_t_kernel, grid_for2(B.m, n), dim3(256), 0, stream, B, V, d_lm);
                                hipLaunchKernelGGL(spt_refine_rhs_kernel, grid_for2(B.nv, n), dim3(256), 0, stream, B, V, d_lm);
                                hipLaunchKernelGGL(spt_parts_up_kernel, dim3(np, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 1u);
                                hipLaunchKernelGGL(spt_top_kernel, dim3(1, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 1u, 1u);
                                hipLaunchKernelGGL(spt_parts_down_kernel, dim3(np, n), dim3(TEAM_THREADS), 0, stream, B, V, d_lm, 1u, 1u);
                            }
                            eval(0u);
                        }
                        e = hipGetLastError();
                        if (e != hipSuccess) return e;
                    }
                    if (mf_ranks > 1u) {  // the solved point into set 0, where the block's epilogue looks for it
                        hipLaunchKernelGGL(mf_ladder_finish_kernel, grid_for2(nvt, n), dim3(256), 0, stream, V, Ld, d_lm, nvt);
                        hipLaunchKernelGGL(mf_ladder_finish_state_kernel, grid_for(n), dim3(256), 0, stream, d_lm, n);
                    }
                    hipLaunchKernelGGL(spt_block_end_kernel, grid_for2(std::max(B.nv, 1u), n), dim3(256), 0, stream, B, V, d_lm, d_accum, flags, d.vars, d_off);
                    if (d_prof && mf_build) {
                        unsigned long long h[32];
                        (void)hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, stream);
                        (void)hipStreamSynchronize(stream);
                        fprintf(stderr, "[fiksi_amd]   fronts up by phase (us; thread 0): part 0: tile + A %.1f children %.1f registers + pivots %.1f stores %.1f barrier %.1f | top: %.1f %.1f %.1f %.1f %.1f\n",
                                h[16] * 0.01, h[17] * 0.01, h[18] * 0.01, h[19] * 0.01, h[20] * 0.01, h[24] * 0.01, h[25] * 0.01, h[26] * 0.01, h[27] * 0.01, h[28] * 0.01);
                        fprintf(stderr, "[fiksi_amd] mf_parts_up (us over %llu launches): part 0: load %.1f form %.1f fronts up %.1f ticket %.1f | top: load+form %.1f fronts up %.1f "
                                        "down %.1f rest %.1f\n", h[15], h[0] * 0.01, h[1] * 0.01, h[2] * 0.01, h[3] * 0.01, h[4] * 0.01, h[5] * 0.01, h[6] * 0.01, h[8] * 0.01);
                    } else if (d_prof) {
                        unsigned long long h[16];
                        (void)hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, stream);
                        (void)hipStreamSynchronize(stream);
                        fprintf(stderr, "[fiksi_amd] parts_up, part 0 (us over %llu launches): load %.1f | factor+forward %.1f (wavefront 0: level 0 walk %.1f (%llu columns) wait %.1f, above walk %.1f wait %.1f) | "
                                        "store + contributions %.1f\n", h[7], h[0] * 0.01, h[1] * 0.01, h[8] * 0.01, h[12], h[9] * 0.01, h[10] * 0.01, h[11] * 0.01, h[2] * 0.01);
                    }
                    if (trace) trace_block(blk, h_lm[0].trials, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lm0).count());
                }
                if (trace && (lbfgs || !two_tier)) {
                    (void)hipStreamSynchronize(stream);
                    trace_block(blk, 0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lm0).count());
                }
            }
        }
        // ---- post-solve check on the unscaled variables, the result records
        if (finish_done) {
        } else if (rows.has_pose)
            hipLaunchKernelGGL(spg_finish_kernel<true>, dim3(n), dim3(TEAM_THREADS), 0, stream, rows, stride, V.scal, d_accum, d.vars, d_off, n, d.results);
        else
            hipLaunchKernelGGL(spg_finish_kernel<false>, dim3(n), dim3(TEAM_THREADS), 0, stream, rows, stride, V.scal, d_accum, d.vars, d_off, n, d.results);
        e = hipGetLastError();
        // the slice's slab is handed to the next one, the host's offsets die with this frame — unless neither is the case
        if (e == hipSuccess && !off_cached) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
    }
    // the slab stays with the plan for the next solve of this structure — up to a bound for the context's own plans (one
    // big one-shot group must not pin a gigabyte of HBM per cached structure until the context goes); a resident batch's
    // plans keep theirs whole until the batch is freed (20 000 Systems of 66 variables: 272 MB, and giving it back and
    // asking for it again cost 12 ms per solve of 3.8 ms)
    cache->values.trim(cache->keep_values);  // (a solve that returned with its launches in flight is under the bound: nothing to trim)
    return hipSuccess;
}

}  // namespace fx
