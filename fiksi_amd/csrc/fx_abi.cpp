// C ABI of libfiksi_amd.so (include/fiksi_amd.h): validation, Jacobian structure, HBM residency,
// kernel launches. Host logic only; every numeric result comes from the HIP kernels in
// fx_kernels.hip. There is deliberately no CPU compute path in this library.
#ifdef FX_HOST_ONLY
#include "fx_hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fx_decompose.h"
#include "fx_device.h"
#include "fx_expr.h"
#include "fx_qrplan.h"
#include "fx_sparse.h"

namespace {

thread_local std::string g_last_error;
// set by fx_cluster_solve_batch while it uploads: the two pose-row tags of fx_expr.h are legal in that batch only
thread_local bool g_allow_pose = false;
// fx_ctx_set_wide_routing of the context the running call belongs to (-1 by cost, 0 team kernels, 1 wide kernel)
thread_local int g_wide_routing = -1;
// fx_system_solve_batch_multi: the choice made once on the whole batch, for every shard (-2: none)
thread_local int g_wide_routing_pinned = -2;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// FIKSI_AMD_TRACE=1: where the wall time of a host-buffer call goes (one line per phase on stderr)
struct PhaseTrace {
    bool on = std::getenv("FIKSI_AMD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void stamp(const char* what, uint32_t n) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fiksi_amd] host call, %u Systems: %-24s %8.3f ms\n", n, what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

}  // namespace
namespace fx {
// the builder (fx_builder.cpp) reports through the same thread-local text fx_last_error() returns
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace fx
namespace {

#define FX_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(FX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// A vector whose resize() leaves new elements unwritten: the analysis fills every entry of its big arrays, and zeroing
// 40 MB first (100k Systems) costs as much as a third of the analysis.
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using RawVec = std::vector<T, NoInitAlloc<T>>;

// Host-side analysis of a batch: everything the device needs besides the raw arrays.
struct HostPlan {
    uint32_t n_systems = 0, n_vars = 0, n_exprs = 0;
    uint64_t nnz = 0;
    uint32_t max_free = 0, max_rows = 0, max_vars = 0, max_exprs = 0, max_vars_all = 0, max_exprs_all = 0;
    uint32_t max_pairs = 0, max_ents = 0, max_pairs_large = 0, max_ents_large = 0, max_pairs_tri = 0;
    uint32_t uniform = 0;  // every System has the same structure (one sketch, many parameter sets)
    std::vector<uint32_t> sys_class;  // not uniform: the first System with this System's structure (empty: not computed)
    std::vector<uint16_t> sys_ncomp;
    std::vector<uint8_t> sys_large;  // 0 fused kernel, 2 wide kernel (65..128 free variables), 1 sparse path
    uint32_t n_large = 0;            // Systems with sys_large != 0
    std::vector<uint32_t> wide_list;
    uint32_t w_max_free = 0, w_max_vars = 0, w_max_rows = 0;
    int wide_decision = -1;  // the batch holds components of 65 ... 128 columns and they go to: 0 the team kernels, 1 the wide kernel
    RawVec<uint16_t> var_info;
    RawVec<uint16_t> expr_comp;
    RawVec<uint16_t> expr_idx16;
    RawVec<uint8_t> expr_tagx;   // tag | 0x80 when a free column repeats inside the row
    std::vector<uint8_t> same_as_prev;  // System s has the raw structure of System s - 1 (its analysis was copied)
};

// What only the row-parallel kernels need (eval_rows_kernel, identity_residual_kernel): built on first use from
// the compact arrays — a batch that is only ever solved neither computes nor uploads these 19 MB per 100k Systems.
struct EvalPlan {
    std::vector<uint32_t> expr_var0;  // var_off of the owning System
    std::vector<uint8_t> row_perm;    // tag-sorted order of each 256-row block
    std::vector<uint8_t> row_sysoff;  // owning System minus the block's first System
    std::vector<fx::BlockInfo> blk_info;
};

// CSR structure of the Jacobian (fixed pattern): built on demand, from the compact arrays.
struct CsrPlan {
    std::vector<uint32_t> jrow_ptr, jcol, jslot;
};

// Runs fn(t, begin, end) over [0, n) cut into contiguous ranges, on up to 16 host threads when the
// work is worth it (the per-expression analysis is ~0.1 us; a thread costs ~50 us to start).
template <typename F>
void parallel_ranges(uint32_t n, uint64_t work_items, F&& fn, uint32_t* n_ranges_out = nullptr, uint64_t min_items = 200000) {
    uint32_t nt = std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (work_items < min_items) nt = 1;
    nt = std::min<uint32_t>(nt, std::max(1u, n));
    if (n_ranges_out) *n_ranges_out = nt;
    if (nt == 1) {
        fn(0u, 0u, n);
        return;
    }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nt; ++t) {
        uint32_t lo = (uint32_t)((uint64_t)n * t / nt), hi = (uint32_t)((uint64_t)n * (t + 1) / nt);
        th.emplace_back([&fn, t, lo, hi] { fn(t, lo, hi); });
    }
    for (auto& x : th) x.join();
}
constexpr uint32_t MAX_RANGES = 16;

// distinct free columns of a row (ascending) and the slot of each gradient entry; returns the count
inline int row_columns(const uint32_t vars8[8], int k, const int32_t* free_rank, int32_t cols[8], uint32_t* slots_out) {
    int ncols = 0;
    for (int q = 0; q < k; ++q) {
        int32_t col = free_rank[vars8[q]];
        if (col < 0) continue;
        bool seen = false;
        for (int t = 0; t < ncols; ++t) seen = seen || cols[t] == col;
        if (!seen) cols[ncols++] = col;
    }
    std::sort(cols, cols + ncols);
    if (slots_out) {
        uint32_t slots = 0;
        for (int q = 0; q < 8; ++q) {
            uint32_t sl = 0xFu;
            if (q < k) {
                int32_t col = free_rank[vars8[q]];
                if (col >= 0) sl = (uint32_t)(std::find(cols, cols + ncols, col) - cols);
            }
            slots |= sl << (4 * q);
        }
        *slots_out = slots;
    }
    return ncols;
}

// CSR pattern from the compact per-variable / per-expression arrays (host copies of what the device
// holds): columns = system-local rank among the free variables, ascending inside a row, duplicates
// merged (sparse_col_mat.rs:690-737 after the row/column transposition).
void build_csr(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info,
               const uint8_t* expr_tag, const uint16_t* expr_idx16, CsrPlan& out) {
    const uint32_t ne = n ? expr_off[n] : 0;
    out.jrow_ptr.assign((size_t)ne + 1, 0);
    out.jslot.assign(ne, 0xFFFFFFFFu);
    std::vector<uint32_t> local_cols[MAX_RANGES];
    uint32_t range_lo[MAX_RANGES + 1] = {0};
    uint32_t nr = 1;
    parallel_ranges(n, ne, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        range_lo[t] = s_lo;
        std::vector<int32_t> free_rank;
        auto& lc = local_cols[t];
        lc.reserve((size_t)(expr_off[s_hi] - expr_off[s_lo]) * 5);
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0;
            const uint32_t e0 = expr_off[s], net = expr_off[s + 1] - e0;
            free_rank.assign(nvt, -1);
            int32_t rank = 0;
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t info = var_info[v0 + i];
                if ((info & fx::VAR_COMP_MASK) != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT)) free_rank[i] = rank++;
            }
            for (uint32_t i = 0; i < net; ++i) {
                const uint32_t e = e0 + i;
                uint32_t vars8[8];
                int k = fx::expand_vars<true>((int)(expr_tag[e] & 0x7F), expr_idx16 + 4 * (size_t)e, vars8);
                int32_t cols[8];
                uint32_t slots;
                int ncols = row_columns(vars8, k, free_rank.data(), cols, &slots);
                out.jslot[e] = slots;
                for (int q = 0; q < ncols; ++q) lc.push_back((uint32_t)cols[q]);
                out.jrow_ptr[(size_t)e + 1] = (uint32_t)ncols;  // count; prefix-summed below
            }
        }
    }, &nr);
    for (uint32_t e = 0; e < ne; ++e) out.jrow_ptr[e + 1] += out.jrow_ptr[e];
    out.jcol.resize(out.jrow_ptr[ne]);
    for (uint32_t t = 0; t < nr; ++t)
        if (!local_cols[t].empty())
            std::copy(local_cols[t].begin(), local_cols[t].end(), out.jcol.begin() + out.jrow_ptr[expr_off[range_lo[t]]]);
}

void build_eval_plan(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info,
                     const uint8_t* expr_tagx, const uint16_t* expr_idx16, EvalPlan& out) {
    const uint32_t ne = n ? expr_off[n] : 0;
    out.expr_var0.assign(ne, 0);
    out.row_perm.assign(ne, 0);
    out.row_sysoff.assign(ne, 0);
    std::vector<uint32_t> expr_sys(ne, 0);
    std::vector<uint8_t> row_simple(ne, 0);
    parallel_ranges(n, ne, [&](uint32_t, uint32_t s_lo, uint32_t s_hi) {
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const uint32_t v0 = var_off[s];
            for (uint32_t e = expr_off[s]; e < expr_off[s + 1]; ++e) {
                out.expr_var0[e] = v0;
                expr_sys[e] = s;
                uint32_t vars8[8];
                const int k = fx::expand_vars<true>((int)(expr_tagx[e] & 0x7F), expr_idx16 + 4 * (size_t)e, vars8);
                bool all_free = true, distinct = true;  // "simple": every variable free, none read twice
                for (int q = 0; q < k; ++q) {
                    const uint16_t info = var_info[v0 + vars8[q]];
                    all_free = all_free && (info & fx::VAR_COMP_MASK) != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT);
                    for (int u = q + 1; u < k; ++u) distinct = distinct && vars8[q] != vars8[u];
                }
                row_simple[e] = (all_free && distinct) ? 1 : 0;
            }
        }
    });
    // tag-sorted thread -> row assignment inside every block of 256 rows (stable counting sort)
    const uint32_t nblk = (ne + 255u) / 256u;
    out.blk_info.assign(nblk, fx::BlockInfo{});
    parallel_ranges(nblk, ne, [&](uint32_t, uint32_t b_lo, uint32_t b_hi) {
        for (uint32_t blk = b_lo; blk < b_hi; ++blk) {
            const uint32_t r0 = blk * 256u;
            uint32_t nrw = std::min<uint32_t>(256, ne - r0), t = 0;
            uint8_t sorted[256];
            for (int tag = 0; tag < FX_NTAGS_POSE; ++tag)
                for (uint32_t i = 0; i < nrw; ++i)
                    if ((expr_tagx[r0 + i] & 0x7F) == tag) sorted[t++] = (uint8_t)i;
            // The sorted order, rotated by a wavefront per block: wavefront w of every workgroup lands on the same SIMD of its
            // CU, so without the rotation the expensive kinds (the angle rows: two atan2, three times a distance row's
            // instructions) of all eight resident blocks pile up on one SIMD while the other three idle.
            const uint32_t shift = nrw == 256u ? 64u * (blk & 3u) : 0u;
            for (uint32_t i = 0; i < nrw; ++i) out.row_perm[r0 + i] = sorted[(i + shift) % nrw];
            fx::BlockInfo bi{};
            bi.sys0 = expr_sys[r0];
            bool simple = true;
            for (uint32_t i = 0; i < nrw; ++i) {
                simple = simple && row_simple[r0 + i];
                // one byte per row: fits while the block's rows belong to at most 256 consecutive Systems. Systems
                // without expressions take an index without taking a row, so a block can span more — such a block
                // is not "simple": its rows then read their System's first variable from expr_var0 instead
                const uint32_t off = expr_sys[r0 + i] - bi.sys0;
                simple = simple && off <= 255u;
                out.row_sysoff[r0 + i] = (uint8_t)(off & 0xFFu);
            }
            bi.flags = simple ? 1u : 0u;  // jbase / jcount are filled when the CSR structure is built
            out.blk_info[blk] = bi;
        }
    });
}

// Checks the batch and builds the plan. Mirrors the data invariants the reference enforces by
// construction (handles of the same System, indices < variables.len()).
int analyze(const fx_batch* b, HostPlan* plan) {
    if (!b) return fail(FX_ERR_INVALID, "batch is NULL");
    const uint32_t n = b->n_systems;
    if (n > 0 && (!b->var_off || !b->expr_off)) return fail(FX_ERR_INVALID, "var_off/expr_off is NULL");
    if (n == 0) {
        if (plan) *plan = HostPlan();
        return FX_OK;
    }
    if (b->var_off[0] != 0 || b->expr_off[0] != 0) return fail(FX_ERR_INVALID, "offset arrays must start at 0");
    for (uint32_t s = 0; s < n; ++s) {
        if (b->var_off[s + 1] < b->var_off[s] || b->expr_off[s + 1] < b->expr_off[s])
            return fail(FX_ERR_INVALID, "offsets of system %u decrease", s);
    }
    const uint32_t nv = b->var_off[n], ne = b->expr_off[n];
    if (nv > 0 && (!b->vars || !b->var_fixed)) return fail(FX_ERR_INVALID, "vars/var_fixed is NULL");
    if (ne > 0 && (!b->expr_tag || !b->expr_idx || !b->expr_param)) return fail(FX_ERR_INVALID, "expr_* is NULL");

    HostPlan local;
    HostPlan& p = plan ? *plan : local;
    p = HostPlan();
    p.n_systems = n;
    p.n_vars = nv;
    p.n_exprs = ne;
    p.sys_ncomp.assign(n, 0);
    p.sys_large.assign(n, 0);
    p.var_info.resize(nv);
    p.expr_comp.resize(ne);
    p.expr_idx16.resize(4 * (size_t)ne);
    p.expr_tagx.resize(ne);
    p.same_as_prev.assign(n, 0);

    struct Partial {
        uint64_t nnz = 0;
        uint32_t max_free = 0, max_rows = 0, max_vars = 0, max_exprs = 0, max_vars_all = 0, max_exprs_all = 0, n_large = 0;
        uint32_t w_max_free = 0, w_max_rows = 0, w_max_vars = 0;
        uint32_t max_pairs = 0, max_ents = 0, max_pairs_large = 0, max_ents_large = 0, max_pairs_tri = 0;
        int err = FX_OK;
        uint32_t err_system = 0;
        uint32_t first = 0;      // first System of the range
        bool chain_same = true;  // every System of the range came out as the one before it (sizes and analysed arrays)
        char msg[192] = {0};
    } part[MAX_RANGES];
    uint32_t nr = 1;
    const int n_tags = g_allow_pose ? FX_NTAGS_POSE : FX_NTAGS;  // read here: the ranges below may run on other threads
    // One sketch, many parameter sets (the last System has the first one's sizes and kinds): nearly every System will take
    // the analysis of the one before it, a few nanoseconds per item — threads then pay from some millions of items on
    uint64_t min_items = 200000;
    if (n >= 2 && b->var_off[1] == nv / n && b->expr_off[1] == ne / n && b->var_off[n - 1] == (uint64_t)(n - 1) * b->var_off[1] &&
        b->expr_off[n - 1] == (uint64_t)(n - 1) * b->expr_off[1] && nv == (uint64_t)n * b->var_off[1] && ne == (uint64_t)n * b->expr_off[1] &&
        memcmp(b->expr_tag, b->expr_tag + b->expr_off[n - 1], b->expr_off[1] * sizeof(*b->expr_tag)) == 0)
        min_items = 4000000;
    parallel_ranges(n, (uint64_t)ne + nv, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Partial& pt = part[t];
        auto bad = [&](int code, uint32_t s, const char* fmt, unsigned a0 = 0, unsigned a1 = 0, unsigned a2 = 0, unsigned a3 = 0) {
            pt.err = code;
            pt.err_system = s;
            snprintf(pt.msg, sizeof(pt.msg), fmt, a0, a1, a2, a3);
        };
        std::vector<int32_t> free_rank;  // per variable of the current system: system-wide free rank
        std::vector<uint32_t> comp_free, comp_rows, comp_pairs, comp_ents, comp_tri;
        pt.first = s_lo;
        // One sketch with many parameter sets is the common batch: a System whose raw structure arrays are those of the
        // System before it takes that System's analysis (two memcmp / memcpy passes instead of the walk below; every
        // statistic of the range is a maximum — unchanged — or a sum of per-System terms kept here).
        uint64_t prev_nnz = 0;
        uint32_t prev_large = 0;
        auto outputs_equal = [&](uint32_t s, uint32_t q, uint32_t nvt, uint32_t net) {
            const uint32_t v0 = b->var_off[s], e0 = b->expr_off[s], pv0 = b->var_off[q], pe0 = b->expr_off[q];
            return memcmp(&p.var_info[v0], &p.var_info[pv0], nvt * sizeof(uint16_t)) == 0 && memcmp(&p.expr_tagx[e0], &p.expr_tagx[pe0], net) == 0 &&
                   memcmp(&p.expr_comp[e0], &p.expr_comp[pe0], net * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_idx16[4 * (size_t)e0], &p.expr_idx16[4 * (size_t)pe0], 4 * (size_t)net * sizeof(uint16_t)) == 0;
        };
        for (uint32_t s = s_lo; s < s_hi && pt.err == FX_OK; ++s) {
            const uint32_t v0 = b->var_off[s], nvt = b->var_off[s + 1] - v0;
            const uint32_t e0 = b->expr_off[s], net = b->expr_off[s + 1] - e0;
            if (s > s_lo) {
                const uint32_t pv0 = b->var_off[s - 1], pe0 = b->expr_off[s - 1];
                if (nvt == v0 - pv0 && net == e0 - pe0 && memcmp(b->var_fixed + v0, b->var_fixed + pv0, nvt) == 0 &&
                    memcmp(b->expr_tag + e0, b->expr_tag + pe0, net * sizeof(*b->expr_tag)) == 0 &&
                    memcmp(b->expr_idx + 4 * (size_t)e0, b->expr_idx + 4 * (size_t)pe0, 4 * (size_t)net * sizeof(uint32_t)) == 0 &&
                    (!b->var_comp || memcmp(b->var_comp + v0, b->var_comp + pv0, nvt * sizeof(*b->var_comp)) == 0) &&
                    (!b->expr_comp || memcmp(b->expr_comp + e0, b->expr_comp + pe0, net * sizeof(*b->expr_comp)) == 0)) {
                    memcpy(&p.var_info[v0], &p.var_info[pv0], nvt * sizeof(uint16_t));
                    memcpy(&p.expr_tagx[e0], &p.expr_tagx[pe0], net);
                    memcpy(&p.expr_comp[e0], &p.expr_comp[pe0], net * sizeof(uint16_t));
                    memcpy(&p.expr_idx16[4 * (size_t)e0], &p.expr_idx16[4 * (size_t)pe0], 4 * (size_t)net * sizeof(uint16_t));
                    p.sys_ncomp[s] = p.sys_ncomp[s - 1];
                    p.sys_large[s] = p.sys_large[s - 1];
                    p.same_as_prev[s] = 1;
                    pt.nnz += prev_nnz;
                    pt.n_large += prev_large;
                    continue;
                }
            }
            const uint64_t nnz_before = pt.nnz;
            const uint32_t large_before = pt.n_large;
            if (nvt > FX_MAX_LARGE_SYSTEM_VARS) {
                bad(FX_ERR_TOO_LARGE, s, "system %u has %u variables (limit %u)", s, nvt, FX_MAX_LARGE_SYSTEM_VARS);
                break;
            }
            bool large = nvt > FX_MAX_SYSTEM_VARS;
            pt.max_vars_all = std::max(pt.max_vars_all, nvt);
            pt.max_exprs_all = std::max(pt.max_exprs_all, net);

            uint32_t ncomp = 0;
            free_rank.assign(nvt, -1);
            int32_t rank = 0;
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t c = b->var_comp ? b->var_comp[v0 + i] : 0;
                bool fixed = b->var_fixed[v0 + i] != 0;
                uint16_t info;
                if (c == FX_NO_COMPONENT) {
                    info = fx::VAR_COMP_NONE;
                } else {
                    if (c >= fx::VAR_COMP_NONE) {
                        bad(FX_ERR_INVALID, s, "system %u: component id %u too large", s, c);
                        break;
                    }
                    info = c;
                    ncomp = std::max<uint32_t>(ncomp, c + 1u);
                    if (!fixed) free_rank[i] = rank++;
                }
                if (fixed) info |= fx::VAR_FIXED_BIT;
                p.var_info[v0 + i] = info;
            }
            for (uint32_t i = 0; i < net && pt.err == FX_OK; ++i) {
                uint16_t c = b->expr_comp ? b->expr_comp[e0 + i] : 0;
                if (c == FX_NO_COMPONENT) {
                    c = fx::VAR_COMP_NONE;  // never selected by any component loop
                } else {
                    if (c >= fx::VAR_COMP_NONE) {
                        bad(FX_ERR_INVALID, s, "system %u: component id %u too large", s, c);
                        break;
                    }
                    ncomp = std::max<uint32_t>(ncomp, c + 1u);
                }
                p.expr_comp[e0 + i] = c;
            }
            if (pt.err != FX_OK) break;
            p.sys_ncomp[s] = (uint16_t)ncomp;

            comp_free.assign(ncomp, 0);
            comp_rows.assign(ncomp, 0);
            comp_pairs.assign(ncomp, 0);
            comp_ents.assign(ncomp, 0);
            comp_tri.assign(ncomp, 0);
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t info = p.var_info[v0 + i];
                uint16_t c = info & fx::VAR_COMP_MASK;
                if (c != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT)) comp_free[c] += 1;
            }

            for (uint32_t i = 0; i < net; ++i) {
                const uint32_t e = e0 + i;
                const int tag = b->expr_tag[e];
                if (tag < 0 || tag >= n_tags) {
                    bad(FX_ERR_INVALID, s, "expression %u of system %u: bad tag %d", i, s, (unsigned)tag);
                    break;
                }
                const uint32_t* f = b->expr_idx + 4 * (size_t)e;
                uint32_t vars8[8];
                int k = fx::expand_vars<true>(tag, f, vars8);
                bool in_range = true;
                for (int q = 0; q < k; ++q) {
                    if (vars8[q] >= nvt) {
                        bad(FX_ERR_INVALID, s, "expression %u of system %u reads variable %u >= %u", i, s, vars8[q], nvt);
                        in_range = false;
                        break;
                    }
                }
                if (!in_range) break;
                for (int q = 0; q < 4; ++q) p.expr_idx16[4 * (size_t)e + q] = (uint16_t)(f[q] < nvt ? f[q] : 0);
                p.expr_tagx[e] = (uint8_t)tag;
                bool dup = false, all_free = true, distinct = true;
                for (int q = 0; q < k; ++q) {
                    all_free = all_free && free_rank[vars8[q]] >= 0;
                    for (int u = q + 1; u < k; ++u) {
                        distinct = distinct && vars8[q] != vars8[u];
                        dup = dup || (vars8[q] == vars8[u] && free_rank[vars8[q]] >= 0);
                    }
                }
                if (dup) p.expr_tagx[e] |= 0x80;
                uint16_t c = p.expr_comp[e];
                if (c != fx::VAR_COMP_NONE) {
                    comp_rows[c] += 1;
                    uint32_t kf = 0;  // entries with a free variable (an upper bound inside the kernel's component)
                    for (int q = 0; q < k; ++q) kf += free_rank[vars8[q]] >= 0;
                    comp_pairs[c] += kf * kf;
                    comp_ents[c] += kf;
                    // the lower triangle only: unordered pairs, plus once more for two entries on one column
                    uint32_t twice = 0;
                    for (int q = 0; q < k; ++q)
                        for (int u = q + 1; u < k; ++u) twice += vars8[q] == vars8[u] && free_rank[vars8[q]] >= 0;
                    comp_tri[c] += kf * (kf + 1u) / 2u + twice;
                }
                if (all_free && distinct) {
                    pt.nnz += (uint64_t)k;
                } else {
                    int32_t cols[8];
                    pt.nnz += (uint64_t)row_columns(vars8, k, free_rank.data(), cols, nullptr);
                }
            }
            if (pt.err != FX_OK) break;
            uint32_t cp_max = 0, ce_max = 0, ct_max = 0;
            for (uint32_t c = 0; c < ncomp; ++c) {
                cp_max = std::max(cp_max, comp_pairs[c]);
                ce_max = std::max(ce_max, comp_ents[c]);
                ct_max = std::max(ct_max, comp_tri[c]);
            }
            bool wide = false;  // more than one wavefront's columns, but still an LDS-resident dense problem
            uint32_t cf_max = 0, cr_max = 0;
            for (uint32_t c = 0; c < ncomp; ++c) {
                cf_max = std::max(cf_max, comp_free[c]);
                cr_max = std::max(cr_max, comp_rows[c]);
            }
            if (!large && cr_max <= FX_MAX_ROWS && cf_max > FX_MAX_FREE_VARS && cf_max <= FX_MAX_WIDE_FREE_VARS) wide = true;
            large = large || cf_max > FX_MAX_FREE_VARS || cr_max > FX_MAX_ROWS;
            if (large) {
                p.sys_large[s] = wide ? 2 : 1;
                pt.n_large += 1;
                pt.max_pairs_large = std::max(pt.max_pairs_large, cp_max);
                pt.max_ents_large = std::max(pt.max_ents_large, ce_max);
                if (wide) {
                    pt.w_max_free = std::max(pt.w_max_free, cf_max);
                    pt.w_max_rows = std::max(pt.w_max_rows, cr_max);
                    pt.w_max_vars = std::max(pt.w_max_vars, nvt);
                }
            } else {  // LDS layout and kernel instantiation are sized by the one-wavefront systems only
                pt.max_vars = std::max(pt.max_vars, nvt);
                pt.max_exprs = std::max(pt.max_exprs, net);
                pt.max_pairs = std::max(pt.max_pairs, cp_max);
                pt.max_ents = std::max(pt.max_ents, ce_max);
                pt.max_pairs_tri = std::max(pt.max_pairs_tri, ct_max);
                for (uint32_t c = 0; c < ncomp; ++c) {
                    pt.max_free = std::max(pt.max_free, comp_free[c]);
                    pt.max_rows = std::max(pt.max_rows, comp_rows[c]);
                }
            }
            prev_nnz = pt.nnz - nnz_before;
            prev_large = pt.n_large - large_before;
            if (s > s_lo && pt.chain_same)
                pt.chain_same = nvt == v0 - b->var_off[s - 1] && net == e0 - b->expr_off[s - 1] && outputs_equal(s, s - 1, nvt, net);
        }
    }, &nr, min_items);
    for (uint32_t t = 0; t < nr; ++t)  // ranges are in system order: the first failing System is reported
        if (part[t].err != FX_OK) return fail(part[t].err, "%s", part[t].msg);
    for (uint32_t t = 0; t < nr; ++t) {
        p.nnz += part[t].nnz;
        p.n_large += part[t].n_large;
        p.max_free = std::max(p.max_free, part[t].max_free);
        p.max_rows = std::max(p.max_rows, part[t].max_rows);
        p.max_vars = std::max(p.max_vars, part[t].max_vars);
        p.max_exprs = std::max(p.max_exprs, part[t].max_exprs);
        p.max_vars_all = std::max(p.max_vars_all, part[t].max_vars_all);
        p.max_exprs_all = std::max(p.max_exprs_all, part[t].max_exprs_all);
        p.max_pairs = std::max(p.max_pairs, part[t].max_pairs);
        p.max_ents = std::max(p.max_ents, part[t].max_ents);
        p.max_pairs_tri = std::max(p.max_pairs_tri, part[t].max_pairs_tri);
        p.max_pairs_large = std::max(p.max_pairs_large, part[t].max_pairs_large);
        p.max_ents_large = std::max(p.max_ents_large, part[t].max_ents_large);
        p.w_max_free = std::max(p.w_max_free, part[t].w_max_free);
        p.w_max_rows = std::max(p.w_max_rows, part[t].w_max_rows);
        p.w_max_vars = std::max(p.w_max_vars, part[t].w_max_vars);
    }
    for (uint32_t s = 0; s < n; ++s)
        if (p.sys_large[s] == 2) p.wide_list.push_back(s);
    // Components of 65 ... 128 columns have two homes. The wide kernel (fx_wide.hip: one wavefront per System, dense packed
    // factor in LDS) holds 4 / 2 / 1 Systems per CU; the team kernels (fx_sparse_team.h: a workgroup of 16 wavefronts per
    // System, sparse factor, the whole solve in one launch) finish ONE such System in half the time (66 variables: 0.17
    // against 0.36 ms) and cost half a microsecond per System at any size, which only the wide kernel's four-per-CU case
    // beats, and only from a thousand Systems on. By cost, measured on the reference's hinged-triangle sketches of 66 / 98 /
    // 126 variables, 1 ... 20 000 per batch (tools/hinged_batch.py):
    //   team  = max(0.18 ms, n (0.49 us + 0.0011 (c - 66)))   below 768 Systems (16 wavefronts per System);
    //           max(0.30 ms, n (0.19 us + 0.0014 (c - 66)))   from there on (2 / 4 wavefronts per System, round 4)
    //   wide  = ceil(n / (256 CUs x Systems per CU)) x (0.37 ms + 0.0123 (c - 66))
    // Both follow the reference's iteration path; they sum in different orders, so which one ran shows in the last bits (as it does
    // for a large System alone / among seven others): fx_ctx_set_wide_routing(ctx, 0 | 1) pins it.
    if (!p.wide_list.empty()) {
        const int routing = g_wide_routing_pinned != -2 ? g_wide_routing_pinned : g_wide_routing;
        bool team = routing == 0;
        if (routing < 0) {
            fx::DeviceBatch probe{};
            probe.w_max_free = p.w_max_free;
            probe.w_max_vars = p.w_max_vars;
            probe.w_max_rows = p.w_max_rows;
            const size_t lds = fx::wide_lds_bytes(probe);
            const double per_cu = lds ? std::min<double>(4., std::floor(160. * 1024. / (double)lds)) : 1.;
            const double c = (double)p.w_max_free - 66., nw = (double)p.wide_list.size();
            // (round 4: from 768 Systems on the team kernels run 2 / 4 wavefronts per System instead of 16 — fx_sparse.hip:
            // team_waves_for — 0.19 us per 66-variable System, 0.27 per 126-variable one)
            const double team_ms = nw < 768. ? std::max(0.18, nw * (0.49e-3 + 1.1e-6 * c)) : std::max(0.30, nw * (0.19e-3 + 1.4e-6 * c));
            const double wide_ms = std::ceil(nw / (256. * std::max(per_cu, 1.))) * (0.37 + 0.0123 * c);
            team = team_ms < wide_ms;
        }
        p.wide_decision = team ? 0 : 1;
        if (team) {
            for (uint32_t s : p.wide_list) p.sys_large[s] = 1;
            p.wide_list.clear();
            p.w_max_free = p.w_max_rows = p.w_max_vars = 0;
        }
    }
    // Same structure everywhere? (sizes, components, fixed flags, kinds and element fields of System 0.) The
    // grouped kernel then builds its per-System lists once per lane row instead of once per System.
    if (n >= 2) {
        const uint32_t nv0 = b->var_off[1] - b->var_off[0], ne0 = b->expr_off[1] - b->expr_off[0];
        bool same = (uint64_t)nv0 * n == nv && (uint64_t)ne0 * n == ne;
        // (inside a range the walk above has compared every System with the one before it: the ranges' first Systems
        // are left; sizes equal all along and the totals above make the offsets regular)
        for (uint32_t t = 0; same && t < nr; ++t) {
            const uint32_t s = part[t].first;
            same = part[t].chain_same && b->var_off[s + 1] - b->var_off[s] == nv0 && b->expr_off[s + 1] - b->expr_off[s] == ne0 &&
                   b->var_off[s] == s * nv0 && b->expr_off[s] == s * ne0 &&
                   memcmp(&p.var_info[(size_t)s * nv0], &p.var_info[0], nv0 * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_tagx[(size_t)s * ne0], &p.expr_tagx[0], ne0) == 0 &&
                   memcmp(&p.expr_comp[(size_t)s * ne0], &p.expr_comp[0], ne0 * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_idx16[4 * (size_t)s * ne0], &p.expr_idx16[0], 4 * (size_t)ne0 * sizeof(uint16_t)) == 0;
        }
        p.uniform = same ? 1u : 0u;
    }
    // Not one structure, but a batch the grouped kernel will take: the structure classes (a few sketches, each with
    // many parameter sets, is the other common batch). Class = first System with the same sizes, components, fixed
    // flags, kinds and element fields: found by a 64-bit hash of those arrays, confirmed by comparing them.
    if (!p.uniform && n >= 1024 && p.max_free > 0 && p.max_free <= 48) {
        auto slices = [&](uint32_t s, const void* ptr[4], size_t len[4]) {
            const uint32_t v0 = b->var_off[s], nvs = b->var_off[s + 1] - v0, e0 = b->expr_off[s], nes = b->expr_off[s + 1] - e0;
            ptr[0] = &p.var_info[v0];            len[0] = nvs * sizeof(uint16_t);
            ptr[1] = &p.expr_tagx[e0];           len[1] = nes;
            ptr[2] = &p.expr_comp[e0];           len[2] = nes * sizeof(uint16_t);
            ptr[3] = &p.expr_idx16[4 * (size_t)e0]; len[3] = 4 * (size_t)nes * sizeof(uint16_t);
        };
        std::vector<uint64_t> hash(n);
        parallel_ranges(n, (uint64_t)nv + ne, [&](uint32_t, uint32_t lo, uint32_t hi) {
            for (uint32_t s = lo; s < hi; ++s) {
                if (s > lo && p.same_as_prev[s]) {
                    hash[s] = hash[s - 1];
                    continue;
                }
                const void* ptr[4];
                size_t len[4];
                slices(s, ptr, len);
                uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)len[0] << 32) ^ len[1];
                for (int k = 0; k < 4; ++k) {
                    const unsigned char* q = static_cast<const unsigned char*>(ptr[k]);
                    size_t i = 0;
                    for (; i + 8 <= len[k]; i += 8) {
                        uint64_t w;
                        memcpy(&w, q + i, 8);
                        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
                        h ^= h >> 29;
                    }
                    uint64_t w = 0;
                    if (i < len[k]) memcpy(&w, q + i, len[k] - i);
                    h = (h ^ w ^ (uint64_t)k) * 0xC4CEB9FE1A85EC53ull;
                    h ^= h >> 32;
                }
                hash[s] = h;
            }
        });
        std::unordered_map<uint64_t, uint32_t> first;
        first.reserve(1024);
        p.sys_class.resize(n);
        for (uint32_t s = 0; s < n; ++s) {
            if (s && p.same_as_prev[s]) {
                p.sys_class[s] = p.sys_class[s - 1];
                continue;
            }
            auto it = first.find(hash[s]);
            if (it == first.end()) {
                first.emplace(hash[s], s);
                p.sys_class[s] = s;
                continue;
            }
            const void *pa[4], *pb[4];
            size_t la[4], lb[4];
            slices(s, pa, la);
            slices(it->second, pb, lb);
            bool eq = true;
            for (int k = 0; eq && k < 4; ++k) eq = la[k] == lb[k] && memcmp(pa[k], pb[k], la[k]) == 0;
            p.sys_class[s] = eq ? it->second : s;  // (a colliding hash: the System is its own class)
        }
    }
    return FX_OK;
}

}  // namespace

struct fx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    char name[128] = {0};
    char arch[64] = {0};
    // Device blocks released by freed batches, kept for the next upload: hipMalloc / hipFree synchronise
    // the device and cost more than a small solve (one System::solve = one upload + one free).
    struct Block { void* p; size_t size; };
    std::vector<Block> free_blocks;
    size_t free_bytes = 0;
    // routing of batches of small Systems (fx_ctx_set_routing)
    int route_grouped = -1;
    int grouped_one_structure = 1;  // the grouped kernel's build for batches of one structure (FIKSI_AMD_GROUPED_C=0: never)
    uint32_t grouped_min_systems = 8u;
    int presort = 1;                       // fx_ctx_set_presort
    uint32_t hold_passes = 2u;             // fx_ctx_set_hold_passes
    uint32_t ladder = 1u, ladder_k = 8u, ladder_tail = 0xFFFFFFFFu, ladder_spread = 1u;  // fx_ctx_set_ladder
    int wide_routing = -1;                 // fx_ctx_set_wide_routing
    uint32_t host_threads = 8u;            // fx_ctx_set_host_threads: groups of large Systems (one structure each) solved side by side
    std::vector<hipStream_t> worker_streams;  // ... a stream per extra host thread
    // Page-locked staging for one-shot solves up to 8 MB of batch (System::solve on one sketch ... some ten thousand small
    // Systems): first half carries the packed upload, second half the read-back — both copies are then truly
    // asynchronous, one each, and the call waits on the stream once.
    unsigned char* pinned = nullptr;
    static constexpr size_t PINNED_HALF = size_t(8) << 20;
    bool pinned_busy = false;  // an upload from the first half may still be in flight (ev_pinned follows its copy)
    hipEvent_t ev_pinned = nullptr;
    hipStream_t pinned_stream = nullptr;
    // a second stream: fx_system_solve_batch on a big batch of small Systems works in chunks, chunk k + 1 analysed and
    // uploaded while chunk k is solved (solve_host_chunked)
    hipStream_t stream2 = nullptr;  // the copies
    hipStream_t stream3 = nullptr;  // every other chunk's solve (the end of one chunk's solve overlaps the next one's start)
    hipEvent_t ev_chunk = nullptr;
    void wait_pinned() {
        if (pinned_busy) (void)hipEventSynchronize(ev_pinned);
        pinned_busy = false;
    }
    void stream_synced() {  // ctx->stream has just been waited for
        if (pinned_stream == stream) pinned_busy = false;
    }
    // Plans of the sparse path for one-shot calls (System::solve on a large sketch, again and again while it is dragged):
    // keyed by the System's structure and the solve mode, a handful kept, least recently used dropped. Values never
    // enter a plan, so a hit only skips the host planning and the upload of its index arrays.
    struct PlanEntry {
        std::vector<unsigned char> key;
        fx::SparsePlanCache* plan;
        uint64_t used;
    };
    std::vector<PlanEntry> plan_cache;
    uint64_t plan_clock = 0;
    static constexpr size_t MAX_PLANS = 8;
    // `call_clock` = plan_clock when the calling solve began: entries used since then belong to it and stay. When all
    // MAX_PLANS entries are this call's, the structure gets no cached plan (nullptr: the solve plans for itself).
    fx::SparsePlanCache* plan_for(std::vector<unsigned char>&& key, uint64_t call_clock) {
        for (PlanEntry& e : plan_cache)
            if (e.key == key) {
                e.used = ++plan_clock;
                return e.plan;
            }
        if (plan_cache.size() >= MAX_PLANS) {
            size_t old = plan_cache.size();
            for (size_t i = 0; i < plan_cache.size(); ++i)
                if (plan_cache[i].used <= call_clock && (old == plan_cache.size() || plan_cache[i].used < plan_cache[old].used)) old = i;
            if (old == plan_cache.size()) return nullptr;
            fx::sparse_cache_free(plan_cache[old].plan);
            plan_cache.erase(plan_cache.begin() + (long)old);
        }
        plan_cache.push_back({std::move(key), fx::sparse_cache_new(), ++plan_clock});
        return plan_cache.back().plan;
    }
    void drop_plans() {
        for (PlanEntry& e : plan_cache) fx::sparse_cache_free(e.plan);
        plan_cache.clear();
    }
    // One-shot solves of a handful of small Systems (System::solve on one sketch): the batch's one block lives in
    // host-coherent page-locked memory that the kernel reads and writes directly — no copy call either way, one launch
    // and one wait on the stream per call (a copy call costs more than such a kernel runs).
    unsigned char* zc = nullptr;
    static constexpr size_t ZC_BYTES = size_t(64) << 10;
    bool ensure_zc() {
        if (zc) return true;
        void* p = nullptr;
        if (hipHostMalloc(&p, ZC_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
        zc = static_cast<unsigned char*>(p);
        return true;
    }
    bool ensure_pinned() {
        if (pinned) return true;
        void* p = nullptr;
        if (!ev_pinned && hipEventCreateWithFlags(&ev_pinned, hipEventDisableTiming) != hipSuccess) return false;
        if (hipHostMalloc(&p, 2 * PINNED_HALF, 0) != hipSuccess) return false;
        pinned = static_cast<unsigned char*>(p);
        return true;
    }
    uint32_t presort_min_systems = 8192u;
    void route(fx::LmParams& p) const {
        p.route_grouped = route_grouped;
        p.grouped_one_structure = grouped_one_structure;
        p.grouped_min_systems = grouped_min_systems;
        p.hold_passes = hold_passes;
        p.ladder = ladder;
        p.ladder_k = ladder_k;
        p.ladder_tail = ladder_tail;
        p.spread = ladder_spread;
    }
    static constexpr size_t MAX_CACHED_BYTES = size_t(4) << 30;  // beyond this, released blocks go back to the driver

    void* take(size_t bytes, hipError_t& err) {
        err = hipSuccess;
        size_t best = free_blocks.size();
        for (size_t i = 0; i < free_blocks.size(); ++i)
            if (free_blocks[i].size >= bytes && free_blocks[i].size <= 2 * bytes + 4096 &&
                (best == free_blocks.size() || free_blocks[i].size < free_blocks[best].size))
                best = i;
        if (best != free_blocks.size()) {
            void* p = free_blocks[best].p;
            free_bytes -= free_blocks[best].size;
            free_blocks.erase(free_blocks.begin() + (long)best);  // (keeps the list in order of release: give_back evicts from the front)
            return p;
        }
        void* p = nullptr;
        err = hipMalloc(&p, bytes);
        if (err == hipErrorOutOfMemory && !free_blocks.empty()) {  // give the cache back and retry once
            drop_cache();
            err = hipMalloc(&p, bytes);
        }
        if (err == hipErrorOutOfMemory && !plan_cache.empty()) {  // ... then the kept plans of one-shot calls
            drop_plans();
            err = hipMalloc(&p, bytes);
        }
        return err == hipSuccess ? p : nullptr;
    }
    void give_back(void* p, size_t bytes) {
        if (bytes > MAX_CACHED_BYTES) {
            (void)hipFree(p);
            return;
        }
        // full: the blocks cached longest go back to the driver, not the one that was in use a moment ago (a list full of
        // small blocks used to turn every big batch's block into a hipFree + hipMalloc pair per call)
        while (!free_blocks.empty() && (free_bytes + bytes > MAX_CACHED_BYTES || free_blocks.size() >= 256)) {
            (void)hipFree(free_blocks.front().p);
            free_bytes -= free_blocks.front().size;
            free_blocks.erase(free_blocks.begin());
        }
        free_blocks.push_back({p, bytes});
        free_bytes += bytes;
    }
    void drop_cache() {
        for (auto& b : free_blocks) (void)hipFree(b.p);
        free_blocks.clear();
        free_bytes = 0;
    }
};

struct fx_dbatch {
    fx::DeviceBatch d{};
    std::vector<fx_ctx::Block> allocations;
    // A batch of SEVERAL structures (a few sketches, many parameter sets each): its big structure classes, each solved by a
    // launch of the grouped kernel's one-structure build over the class's member list (launch_class_solves); `rest`: everyone else
    std::vector<fx::GcClass> classes;  // (programs inside cl_words, members inside cl_lists)
    fx::GcClass* cl_desc = nullptr;    // ... on the device
    uint32_t* cl_words = nullptr;
    uint32_t* cl_lists = nullptr;
    uint32_t cl_max_words_all = 0;
    uint32_t cl_rc = 0;
    uint32_t cl_nc = 0, cl_max_words = 0, cl_max_slots = 0, cl_max_ng = 0, cl_systems = 0;  // the classes' common build, the largest program, their Systems in all
    uint32_t rest_off = 0, rest_count = 0;
    // host copy of the batch, kept only when some System needs the sparse path
    std::vector<uint32_t> h_var_off, h_expr_off, h_expr_idx;
    std::vector<double> h_vars, h_expr_param;
    std::vector<uint8_t> h_var_fixed, h_expr_tag, h_sys_large;
    std::vector<uint8_t> h_units_on_device;  // SinglePass: large Systems the GLOBAL kernel instantiation walks
    std::vector<uint8_t> h_qr_wide;          // FX_STEP_QR: Systems beyond one wavefront the wide kernel's QR build solves (ensure_qr_plans)
    bool qr_wide_active = false;             // ... and they have just been solved that way: the sparse path leaves them alone
    // sparse-path plans of the batch's large Systems, one per structure and decomposer mode (hash -> candidates)
    struct ResidentPlan {
        std::vector<unsigned char> key;
        fx::SparsePlanCache* plan;
    };
    std::multimap<uint64_t, ResidentPlan> sparse_plans;
    // ... and the grouping of those Systems by structure, per solve mode: it reads structure only, so a resident batch
    // works it out once (building and hashing a System's key is ~4 us per 258-variable sketch — more than the solve of a
    // batch of them once that is one launch)
    struct StructureGroup {
        std::vector<unsigned char> key;
        uint64_t hash = 0;
        std::vector<uint32_t> systems;
    };
    std::map<uint32_t, std::vector<StructureGroup>> large_groups;
    // Decomposer::None on large Systems made of small components: the component walk (a DeviceBatch
    // whose unit arrays list whole components), built on first use
    fx::DeviceBatch comp_walk{};
    bool comp_walk_built = false;
    std::vector<uint8_t> h_comp_walk;  // per System: 1 = walked on the device
    uint32_t n_units = 0, n_unit_rows = 0, n_unit_vars = 0;  // sizes of the SinglePass block arrays on the device
    uint32_t* d_order = nullptr;  // fx_batch_schedule_by_last_solve
    // longest-first hand-out from a scout pass (fx_presort.hip): keys / ids [2][n], the sort's workspace
    float* ps_keys = nullptr;
    uint32_t* ps_ids = nullptr;
    unsigned char* ps_temp = nullptr;
    size_t ps_temp_bytes = 0;
    unsigned char* packed_base = nullptr;  // small batches: the one block all arrays live in
    size_t packed_bytes = 0;
    bool upload_pending = false;           // ... and its copy from the context's page-locked staging was not waited for
    bool zero_copy = false;                // one-shot solves of a few small Systems: the host writes the block's image into the context's
    unsigned char* zc_image = nullptr;     // host-coherent region, a kernel pulls it over, another pushes vars / results (the block's end,
    size_t zc_front = 0;                   // from zc_front on) back
    bool resident = false;  // uploaded by the caller (plans are worth keeping); false for the one-shot host entry points
    std::vector<uint16_t> h_var_comp, h_expr_comp;
    fx_batch h_batch{};
    uint32_t n_large = 0;
};

namespace {

template <typename T>
int dev_alloc_copy(fx_ctx* ctx, fx_dbatch* db, T** out, const T* host, size_t count) {
    *out = nullptr;
    size_t bytes = ((std::max<size_t>(count, 1) * sizeof(T)) + 255u) & ~size_t(255);
    hipError_t e = hipSuccess;
    void* p = ctx->take(bytes, e);
    if (!p) return fail(e == hipErrorOutOfMemory ? FX_ERR_NOMEM : FX_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    db->allocations.push_back({p, bytes});
    if (host && count) {
        FX_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    } else {
        FX_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream));
    }
    *out = static_cast<T*>(p);
    return FX_OK;
}

int bind(fx_ctx* ctx) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    FX_HIP(hipSetDevice(ctx->device));
    return FX_OK;
}

}  // namespace

namespace {
// The arrays of the row-parallel kernels (EvalPlan) and the residual buffer, on first use: the structure is read back
// from the device's own compact arrays (nothing is kept on the host for batches that are only ever solved).
int ensure_resid(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.resid) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1, 0), expr_off((size_t)n + 1, 0);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs);
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    EvalPlan ep;
    build_eval_plan(n, var_off.data(), expr_off.data(), var_info.data(), expr_tag.data(), expr_idx.data(), ep);
    int rc = dev_alloc_copy(ctx, db, &d.expr_var0, ep.expr_var0.data(), ep.expr_var0.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.row_perm, ep.row_perm.data(), ep.row_perm.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.row_sysoff, ep.row_sysoff.data(), ep.row_sysoff.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.blk_info, ep.blk_info.data(), ep.blk_info.size());
    double* resid = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &resid, (const double*)nullptr, d.n_exprs);
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    d.resid = resid;  // set last: marks the arrays as complete
    return FX_OK;
}

// The CSR Jacobian structure of a resident batch, built on first use from the device's own compact
// arrays (nothing is kept on the host for batches that are only ever solved).
int ensure_csr(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.jrow_ptr) return FX_OK;
    int rc = ensure_resid(ctx, db);
    if (rc) return rc;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1, 0), expr_off((size_t)n + 1, 0);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs);
    const size_t nblk = ((size_t)d.n_exprs + 255) / 256;
    std::vector<fx::BlockInfo> blk(nblk);
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(blk.data(), d.blk_info, nblk * sizeof(fx::BlockInfo), hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    CsrPlan csr;
    build_csr(n, var_off.data(), expr_off.data(), var_info.data(), expr_tag.data(), expr_idx.data(), csr);
    if (csr.jcol.size() != d.nnz) return fail(FX_ERR_INVALID, "internal: CSR size %zu != counted %llu", csr.jcol.size(), (unsigned long long)d.nnz);
    for (size_t k = 0; k < nblk; ++k) {
        const size_t r0 = k * 256, r1 = std::min<size_t>(r0 + 256, d.n_exprs);
        blk[k].jbase = csr.jrow_ptr[r0];
        blk[k].jcount = csr.jrow_ptr[r1] - csr.jrow_ptr[r0];
    }
    uint32_t* jrow_ptr = nullptr;
    rc = dev_alloc_copy(ctx, db, &jrow_ptr, csr.jrow_ptr.data(), csr.jrow_ptr.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jcol, csr.jcol.data(), csr.jcol.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jslot, csr.jslot.data(), csr.jslot.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jvals, (const double*)nullptr, d.nnz);
    if (rc) return rc;
    if (nblk) FX_HIP(hipMemcpyAsync(d.blk_info, blk.data(), nblk * sizeof(fx::BlockInfo), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    d.jrow_ptr = jrow_ptr;  // set last: marks the structure as complete
    return FX_OK;
}

// Builds the SinglePass blocks of every System that runs in the fused kernel (once per batch; the
// structure is read back from the device arrays, so nothing extra is kept on the host for batches
// that never ask for it). Large Systems get theirs inside the sparse path.
int ensure_units(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.sys_unit_off) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1), expr_off((size_t)n + 1);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs), sys_large(n);
    std::vector<uint16_t> sys_ncomp(n);
    if (db->packed_base && (ctx->pinned || db->zero_copy) && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        // a small batch is one block on the device: one copy of it (page-locked, second half of the staging area) instead
        // of seven small ones — the structure arrays are then taken from that image
        // (a zero-copy block is host memory already)
        unsigned char* img = db->zero_copy ? db->zc_image : ctx->pinned + fx_ctx::PINNED_HALF;
        if (!db->zero_copy) FX_HIP(hipMemcpyAsync(img, db->packed_base, db->packed_bytes, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stream_synced();
        auto grab = [&](void* dst, const void* dev, size_t bytes) {
            if (bytes) memcpy(dst, img + (reinterpret_cast<const unsigned char*>(dev) - db->packed_base), bytes);
        };
        grab(var_off.data(), d.var_off, var_off.size() * 4);
        grab(expr_off.data(), d.expr_off, expr_off.size() * 4);
        grab(var_info.data(), d.var_info, var_info.size() * 2);
        grab(expr_idx.data(), d.expr_idx, expr_idx.size() * 2);
        grab(expr_tag.data(), d.expr_tag, expr_tag.size());
        grab(sys_large.data(), d.sys_large, n);
        grab(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2);
    } else {
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n) {
        FX_HIP(hipMemcpyAsync(sys_large.data(), d.sys_large, n, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    }

    std::vector<uint32_t> sys_unit_off((size_t)n + 1, 0);
    std::vector<fx::UnitDesc> desc;
    std::vector<uint32_t> unit_rows;
    std::vector<uint16_t> unit_vars;
    uint32_t max_unit_free = 0, max_unit_rows = 0;
    // Large Systems: if every block fits the one-wavefront limits the System is walked on the device
    // with its System-wide vectors in an HBM scratch area (the GLOBAL instantiation); otherwise it
    // stays with the host-driven sparse path.
    std::vector<uint32_t> g_list, g_off;
    std::vector<uint8_t> g_ok(n, 0);
    uint32_t g_total = 0, max_unit_free_g = 0, max_unit_rows_g = 0;

    // Systems are independent: ranges of them are decomposed on separate host threads into local
    // lists (offsets relative to the range), stitched together in System order afterwards.
    struct Range {
        uint32_t s_lo = 0, s_hi = 0;
        std::vector<fx::UnitDesc> desc;
        std::vector<uint32_t> rows, desc_count;  // desc_count[s - s_lo]: entries of System s
        std::vector<uint16_t> vars;
        std::vector<uint32_t> g_sys, g_nvt;
        uint32_t max_free = 0, max_rows = 0, max_free_g = 0, max_rows_g = 0;
    } ranges[MAX_RANGES];
    uint32_t n_ranges = 1;
    parallel_ranges(n, (uint64_t)d.n_exprs * 4, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Range& R = ranges[t];
        R.s_lo = s_lo;
        R.s_hi = s_hi;
        R.desc_count.assign(s_hi - s_lo, 0);
        fx::Incidence inc;
        fx::UnitList units;
        std::vector<uint32_t> free_sorted;
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const size_t desc_mark = R.desc.size(), rows_mark = R.rows.size(), vars_mark = R.vars.size();
            bool fits = true;
            uint32_t mf = 0, mrw = 0;
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0;
            const uint32_t e0 = expr_off[s], net = expr_off[s + 1] - e0;
            inc.build(nvt, net, expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0);
            fx::SinglePassDecomposer dec(inc);
            for (uint32_t c = 0; c < sys_ncomp[s]; ++c) {
                free_sorted.clear();
                bool any_var = false;
                for (uint32_t i = 0; i < nvt; ++i) {
                    uint16_t info = var_info[v0 + i];
                    if ((info & fx::VAR_COMP_MASK) != c) continue;
                    any_var = true;
                    if (!(info & fx::VAR_FIXED_BIT)) free_sorted.push_back(i);
                }
                if (!any_var) continue;  // skipped by the reference (`elements.is_empty()`)
                dec.run(free_sorted, units);
                if (units.count() == 0) {
                    R.desc.push_back(fx::UnitDesc{0, 0, 0, 0, (uint16_t)c, (uint16_t)(fx::UNIT_FIRST | fx::UNIT_EMPTY)});
                    continue;
                }
                for (uint32_t u = 0; u < units.count(); ++u) {
                    fx::UnitDesc ud{};
                    ud.row_off = (uint32_t)R.rows.size();
                    ud.var_off = (uint32_t)R.vars.size();
                    ud.nrows = (uint16_t)(units.row_off[u + 1] - units.row_off[u]);
                    ud.nvars = (uint16_t)(units.var_off[u + 1] - units.var_off[u]);
                    ud.comp = (uint16_t)c;
                    ud.flags = u == 0 ? fx::UNIT_FIRST : 0;
                    for (uint32_t k = units.row_off[u]; k < units.row_off[u + 1]; ++k) R.rows.push_back(units.rows[k]);
                    for (uint32_t k = units.var_off[u]; k < units.var_off[u + 1]; ++k) R.vars.push_back((uint16_t)units.vars[k]);
                    const uint32_t bf = units.var_off[u + 1] - units.var_off[u], br = units.row_off[u + 1] - units.row_off[u];
                    fits = fits && bf <= FX_MAX_FREE_VARS && br <= FX_MAX_ROWS;
                    mf = std::max(mf, bf);
                    mrw = std::max(mrw, br);
                    R.desc.push_back(ud);
                }
            }
            if (!sys_large[s]) {
                R.max_free = std::max(R.max_free, mf);
                R.max_rows = std::max(R.max_rows, mrw);
            } else if (fits) {
                R.g_sys.push_back(s);
                R.g_nvt.push_back(nvt);
                R.max_free_g = std::max(R.max_free_g, mf);
                R.max_rows_g = std::max(R.max_rows_g, mrw);
            } else {  // some block is itself too large: the System keeps the sparse path, drop its entries
                R.desc.resize(desc_mark);
                R.rows.resize(rows_mark);
                R.vars.resize(vars_mark);
            }
            R.desc_count[s - s_lo] = (uint32_t)(R.desc.size() - desc_mark);
        }
    }, &n_ranges);
    for (uint32_t t = 0; t < n_ranges; ++t) {
        Range& R = ranges[t];
        const uint32_t row_base = (uint32_t)unit_rows.size(), var_base = (uint32_t)unit_vars.size();
        uint32_t at = (uint32_t)desc.size();
        for (uint32_t s = R.s_lo; s < R.s_hi; ++s) {
            sys_unit_off[s] = at;
            at += R.desc_count[s - R.s_lo];
        }
        for (fx::UnitDesc ud : R.desc) {
            ud.row_off += row_base;
            ud.var_off += var_base;
            desc.push_back(ud);
        }
        unit_rows.insert(unit_rows.end(), R.rows.begin(), R.rows.end());
        unit_vars.insert(unit_vars.end(), R.vars.begin(), R.vars.end());
        for (size_t k = 0; k < R.g_sys.size(); ++k) {
            g_ok[R.g_sys[k]] = 1;
            g_list.push_back(R.g_sys[k]);
            g_off.push_back(g_total);
            g_total += R.g_nvt[k];
        }
        max_unit_free = std::max(max_unit_free, R.max_free);
        max_unit_rows = std::max(max_unit_rows, R.max_rows);
        max_unit_free_g = std::max(max_unit_free_g, R.max_free_g);
        max_unit_rows_g = std::max(max_unit_rows_g, R.max_rows_g);
    }
    sys_unit_off[n] = (uint32_t)desc.size();
    int rc = dev_alloc_copy(ctx, db, &d.unit_desc, desc.data(), desc.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.unit_rows, unit_rows.data(), unit_rows.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.unit_vars, unit_vars.data(), unit_vars.size());
    uint32_t* off = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &off, sys_unit_off.data(), sys_unit_off.size());
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    if (!g_list.empty()) {
        rc = dev_alloc_copy(ctx, db, &d.g_list, g_list.data(), g_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_off, g_off.data(), g_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_xs, (const double*)nullptr, 2 * (size_t)g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_vout, (const double*)nullptr, g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_colof, (const int16_t*)nullptr, g_total);
        if (rc) return rc;
        FX_HIP(hipStreamSynchronize(ctx->stream));
        d.n_g = (uint32_t)g_list.size();
        d.max_unit_free_g = max_unit_free_g;
        d.max_unit_rows_g = max_unit_rows_g;
    }
    db->h_units_on_device = g_ok;
    d.max_unit_free = max_unit_free;
    d.max_unit_rows = max_unit_rows;
    if (fx::solve_lds_bytes_units(d) > 160u * 1024u)
        return fail(FX_ERR_TOO_LARGE, "SinglePass blocks need %zu bytes of LDS per wavefront (limit 163840)", fx::solve_lds_bytes_units(d));
    db->n_units = (uint32_t)desc.size();
    db->n_unit_rows = (uint32_t)unit_rows.size();
    db->n_unit_vars = (uint32_t)unit_vars.size();
    d.sys_unit_off = off;  // set last: marks the plan as complete
    return FX_OK;
}

// ---- FX_STEP_QR: plans of the reference's sparse QR (fx_qrplan.h) ---------------------------------------
// One component (or SinglePass block) of one System: rows = its expressions in row order, free = its free
// variables in column order (both system-local). The pattern of the augmented matrix [J; sqrt(lambda) I] as
// lm.rs:81-98 builds it: column c holds the rows that read free[c] (ascending, an expression reading it twice
// once) and, last, the damping row m + c.
struct QrHostPlan {
    uint32_t n = 0, m = 0;
    std::vector<uint16_t> u16;  // colperm[n], rowperm[m + n], hptr[n + 1], hrows[nnzh]
    std::vector<uint64_t> u64;  // colmask[n], rowmask[n]
    uint32_t nnzh = 0;
    bool ok = false;
};

bool build_qr_plan(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_,
                   uint32_t n, uint32_t nvt, QrHostPlan& out) {
    out = QrHostPlan();
    out.n = n;
    out.m = m;
    if (n > 64u) return false;
    if (n == 0) {  // every variable of the component is fixed: the reference's LM takes one trial with an empty step
        for (uint32_t i = 0; i < m; ++i) out.u16.push_back((uint16_t)i);
        out.u16.push_back(0);
        out.ok = true;
        return true;
    }
    std::vector<int32_t> colof(nvt, -1);
    for (uint32_t c = 0; c < n; ++c) colof[free_[c]] = (int32_t)c;
    std::vector<std::vector<int>> cols(n);
    for (uint32_t r = 0; r < m; ++r) {
        uint32_t vars8[8];
        const int k = fx::expand_vars<true>((int)(expr_tag[rows[r]] & 0x7F), expr_idx16 + 4 * (size_t)rows[r], vars8);
        for (int q = 0; q < k; ++q) {
            const int32_t c = vars8[q] < nvt ? colof[vars8[q]] : -1;
            if (c >= 0 && (cols[c].empty() || cols[c].back() != (int)r)) cols[c].push_back((int)r);
        }
    }
    fx::qr::Csc a;
    a.nrows = (int)(m + n);
    a.ncols = (int)n;
    a.ptr.assign(1, 0);
    for (uint32_t c = 0; c < n; ++c) {
        a.idx.insert(a.idx.end(), cols[c].begin(), cols[c].end());
        a.idx.push_back((int)(m + c));
        a.ptr.push_back((int)a.idx.size());
    }
    fx::qr::Symbolic sy;
    if (!fx::qr::analyze(a, true, sy)) return false;
    if (sy.hrows.size() > 0xFFFFu) return false;
    // The rows of every Householder vector below its diagonal entry are padded to a multiple of eight with row m + n — the
    // row of zeros the kernel keeps under the matrix: its register window then loads, multiplies and stores whole blocks of
    // eight with no per-entry select (a padded slot adds 0 * x = +0.0 to a sum that is never -0.0, and writes back the
    // zero it read). hptr counts the padded entries.
    std::vector<uint16_t> hptr_p(n + 1, 0), hrows_p;
    for (uint32_t j = 0; j < n; ++j) {
        hptr_p[j] = (uint16_t)hrows_p.size();
        const int b = sy.hptr[j], e = sy.hptr[j + 1];
        for (int q = b; q < e; ++q) hrows_p.push_back((uint16_t)sy.hrows[q]);
        const int below = e - b - 1;
        for (int q = below; q < ((below + 7) & ~7); ++q) hrows_p.push_back((uint16_t)(m + n));
        if (hrows_p.size() > 0xFFFFu) return false;
    }
    hptr_p[n] = (uint16_t)hrows_p.size();
    out.nnzh = (uint32_t)hrows_p.size();
    out.u16.reserve(n + (m + n) + (n + 1) + hrows_p.size());
    for (uint32_t j = 0; j < n; ++j) out.u16.push_back((uint16_t)sy.col_perm[j]);
    for (uint32_t i = 0; i < m + n; ++i) out.u16.push_back((uint16_t)sy.row_perm[i]);
    out.u16.insert(out.u16.end(), hptr_p.begin(), hptr_p.end());
    out.u16.insert(out.u16.end(), hrows_p.begin(), hrows_p.end());
    out.u64.assign(2 * (size_t)n, 0);
    for (uint32_t j = 0; j < n; ++j)
        for (int p = sy.rptr[j]; p < sy.rptr[j + 1] - 1; ++p) {
            const uint32_t k = (uint32_t)sy.rrows[p];
            out.u64[j] |= 1ull << k;
            out.u64[n + k] |= 1ull << j;
        }
    out.ok = true;
    return true;
}

// The program of the grouped kernel's one-structure build (fx_grouped_c.hip): everything about a System's STRUCTURE that kernel
// needs, written once for a batch whose Systems all share it — one component, at most NV = 32 (48) variables and expressions
// (every expression a row of the component), 17 ... 32 (33 ... 48) free variables: two (three) matrix columns per lane. Jt J is kept by its pattern: a slot per structural non-zero of
// the lower triangle (all NV diagonal entries included: the columns past the free variables are identity padding), one slot of
// zero behind them. Words:
// [0] version [1] variables [2] expressions [3] free variables [4] products (padded to 64) [5] right-hand-side entries (padded
// to 64) [6] slots (even, the zero slot included) [7] compact Jacobian entries (even) [8] the zero slot [17] words in all [18] words
// of the part the f64 builds copy [9 ... 12] byte offsets of the f32 build's gather tables behind it;
// then, at the byte offsets of fx_device.h's GcTable: vcol (i8 [NV]: variable -> free column, -1 = fixed), fidx (u8 [NV]: free
// column -> variable), rtag (u8 [NV]), gbase (u16 [NV]: first compact entry of a row), gvar (u8 [NV][8]: the variables a row
// reads, gradient order), the load table (u8 [16][NC NV]: lane l's element i of column l + 16 q at [l][NV q + i] — the slot of
// (max, min), or the zero slot), the right-hand side (u32: entry | row << 8 | column << 16) and behind it the products (u32:
// entry a | entry b << 8 | slot << 16, 0xFFFFFFFF = padding; rows ascending, a ascending, b from a upward, a pair of entries on
// one column twice — the order fx_grouped.hip builds its lists in, and so the order of the additions).
struct GcHostProgram {
    std::vector<uint32_t> words;
    uint32_t nslots = 0, ng = 0, nc = 0, rc = 0, words_f64 = 0;  // (words_f64: the part the f64 builds use)
};
template <int NC, int RC>
static bool build_gc_program_t(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                               uint32_t nvt, uint32_t net, GcHostProgram& out) {  // (free variables: 1 ... NV; the caller picks the smallest build)
    using TK = fx::GcTable<NC, RC>;
    constexpr uint32_t NV = TK::NV, NR = TK::NR;
    out = GcHostProgram();
    out.nc = NC;
    out.rc = RC;
    if (nvt == 0 || nvt > NV || net == 0 || net > NR) return false;
    int8_t vcol[NV];
    uint8_t fidx[NV] = {0}, rtag[NR] = {0}, gvar[NR][8] = {{0}};
    uint16_t gbase[NR] = {0};
    uint32_t nfree = 0;
    for (uint32_t i = 0; i < NV; ++i) vcol[i] = -1;
    for (uint32_t i = 0; i < nvt; ++i) {
        if ((var_info[i] & fx::VAR_COMP_MASK) != 0) return false;  // (a variable of no component carries another number)
        if (!(var_info[i] & fx::VAR_FIXED_BIT)) {
            vcol[i] = (int8_t)nfree;
            fidx[nfree++] = (uint8_t)i;
        }
    }
    if (nfree == 0 || nfree > NV) return false;
    int gcol[NR][8];
    uint32_t ng = 0;
    for (uint32_t r = 0; r < net; ++r) {
        if (expr_comp[r] != 0) return false;
        const int tag = (int)(expr_tag[r] & 0x7F);
        if (tag >= FX_TAG_POSE_X) return false;
        uint32_t vars8[8];
        const int k = fx::expand_vars(tag, expr_idx16 + 4 * (size_t)r, vars8);
        rtag[r] = (uint8_t)tag;
        gbase[r] = (uint16_t)ng;
        for (int e = 0; e < 8; ++e) {
            if (vars8[e] >= nvt) return false;
            gvar[r][e] = (uint8_t)vars8[e];
            gcol[r][e] = e < k ? (int)vcol[vars8[e]] : -1;
        }
        ng += (uint32_t)k;
    }
    if (ng > 256u) return false;  // (a byte per compact Jacobian entry in the lists)
    // the pattern of the lower triangle, slots in packed-triangle order
    std::vector<int32_t> slot_of(NV * (NV + 1u) / 2u, -1);
    auto tri = [](uint32_t hi, uint32_t lo) { return hi * (hi + 1u) / 2u + lo; };
    for (uint32_t j = 0; j < NV; ++j) slot_of[tri(j, j)] = 0;
    for (uint32_t r = 0; r < net; ++r)
        for (int a = 0; a < 8; ++a)
            for (int bb = a; bb < 8; ++bb)
                if (gcol[r][a] >= 0 && gcol[r][bb] >= 0) {
                    const uint32_t ca = (uint32_t)gcol[r][a], cb = (uint32_t)gcol[r][bb];
                    slot_of[tri(std::max(ca, cb), std::min(ca, cb))] = 0;
                }
    uint32_t nslots = 0;
    for (int32_t& sl : slot_of)
        if (sl == 0) sl = (int32_t)nslots++;
    const uint32_t zero = nslots++;
    nslots = (nslots + 3u) & ~3u;  // (whole 16-byte vectors in f32 too)
    if (nslots > 256u) return false;
    std::vector<uint32_t> pw, pe;
    for (uint32_t r = 0; r < net; ++r)
        for (int a = 0; a < 8; ++a) {
            if (gcol[r][a] < 0) continue;
            const uint32_t ca = (uint32_t)gcol[r][a];
            pe.push_back((gbase[r] + (uint32_t)a) | (r << 8) | (ca << 16));
            for (int bb = a; bb < 8; ++bb) {
                if (gcol[r][bb] < 0) continue;
                const uint32_t cb = (uint32_t)gcol[r][bb];
                const uint32_t w = (gbase[r] + (uint32_t)a) | ((gbase[r] + (uint32_t)bb) << 8) |
                                   ((uint32_t)slot_of[tri(std::max(ca, cb), std::min(ca, cb))] << 16);
                pw.push_back(w);
                if (a != bb && ca == cb) pw.push_back(w);
            }
        }
    while (pw.size() % 64u) pw.push_back(0xFFFFFFFFu);
    while (pe.size() % 64u) pe.push_back(0xFFFFFFFFu);
    std::vector<uint8_t> lt((size_t)16 * NC * NV);
    for (uint32_t l = 0; l < 16u; ++l)
        for (uint32_t q = 0; q < (uint32_t)NC; ++q)
            for (uint32_t i = 0; i < NV; ++i) {
                const uint32_t j = l + 16u * q;
                const int32_t sl = slot_of[tri(std::max(i, j), std::min(i, j))];
                lt[(size_t)l * NC * NV + NV * q + i] = (uint8_t)(sl >= 0 ? (uint32_t)sl : zero);
            }
    std::vector<uint32_t>& w = out.words;
    w.assign(20, 0);
    auto put = [&](const void* src, size_t bytes) -> uint32_t {
        const uint32_t at = (uint32_t)w.size() * 4u;
        w.resize(w.size() + (bytes + 15u) / 16u * 4u, 0u);
        memcpy(reinterpret_cast<unsigned char*>(w.data()) + at, src, bytes);
        return at;
    };
    // (the tables are a multiple of 16 bytes each, so `put` places them back to back where GcTable says)
    bool placed = put(vcol, sizeof(vcol)) == TK::VCOL;
    placed = put(fidx, sizeof(fidx)) == TK::FIDX && placed;
    placed = put(rtag, sizeof(rtag)) == TK::RTAG && placed;
    placed = put(gbase, sizeof(gbase)) == TK::GBASE && placed;
    placed = put(gvar, sizeof(gvar)) == TK::GVAR && placed;
    placed = put(lt.data(), lt.size()) == TK::LT && placed;
    placed = put(pe.data(), pe.size() * 4u) == TK::PE && placed;
    (void)put(pw.data(), pw.size() * 4u);
    if (!placed) return false;
    w[18] = (uint32_t)w.size();  // what the f64 builds copy; behind it, for the f32 build:
    // the same products and right-hand-side entries by TARGET — slot by slot, column by column, each target's in list order — for
    // an assembly without LDS float atomics (ds_add_f32 costs 192 cycles an instruction on gfx950, ds_add_f64 14:
    // tools/probes/lds_atomic_f32_probe.hip). u16 each: entry a | entry b << 8; entry | row << 8.
    {
        std::vector<std::vector<uint16_t>> by_slot(nslots), by_col(NV);
        for (uint32_t x : pw)
            if (x != 0xFFFFFFFFu) by_slot[x >> 16].push_back((uint16_t)(x & 0xFFFFu));
        for (uint32_t x : pe)
            if (x != 0xFFFFFFFFu) by_col[x >> 16].push_back((uint16_t)(x & 0xFFFFu));
        std::vector<uint16_t> sptr(1, 0), spw, cptr(1, 0), cpe;
        for (auto& v : by_slot) {
            spw.insert(spw.end(), v.begin(), v.end());
            sptr.push_back((uint16_t)spw.size());
        }
        for (auto& v : by_col) {
            cpe.insert(cpe.end(), v.begin(), v.end());
            cptr.push_back((uint16_t)cpe.size());
        }
        w[9] = put(sptr.data(), sptr.size() * 2u);
        w[10] = put(spw.data(), spw.size() * 2u);
        w[11] = put(cptr.data(), cptr.size() * 2u);
        w[12] = put(cpe.data(), cpe.size() * 2u);
    }
    w[0] = 1u;
    w[1] = nvt;
    w[2] = net;
    w[3] = nfree;
    w[4] = (uint32_t)pw.size();
    w[5] = (uint32_t)pe.size();
    w[6] = nslots;
    w[7] = (ng + 3u) & ~3u;
    w[8] = zero;
    w[17] = (uint32_t)w.size();
    out.nslots = nslots;
    out.ng = (ng + 3u) & ~3u;
    out.words_f64 = w[18];
    return true;
}
static bool build_gc_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                             uint32_t nvt, uint32_t net, uint32_t max_free, GcHostProgram& out) {
    // the smallest build that holds the structure: its free variables decide, unless its variables (fixed ones included) or its
    // expressions need the next one's tables — the columns past the free variables are identity padding either way
    // (... and an over-constrained structure the instantiation with twice the rows)
    if (max_free <= 16u && build_gc_program_t<1, 1>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 16u && build_gc_program_t<1, 2>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 32u && build_gc_program_t<2, 2>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 32u && build_gc_program_t<2, 4>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    return build_gc_program_t<3, 3>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out);
}

// The program of the grouped kernel's SPARSE build (fx_grouped_s.hip): batches of one structure whose single component is too
// wide for a register-resident factor (49 free variables and more) but whose Cholesky factor is small — the reference's bench
// sketch of 16 hinged triangles has 66 variables and a factor of 291 entries. Everything the kernel does is a walk over tables:
// the row lists and product lists of fx_grouped_c.hip, and the factorisation as a level schedule — a minimum-degree order of the
// columns, the factor's pattern by columns (column k: its diagonal slot, then its rows ascending), the levels of its
// elimination tree (the columns of a level are independent), per level the update triples L(i, j) -= L(i, k) L(j, k) and the
// (slot, column, row) entries of the triangular solves. Words: [0] version [1] variables [2] expressions [3] free variables
// [4] products (padded to 64) [5] right-hand-side entries (padded to 64) [6] factor slots (even) [7] compact Jacobian entries
// (even) [8] levels [9] update triples [10] below-diagonal entries [11 ...] byte offsets of the tables, in the order of the
// `put` calls below [31] words in all.
struct GsHostProgram {
    std::vector<uint32_t> words;
    uint32_t nl = 0, ng = 0, nvt = 0, net = 0, nfree = 0;
};
static bool build_gs_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                             uint32_t nvt, uint32_t net, GsHostProgram& out) {
    out = GsHostProgram();
    if (nvt == 0 || nvt > 255u || net == 0 || net > 255u) return false;  // (a byte per variable / row / column id in the tables)
    std::vector<int16_t> vcol(nvt, -1);
    std::vector<uint16_t> fidx;
    for (uint32_t i = 0; i < nvt; ++i) {
        if ((var_info[i] & fx::VAR_COMP_MASK) != 0) return false;
        if (!(var_info[i] & fx::VAR_FIXED_BIT)) {
            vcol[i] = (int16_t)fidx.size();
            fidx.push_back((uint16_t)i);
        }
    }
    const uint32_t n = (uint32_t)fidx.size();
    if (n <= 32u || n > 255u) return false;
    std::vector<uint8_t> rtag(net), gvar((size_t)net * 8, 0);
    std::vector<uint16_t> gbase(net);
    std::vector<int> gcol((size_t)net * 8, -1);
    uint32_t ng = 0;
    std::vector<uint8_t> adj((size_t)n * n, 0);  // pattern of Jt J
    for (uint32_t r = 0; r < net; ++r) {
        if (expr_comp[r] != 0) return false;
        const int tag = (int)(expr_tag[r] & 0x7F);
        if (tag >= FX_TAG_POSE_X) return false;
        uint32_t vars8[8];
        const int k = fx::expand_vars(tag, expr_idx16 + 4 * (size_t)r, vars8);
        rtag[r] = (uint8_t)tag;
        gbase[r] = (uint16_t)ng;
        for (int e = 0; e < 8; ++e) {
            if (vars8[e] >= nvt) return false;
            gvar[(size_t)r * 8 + e] = (uint8_t)vars8[e];
            gcol[(size_t)r * 8 + e] = e < k ? (int)vcol[vars8[e]] : -1;
        }
        // (a distance row keeps TWO of its four entries: the other two are their exact negatives — expressions.rs:291-317 —,
        // and the lists below carry the sign)
        ng += tag == FX_TAG_PPD ? 2u : (uint32_t)k;
        for (int a = 0; a < k; ++a)
            for (int bb = 0; bb < k; ++bb)
                if (gcol[(size_t)r * 8 + a] >= 0 && gcol[(size_t)r * 8 + bb] >= 0)
                    adj[(size_t)gcol[(size_t)r * 8 + a] * n + (size_t)gcol[(size_t)r * 8 + bb]] = 1;
    }
    if (ng > 1023u) return false;
    // ---- minimum-degree order on the graph of Jt J (ties: the lower column), eliminating on a copy
    std::vector<uint32_t> order, pos(n, 0);
    {
        std::vector<uint8_t> g = adj;
        std::vector<uint8_t> gone(n, 0);
        std::vector<uint32_t> deg(n, 0), nbr;
        auto degree = [&](uint32_t c) {
            uint32_t dg = 0;
            for (uint32_t e = 0; e < n; ++e) dg += (!gone[e] && e != c && g[(size_t)c * n + e]) ? 1u : 0u;
            return dg;
        };
        for (uint32_t c = 0; c < n; ++c) deg[c] = degree(c);
        for (uint32_t step = 0; step < n; ++step) {
            uint32_t best = n, bdeg = 0xFFFFFFFFu;
            for (uint32_t c = 0; c < n; ++c)
                if (!gone[c] && deg[c] < bdeg) {
                    bdeg = deg[c];
                    best = c;
                }
            gone[best] = 1;
            pos[best] = step;
            order.push_back(best);
            nbr.clear();
            for (uint32_t a = 0; a < n; ++a)
                if (!gone[a] && g[(size_t)best * n + a]) nbr.push_back(a);
            for (uint32_t a : nbr)
                for (uint32_t bb : nbr) g[(size_t)a * n + bb] = 1;
            for (uint32_t a : nbr) deg[a] = degree(a);  // (only the eliminated column's neighbours change their degree)
        }
    }
    // ---- the factor's pattern in elimination order: lp[i][k] (positions), with fill
    std::vector<uint8_t> lp((size_t)n * n, 0);
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t bb = 0; bb < n; ++bb)
            if (a == bb || adj[(size_t)a * n + bb]) {
                const uint32_t pi = std::max(pos[a], pos[bb]), pk = std::min(pos[a], pos[bb]);
                lp[(size_t)pi * n + pk] = 1;
            }
    for (uint32_t k = 0; k < n; ++k)
        for (uint32_t i = k + 1; i < n; ++i)
            if (lp[(size_t)i * n + k])
                for (uint32_t j = k + 1; j <= i; ++j)
                    if (lp[(size_t)j * n + k]) lp[(size_t)i * n + j] = 1;
    // slots: column position k holds its diagonal, then its rows (positions ascending)
    std::vector<uint16_t> cbase(n + 1, 0);
    std::vector<int32_t> slot((size_t)n * n, -1);
    std::vector<uint8_t> rowof;  // free COLUMN id of a slot's row
    uint32_t nl = 0;
    for (uint32_t k = 0; k < n; ++k) {
        cbase[k] = (uint16_t)nl;
        for (uint32_t i = k; i < n; ++i)
            if (lp[(size_t)i * n + k]) {
                slot[(size_t)i * n + k] = (int32_t)nl++;
                rowof.push_back((uint8_t)order[i]);
            }
        if (nl > 1023u) return false;
    }
    cbase[n] = (uint16_t)nl;
    // levels of the elimination tree: a column waits for every column that updates it
    std::vector<uint32_t> level(n, 0);
    uint32_t nlev = 0;
    for (uint32_t k = 0; k < n; ++k) {
        for (uint32_t j = 0; j < k; ++j)
            if (lp[(size_t)k * n + j]) level[k] = std::max(level[k], level[j] + 1u);
        nlev = std::max(nlev, level[k] + 1u);
    }
    std::vector<uint8_t> lcol;      // column POSITIONS in level order
    std::vector<uint16_t> lptr(1, 0);
    std::vector<uint32_t> uptr(1, 0), eptr(1, 0), upd, ent;
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        for (uint32_t k = 0; k < n; ++k) {
            if (level[k] != lv) continue;
            lcol.push_back((uint8_t)k);
            for (uint32_t i = k + 1; i < n; ++i) {
                if (!lp[(size_t)i * n + k]) continue;
                // (slot | column id of k << 10 | column id of the row << 18): the triangular solves' entries
                ent.push_back((uint32_t)slot[(size_t)i * n + k] | (order[k] << 10) | (order[i] << 18));
                for (uint32_t j = k + 1; j <= i; ++j)
                    if (lp[(size_t)j * n + k])
                        upd.push_back((uint32_t)slot[(size_t)i * n + j] | ((uint32_t)slot[(size_t)i * n + k] << 10) | ((uint32_t)slot[(size_t)j * n + k] << 20));
            }
        }
        lptr.push_back((uint16_t)lcol.size());
        uptr.push_back((uint32_t)upd.size());
        eptr.push_back((uint32_t)ent.size());
    }
    // per column position: its column id; per column id: the slot of its diagonal
    std::vector<uint8_t> colid(n);
    std::vector<uint16_t> dslot(n);
    for (uint32_t k = 0; k < n; ++k) {
        colid[k] = (uint8_t)order[k];
        dslot[order[k]] = cbase[k];
    }
    // products of Jt J into the factor's slots, right-hand side entries (the order of fx_grouped_c.hip's lists)
    std::vector<uint32_t> pw, pe;
    for (uint32_t r = 0; r < net; ++r) {
        const bool ppd = rtag[r] == FX_TAG_PPD;
        auto gent = [&](int e) -> uint32_t { return gbase[r] + (uint32_t)(ppd && e >= 2 ? e - 2 : e); };  // where entry e's value (or its negative) is kept
        auto gneg = [&](int e) -> uint32_t { return ppd && e >= 2 ? 1u : 0u; };
        for (int a = 0; a < 8; ++a) {
            const int ca = gcol[(size_t)r * 8 + a];
            if (ca < 0) continue;
            pe.push_back(gent(a) | (r << 10) | ((uint32_t)ca << 20) | (gneg(a) << 31));
            for (int bb = a; bb < 8; ++bb) {
                const int cb = gcol[(size_t)r * 8 + bb];
                if (cb < 0) continue;
                const uint32_t pi = std::max(pos[(uint32_t)ca], pos[(uint32_t)cb]), pk = std::min(pos[(uint32_t)ca], pos[(uint32_t)cb]);
                const uint32_t w = gent(a) | (gent(bb) << 10) | ((uint32_t)slot[(size_t)pi * n + pk] << 20) | ((gneg(a) ^ gneg(bb)) << 31);
                pw.push_back(w);
                if (a != bb && ca == cb) pw.push_back(w);
            }
        }
    }
    while (pw.size() % 64u) pw.push_back(0xFFFFFFFFu);
    while (pe.size() % 64u) pe.push_back(0xFFFFFFFFu);
    std::vector<uint32_t>& w = out.words;
    w.assign(32, 0);
    auto put = [&](const void* src, size_t bytes) -> uint32_t {
        const uint32_t at = (uint32_t)w.size() * 4u;
        w.resize(w.size() + (bytes + 15u) / 16u * 4u, 0u);
        if (bytes) memcpy(reinterpret_cast<unsigned char*>(w.data()) + at, src, bytes);
        return at;
    };
    w[11] = put(vcol.data(), vcol.size() * 2);
    w[12] = put(fidx.data(), fidx.size() * 2);
    w[13] = put(rtag.data(), rtag.size());
    w[14] = put(gbase.data(), gbase.size() * 2);
    w[15] = put(gvar.data(), gvar.size());
    w[16] = put(dslot.data(), dslot.size() * 2);
    w[17] = put(cbase.data(), cbase.size() * 2);
    w[18] = put(rowof.data(), rowof.size());
    w[19] = put(lcol.data(), lcol.size());
    w[20] = put(lptr.data(), lptr.size() * 2);
    w[21] = put(uptr.data(), uptr.size() * 4);
    w[22] = put(eptr.data(), eptr.size() * 4);
    w[23] = put(upd.data(), upd.size() * 4);
    w[24] = put(ent.data(), ent.size() * 4);
    w[25] = put(pw.data(), pw.size() * 4);
    w[26] = put(pe.data(), pe.size() * 4);
    w[27] = put(colid.data(), colid.size());
    w[0] = 1u;
    w[1] = nvt;
    w[2] = net;
    w[3] = n;
    w[4] = (uint32_t)pw.size();
    w[5] = (uint32_t)pe.size();
    w[6] = (nl + 1u) & ~1u;
    w[7] = (ng + 1u) & ~1u;
    w[8] = nlev;
    w[9] = (uint32_t)upd.size();
    w[10] = (uint32_t)ent.size();
    w[31] = (uint32_t)w.size();
    out.nl = (nl + 1u) & ~1u;
    out.ng = (ng + 1u) & ~1u;
    out.nvt = nvt;
    out.net = net;
    out.nfree = n;
    return true;
}

// The same analysis compiled into a table-driven program for the grouped FX_STEP_QR build (fx_grouped.hip: four Systems per
// wavefront, one per row of 16 lanes; batches of ONE structure, so one program serves every System). The permuted augmented
// matrix [J | -r; sqrt(lambda) I | 0] is stored by its symbolic patterns — per column position j the rows of R(:, j) above the
// diagonal and of the Householder vector H(:, j) from it down — plus the dense right-hand side and one slot of zero for the
// padding. Every access of the factorisation is then an offset from a table: per Householder step k its active columns
// (those with k in R's pattern, and the right-hand side), one lane each, and per (active column, vector entry) one word
// holding both offsets of the multiply-add. The arithmetic and its order are the one-wavefront QR kernel's (fx_kernels.hip).
// Words: [0] n [1] m [2] nx (doubles per System, even) [3] zero slot [4] scat (u16 [m][8], 0xFFFF = dropped) [5] rhs_off (u16 [m])
// [6] damp_off (u16 [n]) [7] cpos (u16 [n]: free column -> position) [8] steps ([n][3]: diag | len << 16, entries' first word,
// active columns) [9] bptr (u16 [n + 1]) [10] bent (row << 16 | offset of R(row, i)) [11] words in all [12] first right-hand
// side entry [13] longest vector (entries below the diagonal, padded to fours).
struct QrgHostProgram {
    std::vector<uint32_t> words;
    uint32_t n = 0, m = 0, nx = 0, ng = 0;
    bool ok = false;
};
// `wide`: the program of the one-wavefront QR build of the wide kernel (fx_wide.hip: components of up to 128 columns and
// 256 rows, Householder vectors of any length) — the same tables with offsets in ELEMENTS (the matrix may pass 64 KB).
bool build_qrg_program(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_,
                       uint32_t n, uint32_t nvt, QrgHostProgram& out, bool wide = false) {
    out = QrgHostProgram();
    out.n = n;
    out.m = m;
    if (n == 0 || n > (wide ? 128u : 32u) || m == 0 || m > (wide ? 256u : 64u)) return false;
    std::vector<int32_t> colof(nvt, -1);
    for (uint32_t c = 0; c < n; ++c) colof[free_[c]] = (int32_t)c;
    std::vector<std::vector<int>> cols(n);
    std::vector<int32_t> gcol((size_t)m * 8, -1);
    for (uint32_t r = 0; r < m; ++r) {
        uint32_t vars8[8];
        const int k = fx::expand_vars<true>((int)(expr_tag[rows[r]] & 0x7F), expr_idx16 + 4 * (size_t)rows[r], vars8);
        for (int q = 0; q < k; ++q) {
            const int32_t c = vars8[q] < nvt ? colof[vars8[q]] : -1;
            gcol[(size_t)r * 8 + q] = c;
            if (c >= 0 && (cols[c].empty() || cols[c].back() != (int)r)) cols[c].push_back((int)r);
        }
    }
    fx::qr::Csc a;
    a.nrows = (int)(m + n);
    a.ncols = (int)n;
    a.ptr.assign(1, 0);
    for (uint32_t c = 0; c < n; ++c) {
        a.idx.insert(a.idx.end(), cols[c].begin(), cols[c].end());
        a.idx.push_back((int)(m + c));
        a.ptr.push_back((int)a.idx.size());
    }
    fx::qr::Symbolic sy;
    if (!fx::qr::analyze(a, true, sy)) return false;
    const uint32_t Mq = m + n;
    std::vector<uint32_t> cpos(n, 0);
    for (uint32_t j = 0; j < n; ++j) cpos[(uint32_t)sy.col_perm[j]] = j;
    // storage: column position j holds the rows of R(:, j) above the diagonal, then those of H(:, j) (j first)
    std::vector<std::vector<int>> prow(n);
    std::vector<uint32_t> cbase(n + 1, 0);
    for (uint32_t j = 0; j < n; ++j) {
        for (int p = sy.rptr[j]; p < sy.rptr[j + 1] - 1; ++p) prow[j].push_back(sy.rrows[p]);
        for (int p = sy.hptr[j]; p < sy.hptr[j + 1]; ++p) prow[j].push_back(sy.hrows[p]);
        if (!std::is_sorted(prow[j].begin(), prow[j].end()) || std::adjacent_find(prow[j].begin(), prow[j].end()) != prow[j].end()) return false;
        cbase[j + 1] = cbase[j] + (uint32_t)prow[j].size();
    }
    const uint32_t rhsbase = cbase[n];
    uint32_t nx = rhsbase + Mq + 1u;
    const uint32_t zero = nx - 1u;
    nx = (nx + 1u) & ~1u;
    const uint32_t osc = wide ? 1u : 8u;  // offsets in elements / in bytes
    if (osc * nx > 0xFFF0u) return false;
    bool bad = false;
    auto at = [&](int r, uint32_t j) -> uint32_t {  // offset of entry (permuted row r, column position j; j == n: right-hand side)
        if (j == n) return rhsbase + (uint32_t)r;
        auto it = std::lower_bound(prow[j].begin(), prow[j].end(), r);
        if (it == prow[j].end() || *it != r) {
            bad = true;
            return zero;
        }
        return cbase[j] + (uint32_t)(it - prow[j].begin());
    };
    auto atb = [&](int r, uint32_t j) -> uint32_t { return osc * at(r, j); };  // ... as the kernel takes them
    std::vector<uint16_t> scat((size_t)m * 8, 0xFFFFu), rhs_off(m), damp(n), cpos16(n), bptr(n + 1, 0);
    for (uint32_t r = 0; r < m; ++r) {
        for (int q = 0; q < 8; ++q)
            if (gcol[(size_t)r * 8 + q] >= 0) scat[(size_t)r * 8 + q] = (uint16_t)atb(sy.row_perm[r], cpos[(uint32_t)gcol[(size_t)r * 8 + q]]);
        rhs_off[r] = (uint16_t)atb(sy.row_perm[r], n);
    }
    for (uint32_t c = 0; c < n; ++c) {
        damp[c] = (uint16_t)atb(sy.row_perm[m + c], cpos[c]);
        cpos16[c] = (uint16_t)cpos[c];
    }
    std::vector<uint32_t> steps(3 * (size_t)n, 0), ent, bent;
    uint32_t max_len = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const int hb = sy.hptr[k], he = sy.hptr[k + 1];
        if (he <= hb || sy.hrows[hb] != (int)k) return false;
        const uint32_t below = (uint32_t)(he - hb - 1), len = (below + 3u) & ~3u;
        max_len = std::max(max_len, len);
        std::vector<uint32_t> active;  // column positions the vector is applied to, ascending, then the right-hand side
        for (uint32_t j = k + 1; j < n; ++j)
            if (std::binary_search(sy.rrows.begin() + sy.rptr[j], sy.rrows.begin() + sy.rptr[j + 1] - 1, (int)k)) active.push_back(j);
        active.push_back(n);
        const uint32_t na = (uint32_t)active.size();
        steps[3 * k] = atb((int)k, k) | (len << 16);
        steps[3 * k + 1] = (uint32_t)ent.size();
        steps[3 * k + 2] = na;
        const size_t e0 = ent.size();
        ent.resize(e0 + (size_t)(len + 1u) * na, (osc * zero) | ((osc * zero) << 16));
        for (uint32_t i = 0; i < na; ++i) {
            ent[e0 + i] = atb((int)k, active[i]);
            for (uint32_t u = 0; u < below; ++u) {
                const int r = sy.hrows[hb + 1 + (int)u];
                ent[e0 + (size_t)(1u + u) * na + i] = atb(r, active[i]) | (atb(r, k) << 16);
            }
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        bptr[i] = (uint16_t)bent.size();
        for (int p = sy.rptr[i]; p < sy.rptr[i + 1] - 1; ++p) bent.push_back(((osc * (uint32_t)sy.rrows[p]) << 16) | atb(sy.rrows[p], i));
    }
    bptr[n] = (uint16_t)bent.size();
    if (bad || (!wide && max_len > 32u) || max_len > 0xFFFFu) return false;
    std::vector<uint32_t>& w = out.words;
    w.assign(16, 0);
    auto put16 = [&](const std::vector<uint16_t>& v) -> uint32_t {
        const uint32_t o = (uint32_t)w.size();
        w.resize(o + (v.size() + 1) / 2, 0);
        memcpy(w.data() + o, v.data(), v.size() * 2);
        return o;
    };
    auto put32 = [&](const std::vector<uint32_t>& v) -> uint32_t {
        const uint32_t o = (uint32_t)w.size();
        w.insert(w.end(), v.begin(), v.end());
        return o;
    };
    w[0] = n; w[1] = m; w[2] = nx; w[3] = zero;
    w[4] = put16(scat); w[5] = put16(rhs_off); w[6] = put16(damp); w[7] = put16(cpos16);
    const uint32_t o_ent_rel = 0;
    (void)o_ent_rel;
    w[8] = put32(steps);
    w[9] = put16(bptr);
    w[10] = put32(bent);
    {  // the Jacobian rows are kept compact: row r's entries start at gbase[r], one per variable of its expression kind
        std::vector<uint16_t> gbase(m);
        uint32_t ng = 0;
        for (uint32_t r = 0; r < m; ++r) {
            gbase[r] = (uint16_t)ng;
            ng += (uint32_t)fx::tag_nvars<true>((int)(expr_tag[rows[r]] & 0x7F));
        }
        w[15] = put16(gbase);
        out.ng = (ng + 3u) & ~3u;
    }
    w.resize((w.size() + 3u) & ~size_t(3), 0);
    w[14] = (uint32_t)w.size();  // the small tables end here (the kernel keeps them in LDS); the per-entry words stay in global memory
    const uint32_t o_ent = put32(ent);
    for (uint32_t k = 0; k < n; ++k) w[w[8] + 3 * k + 1] += o_ent;  // entries' first word, from the start of the program
    w.resize((w.size() + 3u) & ~size_t(3), 0);
    w[11] = (uint32_t)w.size();
    w[12] = osc * rhsbase;
    w[13] = max_len;
    out.nx = nx;
    out.ok = true;
    return true;
}

// Builds (once per resident batch and decomposer) the QR plans of every System the one-wavefront kernel takes.
// Systems of one structure share a plan: a batch of one sketch with many parameter sets is analysed once.
int ensure_qr_plans(fx_ctx* ctx, fx_dbatch* db, bool units) {
    fx::DeviceBatch& d = db->d;
    fx::QrPlans& Q = units ? d.qr_units : d.qr_none;
    if (Q.desc) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1), expr_off((size_t)n + 1), sys_class;
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs), expr_comp(d.n_exprs), sys_ncomp(n);
    std::vector<uint8_t> expr_tag(d.n_exprs), sys_large(n);
    std::vector<uint32_t> sys_unit_off, unit_rows;
    std::vector<fx::UnitDesc> unit_desc;
    std::vector<uint16_t> unit_vars;
    if (d.sys_class && !units) sys_class.resize(n);
    if (db->packed_base && (ctx->pinned || db->zero_copy) && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        // a small batch is one block on the device: its image in one page-locked copy instead of nine small ones
        // (a zero-copy block is host memory already)
        unsigned char* img = db->zero_copy ? db->zc_image : ctx->pinned + fx_ctx::PINNED_HALF;
        if (!db->zero_copy) FX_HIP(hipMemcpyAsync(img, db->packed_base, db->packed_bytes, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stream_synced();
        auto grab = [&](void* dst, const void* dev, size_t bytes) {
            if (bytes) memcpy(dst, img + (reinterpret_cast<const unsigned char*>(dev) - db->packed_base), bytes);
        };
        grab(var_off.data(), d.var_off, var_off.size() * 4);
        grab(expr_off.data(), d.expr_off, expr_off.size() * 4);
        grab(var_info.data(), d.var_info, var_info.size() * 2);
        grab(expr_idx.data(), d.expr_idx, expr_idx.size() * 2);
        grab(expr_tag.data(), d.expr_tag, expr_tag.size());
        grab(expr_comp.data(), d.expr_comp, expr_comp.size() * 2);
        grab(sys_large.data(), d.sys_large, n);
        grab(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2);
        if (!sys_class.empty()) grab(sys_class.data(), d.sys_class, (size_t)n * 4);
    } else {
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_comp.data(), d.expr_comp, expr_comp.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n) {
        FX_HIP(hipMemcpyAsync(sys_large.data(), d.sys_large, n, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
        if (!sys_class.empty()) FX_HIP(hipMemcpyAsync(sys_class.data(), d.sys_class, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    }
    if (units) {
        if (!d.sys_unit_off) return fail(FX_ERR_INVALID, "internal: SinglePass blocks are not built yet");
        sys_unit_off.resize((size_t)n + 1);
        unit_desc.resize(db->n_units);
        unit_rows.resize(db->n_unit_rows);
        unit_vars.resize(db->n_unit_vars);
        FX_HIP(hipMemcpyAsync(sys_unit_off.data(), d.sys_unit_off, sys_unit_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_units) FX_HIP(hipMemcpyAsync(unit_desc.data(), d.unit_desc, unit_desc.size() * sizeof(fx::UnitDesc), hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_unit_rows) FX_HIP(hipMemcpyAsync(unit_rows.data(), d.unit_rows, unit_rows.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_unit_vars) FX_HIP(hipMemcpyAsync(unit_vars.data(), d.unit_vars, unit_vars.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));

    // which System's analysis a System uses (itself, System 0 of a uniform batch, or its structure class)
    auto owner = [&](uint32_t s) -> uint32_t {
        if (units) return s;  // blocks are indexed per block
        if (d.uniform) return 0u;
        if (!sys_class.empty()) return sys_class[s];
        return s;
    };
    struct Range {
        std::vector<fx::QrDesc> desc;
        std::vector<uint16_t> u16;
        std::vector<uint64_t> u64;
        std::vector<uint32_t> sys_first;  // per System of the range: its first desc (relative), or UINT32_MAX when it has none
        uint32_t s_lo = 0, max_m = 0, max_h = 0;
        bool failed = false;
    } ranges[MAX_RANGES];
    uint32_t n_ranges = 1;
    parallel_ranges(n, (uint64_t)d.n_exprs * 64, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Range& R = ranges[t];
        R.s_lo = s_lo;
        R.sys_first.assign(s_hi - s_lo, 0xFFFFFFFFu);
        std::vector<uint32_t> rows, free_;
        QrHostPlan hp;
        auto emit = [&](const uint32_t* rw, uint32_t m, const uint32_t* fr, uint32_t nf, uint32_t s, bool skip) {
            const uint32_t e0 = expr_off[s], nvt = var_off[s + 1] - var_off[s];
            fx::QrDesc qd{};
            qd.u16_off = (uint32_t)R.u16.size();
            qd.u64_off = (uint32_t)R.u64.size();
            qd.n = (uint16_t)nf;
            qd.m = (uint16_t)m;
            if (!skip && build_qr_plan(expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0, rw, m, fr, nf, nvt, hp)) {
                qd.nnzh = (uint16_t)hp.nnzh;
                qd.ok = 1;
                R.u16.insert(R.u16.end(), hp.u16.begin(), hp.u16.end());
                R.u64.insert(R.u64.end(), hp.u64.begin(), hp.u64.end());
                R.max_m = std::max(R.max_m, m + nf);
                R.max_h = std::max(R.max_h, hp.nnzh);
            } else if (!skip) {
                R.failed = true;
            }
            R.desc.push_back(qd);
        };
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            if (sys_large[s] || owner(s) != s) continue;
            R.sys_first[s - s_lo] = (uint32_t)R.desc.size();
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0, e0 = expr_off[s], net = expr_off[s + 1] - e0;
            if (units) {
                for (uint32_t u = sys_unit_off[s]; u < sys_unit_off[s + 1]; ++u) {
                    const fx::UnitDesc& ud = unit_desc[u];
                    rows.assign(unit_rows.begin() + ud.row_off, unit_rows.begin() + ud.row_off + ud.nrows);
                    free_.clear();
                    for (uint32_t k = 0; k < ud.nvars; ++k) free_.push_back(unit_vars[ud.var_off + k]);
                    emit(rows.data(), ud.nrows, free_.data(), ud.nvars, s, (ud.flags & fx::UNIT_EMPTY) != 0);
                }
            } else {
                for (uint32_t c = 0; c < sys_ncomp[s]; ++c) {
                    rows.clear();
                    free_.clear();
                    bool any_var = false;  // a component without variables is skipped by the kernel (and by the reference)
                    for (uint32_t i = 0; i < nvt; ++i) {
                        const uint16_t info = var_info[v0 + i];
                        any_var = any_var || (info & fx::VAR_COMP_MASK) == c;
                        if ((info & fx::VAR_COMP_MASK) == c && !(info & fx::VAR_FIXED_BIT)) free_.push_back(i);
                    }
                    for (uint32_t i = 0; i < net; ++i)
                        if (expr_comp[e0 + i] == c) rows.push_back(i);
                    emit(rows.data(), (uint32_t)rows.size(), free_.data(), (uint32_t)free_.size(), s, !any_var);
                }
            }
        }
    }, &n_ranges);
    std::vector<fx::QrDesc> desc;
    std::vector<uint16_t> u16;
    std::vector<uint64_t> u64;
    std::vector<uint32_t> sys_first(n, 0);
    uint32_t max_m = 0, max_h = 0;
    for (uint32_t t = 0; t < n_ranges; ++t) {
        Range& R = ranges[t];
        if (R.failed) return fail(FX_ERR_UNSUPPORTED, "FX_STEP_QR: the symbolic analysis of a component failed (more than 64 columns or a malformed pattern)");
        const uint32_t dbase = (uint32_t)desc.size(), b16 = (uint32_t)u16.size(), b64 = (uint32_t)u64.size();
        for (fx::QrDesc qd : R.desc) {
            qd.u16_off += b16;
            qd.u64_off += b64;
            desc.push_back(qd);
        }
        u16.insert(u16.end(), R.u16.begin(), R.u16.end());
        u64.insert(u64.end(), R.u64.begin(), R.u64.end());
        for (size_t k = 0; k < R.sys_first.size(); ++k)
            if (R.sys_first[k] != 0xFFFFFFFFu) sys_first[R.s_lo + k] = dbase + R.sys_first[k];
        max_m = std::max(max_m, R.max_m);
        max_h = std::max(max_h, R.max_h);
    }
    std::vector<uint32_t> index;
    if (units) {  // one desc per block, in block order
        index.assign(db->n_units, 0);
        for (uint32_t s = 0; s < n; ++s) {
            if (sys_large[s]) continue;
            for (uint32_t u = sys_unit_off[s]; u < sys_unit_off[s + 1]; ++u) index[u] = sys_first[s] + (u - sys_unit_off[s]);
        }
    } else {
        index.assign(n, 0);
        for (uint32_t s = 0; s < n; ++s) index[s] = sys_first[owner(s)];
    }
    fx::QrPlans q;
    {  // the kernel keeps a component's augmented matrix in LDS: does the largest one fit? Asked before anything is uploaded
       // (a batch that fails here fails again on every later FX_STEP_QR solve — nothing may pile up on the device)
        Q.max_m = max_m;
        Q.max_h = max_h;
        const size_t need = fx::solve_lds_bytes_qr(d, units);
        if (need > 160u * 1024u) {
            Q.max_m = Q.max_h = 0;
            return fail(FX_ERR_TOO_LARGE, "FX_STEP_QR keeps the (expressions + free variables) x (free variables + 1) matrix of a component in LDS: %zu bytes needed (limit 163840)",
                        need);
        }
    }
    // a batch of ONE structure with a single component of at most 32 columns: the program of the grouped build as well
    QrgHostProgram prog;
    if (!units && d.uniform && d.u_ncomp == 1u && n >= 1 && !sys_large[0] && sys_ncomp[0] == 1u) {
        const uint32_t nvt = var_off[1] - var_off[0], net = expr_off[1] - expr_off[0];
        std::vector<uint32_t> rows, free_;
        for (uint32_t i = 0; i < nvt; ++i)
            if ((var_info[i] & fx::VAR_COMP_MASK) == 0 && !(var_info[i] & fx::VAR_FIXED_BIT)) free_.push_back(i);
        for (uint32_t i = 0; i < net; ++i)
            if (expr_comp[i] == 0) rows.push_back(i);
        (void)build_qrg_program(expr_tag.data(), expr_idx.data(), rows.data(), (uint32_t)rows.size(), free_.data(), (uint32_t)free_.size(), nvt, prog);
    }
    // Systems beyond one wavefront whose components all fit the wide kernel (at most 128 columns, 256 rows, 512 variables):
    // a program per component for its FX_STEP_QR build (fx_wide.hip). Systems of one structure share theirs. All or nothing
    // for the Systems the analysis marked for the wide kernel (they have no other reference-numerics home), and within a
    // budget of program words — beyond it, as for everything larger, FX_STEP_QR stays FX_STEP_CHOLESKY_REFINED.
    std::vector<uint32_t> w_list, w_comp_off(1, 0), w_prog_off, w_words;
    uint32_t w_nx = 0, w_free = 0, w_vars = 0, w_rows = 0;
    db->h_qr_wide.assign(n, 0);
    if (!units) {
        bool all_marked = true;
        std::vector<std::pair<uint32_t, uint32_t>> first_of;  // owner System -> first entry of w_prog_off
        constexpr size_t WORD_BUDGET = size_t(48) << 20;       // 192 MB of tables per batch
        for (uint32_t s = 0; s < n && w_words.size() <= WORD_BUDGET; ++s) {
            if (!sys_large[s]) continue;
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0, e0 = expr_off[s], net = expr_off[s + 1] - e0;
            bool fits = nvt <= 512u && sys_ncomp[s] >= 1u;
            const uint32_t own = owner(s);
            uint32_t first = 0xFFFFFFFFu;
            if (fits && own != s)
                for (auto& kv : first_of)
                    if (kv.first == own) first = kv.second;
            std::vector<uint32_t> offs;
            if (fits && first == 0xFFFFFFFFu) {
                std::vector<uint32_t> rows, free_;
                for (uint32_t c = 0; c < sys_ncomp[s] && fits; ++c) {
                    rows.clear();
                    free_.clear();
                    bool any_var = false;
                    for (uint32_t i = 0; i < nvt; ++i) {
                        const uint16_t info = var_info[v0 + i];
                        any_var = any_var || (info & fx::VAR_COMP_MASK) == c;
                        if ((info & fx::VAR_COMP_MASK) == c && !(info & fx::VAR_FIXED_BIT)) free_.push_back(i);
                    }
                    for (uint32_t i = 0; i < net; ++i)
                        if (expr_comp[e0 + i] == c) rows.push_back(i);
                    if (!any_var) {  // skipped by the kernel, as by the reference
                        offs.push_back(0xFFFFFFFFu);
                        continue;
                    }
                    QrgHostProgram wp;
                    if (free_.empty() || rows.empty() ||
                        !build_qrg_program(expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0, rows.data(), (uint32_t)rows.size(), free_.data(),
                                           (uint32_t)free_.size(), nvt, wp, /*wide=*/true)) {
                        fits = false;
                        break;
                    }
                    offs.push_back((uint32_t)w_words.size());
                    w_words.insert(w_words.end(), wp.words.begin(), wp.words.end());
                    w_nx = std::max(w_nx, wp.nx);
                    w_free = std::max(w_free, wp.n);
                    w_rows = std::max(w_rows, wp.m);
                }
                if (fits) {
                    first = (uint32_t)w_prog_off.size();
                    w_prog_off.insert(w_prog_off.end(), offs.begin(), offs.end());
                    first_of.push_back({s, first});
                }
            }
            if (fits && first != 0xFFFFFFFFu) {
                if (own != s) {  // a System of a structure already planned: its own row of offsets (the same values)
                    const uint32_t nc = sys_ncomp[s];
                    const uint32_t at = (uint32_t)w_prog_off.size();
                    for (uint32_t c = 0; c < nc; ++c) {
                        const uint32_t po = w_prog_off[first + c];
                        w_prog_off.push_back(po);
                    }
                    first = at;
                }
                w_list.push_back(s);
                w_comp_off.push_back(first);
                w_vars = std::max(w_vars, nvt);
                db->h_qr_wide[s] = 1;
            } else if (sys_large[s] == 2) {
                all_marked = false;
            }
        }
        if (!all_marked || w_words.size() > WORD_BUDGET ||
            (!w_list.empty() && fx::wide_qr_lds_bytes(w_free, w_vars, w_rows, w_nx) > 160u * 1024u)) {
            w_list.clear();
            db->h_qr_wide.assign(n, 0);
        }
    }
    unsigned long long* d64 = nullptr;
    int rc = dev_alloc_copy(ctx, db, &q.u16, u16.data(), u16.size());
    if (!rc && !w_list.empty()) {
        // (w_comp_off holds, per listed System, the first entry of ITS components in w_prog_off)
        w_comp_off.erase(w_comp_off.begin());
        rc = dev_alloc_copy(ctx, db, &q.qrw_list, w_list.data(), w_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_comp_first, w_comp_off.data(), w_comp_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_prog_off, w_prog_off.data(), w_prog_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_words, w_words.data(), w_words.size());
        q.n_qrw = (uint32_t)w_list.size();
        q.qrw_nx = w_nx;
        q.qrw_free = w_free;
        q.qrw_vars = w_vars;
        q.qrw_rows = w_rows;
    }
    if (!rc && prog.ok) {
        rc = dev_alloc_copy(ctx, db, &q.qrg, prog.words.data(), prog.words.size());
        q.qrg_words = (uint32_t)prog.words.size();
        q.qrg_small = prog.words[14];
        q.qrg_ng = prog.ng;
        q.qrg_nx = prog.nx;
        q.qrg_n = prog.n;
        q.qrg_m = prog.m;
    }
    if (!rc) rc = dev_alloc_copy(ctx, db, &d64, reinterpret_cast<const unsigned long long*>(u64.data()), u64.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &q.index, index.data(), index.size());
    fx::QrDesc* ddesc = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &ddesc, desc.data(), desc.size());
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    q.u64 = d64;
    q.max_m = max_m;
    q.max_h = max_h;
    q.desc = ddesc;  // set last: marks the plans as complete
    Q = q;
    return FX_OK;
}

// launch_solve, with the Systems of a big batch of small Systems handed out longest-first (fx_presort.hip) unless the
// caller chose a schedule (fx_batch_schedule_by_last_solve) or switched it off (fx_ctx_set_presort)
// A batch of several structures whose big classes have programs (upload_planned): ONE launch of the one-structure build over
// all of them — a wavefront works through the queue of its class, then loads the next class's program and helps there
// (fx_grouped_c.hip) —, and the general build over everyone else. Results are each System's own: the bits of the general build.
// Returns false when the batch or the options do not qualify (nothing launched).
static bool launch_class_solves(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p, int* rc) {
    fx::DeviceBatch& d = db->d;
    *rc = FX_OK;
    if (db->classes.empty() || d.order || !p.grouped_one_structure || !fx::grouped_applies(d, p)) return false;
    fx::DeviceBatch dc = d;
    dc.gc_tab = db->cl_words;
    dc.gc_words = db->cl_max_words;
    dc.gc_words_all = db->cl_max_words_all;
    dc.gc_nslots = db->cl_max_slots;
    dc.gc_ng = db->cl_max_ng;
    dc.gc_nc = db->cl_nc;
    dc.gc_rc = db->cl_rc;
    dc.gc_classes = db->cl_desc;
    dc.gc_nclasses = (uint32_t)db->classes.size();
    dc.order = db->cl_lists;
    dc.n_systems = db->cl_systems;
    dc.work_counter = d.work_counter + 1;
    if (!fx::grouped_c_applies(dc, p)) return false;  // (f32 beyond the 32-column shape, the stamped build, ...: the general build for all)
    // every list longest-first, as a whole batch would be (fx_presort.hip): the scout pass once, a ranking per list
    uint32_t* lists = db->cl_lists;
    if (ctx->presort && d.n_systems >= ctx->presort_min_systems) {
        const uint32_t n = d.n_systems;
        if (!db->ps_keys) {
            db->ps_temp_bytes = fx::presort_temp_bytes(n);
            int r2 = dev_alloc_copy(ctx, db, &db->ps_keys, (const float*)nullptr, 2 * (size_t)n);
            if (!r2) r2 = dev_alloc_copy(ctx, db, &db->ps_ids, (const uint32_t*)nullptr, 2 * (size_t)n);
            if (!r2) r2 = dev_alloc_copy(ctx, db, &db->ps_temp, (const unsigned char*)nullptr, db->ps_temp_bytes);
            if (r2) {
                *rc = r2;
                return true;
            }
        }
        std::vector<uint32_t> offs, counts;
        for (const fx::GcClass& cl : db->classes) {
            offs.push_back(cl.list_off);
            counts.push_back(cl.count);
        }
        offs.push_back(db->rest_off);
        counts.push_back(db->rest_count);
        hipError_t e0 = fx::launch_presort_lists(d, db->ps_keys, db->cl_lists, offs.data(), counts.data(), (uint32_t)offs.size(), db->ps_ids + n, ctx->stream);
        if (e0 != hipSuccess) {
            *rc = fail(FX_ERR_HIP, "presort launch failed: %s", hipGetErrorString(e0));
            return true;
        }
        lists = db->ps_ids + n;
        dc.order = lists;
    }
    hipError_t e = fx::launch_solve_grouped_c(dc, p, ctx->stream);
    if (e == hipSuccess && db->rest_count) {
        fx::DeviceBatch dr = d;
        dr.order = lists + db->rest_off;
        dr.n_systems = db->rest_count;
        dr.work_counter = d.work_counter + 1 + dc.gc_nclasses;
        e = fx::launch_solve_grouped_general(dr, p, ctx->stream);
    }
    if (e != hipSuccess) *rc = fail(FX_ERR_HIP, "class launch failed: %s", hipGetErrorString(e));
    return true;
}

int launch_solve_scheduled(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p) {
    fx::DeviceBatch& d = db->d;
    {
        int rc = FX_OK;
        if (launch_class_solves(ctx, db, p, &rc)) return rc;
    }
    if (ctx->presort && !d.order && !p.prof && d.n_systems >= ctx->presort_min_systems && fx::grouped_applies(d, p)) {
        const uint32_t n = d.n_systems;
        if (!db->ps_keys) {
            db->ps_temp_bytes = fx::presort_temp_bytes(n);
            int rc = dev_alloc_copy(ctx, db, &db->ps_keys, (const float*)nullptr, 2 * (size_t)n);
            if (!rc) rc = dev_alloc_copy(ctx, db, &db->ps_ids, (const uint32_t*)nullptr, 2 * (size_t)n);
            if (!rc) rc = dev_alloc_copy(ctx, db, &db->ps_temp, (const unsigned char*)nullptr, db->ps_temp_bytes);
            if (rc) return rc;
        }
        FX_HIP(fx::launch_presort(d, db->ps_keys, db->ps_ids, db->ps_temp, db->ps_temp_bytes, ctx->stream));
        fx::DeviceBatch dd = d;
        dd.order = db->ps_ids + n;
        FX_HIP(fx::launch_solve(dd, p, ctx->stream));
        return FX_OK;
    }
    FX_HIP(fx::launch_solve(d, p, ctx->stream));
    return FX_OK;
}

// Decomposer::None on a System too large for LDS but made of components that each fit one wavefront (a
// sketch of many separate features): the GLOBAL block walker takes the components as its blocks — rows
// and free variables ascending, snapshot restore after each (quirk Q2) — instead of the host-driven
// sparse path going through them one by one.
int ensure_component_walk(fx_ctx* ctx, fx_dbatch* db) {
    if (db->comp_walk_built) return FX_OK;
    const fx_batch& hb = db->h_batch;
    const uint32_t n = db->d.n_systems;
    std::vector<uint32_t> unit_off((size_t)n + 1, 0), unit_rows, g_list, g_off;
    std::vector<uint16_t> unit_vars;
    std::vector<fx::UnitDesc> desc;
    db->h_comp_walk.assign(n, 0);
    uint32_t g_total = 0, mf = 0, mr = 0, mp = 0, me = 0;
    for (uint32_t s = 0; s < n; ++s) {
        unit_off[s] = (uint32_t)desc.size();
        if (db->h_sys_large.empty() || db->h_sys_large[s] != 1) continue;
        const uint32_t v0 = hb.var_off[s], nvt = hb.var_off[s + 1] - v0;
        const uint32_t e0 = hb.expr_off[s], net = hb.expr_off[s + 1] - e0;
        uint32_t ncomp = 0;
        for (uint32_t i = 0; i < nvt; ++i) {
            uint16_t c = hb.var_comp ? hb.var_comp[v0 + i] : 0;
            if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
        }
        std::vector<std::vector<uint32_t>> rows(ncomp), fvars(ncomp);
        std::vector<uint8_t> has_var(ncomp, 0);
        std::vector<uint32_t> pairs(ncomp, 0), ents(ncomp, 0);
        for (uint32_t i = 0; i < nvt; ++i) {
            uint16_t c = hb.var_comp ? hb.var_comp[v0 + i] : 0;
            if (c == FX_NO_COMPONENT) continue;
            has_var[c] = 1;
            if (!hb.var_fixed[v0 + i]) fvars[c].push_back(i);
        }
        for (uint32_t i = 0; i < net; ++i) {
            uint16_t c = hb.expr_comp ? hb.expr_comp[e0 + i] : 0;
            if (c == FX_NO_COMPONENT || c >= ncomp) continue;
            rows[c].push_back(i);
            uint32_t k = (uint32_t)fx::tag_nvars<true>((int)hb.expr_tag[e0 + i]);
            pairs[c] += k * k;
            ents[c] += k;
        }
        bool fits = nvt <= 0xFFFFu;
        for (uint32_t c = 0; c < ncomp && fits; ++c) fits = fvars[c].size() <= FX_MAX_FREE_VARS && rows[c].size() <= FX_MAX_ROWS;
        if (!fits) continue;
        for (uint32_t c = 0; c < ncomp; ++c) {
            if (!has_var[c]) continue;  // skipped by the reference
            fx::UnitDesc ud{};
            ud.row_off = (uint32_t)unit_rows.size();
            ud.var_off = (uint32_t)unit_vars.size();
            ud.nrows = (uint16_t)rows[c].size();
            ud.nvars = (uint16_t)fvars[c].size();
            ud.comp = (uint16_t)c;
            ud.flags = (uint16_t)(fx::UNIT_FIRST | fx::UNIT_RESTORE);
            unit_rows.insert(unit_rows.end(), rows[c].begin(), rows[c].end());
            for (uint32_t v : fvars[c]) unit_vars.push_back((uint16_t)v);
            desc.push_back(ud);
            mf = std::max<uint32_t>(mf, ud.nvars);
            mr = std::max<uint32_t>(mr, ud.nrows);
            mp = std::max(mp, pairs[c]);
            me = std::max(me, ents[c]);
        }
        db->h_comp_walk[s] = 1;
        g_list.push_back(s);
        g_off.push_back(g_total);
        g_total += nvt;
    }
    unit_off[n] = (uint32_t)desc.size();
    fx::DeviceBatch w = db->d;
    w.n_g = 0;
    if (!g_list.empty()) {
        int rc = dev_alloc_copy(ctx, db, &w.unit_desc, desc.data(), desc.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.unit_rows, unit_rows.data(), unit_rows.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.unit_vars, unit_vars.data(), unit_vars.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.sys_unit_off, unit_off.data(), unit_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_list, g_list.data(), g_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_off, g_off.data(), g_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_xs, (const double*)nullptr, 2 * (size_t)g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_vout, (const double*)nullptr, g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_colof, (const int16_t*)nullptr, g_total);
        if (rc) return rc;
        FX_HIP(hipStreamSynchronize(ctx->stream));
        w.n_g = (uint32_t)g_list.size();
        w.max_unit_free_g = mf;
        w.max_unit_rows_g = mr;
        w.max_pairs_g = mp;
        w.max_ents_g = me;
    }
    db->comp_walk = w;
    db->comp_walk_built = true;
    return FX_OK;
}

// Systems beyond the one-wavefront limits: host-driven LM with device numerics (fx_sparse.hip).
// the wide kernel covers Levenberg-Marquardt without a decomposer. Every path for Systems beyond the one-wavefront
// limits computes in f64 (the sparse path always did): a request for f32 compute keeps its options (ftol, max_outer)
// and gets the f64 device kernels here rather than the host-driven loop.
bool wide_kernel_applies(const fx::LmParams& p) {
    return !(p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS));
}

int solve_large_systems(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p) {
    if (!db->n_large) return FX_OK;
    // cluster problems with pose rows: only the one-wavefront pose build and the sparse path evaluate those
    const bool pose = db->d.has_pose != 0;
    if (!pose && wide_kernel_applies(p) && db->d.n_wide && !db->qr_wide_active) {  // (FX_STEP_QR: the QR build has solved them)
        hipError_t e = fx::launch_solve_wide(db->d, p, ctx->stream);
        if (e != hipSuccess) return fail(FX_ERR_HIP, "wide kernel launch failed: %s", hipGetErrorString(e));
    }
    const bool device_units = (p.mode & fx::MODE_UNITS) && !(p.mode & fx::MODE_LBFGS);
    // Decomposer::None, f64 LM: large Systems made of small components are walked on the device
    const bool comp_walk = !pose && wide_kernel_applies(p);
    if (comp_walk) {
        int rc = ensure_component_walk(ctx, db);
        if (rc) return rc;
        if (db->comp_walk.n_g) {
            fx::LmParams pw = p;
            pw.mode |= fx::MODE_UNITS;  // the walker's block loop; the RESTORE flag keeps None semantics
            hipError_t e = fx::launch_solve_walk(db->comp_walk, pw, ctx->stream);
            if (e != hipSuccess) return fail(FX_ERR_HIP, "component walk launch failed: %s", hipGetErrorString(e));
        }
    }
    // ---- Systems of one STRUCTURE (fixed flags, tags, fields, components) share a plan and are solved together
    const fx_batch& hb = db->h_batch;
    const bool wide_done = !pose && wide_kernel_applies(p);
    const uint32_t groups_of = (p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS)) | (wide_done ? 0x100u : 0u) | (device_units ? 0x200u : 0u) |
                               (comp_walk ? 0x400u : 0u) | (pose ? 0x800u : 0u) | (db->qr_wide_active ? 0x1000u : 0u);
    std::vector<fx_dbatch::StructureGroup> local_groups;
    const bool cached = db->resident && db->large_groups.count(groups_of) != 0;
    std::vector<fx_dbatch::StructureGroup>& groups = db->resident ? db->large_groups[groups_of] : local_groups;
    if (!cached) {
        std::vector<uint32_t> todo;
        for (uint32_t s = 0; s < db->d.n_systems; ++s) {
            if (!db->h_sys_large[s]) continue;
            if (db->h_sys_large[s] == 2 && wide_done) continue;                                            // done by the wide kernel
            if (db->qr_wide_active && s < db->h_qr_wide.size() && db->h_qr_wide[s]) continue;             // done by its QR build
            if (device_units && s < db->h_units_on_device.size() && db->h_units_on_device[s]) continue;  // done by the kernel
            if (comp_walk && s < db->h_comp_walk.size() && db->h_comp_walk[s]) continue;                  // done by the walker
            todo.push_back(s);
        }
        auto slices = [&](uint32_t s, const void* ptr[5], size_t len[5]) {
            const uint32_t v0 = hb.var_off[s], nvt = hb.var_off[s + 1] - v0, e0 = hb.expr_off[s], net = hb.expr_off[s + 1] - e0;
            ptr[0] = hb.var_fixed + v0;                  len[0] = nvt;
            ptr[1] = hb.expr_tag + e0;                   len[1] = net;
            ptr[2] = hb.expr_idx + 4 * (size_t)e0;       len[2] = 4 * (size_t)net * sizeof(uint32_t);
            ptr[3] = hb.var_comp ? hb.var_comp + v0 : nullptr;   len[3] = hb.var_comp ? nvt * sizeof(uint16_t) : 0;
            ptr[4] = hb.expr_comp ? hb.expr_comp + e0 : nullptr; len[4] = hb.expr_comp ? net * sizeof(uint16_t) : 0;
        };
        auto structure_key = [&](uint32_t s) {
            const void* ptr[5];
            size_t len[5];
            slices(s, ptr, len);
            std::vector<unsigned char> key;
            const uint32_t head[4] = {p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS), 0u, hb.var_off[s + 1] - hb.var_off[s],
                                      hb.expr_off[s + 1] - hb.expr_off[s]};
            key.reserve(sizeof(head) + len[0] + len[1] + len[2] + len[3] + len[4]);
            key.insert(key.end(), reinterpret_cast<const unsigned char*>(head), reinterpret_cast<const unsigned char*>(head) + sizeof(head));
            for (int k = 0; k < 5; ++k)
                if (len[k]) key.insert(key.end(), static_cast<const unsigned char*>(ptr[k]), static_cast<const unsigned char*>(ptr[k]) + len[k]);
            return key;
        };
        auto same_structure = [&](uint32_t x, uint32_t y) {  // the raw arrays of two Systems, slice by slice
            const void *px[5], *py[5];
            size_t lx[5], ly[5];
            slices(x, px, lx);
            slices(y, py, ly);
            for (int k = 0; k < 5; ++k)
                if (lx[k] != ly[k] || (lx[k] && memcmp(px[k], py[k], lx[k]) != 0)) return false;
            return true;
        };
        auto hash_of = [](const std::vector<unsigned char>& key) {  // eight bytes at a time
            uint64_t h = 1469598103934665603ull;
            size_t i = 0;
            for (; i + 8 <= key.size(); i += 8) {
                uint64_t w;
                memcpy(&w, key.data() + i, 8);
                h = (h ^ w) * 0xFF51AFD7ED558CCDull;
                h ^= h >> 29;
            }
            for (; i < key.size(); ++i) h = (h ^ key[i]) * 1099511628211ull;
            return h;
        };
        // groups in order of their first System; a System with the structure of the one before it joins that one's group
        // (one sketch, many parameter sets: two memcmp passes instead of a key), otherwise a 64-bit hash finds the candidates
        // and the bytes decide
        std::map<uint64_t, std::vector<size_t>> by_hash;
        size_t last_group = 0;
        uint32_t last_system = 0;
        bool have_last = false;
        for (uint32_t s : todo) {
            if (have_last && same_structure(s, last_system)) {
                groups[last_group].systems.push_back(s);
                continue;
            }
            std::vector<unsigned char> key = structure_key(s);
            const uint64_t h = hash_of(key);
            std::vector<size_t>& cand = by_hash[h];
            size_t g = groups.size();
            for (size_t i : cand)
                if (groups[i].key == key) g = i;
            if (g == groups.size()) {
                cand.push_back(g);
                groups.emplace_back();
                groups.back().key = std::move(key);
                groups.back().hash = h;
            }
            groups[g].systems.push_back(s);
            last_group = g;
            last_system = s;
            have_last = true;
        }
    }
    if (groups.empty()) return FX_OK;
    // plans: a resident batch keeps its own (per structure and decomposer mode); one-shot calls share the context's
    // (fx_ctx::plan_for — entries this call has touched are never evicted under it). Cluster problems of
    // RecursiveAssembly differ from step to step: nothing to keep.
    const uint64_t call_clock = ctx->plan_clock;
    std::vector<fx::SparsePlanCache*> group_plan(groups.size(), nullptr);
    for (size_t g = 0; g < groups.size(); ++g) {
        if (db->resident) {
            const uint64_t h = groups[g].hash ^ ((p.mode & fx::MODE_UNITS) ? 0x9E3779B97F4A7C15ull : 0ull);
            auto range = db->sparse_plans.equal_range(h);
            fx_dbatch::ResidentPlan* found = nullptr;
            for (auto it = range.first; it != range.second; ++it)
                if (it->second.key == groups[g].key) found = &it->second;
            if (!found) {
                found = &db->sparse_plans.emplace(h, fx_dbatch::ResidentPlan{groups[g].key, fx::sparse_cache_new()})->second;
                fx::sparse_cache_keep_slab(found->plan, SIZE_MAX);  // the batch is there to be solved again: its slab goes with it
            }
            group_plan[g] = found->plan;
        } else if (!pose) {
            group_plan[g] = ctx->plan_for(std::vector<unsigned char>(groups[g].key), call_clock);
        }
    }
    // every launch covers a whole group, control flow on the device (fx_sparse_team.h): Levenberg-Marquardt or L-BFGS.
    // Groups are independent (their own plans, slabs and Systems): several of them run side by side, a host thread and a
    // stream each (fx_ctx_set_host_threads) — one structure's launches leave most of the chip idle (32 different
    // 150-variable sketches: 9.5 ms one after the other).
    const uint32_t n_workers = (uint32_t)std::min<size_t>(std::max(1u, ctx->host_threads), groups.size());
    if (n_workers <= 1) {
        for (size_t g = 0; g < groups.size(); ++g) {
            hipError_t e = fx::sparse_solve_group(&hb, db->d, groups[g].systems.data(), (uint32_t)groups[g].systems.size(), p, ctx->stream,
                                                  group_plan[g]);
            if (e != hipSuccess)
                return fail(FX_ERR_HIP, "sparse path failed on the group of system %u: %s", groups[g].systems[0], hipGetErrorString(e));
        }
        return FX_OK;
    }
    while (ctx->worker_streams.size() + 1 < n_workers) {
        hipStream_t st = nullptr;
        FX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->worker_streams.push_back(st);
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the batch's upload and whatever else the context's stream still holds
    ctx->stream_synced();
    std::atomic<size_t> next{0};
    std::vector<hipError_t> errs(n_workers, hipSuccess);
    std::vector<size_t> err_group(n_workers, 0);
    auto work = [&](uint32_t w) {
        (void)hipSetDevice(ctx->device);
        hipStream_t st = w == 0 ? ctx->stream : ctx->worker_streams[w - 1];
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups.size() || errs[w] != hipSuccess) break;
            errs[w] = fx::sparse_solve_group(&hb, db->d, groups[g].systems.data(), (uint32_t)groups[g].systems.size(), p, st, group_plan[g]);
            err_group[w] = g;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t w = 1; w < n_workers; ++w) th.emplace_back(work, w);
    work(0);
    for (auto& t : th) t.join();
    for (uint32_t w = 0; w < n_workers; ++w)
        if (errs[w] != hipSuccess)
            return fail(FX_ERR_HIP, "sparse path failed on the group of system %u: %s", groups[err_group[w]].systems[0], hipGetErrorString(errs[w]));
    return FX_OK;
}
}  // namespace

extern "C" {

int fx_abi_version(void) { return FX_ABI_VERSION; }

const char* fx_last_error(void) { return g_last_error.c_str(); }

int fx_device_count(int* count) {
    if (!count) return fail(FX_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(FX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return FX_OK;
}

int fx_ctx_create(fx_ctx** out, int device) {
    if (!out) return fail(FX_ERR_INVALID, "ctx out-pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(FX_ERR_NO_DEVICE, "no HIP device available (%s); fiksi_amd has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(FX_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    FX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FX_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    fx_ctx* ctx = new (std::nothrow) fx_ctx();
    if (!ctx) return fail(FX_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    if (const char* sw = getenv("FIKSI_AMD_GROUPED_C"))  // A / B and tests: 0 keeps every batch on the grouped kernel's general build
        if (sw[0] == '0') ctx->grouped_one_structure = 0;
    if (const char* sw = getenv("FIKSI_AMD_GROUPED")) {  // the default of fx_ctx_set_routing's first option
        if (sw[0] == '0') ctx->route_grouped = 0;
        if (sw[0] == '1') ctx->route_grouped = 1;
    }
    snprintf(ctx->name, sizeof(ctx->name), "%s", prop.name);
    snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_end);
    if (e != hipSuccess) {
        fx_ctx_destroy(ctx);
        return fail(FX_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return FX_OK;
}

void fx_ctx_destroy(fx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    for (hipStream_t st : ctx->worker_streams) (void)hipStreamDestroy(st);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->ev_chunk) (void)hipEventDestroy(ctx->ev_chunk);
    if (ctx->ev_pinned) (void)hipEventDestroy(ctx->ev_pinned);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    ctx->drop_plans();
    ctx->drop_cache();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->zc) (void)hipHostFree(ctx->zc);
    delete ctx;
}

int fx_ctx_set_routing(fx_ctx* ctx, int grouped, uint32_t grouped_min_systems) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (grouped < -1 || grouped > 1) return fail(FX_ERR_INVALID, "grouped must be -1 (by batch size), 0 or 1");
    ctx->route_grouped = grouped;
    if (grouped_min_systems) ctx->grouped_min_systems = grouped_min_systems;
    return FX_OK;
}

int fx_ctx_set_one_structure_builds(fx_ctx* ctx, int enable) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->grouped_one_structure = enable ? 1 : 0;
    return FX_OK;
}

int fx_ctx_set_hold_passes(fx_ctx* ctx, uint32_t passes) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->hold_passes = passes;
    return FX_OK;
}

int fx_ctx_set_ladder(fx_ctx* ctx, int enable, uint32_t tail_systems, uint32_t min_trials, int spread) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->ladder = enable ? 1u : 0u;
    ctx->ladder_tail = tail_systems;
    ctx->ladder_k = min_trials;
    ctx->ladder_spread = spread ? 1u : 0u;
    return FX_OK;
}

int fx_ctx_set_wide_routing(fx_ctx* ctx, int wide) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (wide < -1 || wide > 1) return fail(FX_ERR_INVALID, "wide must be -1 (by cost), 0 (team kernels) or 1 (wide kernel)");
    ctx->wide_routing = wide;
    return FX_OK;
}

int fx_ctx_set_host_threads(fx_ctx* ctx, uint32_t threads) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->host_threads = threads ? std::min(threads, 64u) : 8u;
    return FX_OK;
}

int fx_ctx_set_presort(fx_ctx* ctx, int enable, uint32_t min_systems) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->presort = enable ? 1 : 0;
    if (min_systems) ctx->presort_min_systems = min_systems;
    return FX_OK;
}

int fx_ctx_synchronize(fx_ctx* ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_ctx_device_name(fx_ctx* ctx, char* buf, size_t len) {
    if (!ctx || !buf || len == 0) return fail(FX_ERR_INVALID, "bad argument");
    snprintf(buf, len, "%s (%s)", ctx->name, ctx->arch);
    return FX_OK;
}

void fx_lm_opts_default(fx_lm_opts* o) {
    if (!o) return;
    o->lambda0 = 0.5;
    o->sse_tol = 1e-8;
    o->step_tol = 1e-12;
    o->ftol = 1e-6;
    o->accept_factor = 0.125;
    o->reject_factor = 2.0;
    o->singular_factor = 8.0;
    o->lambda_min = 1e-50;
    o->max_outer = 100;
    o->max_trials = 4096;
    o->solver = FX_STEP_CHOLESKY;
    o->precision = 64;
}

void fx_lm_opts_default_f32(fx_lm_opts* o) {
    if (!o) return;
    fx_lm_opts_default(o);
    o->ftol = 1e-4;        // an f32 SSE carries round-off near 1e-5 ... 1e-4 relative once the residuals are small against
                           // the coordinates: below that an "improvement" is noise, and a solve that keeps accepting noise
                           // runs to max_outer (measured: 100k ring16 sketches 5.8 ms with 1e-5, 3.8 ms with 1e-4)
    o->lambda_min = 1e-7;  // keeps JtJ + lambda I numerically positive definite in f32
    o->max_outer = 40;     // the f64 solve of cfg5's batch never takes more than 56 accepted steps (99.9 %: 16); an f32 solve
                           // still improving by more than ftol after 40 is crawling on round-off (1 System in 125 000 used
                           // to take all 100 and, alone, a third of the batch's time). Same SSE statistics, 6.0 -> 4.0 ms
    o->precision = 32;
}

void fx_solving_opts_default(fx_solving_opts* o) {
    if (!o) return;
    o->optimizer = 0;
    o->decomposer = 0;
    o->perturb = 1;
    o->plan_budget = 0;
    fx_lm_opts_default(&o->lm);
}

int fx_batch_validate(const fx_batch* batch) { return analyze(batch, nullptr); }

int fx_jacobian_structure(const fx_batch* batch, uint64_t* nnz, uint32_t* row_ptr, uint32_t* col_idx) {
    HostPlan p;
    int rc = analyze(batch, &p);
    if (rc) return rc;
    if (nnz) *nnz = p.nnz;
    if (!row_ptr && !col_idx) return FX_OK;
    CsrPlan csr;
    build_csr(p.n_systems, batch->var_off, batch->expr_off, p.var_info.data(), p.expr_tagx.data(), p.expr_idx16.data(), csr);
    if (row_ptr) std::copy(csr.jrow_ptr.begin(), csr.jrow_ptr.end(), row_ptr);
    if (col_idx) std::copy(csr.jcol.begin(), csr.jcol.end(), col_idx);
    return FX_OK;
}

// Systems [s0, s1) of an analysed batch onto the device (the whole batch, or one chunk of solve_host_chunked: the limits
// that size kernels and LDS are the whole batch's either way, so a chunk runs the very kernels the whole batch would).
static int upload_planned(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, uint32_t s0, uint32_t s1, fx_dbatch** out,
                          bool one_shot = false) {
    *out = nullptr;
    int rc = FX_OK;
    const bool whole = s0 == 0 && s1 == p.n_systems;
    const uint32_t n_sys = s1 - s0;
    const uint32_t v0 = n_sys ? batch->var_off[s0] : 0, e0 = n_sys ? batch->expr_off[s0] : 0;
    const uint32_t n_vars = n_sys ? batch->var_off[s1] - v0 : 0, n_exprs = n_sys ? batch->expr_off[s1] - e0 : 0;
    fx_dbatch* db = new (std::nothrow) fx_dbatch();
    if (!db) return fail(FX_ERR_NOMEM, "out of host memory");
    db->resident = true;
    fx::DeviceBatch& d = db->d;
    d.n_systems = n_sys;
    d.n_vars = n_vars;
    d.n_exprs = n_exprs;
    d.nnz = p.nnz;  // (of the whole batch: an upper bound for a chunk, which never builds its CSR)
    d.max_free = p.max_free;
    d.max_rows = p.max_rows;
    d.max_vars = p.max_vars;
    d.max_exprs = p.max_exprs;
    d.max_vars_all = p.max_vars_all;
    d.max_exprs_all = p.max_exprs_all;
    d.max_pairs = p.max_pairs;
    d.max_ents = p.max_ents;
    d.max_pairs_tri = p.max_pairs_tri;
    d.uniform = p.uniform && n_sys >= 2 ? 1u : 0u;
    d.u_nvars = d.uniform ? batch->var_off[1] : 0;
    d.u_nexprs = d.uniform ? batch->expr_off[1] : 0;
    d.u_ncomp = d.uniform ? p.sys_ncomp[0] : 0;
    d.max_pairs_g = p.max_pairs_large;  // blocks of a large System hold at most its components' products
    d.max_ents_g = p.max_ents_large;
    const uint32_t zero_off[1] = {0};
    const uint32_t* voff = n_sys ? batch->var_off : zero_off;
    const uint32_t* eoff = n_sys ? batch->expr_off : zero_off;
    std::vector<uint32_t> voff_r, eoff_r, class_r;
    if (!whole) {  // a chunk's offsets start at 0; a structure class is the first System OF THE CHUNK with the structure
        voff_r.resize((size_t)n_sys + 1);
        eoff_r.resize((size_t)n_sys + 1);
        for (uint32_t s = 0; s <= n_sys; ++s) {
            voff_r[s] = batch->var_off[s0 + s] - v0;
            eoff_r[s] = batch->expr_off[s0 + s] - e0;
        }
        voff = voff_r.data();
        eoff = eoff_r.data();
        if (!p.sys_class.empty()) {
            class_r.resize(n_sys);
            std::unordered_map<uint32_t, uint32_t> first;
            for (uint32_t s = 0; s < n_sys; ++s) {
                if (s && p.sys_class[s0 + s] == p.sys_class[s0 + s - 1]) {
                    class_r[s] = class_r[s - 1];
                    continue;
                }
                class_r[s] = first.emplace(p.sys_class[s0 + s], s).first->second;
            }
        }
    }
    const uint32_t* sys_class = p.sys_class.empty() ? nullptr : whole ? p.sys_class.data() : class_r.data();
    // Every array of the batch: (device pointer to set, host source or NULL for zeros, bytes) — ONE device block.
    //  - a small batch (one System::solve) goes up with ONE copy from the page-locked staging area: a dozen-and-a-half
    //    separate hipMemcpy calls cost more than its solve;
    //  - up to PINNED_HALF the same, but `vars` (the working copy of vars0) and the zeroed `results` at the block's end
    //    are made on the device (a copy, a memset) instead of crossing the bus;
    //  - a big batch: one copy per array straight from the caller's memory (no extra pass over 100 MB on the host).
    // (`period`: the array repeats with this period — the structure arrays of a batch of one structure —, so a big batch sends
    // its first `period` bytes over the bus and the device fills in the rest: 45 % of a ring16 batch's bytes)
    struct Req { void** dst; const void* src; size_t bytes; size_t period; };
    std::vector<Req> reqs;
#define FX_UP(field, host, count) \
    reqs.push_back({reinterpret_cast<void**>(&d.field), static_cast<const void*>(host), (size_t)(count) * sizeof(*d.field), 0});
#define FX_UP_PERIODIC(field, host, count, per_system) \
    reqs.push_back({reinterpret_cast<void**>(&d.field), static_cast<const void*>(host), (size_t)(count) * sizeof(*d.field), \
                    d.uniform ? (size_t)(per_system) * sizeof(*d.field) : 0});
    FX_UP(var_off, voff, (size_t)n_sys + 1)
    FX_UP(expr_off, eoff, (size_t)n_sys + 1)
    FX_UP(sys_ncomp, p.sys_ncomp.data() + s0, n_sys)
    FX_UP(sys_large, p.sys_large.data() + s0, n_sys)
    FX_UP(vars0, (const double*)batch->vars + v0, n_vars)
    FX_UP_PERIODIC(var_info, p.var_info.data() + v0, n_vars, d.u_nvars)
    FX_UP_PERIODIC(expr_tag, p.expr_tagx.data() + e0, n_exprs, d.u_nexprs)
    FX_UP_PERIODIC(expr_comp, p.expr_comp.data() + e0, n_exprs, d.u_nexprs)
    FX_UP_PERIODIC(expr_idx, p.expr_idx16.data() + 4 * (size_t)e0, 4 * (size_t)n_exprs, 4 * (size_t)d.u_nexprs)
    FX_UP(expr_param, batch->expr_param + e0, n_exprs)
    FX_UP(work_counter, (const uint32_t*)nullptr, 16)  // (the batch's queue head, then those of up to eight structure classes and of the rest: launch_class_solves)
    if (sys_class) FX_UP(sys_class, sys_class, n_sys)
    // A batch of one structure gets the program of one of the grouped kernel's builds for such batches, when the structure
    // qualifies: the sparse build (fx_grouped_s.hip) for components beyond a register-resident factor — and from 33 free variables
    // on when the factor is sparse (at most a quarter of the dense triangle: the reference's bench sketch of 11 triangles, 46
    // variables, 201 of 1 081 entries: 1.52 ms per 100 000 against 3.14 in the 48-column register build) —, the register build
    // (fx_grouped_c.hip) otherwise.
    GsHostProgram gs;
    GcHostProgram gc;
    bool sparse_build = false;
    if (d.uniform && d.u_ncomp == 1u && d.u_nvars <= 255u && d.u_nexprs <= 255u &&
        build_gs_program(p.var_info.data() + v0, p.expr_tagx.data() + e0, p.expr_comp.data() + e0, p.expr_idx16.data() + 4 * (size_t)e0, d.u_nvars,
                         d.u_nexprs, gs))
        sparse_build = gs.nfree > 48u || 4u * gs.nl <= gs.nfree * (gs.nfree + 1u) / 2u;
    if (sparse_build) {
        FX_UP(gs_tab, gs.words.data(), gs.words.size())
        d.gs_words = (uint32_t)gs.words.size();
        d.gs_nl = gs.nl;
        d.gs_ng = gs.ng;
        d.gs_nfree = gs.nfree;
    } else if (d.uniform && d.u_ncomp == 1u && p.n_large == 0 && p.max_free >= 1u && p.max_free <= 48u &&
               build_gc_program(p.var_info.data() + v0, p.expr_tagx.data() + e0, p.expr_comp.data() + e0, p.expr_idx16.data() + 4 * (size_t)e0,
                                d.u_nvars, d.u_nexprs, p.max_free, gc)) {
        FX_UP(gc_tab, gc.words.data(), gc.words.size())
        d.gc_words = gc.words_f64;
        d.gc_words_all = (uint32_t)gc.words.size();
        d.gc_nslots = gc.nslots;
        d.gc_ng = gc.ng;
        d.gc_nc = gc.nc;
        d.gc_rc = gc.rc;
    }
    // Several structures: the classes with 2 048 members and more (at most eight, the largest first; those of the largest one's
    // build — columns per lane) get a program each
    std::vector<uint32_t> cl_words_h, cl_lists_h;
    if (!d.uniform && sys_class && p.n_large == 0 && p.max_free >= 1u && p.max_free <= 48u) {
        constexpr uint32_t CLASS_MIN = 2048u, MAX_CLASSES = 8u;
        std::vector<uint32_t> count(n_sys, 0u);  // (a class is named by its first System)
        for (uint32_t s = 0; s < n_sys; ++s) count[sys_class[s]] += 1u;
        std::vector<std::pair<uint32_t, uint32_t>> big;  // (members, first System)
        for (uint32_t f = 0; f < n_sys; ++f)
            if (count[f] >= CLASS_MIN) big.push_back({count[f], f});
        std::sort(big.begin(), big.end(), [](auto& a, auto& b2) { return a.first != b2.first ? a.first > b2.first : a.second < b2.second; });
        std::vector<uint32_t>& class_slot = count;  // first System -> index into db->classes (reusing the array: NO_SLOT elsewhere)
        constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
        std::fill(class_slot.begin(), class_slot.end(), NO_SLOT);
        for (auto& pr : big) {
            if (db->classes.size() >= MAX_CLASSES) break;
            const uint32_t f = pr.second;
            if (p.sys_ncomp[s0 + f] != 1u) continue;
            const uint32_t vf = v0 + voff[f], ef = e0 + eoff[f], nvt = voff[f + 1] - voff[f], net = eoff[f + 1] - eoff[f];
            uint32_t nfree = 0;
            for (uint32_t i = 0; i < nvt; ++i) nfree += (p.var_info[vf + i] & fx::VAR_FIXED_BIT) ? 0u : 1u;
            GcHostProgram cp;
            if (!build_gc_program(p.var_info.data() + vf, p.expr_tagx.data() + ef, p.expr_comp.data() + ef, p.expr_idx16.data() + 4 * (size_t)ef, nvt, net,
                                  nfree, cp))
                continue;
            if (db->classes.empty()) {
                db->cl_nc = cp.nc;
                db->cl_rc = cp.rc;
            }
            if (cp.nc != db->cl_nc || cp.rc != db->cl_rc) continue;
            fx::GcClass cl;
            cl.prog_off = (uint32_t)cl_words_h.size();
            cl.words = cp.words_f64;
            cl.words_all = (uint32_t)cp.words.size();
            cl.list_off = 0;
            cl.count = pr.first;
            db->cl_max_words = std::max(db->cl_max_words, cl.words);
            db->cl_max_words_all = std::max(db->cl_max_words_all, cl.words_all);
            db->cl_max_slots = std::max(db->cl_max_slots, cp.nslots);
            db->cl_max_ng = std::max(db->cl_max_ng, cp.ng);
            cl_words_h.insert(cl_words_h.end(), cp.words.begin(), cp.words.end());
            class_slot[f] = (uint32_t)db->classes.size();
            db->classes.push_back(cl);
        }
        if (!db->classes.empty()) {
            uint32_t at = 0;
            for (auto& cl : db->classes) {
                cl.list_off = at;
                at += cl.count;
            }
            db->cl_systems = at;
            db->rest_off = at;
            cl_lists_h.resize(n_sys);
            std::vector<uint32_t> fill(db->classes.size(), 0);
            uint32_t nrest = 0;
            for (uint32_t s = 0; s < n_sys; ++s) {
                const uint32_t slot = class_slot[sys_class[s]];
                if (slot == NO_SLOT) cl_lists_h[db->rest_off + nrest++] = s;
                else cl_lists_h[db->classes[slot].list_off + fill[slot]++] = s;
            }
            db->rest_count = nrest;
            reqs.push_back({reinterpret_cast<void**>(&db->cl_words), cl_words_h.data(), cl_words_h.size() * 4, 0});
            reqs.push_back({reinterpret_cast<void**>(&db->cl_lists), cl_lists_h.data(), cl_lists_h.size() * 4, 0});
            reqs.push_back({reinterpret_cast<void**>(&db->cl_desc), db->classes.data(), db->classes.size() * sizeof(fx::GcClass), 0});
        }
    }
    FX_UP(w_list, p.wide_list.data(), whole ? p.wide_list.size() : 0)
    const size_t n_front = reqs.size();  // the two below end the block, side by side: a one-shot solve reads them back in one copy
    FX_UP(vars, (const double*)batch->vars + v0, n_vars)
    FX_UP(results, (const fx_result*)nullptr, n_sys)
#undef FX_UP
#undef FX_UP_PERIODIC
    auto room_of = [](const Req& r) { return (std::max<size_t>(r.bytes, 1) + 255u) & ~size_t(255); };
    size_t packed = 0, front = 0;
    for (size_t i = 0; i < reqs.size(); ++i) {
        packed += room_of(reqs[i]);
        if (i + 1 == n_front) front = packed;
    }
    hipError_t e1 = hipSuccess;
    // A one-shot solve of a few small Systems (System::solve on one sketch) makes no copy call: the host writes the block's
    // image into the context's host-coherent region, one small kernel pulls it over in a burst and another pushes `vars` and
    // `results` back when the solve is done (2 us each; a solve kernel working on the region in place pays a PCIe round trip
    // per dependent load, and waits for its stores: 40 / 30 us instead of 10 — both measured).
    const bool zero_copy = one_shot && whole && packed <= fx_ctx::ZC_BYTES && p.n_large == 0 && p.wide_list.empty() && ctx->ensure_zc();
    unsigned char* base = static_cast<unsigned char*>(ctx->take(packed, e1));
    if (!base) {
        fx_batch_free(ctx, db);
        return fail(e1 == hipErrorOutOfMemory ? FX_ERR_NOMEM : FX_ERR_HIP, "hipMalloc(%zu): %s", packed, hipGetErrorString(e1));
    }
    db->allocations.push_back({base, packed});
    db->zero_copy = zero_copy;
    db->zc_image = zero_copy ? ctx->zc : nullptr;
    {
        size_t at = 0;
        for (size_t i = 0; i < reqs.size(); ++i) {
            *reqs[i].dst = base + at;
            at += room_of(reqs[i]);
        }
    }
    const bool on_device_tail = !zero_copy && packed > (size_t(256) << 10);  // vars and results made on the device
    auto upload = [&]() -> int {
        if (zero_copy) {
            db->packed_base = base;
            db->packed_bytes = packed;
            unsigned char* img = ctx->zc;
            size_t at = 0;
            for (const Req& r : reqs) {
                const size_t room = room_of(r);
                if (r.src && r.bytes) {
                    memcpy(img + at, r.src, r.bytes);
                    memset(img + at + r.bytes, 0, room - r.bytes);
                } else {
                    memset(img + at, 0, room);
                }
                at += room;
            }
            FX_HIP(fx::launch_pull(base, img, packed, ctx->stream));
            db->zc_front = front;
            db->upload_pending = true;  // (the image outlives this frame: nothing to wait for below)
            return FX_OK;
        }
        if (packed <= fx_ctx::PINNED_HALF) {
            db->packed_base = base;
            db->packed_bytes = packed;
            const size_t staged = on_device_tail ? front : packed;
            std::vector<unsigned char> stage;
            unsigned char* st = nullptr;
            if (ctx->ensure_pinned()) {
                ctx->wait_pinned();  // an earlier upload may still be reading the staging area
                st = ctx->pinned;
                db->upload_pending = true;
            } else {
                stage.resize(staged);
                st = stage.data();
            }
            size_t at = 0;
            for (size_t i = 0; i < reqs.size() && at < staged; ++i) {
                const Req& r = reqs[i];
                const size_t room = room_of(r);
                if (r.src && r.bytes) {
                    memcpy(st + at, r.src, r.bytes);
                    memset(st + at + r.bytes, 0, room - r.bytes);
                } else {
                    memset(st + at, 0, room);
                }
                at += room;
            }
            FX_HIP(hipMemcpyAsync(base, st, staged, hipMemcpyHostToDevice, ctx->stream));
            if (db->upload_pending) {
                (void)hipEventRecord(ctx->ev_pinned, ctx->stream);
                ctx->pinned_stream = ctx->stream;
                ctx->pinned_busy = true;
            } else {
                FX_HIP(hipStreamSynchronize(ctx->stream));  // `stage` goes away with this frame
            }
        } else {
            for (size_t i = 0; i < n_front; ++i) {
                const Req& r = reqs[i];
                if (r.src && r.bytes && r.period && r.period < r.bytes) {
                    FX_HIP(hipMemcpyAsync(*r.dst, r.src, r.period, hipMemcpyHostToDevice, ctx->stream));
                    FX_HIP(fx::launch_replicate(*r.dst, r.period, r.bytes, ctx->stream));
                } else if (r.src && r.bytes) {
                    FX_HIP(hipMemcpyAsync(*r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->stream));
                } else {
                    FX_HIP(hipMemsetAsync(*r.dst, 0, room_of(r), ctx->stream));
                }
            }
        }
        if (on_device_tail) {
            if (n_vars) FX_HIP(hipMemcpyAsync(d.vars, d.vars0, (size_t)n_vars * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            FX_HIP(hipMemsetAsync(d.results, 0, room_of(reqs.back()), ctx->stream));
        }
        return FX_OK;
    };
    rc = upload();
    if (rc) {
        fx_batch_free(ctx, db);
        return rc;
    }
    d.n_wide = whole ? (uint32_t)p.wide_list.size() : 0u;
    d.w_max_free = p.w_max_free;
    d.w_max_vars = p.w_max_vars;
    d.w_max_rows = p.w_max_rows;
    // jrow_ptr / jcol / jslot / jvals (the CSR Jacobian) and resid are only needed by the standalone
    // evaluation entry points: they are built on first use (ensure_csr / ensure_resid), so a plain
    // solve neither computes nor uploads them.
#undef FX_UP
    // the host plan lives on this stack frame: finish the copies before returning (a small batch went through the
    // context's page-locked staging area, which outlives the call: its copy is left in flight)
    if (!db->upload_pending) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            fx_batch_free(ctx, db);
            return fail(FX_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
        }
    }
    if (p.n_large && !whole) {
        fx_batch_free(ctx, db);
        return fail(FX_ERR_INVALID, "internal: a batch with Systems beyond one wavefront is not cut into chunks");
    }
    if (p.n_large) {
        const uint32_t nv = p.n_vars, ne = p.n_exprs, n = p.n_systems;
        db->n_large = p.n_large;
        db->h_sys_large.assign(p.sys_large.begin(), p.sys_large.end());
        db->h_var_off.assign(batch->var_off, batch->var_off + n + 1);
        db->h_expr_off.assign(batch->expr_off, batch->expr_off + n + 1);
        db->h_vars.assign(batch->vars, batch->vars + nv);
        db->h_var_fixed.assign(batch->var_fixed, batch->var_fixed + nv);
        db->h_expr_tag.assign(batch->expr_tag, batch->expr_tag + ne);
        db->h_expr_idx.assign(batch->expr_idx, batch->expr_idx + 4 * (size_t)ne);
        db->h_expr_param.assign(batch->expr_param, batch->expr_param + ne);
        if (batch->var_comp) db->h_var_comp.assign(batch->var_comp, batch->var_comp + nv);
        if (batch->expr_comp) db->h_expr_comp.assign(batch->expr_comp, batch->expr_comp + ne);
        fx_batch& hb = db->h_batch;
        hb.n_systems = n;
        hb.var_off = db->h_var_off.data();
        hb.expr_off = db->h_expr_off.data();
        hb.vars = db->h_vars.data();
        hb.var_fixed = db->h_var_fixed.data();
        hb.expr_tag = db->h_expr_tag.data();
        hb.expr_idx = db->h_expr_idx.data();
        hb.expr_param = db->h_expr_param.data();
        hb.var_comp = batch->var_comp ? db->h_var_comp.data() : nullptr;
        hb.expr_comp = batch->expr_comp ? db->h_expr_comp.data() : nullptr;
    }
    if (fx::solve_lds_bytes(d) > 160u * 1024u) {
        fx_batch_free(ctx, db);
        return fail(FX_ERR_TOO_LARGE, "batch needs %zu bytes of LDS per wavefront (limit 163840)", fx::solve_lds_bytes(d));
    }
    *out = db;
    return FX_OK;
}

int fx_batch_upload(fx_ctx* ctx, const fx_batch* batch, fx_dbatch** out) {
    if (!out) return fail(FX_ERR_INVALID, "out-pointer is NULL");
    *out = nullptr;
    int rc = bind(ctx);
    if (rc) return rc;
    HostPlan p;
    PhaseTrace tr;
    g_wide_routing = ctx->wide_routing;
    rc = analyze(batch, &p);
    if (rc) return rc;
    tr.stamp("  analysis", batch->n_systems);
    return upload_planned(ctx, batch, p, 0, p.n_systems, out);
}

static void free_batch(fx_ctx* ctx, fx_dbatch* db, bool stream_idle) {
    if (!db) return;
    if (ctx && !stream_idle) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->stream_synced();
    }
    for (auto& kv : db->sparse_plans) fx::sparse_cache_free(kv.second.plan);
    for (auto& blk : db->allocations) {
        if (ctx) ctx->give_back(blk.p, blk.size);  // the stream is idle: the blocks can be handed out again
        else (void)hipFree(blk.p);
    }
    delete db;
}

void fx_batch_free(fx_ctx* ctx, fx_dbatch* db) { free_batch(ctx, db, false); }

int fx_batch_set_vars(fx_ctx* ctx, fx_dbatch* db, const double* vars) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !vars) return fail(FX_ERR_INVALID, "bad argument");
    if (db->n_large) std::copy(vars, vars + db->d.n_vars, db->h_vars.begin());
    FX_HIP(hipMemcpyAsync(db->d.vars0, vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipMemcpyAsync(db->d.vars, vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_batch_set_params(fx_ctx* ctx, fx_dbatch* db, const double* expr_param) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !expr_param) return fail(FX_ERR_INVALID, "bad argument");
    if (db->n_large) std::copy(expr_param, expr_param + db->d.n_exprs, db->h_expr_param.begin());
    FX_HIP(hipMemcpyAsync(db->d.expr_param, expr_param, (size_t)db->d.n_exprs * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_batch_schedule_by_last_solve(fx_ctx* ctx, fx_dbatch* db, int enable) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    fx::DeviceBatch& d = db->d;
    if (!enable) {
        d.order = nullptr;  // (the array stays allocated with the batch)
        return FX_OK;
    }
    const uint32_t n = d.n_systems;
    std::vector<fx_result> res(n);
    if (n) FX_HIP(hipMemcpyAsync(res.data(), d.results, (size_t)n * sizeof(fx_result), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> order(n);
    for (uint32_t s = 0; s < n; ++s) order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) { return res[a].trials > res[b2].trials; });
    if (!db->d_order) {
        rc = dev_alloc_copy(ctx, db, &db->d_order, order.data(), order.size());
        if (rc) return rc;
    } else if (n) {
        FX_HIP(hipMemcpyAsync(db->d_order, order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    d.order = db->d_order;
    return FX_OK;
}

int fx_batch_get_vars(fx_ctx* ctx, fx_dbatch* db, double* vars) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !vars) return fail(FX_ERR_INVALID, "bad argument");
    FX_HIP(hipMemcpyAsync(vars, db->d.vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_batch_get_results(fx_ctx* ctx, fx_dbatch* db, fx_result* results) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !results) return fail(FX_ERR_INVALID, "bad argument");
    FX_HIP(hipMemcpyAsync(results, db->d.results, (size_t)db->d.n_systems * sizeof(fx_result), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

uint64_t fx_batch_nnz(const fx_dbatch* db) { return db ? db->d.nnz : 0; }

// Systems beyond one wavefront. FX_STEP_QR: those whose components have at most 128 columns run the wide kernel's QR build
// (the reference's numerics; ensure_qr_plans listed them), everything larger takes the refined step on the sparse path.
static int solve_beyond_one_wavefront(fx_ctx* ctx, fx_dbatch* db, fx::LmParams p) {
    if (p.lm.solver == FX_STEP_QR) {
        const bool qr_wide = !(p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS)) && db->d.qr_none.n_qrw != 0 && !db->d.has_pose;
        if (qr_wide) {
            hipError_t e = fx::launch_solve_wide_qr(db->d, p, ctx->stream);
            if (e != hipSuccess) return fail(FX_ERR_HIP, "wide QR kernel launch failed: %s", hipGetErrorString(e));
        }
        p.lm.solver = FX_STEP_CHOLESKY_REFINED;
        db->qr_wide_active = qr_wide;
        const int rc = solve_large_systems(ctx, db, p);
        db->qr_wide_active = false;
        return rc;
    }
    return solve_large_systems(ctx, db, p);
}

int fx_system_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    fx_solving_opts o;
    if (opts) o = *opts; else fx_solving_opts_default(&o);
    if (o.optimizer > 1) return fail(FX_ERR_UNSUPPORTED, "unknown optimizer %u (0 = LevenbergMarquardt, 1 = LBfgs)", o.optimizer);
    if (o.optimizer == 1 && o.lm.precision == 32)
        return fail(FX_ERR_UNSUPPORTED, "Optimizer::LBfgs runs in f64 only");

    if (o.decomposer == 2)
        return fail(FX_ERR_UNSUPPORTED, "Decomposer::RecursiveAssembly works on a System's elements and constraints: call it through the "
                                        "builder (fxs_system_solve); a flat batch does not carry them");
    if (o.decomposer > 2) return fail(FX_ERR_UNSUPPORTED, "unknown decomposer %u (0 = None, 1 = SinglePass, 2 = RecursiveAssembly)", o.decomposer);
    if (o.lm.solver > FX_STEP_QR) return fail(FX_ERR_UNSUPPORTED, "unknown step solver %u", o.lm.solver);
    fx::LmParams p;
    ctx->route(p);
    p.lm = o.lm;
    p.mode = 1u | (o.perturb ? 2u : 0u) | (o.optimizer == 1 ? fx::MODE_LBFGS : 0u);
    if (o.decomposer == 1) {
        rc = ensure_units(ctx, db);
        if (rc) return rc;
        p.mode |= fx::MODE_UNITS;
    }
    if (p.lm.solver == FX_STEP_QR) {
        if (o.optimizer != 0 || p.lm.precision == 32) return fail(FX_ERR_UNSUPPORTED, "FX_STEP_QR is the f64 Levenberg-Marquardt step");
        rc = ensure_qr_plans(ctx, db, o.decomposer == 1);
        if (rc) return rc;
    }
    if (fx::grouped_s_applies(db->d, p)) {  // one structure, a wide component with a small factor: every System of the batch in one launch
        FX_HIP(fx::launch_solve_grouped_s(db->d, p, ctx->stream));
        return FX_OK;
    }
    rc = launch_solve_scheduled(ctx, db, p);
    if (rc) return rc;
    return solve_beyond_one_wavefront(ctx, db, p);
}

int fx_lm_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_lm_opts* opts) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    fx::LmParams p;
    ctx->route(p);
    if (opts) p.lm = *opts; else fx_lm_opts_default(&p.lm);
    p.mode = 0;
    if (p.lm.solver > FX_STEP_QR) return fail(FX_ERR_UNSUPPORTED, "unknown step solver %u", p.lm.solver);
    if (p.lm.solver == FX_STEP_QR) {
        if (p.lm.precision == 32) return fail(FX_ERR_UNSUPPORTED, "FX_STEP_QR is the f64 Levenberg-Marquardt step");
        rc = ensure_qr_plans(ctx, db, false);
        if (rc) return rc;
    }
    if (fx::grouped_s_applies(db->d, p)) {  // one structure, a wide component with a small factor: every System of the batch in one launch
        FX_HIP(fx::launch_solve_grouped_s(db->d, p, ctx->stream));
        return FX_OK;
    }
    rc = launch_solve_scheduled(ctx, db, p);
    if (rc) return rc;
    return solve_beyond_one_wavefront(ctx, db, p);
}

int fx_debug_solve_route(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* route) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !route) return fail(FX_ERR_INVALID, "bad argument");
    fx_solving_opts o;
    if (opts) o = *opts; else fx_solving_opts_default(&o);
    fx::LmParams p;
    ctx->route(p);
    p.lm = o.lm;
    p.mode = 1u | (o.perturb ? 2u : 0u) | (o.optimizer == 1 ? fx::MODE_LBFGS : 0u) | (o.decomposer == 1 ? fx::MODE_UNITS : 0u);
    *route = fx::grouped_applies(db->d, p) ? 1 : 0;
    return FX_OK;
}

int fx_debug_grouped_build(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* build) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !build) return fail(FX_ERR_INVALID, "bad argument");
    fx_solving_opts o;
    if (opts) o = *opts; else fx_solving_opts_default(&o);
    fx::LmParams p;
    ctx->route(p);
    p.lm = o.lm;
    p.mode = 1u | (o.perturb ? 2u : 0u) | (o.optimizer == 1 ? fx::MODE_LBFGS : 0u) | (o.decomposer == 1 ? fx::MODE_UNITS : 0u);
    *build = fx::grouped_s_applies(db->d, p) ? 2 : !fx::grouped_applies(db->d, p) ? -1 : (p.lm.solver == FX_STEP_CHOLESKY && fx::grouped_c_applies(db->d, p)) ? 1 : 0;
    if (*build == 0 && !db->classes.empty() && !db->d.order && p.grouped_one_structure) {  // several structures: launch_class_solves
        fx::DeviceBatch dc = db->d;
        dc.gc_tab = db->cl_words;
        dc.gc_words = db->cl_max_words;
        dc.gc_words_all = db->cl_max_words_all;
        dc.gc_nslots = db->cl_max_slots;
        dc.gc_ng = db->cl_max_ng;
        dc.gc_nc = db->cl_nc;
        dc.gc_rc = db->cl_rc;
        dc.gc_nclasses = (uint32_t)db->classes.size();
        if (fx::grouped_c_applies(dc, p)) *build = 3;
    }
    return FX_OK;
}

// Diagnostic (not part of the drop-in surface): runs the stamped build of the fused kernel once and
// returns the shader cycles summed over all wavefronts for {setup, eval, form, factor, solve, tail}.
int fx_debug_phase_cycles(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, uint64_t cycles[6]) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !cycles) return fail(FX_ERR_INVALID, "bad argument");
    fx_solving_opts o;
    if (opts) o = *opts; else fx_solving_opts_default(&o);
    unsigned long long* dev = nullptr;
    FX_HIP(hipMalloc((void**)&dev, 6 * sizeof(unsigned long long)));
    FX_HIP(hipMemsetAsync(dev, 0, 6 * sizeof(unsigned long long), ctx->stream));
    fx::LmParams p;
    ctx->route(p);
    p.lm = o.lm;
    p.mode = 1u | (o.perturb ? 2u : 0u);
    p.prof = dev;
    if (p.lm.solver == FX_STEP_QR) {
        rc = ensure_qr_plans(ctx, db, false);
        if (rc) {
            (void)hipFree(dev);
            return rc;
        }
    }
    // a batch of medium Systems only: the wide kernel's stamps; otherwise the fused kernel's (N = 32 build)
    hipError_t e = (db->d.n_wide && db->d.n_wide == db->d.n_systems) ? fx::launch_solve_wide(db->d, p, ctx->stream)
                                                                     : fx::launch_solve(db->d, p, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cycles, dev, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(FX_ERR_HIP, "phase profile failed: %s", hipGetErrorString(e));
    return FX_OK;
}

int fx_eval_residual_jacobian_device(fx_ctx* ctx, fx_dbatch* db, int which) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    rc = ensure_csr(ctx, db);
    if (rc) return rc;
    FX_HIP(fx::launch_eval(db->d, which ? db->d.vars : db->d.vars0, true, ctx->stream));
    return FX_OK;
}

int fx_eval_residual_device(fx_ctx* ctx, fx_dbatch* db, int which) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    rc = ensure_resid(ctx, db);
    if (rc) return rc;
    FX_HIP(fx::launch_eval(db->d, which ? db->d.vars : db->d.vars0, false, ctx->stream));
    return FX_OK;
}

int fx_batch_get_residuals(fx_ctx* ctx, fx_dbatch* db, double* r) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !r) return fail(FX_ERR_INVALID, "bad argument");
    if (!db->d.resid) return fail(FX_ERR_INVALID, "no residuals have been evaluated on this batch yet");
    FX_HIP(hipMemcpyAsync(r, db->d.resid, (size_t)db->d.n_exprs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_batch_get_jacobian_values(fx_ctx* ctx, fx_dbatch* db, double* jvals) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !jvals) return fail(FX_ERR_INVALID, "bad argument");
    if (!db->d.jvals) return fail(FX_ERR_INVALID, "no Jacobian has been evaluated on this batch yet");
    FX_HIP(hipMemcpyAsync(jvals, db->d.jvals, (size_t)db->d.nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}

int fx_timer_begin(fx_ctx* ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    FX_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    return FX_OK;
}

int fx_timer_end(fx_ctx* ctx, float* milliseconds) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!milliseconds) return fail(FX_ERR_INVALID, "milliseconds is NULL");
    FX_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    FX_HIP(hipEventSynchronize(ctx->ev_end));
    FX_HIP(hipEventElapsedTime(milliseconds, ctx->ev_begin, ctx->ev_end));
    return FX_OK;
}

// ---- host-buffer entry points ---------------------------------------------------------------

// After a one-shot solve: solved variables and results back to the caller, then the batch is freed. A small batch sits in
// one block on the device: both come back in ONE copy through the page-locked staging area and the call waits on the
// stream once.
static int read_back_and_free(fx_ctx* ctx, fx_dbatch* db, const fx_batch* batch, fx_result* results, int rc) {
    if (db->zero_copy) {  // vars and results, the end of the block, pushed into the host-coherent image by one small kernel
        if (!rc) {
            hipError_t e = fx::launch_pull(db->zc_image + db->zc_front, db->packed_base + db->zc_front, db->packed_bytes - db->zc_front, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail(FX_ERR_HIP, "solve failed: %s", hipGetErrorString(e));
            else ctx->stream_synced();
        }
        if (!rc && batch->n_systems) {
            const unsigned char* dev0 = db->packed_base;
            memcpy(batch->vars, db->zc_image + (reinterpret_cast<const unsigned char*>(db->d.vars) - dev0), (size_t)db->d.n_vars * sizeof(double));
            if (results)
                memcpy(results, db->zc_image + (reinterpret_cast<const unsigned char*>(db->d.results) - dev0), (size_t)db->d.n_systems * sizeof(fx_result));
        }
        free_batch(ctx, db, /*stream_idle=*/rc == FX_OK);
        return rc;
    }
    if (!rc && batch->n_systems && db->packed_base && ctx->pinned && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        const unsigned char* lo = reinterpret_cast<const unsigned char*>(db->d.vars);
        const unsigned char* hi = reinterpret_cast<const unsigned char*>(db->d.results + db->d.n_systems);
        if (lo >= db->packed_base && hi <= db->packed_base + db->packed_bytes && lo < hi) {
            unsigned char* back = ctx->pinned + fx_ctx::PINNED_HALF;
            auto run = [&]() -> int {
                FX_HIP(hipMemcpyAsync(back, lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
                FX_HIP(hipStreamSynchronize(ctx->stream));
                return FX_OK;
            };
            rc = run();
            ctx->stream_synced();
            if (!rc) {
                memcpy(batch->vars, back, (size_t)db->d.n_vars * sizeof(double));
                if (results) memcpy(results, back + (reinterpret_cast<const unsigned char*>(db->d.results) - lo), (size_t)db->d.n_systems * sizeof(fx_result));
            }
            free_batch(ctx, db, /*stream_idle=*/rc == FX_OK);
            return rc;
        }
    }
    if (!rc && batch->n_systems) rc = fx_batch_get_vars(ctx, db, batch->vars);
    if (!rc && results && batch->n_systems) rc = fx_batch_get_results(ctx, db, results);
    fx_batch_free(ctx, db);
    return rc;
}

// A big batch of one-wavefront Systems, analysed as a whole, goes up and is solved in chunks of some megabytes: chunk k + 1
// is copied up (second stream) while chunk k is being solved; the read-backs follow in order. Every System is solved on its
// own and every chunk runs the kernels the whole batch would, so the cut changes nothing in the results (the tests compare
// the bits). FIKSI_AMD_HOST_CHUNKS=0 switches it off, =k sets the number of chunks.
static int solve_host_chunked(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, uint32_t n_chunks, const fx_solving_opts* sopts,
                              const fx_lm_opts* lopts, bool system_level, fx_result* results) {
    const uint32_t n = p.n_systems;
    struct Chunk {
        fx_batch b{};
        fx_dbatch* db = nullptr;
        hipStream_t solve_stream = nullptr;
        uint32_t s0 = 0;
    };
    std::vector<Chunk> chunks(n_chunks);
    hipStream_t const main_stream = ctx->stream;
    PhaseTrace tr;
    int rc = FX_OK;
    for (uint32_t k = 0; k < n_chunks && rc == FX_OK; ++k) {
        Chunk& c = chunks[k];
        const uint32_t s0 = (uint32_t)((uint64_t)n * k / n_chunks), s1 = (uint32_t)((uint64_t)n * (k + 1) / n_chunks);
        c.s0 = s0;
        c.b.n_systems = s1 - s0;
        c.b.vars = batch->vars + batch->var_off[s0];  // (all the read-back needs of the chunk's host side)
        // the copies on the second stream (never behind a solve), the solve on the context's own, after them
        ctx->stream = ctx->stream2;
        rc = upload_planned(ctx, batch, p, s0, s1, &c.db);
        c.solve_stream = (k & 1u) ? ctx->stream3 : main_stream;
        ctx->stream = c.solve_stream;
        if (rc) break;
        c.db->resident = false;
        if (hipEventRecord(ctx->ev_chunk, ctx->stream2) != hipSuccess || hipStreamWaitEvent(c.solve_stream, ctx->ev_chunk, 0) != hipSuccess) {
            rc = fail(FX_ERR_HIP, "event between the copy and the solve stream failed");
            break;
        }
        rc = system_level ? fx_system_solve_device(ctx, c.db, sopts) : fx_lm_solve_device(ctx, c.db, lopts);
    }
    tr.stamp("chunks: up + launched", n);
    if (rc != FX_OK) {  // nothing has been written to the caller's arrays yet
        (void)hipStreamSynchronize(ctx->stream2);
        for (Chunk& c : chunks)
            if (c.db) {
                ctx->stream = c.solve_stream ? c.solve_stream : main_stream;
                free_batch(ctx, c.db, /*stream_idle=*/false);
            }
        ctx->stream = main_stream;
        return rc;
    }
    for (Chunk& c : chunks) {
        ctx->stream = c.solve_stream;
        int r = read_back_and_free(ctx, c.db, &c.b, results ? results + c.s0 : nullptr, FX_OK);
        if (r && !rc) rc = r;
    }
    ctx->stream = main_stream;
    tr.stamp("chunks: read back", n);
    return rc;
}

static int solve_host(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* sopts, const fx_lm_opts* lopts,
                      bool system_level, fx_result* results) {
    int rc = bind(ctx);
    if (rc) return rc;
    fx_dbatch* db = nullptr;
    PhaseTrace tr;
    {
        HostPlan p;
        g_wide_routing = ctx->wide_routing;
        rc = analyze(batch, &p);
        if (rc) return rc;
        tr.stamp("analysis", p.n_systems);
        // Two chunks from 65 536 Systems on: measured on 100 000 ring16 sketches (tools/host_path.py), 2 chunks
        // 6.3 ms, 3 and 4 chunks 6.8 ms, 8 chunks 8.7 ms, uncut 7.6 ms — every chunk pays its own dozen copies and the slow
        // end of its own solve, so more chunks lose what the earlier start of the first solve wins
        static const int forced = [] { const char* e = std::getenv("FIKSI_AMD_HOST_CHUNKS"); return e ? atoi(e) : -1; }();
        uint32_t n_chunks = p.n_systems >= 65536u ? 2u : 0u;
        if (forced >= 0) n_chunks = std::min<uint32_t>((uint32_t)forced, p.n_systems / 2u);
        if (n_chunks >= 2 && p.n_large == 0 && (ctx->stream2 || hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) == hipSuccess) &&
            (ctx->stream3 || hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) == hipSuccess) &&
            (ctx->ev_chunk || hipEventCreateWithFlags(&ctx->ev_chunk, hipEventDisableTiming) == hipSuccess))
            return solve_host_chunked(ctx, batch, p, n_chunks, sopts, lopts, system_level, results);
        rc = upload_planned(ctx, batch, p, 0, p.n_systems, &db, /*one_shot=*/true);
    }
    if (rc) return rc;
    tr.stamp("upload", batch->n_systems);
    db->resident = false;  // solved once and freed: no point in keeping plans
    rc = system_level ? fx_system_solve_device(ctx, db, sopts) : fx_lm_solve_device(ctx, db, lopts);
    tr.stamp("solve (launches)", batch->n_systems);
    rc = read_back_and_free(ctx, db, batch, results, rc);
    tr.stamp("wait + read back", batch->n_systems);
    return rc;
}

int fx_system_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* opts, fx_result* results) {
    return solve_host(ctx, batch, opts, nullptr, true, results);
}

int fx_lm_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results) {
    return solve_host(ctx, batch, nullptr, opts, false, results);
}

// One batch over several devices (Systems are independent — contiguous shards, no data-path collective): one
// host thread per context, each solving its shard with fx_system_solve_batch on its own device and stream; the
// throughput counters are summed on the host. Shard r of n = Systems [r N / n, (r + 1) N / n) — the rule of
// fiksi_amd/workloads.py: shard, so a result never depends on how many devices took part.
int fx_system_solve_batch_multi(fx_ctx* const* ctxs, uint32_t n_ctx, const fx_batch* batch, const fx_solving_opts* opts, fx_result* results,
                                fx_throughput* total) {
    if (!ctxs || n_ctx == 0 || !batch) return fail(FX_ERR_INVALID, "bad argument");
    for (uint32_t r = 0; r < n_ctx; ++r) {
        if (!ctxs[r]) return fail(FX_ERR_INVALID, "context %u is NULL", r);
        for (uint32_t q = 0; q < r; ++q)
            if (ctxs[q] == ctxs[r]) return fail(FX_ERR_INVALID, "context %u is listed twice (a context is bound to one host thread)", r);
    }
    int rc = fx_batch_validate(batch);
    if (rc) return rc;
    const uint32_t n = batch->n_systems;
    std::vector<fx_result> local;
    if (!results) {
        local.resize(n);
        results = local.data();
    }
    // Components of 65 ... 128 columns go to the wide kernel or to the team kernels by the cost of the batch at hand
    // (analyze), and the two add in different orders: decided per shard, a result's last bits would depend on the number of
    // contexts. The choice is made ONCE, on the whole batch, and pinned for every shard. (Only batches that can hold such a
    // component — a System of more than 64 variables — pay for the extra analysis.)
    int pinned = -2;
    {
        uint32_t biggest = 0;
        for (uint32_t s = 0; s < n; ++s) biggest = std::max(biggest, batch->var_off[s + 1] - batch->var_off[s]);
        if (biggest > 64u) {
            int routing = ctxs[0]->wide_routing;
            for (uint32_t r = 1; r < n_ctx; ++r)
                if (ctxs[r]->wide_routing != routing)
                    return fail(FX_ERR_INVALID, "contexts 0 and %u differ in fx_ctx_set_wide_routing: results would depend on the shard", r);
            if (routing < 0) {
                HostPlan whole;
                g_wide_routing = -1;
                g_wide_routing_pinned = -2;
                rc = analyze(batch, &whole);
                if (rc) return rc;
                if (whole.wide_decision >= 0) pinned = whole.wide_decision;
            }
        }
    }
    std::vector<int> codes(n_ctx, FX_OK);
    std::vector<std::string> messages(n_ctx);
    std::vector<std::thread> workers;
    for (uint32_t r = 0; r < n_ctx; ++r) {
        workers.emplace_back([&, r] {
            g_wide_routing_pinned = pinned;  // (thread-local; the thread ends with the call)
            const uint32_t lo = (uint32_t)((uint64_t)n * r / n_ctx), hi = (uint32_t)((uint64_t)n * (r + 1) / n_ctx);
            if (hi == lo) return;
            const uint32_t v0 = batch->var_off[lo], e0 = batch->expr_off[lo];
            std::vector<uint32_t> var_off(hi - lo + 1), expr_off(hi - lo + 1);
            for (uint32_t s = lo; s <= hi; ++s) {
                var_off[s - lo] = batch->var_off[s] - v0;
                expr_off[s - lo] = batch->expr_off[s] - e0;
            }
            fx_batch sub = *batch;
            sub.n_systems = hi - lo;
            sub.var_off = var_off.data();
            sub.expr_off = expr_off.data();
            sub.vars = batch->vars + v0;  // solved in place: every shard owns its slice
            sub.var_fixed = batch->var_fixed + v0;
            sub.expr_tag = batch->expr_tag + e0;
            sub.expr_idx = batch->expr_idx + 4 * (size_t)e0;
            sub.expr_param = batch->expr_param + e0;
            sub.var_comp = batch->var_comp ? batch->var_comp + v0 : nullptr;
            sub.expr_comp = batch->expr_comp ? batch->expr_comp + e0 : nullptr;
            codes[r] = fx_system_solve_batch(ctxs[r], &sub, opts, results + lo);
            if (codes[r]) messages[r] = fx_last_error();  // (thread-local: carried over to the caller below)
        });
    }
    for (auto& w : workers) w.join();
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (codes[r]) return fail(codes[r], "shard %u of %u: %s", r, n_ctx, messages[r].c_str());
    if (total) {
        fx_throughput t{};
        t.systems = n;
        for (uint32_t s = 0; s < n; ++s) {
            t.converged += results[s].sse_unscaled < 1e-4 ? 1u : 0u;  // fiksi_bench.rs:65-72
            t.accepted += results[s].accepted;
            t.trials += results[s].trials;
        }
        *total = t;
    }
    return FX_OK;
}

// ---- Decomposer::RecursiveAssembly: the device work around the host plan of fx_recursive.h -------------------

int fx_system_prepare_batch(fx_ctx* ctx, const fx_batch* batch, uint32_t perturb, double* out_vars, double* out_params,
                            double* out_scale) {
    if (!batch || !out_vars || !out_params || !out_scale) return fail(FX_ERR_INVALID, "bad argument");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    db->resident = false;
    auto run = [&]() -> int {
        double *d_vars = nullptr, *d_scale = nullptr, *d_params = nullptr;
        int r = dev_alloc_copy<double>(ctx, db, &d_vars, nullptr, db->d.n_vars);
        if (r) return r;
        r = dev_alloc_copy<double>(ctx, db, &d_params, nullptr, db->d.n_exprs);
        if (r) return r;
        r = dev_alloc_copy<double>(ctx, db, &d_scale, nullptr, db->d.n_systems);
        if (r) return r;
        FX_HIP(fx::launch_prepare(db->d, 1u | (perturb ? 2u : 0u), d_vars, d_params, d_scale, ctx->stream));
        if (db->d.n_vars) FX_HIP(hipMemcpyAsync(out_vars, d_vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (db->d.n_exprs)
            FX_HIP(hipMemcpyAsync(out_params, d_params, (size_t)db->d.n_exprs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (db->d.n_systems)
            FX_HIP(hipMemcpyAsync(out_scale, d_scale, (size_t)db->d.n_systems * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        return FX_OK;
    };
    rc = run();
    fx_batch_free(ctx, db);
    return rc;
}

int fx_cluster_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results) {
    fx_lm_opts o;
    if (opts) o = *opts; else fx_lm_opts_default(&o);
    if (o.precision == 32) return fail(FX_ERR_UNSUPPORTED, "cluster problems are solved in f64");
    if (o.solver > FX_STEP_QR) return fail(FX_ERR_UNSUPPORTED, "unknown step solver %u", o.solver);
    fx_dbatch* db = nullptr;
    PhaseTrace tr;
    g_allow_pose = true;
    int rc = fx_batch_upload(ctx, batch, &db);
    g_allow_pose = false;
    if (rc) return rc;
    tr.stamp("analysis + upload", batch->n_systems);
    db->resident = false;
    db->d.has_pose = 1u;
    rc = fx_lm_solve_device(ctx, db, &o);
    tr.stamp("solve (launches)", batch->n_systems);
    rc = read_back_and_free(ctx, db, batch, results, rc);
    tr.stamp("wait + read back", batch->n_systems);
    return rc;
}

int fx_pose_transform_points(fx_ctx* ctx, const double* poses, uint32_t n_poses, const uint32_t* pose_of, const uint32_t* var_idx,
                             uint32_t n_points, double* vars, uint32_t n_vars) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_points == 0) return FX_OK;
    if (!poses || !pose_of || !var_idx || !vars) return fail(FX_ERR_INVALID, "bad argument");
    {
        std::vector<uint8_t> touched(n_vars, 0);  // points are moved in place, side by side: no variable may belong to two of them
        for (uint32_t i = 0; i < n_points; ++i) {
            if (pose_of[i] >= n_poses || (uint64_t)var_idx[i] + 1u >= n_vars) return fail(FX_ERR_INVALID, "point %u out of range", i);
            if (touched[var_idx[i]] || touched[var_idx[i] + 1u]) return fail(FX_ERR_INVALID, "point %u overlaps an earlier point", i);
            touched[var_idx[i]] = touched[var_idx[i] + 1u] = 1;
        }
    }
    fx_dbatch scratch;  // owns the device blocks of this call
    double *d_poses = nullptr, *d_vars = nullptr;
    uint32_t *d_of = nullptr, *d_idx = nullptr;
    auto run = [&]() -> int {
        int r = dev_alloc_copy<double>(ctx, &scratch, &d_poses, poses, 3 * (size_t)n_poses);
        if (!r) r = dev_alloc_copy<double>(ctx, &scratch, &d_vars, vars, n_vars);
        if (!r) r = dev_alloc_copy<uint32_t>(ctx, &scratch, &d_of, pose_of, n_points);
        if (!r) r = dev_alloc_copy<uint32_t>(ctx, &scratch, &d_idx, var_idx, n_points);
        if (r) return r;
        FX_HIP(fx::launch_pose_transform(d_poses, d_of, d_idx, n_points, d_vars, ctx->stream));
        FX_HIP(hipMemcpyAsync(vars, d_vars, (size_t)n_vars * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        return FX_OK;
    };
    rc = run();
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& blk : scratch.allocations) ctx->give_back(blk.p, blk.size);
    return rc;
}

int fx_unscale_vars(fx_ctx* ctx, double scale, const double* scaled, const uint8_t* mask, double* vars, uint32_t n) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n == 0) return FX_OK;
    if (!scaled || !mask || !vars) return fail(FX_ERR_INVALID, "bad argument");
    fx_dbatch scratch;
    double *d_scaled = nullptr, *d_vars = nullptr;
    uint8_t* d_mask = nullptr;
    auto run = [&]() -> int {
        int r = dev_alloc_copy<double>(ctx, &scratch, &d_scaled, scaled, n);
        if (!r) r = dev_alloc_copy<double>(ctx, &scratch, &d_vars, vars, n);
        if (!r) r = dev_alloc_copy<uint8_t>(ctx, &scratch, &d_mask, mask, n);
        if (r) return r;
        FX_HIP(fx::launch_unscale(scale, d_scaled, d_mask, d_vars, n, ctx->stream));
        FX_HIP(hipMemcpyAsync(vars, d_vars, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        return FX_OK;
    };
    rc = run();
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& blk : scratch.allocations) ctx->give_back(blk.p, blk.size);
    return rc;
}

int fx_unscale_vars_strided(fx_ctx* ctx, const double* scales, uint32_t n_systems, uint32_t nvars, const double* scaled, const uint8_t* mask,
                            double* vars) {
    int rc = bind(ctx);
    if (rc) return rc;
    const uint64_t n = (uint64_t)n_systems * nvars;
    if (n == 0) return FX_OK;
    if (!scales || !scaled || !mask || !vars) return fail(FX_ERR_INVALID, "bad argument");
    fx_dbatch scratch;
    double *d_scaled = nullptr, *d_vars = nullptr, *d_scales = nullptr;
    uint8_t* d_mask = nullptr;
    auto run = [&]() -> int {
        int r = dev_alloc_copy<double>(ctx, &scratch, &d_scaled, scaled, n);
        if (!r) r = dev_alloc_copy<double>(ctx, &scratch, &d_vars, vars, n);
        if (!r) r = dev_alloc_copy<double>(ctx, &scratch, &d_scales, scales, n_systems);
        if (!r) r = dev_alloc_copy<uint8_t>(ctx, &scratch, &d_mask, mask, nvars);
        if (r) return r;
        FX_HIP(fx::launch_unscale_strided(d_scales, n_systems, nvars, d_scaled, d_mask, d_vars, ctx->stream));
        FX_HIP(hipMemcpyAsync(vars, d_vars, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        return FX_OK;
    };
    rc = run();
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& blk : scratch.allocations) ctx->give_back(blk.p, blk.size);
    return rc;
}

int fx_eval_residual_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jvals) {
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    rc = jvals ? fx_eval_residual_jacobian_device(ctx, db, 0) : fx_eval_residual_device(ctx, db, 0);
    if (!rc && r && db->d.n_exprs) rc = fx_batch_get_residuals(ctx, db, r);
    if (!rc && jvals && db->d.nnz) rc = fx_batch_get_jacobian_values(ctx, db, jvals);
    fx_batch_free(ctx, db);
    return rc;
}

int fx_analyze_batch(fx_ctx* ctx, const fx_batch* batch, uint8_t* dependent) {
    if (!dependent) return fail(FX_ERR_INVALID, "dependent is NULL");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    const uint32_t ne = db->d.n_exprs;
    if (fx::analyze_lds_bytes(db->d.max_vars_all, db->d.max_exprs_all) > 150u * 1024u) {
        fx_batch_free(ctx, db);
        return fail(FX_ERR_TOO_LARGE, "analyze keeps the dense expressions x variables Jacobian of a System in LDS (limit 150 KB)");
    }
    uint8_t* d_dep = nullptr;
    hipError_t e = hipMalloc((void**)&d_dep, std::max<uint32_t>(ne, 1));
    if (e == hipSuccess) e = fx::launch_analyze(db->d, db->d.vars0, db->d.max_vars_all, db->d.max_exprs_all, d_dep, ctx->stream);
    if (e == hipSuccess && ne) e = hipMemcpyAsync(dependent, d_dep, ne, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_dep) (void)hipFree(d_dep);
    fx_batch_free(ctx, db);
    if (e != hipSuccess) return fail(FX_ERR_HIP, "analyze failed: %s", hipGetErrorString(e));
    return FX_OK;
}

int fx_single_pass_blocks(const fx_batch* batch, uint32_t system, uint32_t* n_blocks, uint32_t* block_comp,
                          uint32_t* row_off, uint32_t* rows, uint32_t* var_off, uint32_t* vars) {
    int rc = analyze(batch, nullptr);
    if (rc) return rc;
    if (system >= batch->n_systems) return fail(FX_ERR_INVALID, "system %u out of range (%u systems)", system, batch->n_systems);
    const uint32_t v0 = batch->var_off[system], nvt = batch->var_off[system + 1] - v0;
    const uint32_t e0 = batch->expr_off[system], net = batch->expr_off[system + 1] - e0;
    fx::Incidence inc;
    inc.build(nvt, net, batch->expr_tag + e0, batch->expr_idx + 4 * (size_t)e0);
    fx::SinglePassDecomposer dec(inc);
    uint32_t ncomp = 0;
    for (uint32_t i = 0; i < nvt; ++i) {
        uint16_t c = batch->var_comp ? batch->var_comp[v0 + i] : 0;
        if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
    }
    uint32_t nb = 0, nr = 0, nv = 0;
    if (row_off) row_off[0] = 0;
    if (var_off) var_off[0] = 0;
    std::vector<uint32_t> free_sorted;
    fx::UnitList units;
    for (uint32_t c = 0; c < ncomp; ++c) {
        free_sorted.clear();
        for (uint32_t i = 0; i < nvt; ++i)
            if ((batch->var_comp ? batch->var_comp[v0 + i] : 0) == c && !batch->var_fixed[v0 + i]) free_sorted.push_back(i);
        dec.run(free_sorted, units);
        if (nb + units.count() > 4u * net || nr + units.rows.size() > 4u * (size_t)net || nv + units.vars.size() > nvt)
            return fail(FX_ERR_INVALID, "system %u: decomposition exceeds the documented capacities", system);
        for (uint32_t u = 0; u < units.count(); ++u) {
            for (uint32_t k = units.row_off[u]; k < units.row_off[u + 1]; ++k, ++nr)
                if (rows) rows[nr] = units.rows[k];
            for (uint32_t k = units.var_off[u]; k < units.var_off[u + 1]; ++k, ++nv)
                if (vars) vars[nv] = units.vars[k];
            if (block_comp) block_comp[nb] = c;
            ++nb;
            if (row_off) row_off[nb] = nr;
            if (var_off) var_off[nb] = nv;
        }
    }
    if (n_blocks) *n_blocks = nb;
    return FX_OK;
}

void fx_atan2_cr_batch(uint64_t n, const double* y, const double* x, double* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = fx::atan2_cr(y[i], x[i]);
}

int fx_qr_symbolic(int32_t nrows, int32_t ncols, const int32_t* colptr, const int32_t* rowidx, int use_colamd,
                   int32_t* col_perm, int32_t* row_perm, int32_t* h_ptr, int32_t* h_rows, int32_t h_cap, int32_t* r_ptr,
                   int32_t* r_rows, int32_t r_cap) {
    if (nrows < 0 || ncols < 0 || !colptr) return fail(FX_ERR_INVALID, "bad argument");
    if (colptr[0] != 0) return fail(FX_ERR_INVALID, "colptr[0] must be 0");
    for (int32_t j = 0; j < ncols; ++j)
        if (colptr[j + 1] < colptr[j]) return fail(FX_ERR_INVALID, "colptr must not decrease (column %d)", j);
    if (colptr[ncols] > 0 && !rowidx) return fail(FX_ERR_INVALID, "rowidx is NULL");
    for (int32_t j = 0; j < ncols; ++j)
        for (int32_t p = colptr[j]; p < colptr[j + 1]; ++p) {
            if (rowidx[p] < 0 || rowidx[p] >= nrows) return fail(FX_ERR_INVALID, "column %d: row %d outside 0 .. %d", j, rowidx[p], nrows - 1);
            if (p > colptr[j] && rowidx[p] <= rowidx[p - 1]) return fail(FX_ERR_INVALID, "column %d: rows must ascend strictly", j);
        }
    fx::qr::Csc a;
    a.nrows = nrows;
    a.ncols = ncols;
    a.ptr.assign(colptr, colptr + ncols + 1);
    if (colptr[ncols] > 0) a.idx.assign(rowidx, rowidx + colptr[ncols]);
    fx::qr::Symbolic sy;
    if (!fx::qr::analyze(a, use_colamd != 0, sy)) return fail(FX_ERR_INVALID, "malformed or structurally rank-deficient pattern");
    if ((h_rows && (int64_t)sy.hrows.size() > h_cap) || (r_rows && (int64_t)sy.rrows.size() > r_cap))
        return fail(FX_ERR_INVALID, "output capacity too small (%zu / %zu entries needed)", sy.hrows.size(), sy.rrows.size());
    if (col_perm) std::copy(sy.col_perm.begin(), sy.col_perm.end(), col_perm);
    if (row_perm) std::copy(sy.row_perm.begin(), sy.row_perm.end(), row_perm);
    if (h_ptr) std::copy(sy.hptr.begin(), sy.hptr.end(), h_ptr);
    if (h_rows) std::copy(sy.hrows.begin(), sy.hrows.end(), h_rows);
    if (r_ptr) std::copy(sy.rptr.begin(), sy.rptr.end(), r_ptr);
    if (r_rows) std::copy(sy.rrows.begin(), sy.rrows.end(), r_rows);
    return FX_OK;
}

int fx_eval_residual_dense_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jac, uint64_t* jac_off,
                                    uint64_t* total) {
    int rc = analyze(batch, nullptr);
    if (rc) return rc;
    const uint32_t n = batch->n_systems;
    const uint32_t nv = n ? batch->var_off[n] : 0, ne = n ? batch->expr_off[n] : 0;
    // free rank per variable (as in fx_jacobian_structure), sizes and offsets of the dense blocks
    std::vector<uint16_t> var_rank(nv, 0xFFFFu), sys_nfree(n, 0);
    std::vector<uint32_t> expr_sys(ne, 0);
    std::vector<uint64_t> off((size_t)n + 1, 0);
    for (uint32_t s = 0; s < n; ++s) {
        uint32_t rank = 0;
        for (uint32_t i = batch->var_off[s]; i < batch->var_off[s + 1]; ++i) {
            const uint16_t c = batch->var_comp ? batch->var_comp[i] : 0;
            if (c != FX_NO_COMPONENT && !batch->var_fixed[i]) var_rank[i] = (uint16_t)rank++;
        }
        sys_nfree[s] = (uint16_t)rank;
        for (uint32_t e = batch->expr_off[s]; e < batch->expr_off[s + 1]; ++e) expr_sys[e] = s;
        off[s + 1] = off[s] + (uint64_t)(batch->expr_off[s + 1] - batch->expr_off[s]) * rank;
    }
    if (total) *total = off[n];
    if (jac_off) std::copy(off.begin(), off.end(), jac_off);
    if (!jac && !r) return FX_OK;  // size query
    if (!jac) return fail(FX_ERR_INVALID, "jac is NULL");
    fx_dbatch* db = nullptr;
    rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    uint16_t *d_rank = nullptr, *d_nfree = nullptr;
    uint32_t* d_sys = nullptr;
    uint64_t* d_off = nullptr;
    double *d_jac = nullptr, *d_r = nullptr;
    rc = dev_alloc_copy(ctx, db, &d_rank, var_rank.data(), var_rank.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d_nfree, sys_nfree.data(), sys_nfree.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d_sys, expr_sys.data(), expr_sys.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d_off, off.data(), off.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d_jac, (const double*)nullptr, (size_t)off[n]);
    if (!rc) rc = dev_alloc_copy(ctx, db, &d_r, (const double*)nullptr, ne);
    if (!rc) {
        hipError_t e = fx::launch_dense_jacobian(db->d, db->d.vars0, d_rank, d_sys, d_nfree, d_off, d_r, d_jac, ctx->stream);
        if (e == hipSuccess && off[n])
            e = hipMemcpyAsync(jac, d_jac, (size_t)off[n] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && r && ne) e = hipMemcpyAsync(r, d_r, (size_t)ne * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(FX_ERR_HIP, "dense Jacobian evaluation failed: %s", hipGetErrorString(e));
    }
    fx_batch_free(ctx, db);
    return rc;
}

int fx_constraint_residuals(fx_ctx* ctx, const fx_batch* batch, double* r) {
    if (!r) return fail(FX_ERR_INVALID, "r is NULL");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    rc = ensure_resid(ctx, db);
    hipError_t e = rc ? hipSuccess : fx::launch_identity_residuals(db->d, db->d.vars0, db->d.resid, ctx->stream);
    if (e != hipSuccess) rc = fail(FX_ERR_HIP, "launch failed: %s", hipGetErrorString(e));
    if (!rc && db->d.n_exprs) rc = fx_batch_get_residuals(ctx, db, r);
    fx_batch_free(ctx, db);
    return rc;
}

}  // extern "C"
