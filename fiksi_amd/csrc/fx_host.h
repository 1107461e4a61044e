// Host side of libfiksi_amd.so, shared declarations: the analysis of a batch (fx_analyze.cpp), the table programs of the
// one-structure kernels (fx_programs.cpp), device residency (fx_upload.cpp), routing and launches (fx_solve.cpp) and the
// guarded extern "C" entry points (fx_entry.cpp). Host logic only: every numeric result comes from the HIP kernels.
#pragma once
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
#include "fx_guard.h"
#include "fx_qrplan.h"
#include "fx_sparse.h"

namespace fxh {

using fx::fail;

// set by fx_cluster_solve_batch while it uploads: the two pose-row tags of fx_expr.h are legal in that batch only
extern thread_local bool g_allow_pose;
// fx_ctx_set_wide_routing of the context the running call belongs to (-1 by cost, 0 team kernels, 1 wide kernel)
extern thread_local int g_wide_routing;
// fx_system_solve_batch_multi: the choice made once on the whole batch, for every shard (-2: none)
extern thread_local int g_wide_routing_pinned;
// the running host-buffer call was told that its batch is of one structure (fx_ctx_set_batch_hints): analyze takes the first System's
// word for the others — the caller of analyze verifies (verify_one_structure) before anything reaches the user's arrays
extern thread_local bool g_hint_one_structure;

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

#define FX_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return ::fx::fail(FX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
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
    bool hinted = false;   // ... on the caller's word (fx_ctx_set_batch_hints): only System 0 was analysed, verify_one_structure is still owed
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
// work is worth it (the per-expression analysis is ~0.1 us; a thread costs ~50 us to start). A range whose thread cannot
// be started runs on the caller; what a range throws is re-thrown here once every thread has been joined (fx_guard.h).
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
    fx::run_workers(nt, [&](uint32_t t) {
        const uint32_t lo = (uint32_t)((uint64_t)n * t / nt), hi = (uint32_t)((uint64_t)n * (t + 1) / nt);
        fn(t, lo, hi);
    });
}
constexpr uint32_t MAX_RANGES = 16;

// ---- fx_analyze.cpp
void build_csr(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info, const uint8_t* expr_tag,
               const uint16_t* expr_idx16, CsrPlan& out);
void build_eval_plan(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info, const uint8_t* expr_tagx,
                     const uint16_t* expr_idx16, EvalPlan& out);
int analyze(const fx_batch* b, HostPlan* plan);
bool verify_one_structure(const fx_batch* b);  // every System's raw structure arrays equal the first System's

}  // namespace fxh

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
    uint32_t sparse_fronts = 1u, sparse_front_ranks = 0u;  // fx_ctx_set_sparse_fronts
    uint32_t batch_hints = 0u;             // fx_ctx_set_batch_hints
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
        plan_cache.reserve(plan_cache.size() + 1);  // (so that a new plan cannot be lost between its making and the list)
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
        p.sparse_fronts = sparse_fronts;
        p.sparse_front_ranks = sparse_front_ranks;
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
    void give_back(void* p, size_t bytes) noexcept {
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
        try {
            free_blocks.push_back({p, bytes});
            free_bytes += bytes;
        } catch (...) {  // (no room for the list entry: the block goes back to the driver — this runs inside destructors)
            (void)hipFree(p);
        }
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
    std::vector<uint32_t> class_first;  // the first System of each class (its structure stands for the class)
    // FX_STEP_QR on such a batch: the grouped QR build's program per class (ensure_qr_plans; prog == null: the class's structure does
    // not qualify), and sys_large with the members of the classes that have one marked — the one-wavefront QR kernel, launched over
    // the whole batch for everybody else, passes them by
    struct QrClassProg {
        uint32_t* prog = nullptr;
        uint32_t words = 0, small_words = 0, ng = 0, nx = 0, n = 0, m = 0;
    };
    std::vector<QrClassProg> qr_class;
    uint8_t* qr_skip = nullptr;
    // host copy of the batch, kept only when some System needs the sparse path
    std::vector<uint32_t> h_var_off, h_expr_off, h_expr_idx;
    std::vector<double> h_vars, h_expr_param;
    std::vector<uint8_t> h_var_fixed, h_expr_tag, h_sys_large;
    std::vector<uint8_t> h_units_on_device;  // SinglePass: large Systems the GLOBAL kernel instantiation walks
    std::vector<uint8_t> h_qr_wide;          // FX_STEP_QR: Systems beyond one wavefront the wide kernel's QR build solves (ensure_qr_plans)
    bool qr_wide_active = false;             // ... and they have just been solved that way: the sparse path leaves them alone
    uint32_t* tiny_left = nullptr;           // [n_systems] the Systems the tiny build hands over to the 16-column build (launch_solve_scheduled)
    bool in_place = false;                   // d.vars_in / param_in are set: vars0 / expr_param are filled by the solve kernel itself (no scout pass may read them first)
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

namespace fxh {

// A batch that is being put together: freed (blocks back to the context's cache) unless release()d — whatever ends the
// upload early, an error code or an exception on its way to the entry point's guard.
void free_batch(fx_ctx* ctx, fx_dbatch* db, bool stream_idle);
struct BatchHolder {
    fx_ctx* ctx;
    fx_dbatch* db;
    BatchHolder(fx_ctx* c, fx_dbatch* d) : ctx(c), db(d) {}
    BatchHolder(const BatchHolder&) = delete;
    BatchHolder& operator=(const BatchHolder&) = delete;
    ~BatchHolder() {
        if (db) free_batch(ctx, db, false);
    }
    fx_dbatch* release() {
        fx_dbatch* d = db;
        db = nullptr;
        return d;
    }
};

template <typename T>
int dev_alloc_copy(fx_ctx* ctx, fx_dbatch* db, T** out, const T* host, size_t count) {
    *out = nullptr;
    size_t bytes = ((std::max<size_t>(count, 1) * sizeof(T)) + 255u) & ~size_t(255);
    hipError_t e = hipSuccess;
    db->allocations.reserve(db->allocations.size() + 1);  // (so that the block cannot be lost between take() and the list)
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

// ---- fx_upload.cpp: device residency
int bind(fx_ctx* ctx);
int ensure_resid(fx_ctx* ctx, fx_dbatch* db);
int ensure_csr(fx_ctx* ctx, fx_dbatch* db);
int ensure_units(fx_ctx* ctx, fx_dbatch* db);
int ensure_qr_plans(fx_ctx* ctx, fx_dbatch* db, bool units);
int ensure_component_walk(fx_ctx* ctx, fx_dbatch* db);
constexpr uint32_t FX_DEFER_VARS = 1u, FX_DEFER_PARAMS = 2u;  // upload_planned: room for vars0 / vars (expr_param), no values yet
int upload_planned(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, uint32_t s0, uint32_t s1, fx_dbatch** out, bool one_shot = false,
                   uint32_t defer = 0);
int fill_deferred(fx_ctx* ctx, fx_dbatch* db, const fx_batch* batch, uint32_t defer);
bool takes_one_structure_build(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* sopts, const fx_lm_opts* lopts, bool system_level);
// fx_host_register's ranges: the device-visible address of [p, p + bytes) when the whole range lies inside one of them, else null
void* registered_range(const void* p, size_t bytes);
int read_back_and_free(fx_ctx* ctx, fx_dbatch* db, const fx_batch* batch, fx_result* results, int rc);

// ---- fx_programs.cpp: the table programs of the kernels for batches of one structure, and the QR plans
struct QrHostPlan {
    uint32_t n = 0, m = 0;
    std::vector<uint16_t> u16;  // colperm[n], rowperm[m + n], hptr[n + 1], hrows[nnzh]
    std::vector<uint64_t> u64;  // colmask[n], rowmask[n]
    uint32_t nnzh = 0;
    bool ok = false;
};
struct GcHostProgram {
    std::vector<uint32_t> words;
    uint32_t nslots = 0, ng = 0, nc = 0, rc = 0, words_f64 = 0;  // (words_f64: the part the f64 builds use)
};
struct GsHostProgram {
    std::vector<uint32_t> words;
    uint32_t nl = 0, ng = 0, nvt = 0, net = 0, nfree = 0;
};
struct QrgHostProgram {
    std::vector<uint32_t> words;
    uint32_t n = 0, m = 0, nx = 0, ng = 0;
    bool ok = false;
};
bool build_qr_plan(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_, uint32_t n,
                   uint32_t nvt, QrHostPlan& out);
bool build_gc_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16, uint32_t nvt,
                      uint32_t net, uint32_t max_free, GcHostProgram& out);
bool build_gs_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16, uint32_t nvt,
                      uint32_t net, GsHostProgram& out);
bool build_qrg_program(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_, uint32_t n,
                       uint32_t nvt, QrgHostProgram& out, bool wide = false);

// ---- fx_solve.cpp: routing and launches
int launch_solve_scheduled(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p);
bool wide_kernel_applies(const fx::LmParams& p);
int solve_large_systems(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p);
int solve_beyond_one_wavefront(fx_ctx* ctx, fx_dbatch* db, fx::LmParams p);

int solve_host(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* sopts, const fx_lm_opts* lopts, bool system_level, fx_result* results,
               bool no_hint = false);

}  // namespace fxh
