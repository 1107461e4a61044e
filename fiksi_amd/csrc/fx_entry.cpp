// C ABI of libfiksi_amd.so (include/fiksi_amd.h): the extern "C" entry points. Every one of them is a function-try-block
// (fx_guard.h): nothing unwinds across the boundary — std::bad_alloc becomes FX_ERR_NOMEM, anything else FX_ERR_INTERNAL,
// with the text in fx_last_error. Host logic only; every numeric result comes from the HIP kernels. There is
// deliberately no CPU compute path in this library.
#include "fx_host.h"

using namespace fxh;

#include <mutex>
#include <stdexcept>
#include <system_error>

namespace fxh {
thread_local bool g_allow_pose = false;
thread_local int g_wide_routing = -1;
thread_local int g_wide_routing_pinned = -2;
thread_local bool g_hint_one_structure = false;
}  // namespace fxh

namespace fx {

namespace {
thread_local char g_last_error[LAST_ERROR_LEN] = {0};
}

char* last_error_buffer() noexcept { return g_last_error; }

void set_last_error(const char* msg) noexcept {
    snprintf(g_last_error, LAST_ERROR_LEN, "%s", msg ? msg : "");
}

int fail(int code, const char* fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, LAST_ERROR_LEN, fmt, ap);
    va_end(ap);
    return code;
}

int translate_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(FX_ERR_NOMEM, "out of host memory");
    } catch (const std::length_error& e) {  // a container asked for more than max_size(): the same thing to the caller
        return fail(FX_ERR_NOMEM, "out of host memory (%s)", e.what());
    } catch (const std::system_error& e) {
        return fail(FX_ERR_INTERNAL, "internal error: %s", e.what());
    } catch (const std::exception& e) {
        return fail(FX_ERR_INTERNAL, "internal error: %s", e.what());
    } catch (...) {
        return fail(FX_ERR_INTERNAL, "internal error: unknown exception");
    }
}

}  // namespace fx

extern "C" {

int fx_abi_version(void) { return FX_ABI_VERSION; }

const char* fx_last_error(void) { return fx::last_error_buffer(); }

int fx_device_count(int* count) try {
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
FX_CATCH_CODE

int fx_ctx_create(fx_ctx** out, int device) try {
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
FX_CATCH_CODE

void fx_ctx_destroy(fx_ctx* ctx) try {
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
FX_CATCH_VOID

int fx_ctx_set_routing(fx_ctx* ctx, int grouped, uint32_t grouped_min_systems) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (grouped < -1 || grouped > 1) return fail(FX_ERR_INVALID, "grouped must be -1 (by batch size), 0 or 1");
    ctx->route_grouped = grouped;
    if (grouped_min_systems) ctx->grouped_min_systems = grouped_min_systems;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_one_structure_builds(fx_ctx* ctx, int enable) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->grouped_one_structure = enable == 2 ? 2 : enable ? 1 : 0;  // (2: on, without the tiny build of fx_grouped_tiny.hip)
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_hold_passes(fx_ctx* ctx, uint32_t passes) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->hold_passes = passes;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_ladder(fx_ctx* ctx, int enable, uint32_t tail_systems, uint32_t min_trials, int spread) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->ladder = enable ? 1u : 0u;
    ctx->ladder_tail = tail_systems;
    ctx->ladder_k = min_trials;
    ctx->ladder_spread = spread ? 1u : 0u;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_wide_routing(fx_ctx* ctx, int wide) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (wide < -1 || wide > 1) return fail(FX_ERR_INVALID, "wide must be -1 (by cost), 0 (team kernels) or 1 (wide kernel)");
    ctx->wide_routing = wide;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_sparse_fronts(fx_ctx* ctx, int enable, uint32_t ranks) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (ranks > 8u) return fail(FX_ERR_INVALID, "ranks must be 0 (by the room on the chip) ... 8");
    ctx->sparse_fronts = enable ? 1u : 0u;
    ctx->sparse_front_ranks = ranks;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_batch_hints(fx_ctx* ctx, uint32_t hints) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    if (hints & ~FX_HINT_ONE_STRUCTURE) return fail(FX_ERR_INVALID, "unknown hint bits 0x%x", hints & ~FX_HINT_ONE_STRUCTURE);
    ctx->batch_hints = hints;
    return FX_OK;
}
FX_CATCH_CODE

// The ranges fx_host_register has page-locked (process-wide, as the registration is), with their device-visible addresses: a
// host-buffer call whose arrays lie inside them lets the solve kernel read and write them in place (fx_solve.cpp: solve_host).
namespace {
struct HostRange {
    const unsigned char* host;
    size_t bytes;
    unsigned char* dev;
};
std::mutex g_ranges_lock;
std::vector<HostRange> g_ranges;
}  // namespace

extern "C++" {
namespace fxh {
void* registered_range(const void* p, size_t bytes) {
    if (!p || !bytes) return nullptr;
    const unsigned char* q = static_cast<const unsigned char*>(p);
    std::lock_guard<std::mutex> hold(g_ranges_lock);
    for (const HostRange& r : g_ranges)
        if (q >= r.host && bytes <= r.bytes && (size_t)(q - r.host) <= r.bytes - bytes) return r.dev + (q - r.host);
    return nullptr;
}
}  // namespace fxh
}  // extern "C++"

int fx_host_register(fx_ctx* ctx, void* ptr, size_t bytes) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ptr || !bytes) return fail(FX_ERR_INVALID, "nothing to register");
    {
        std::lock_guard<std::mutex> hold(g_ranges_lock);
        g_ranges.reserve(g_ranges.size() + 1);  // (before the runtime pins anything: the bookkeeping below cannot fail)
    }
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? FX_ERR_NOMEM : FX_ERR_HIP, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) {  // (page-locked all the same: the copies run at the link's rate)
        (void)hipGetLastError();
        dev = nullptr;
    }
    if (dev) {
        std::lock_guard<std::mutex> hold(g_ranges_lock);
        g_ranges.push_back({static_cast<const unsigned char*>(ptr), bytes, static_cast<unsigned char*>(dev)});
    }
    return FX_OK;
}
FX_CATCH_CODE

int fx_host_unregister(fx_ctx* ctx, void* ptr) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ptr) return fail(FX_ERR_INVALID, "ptr is NULL");
    {
        std::lock_guard<std::mutex> hold(g_ranges_lock);
        for (size_t i = 0; i < g_ranges.size(); ++i)
            if (g_ranges[i].host == ptr) {
                g_ranges.erase(g_ranges.begin() + (long)i);
                break;
            }
    }
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(FX_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(e));
    }
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_host_threads(fx_ctx* ctx, uint32_t threads) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->host_threads = threads ? std::min(threads, 64u) : 8u;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_set_presort(fx_ctx* ctx, int enable, uint32_t min_systems) try {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    ctx->presort = enable ? 1 : 0;
    if (min_systems) ctx->presort_min_systems = min_systems;
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_synchronize(fx_ctx* ctx) try {
    int rc = bind(ctx);
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_ctx_device_name(fx_ctx* ctx, char* buf, size_t len) try {
    if (!ctx || !buf || len == 0) return fail(FX_ERR_INVALID, "bad argument");
    snprintf(buf, len, "%s (%s)", ctx->name, ctx->arch);
    return FX_OK;
}
FX_CATCH_CODE

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

int fx_batch_validate(const fx_batch* batch) try { return analyze(batch, nullptr); } FX_CATCH_CODE

int fx_jacobian_structure(const fx_batch* batch, uint64_t* nnz, uint32_t* row_ptr, uint32_t* col_idx) try {
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
FX_CATCH_CODE


int fx_batch_upload(fx_ctx* ctx, const fx_batch* batch, fx_dbatch** out) try {
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
FX_CATCH_CODE


void fx_batch_free(fx_ctx* ctx, fx_dbatch* db) try { free_batch(ctx, db, false); } FX_CATCH_VOID

int fx_batch_set_vars(fx_ctx* ctx, fx_dbatch* db, const double* vars) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !vars) return fail(FX_ERR_INVALID, "bad argument");
    if (db->n_large) std::copy(vars, vars + db->d.n_vars, db->h_vars.begin());
    FX_HIP(hipMemcpyAsync(db->d.vars0, vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipMemcpyAsync(db->d.vars, vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_batch_set_params(fx_ctx* ctx, fx_dbatch* db, const double* expr_param) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !expr_param) return fail(FX_ERR_INVALID, "bad argument");
    if (db->n_large) std::copy(expr_param, expr_param + db->d.n_exprs, db->h_expr_param.begin());
    FX_HIP(hipMemcpyAsync(db->d.expr_param, expr_param, (size_t)db->d.n_exprs * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_batch_schedule_by_last_solve(fx_ctx* ctx, fx_dbatch* db, int enable) try {
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
FX_CATCH_CODE

int fx_batch_get_vars(fx_ctx* ctx, fx_dbatch* db, double* vars) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !vars) return fail(FX_ERR_INVALID, "bad argument");
    FX_HIP(hipMemcpyAsync(vars, db->d.vars, (size_t)db->d.n_vars * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_batch_get_results(fx_ctx* ctx, fx_dbatch* db, fx_result* results) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !results) return fail(FX_ERR_INVALID, "bad argument");
    FX_HIP(hipMemcpyAsync(results, db->d.results, (size_t)db->d.n_systems * sizeof(fx_result), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

uint64_t fx_batch_nnz(const fx_dbatch* db) { return db ? db->d.nnz : 0; }


int fx_system_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts) try {
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
FX_CATCH_CODE

extern "C++" {
namespace fxh {
// Whether the solve the two entry points above / below would launch for these options is ONE launch of the grouped kernel's
// one-structure build over the whole batch (fx_grouped_c.hip) — the build that can read and write the caller's arrays in place.
bool takes_one_structure_build(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* sopts, const fx_lm_opts* lopts, bool system_level) {
    fx::LmParams p;
    ctx->route(p);
    if (system_level) {
        fx_solving_opts o;
        if (sopts) o = *sopts; else fx_solving_opts_default(&o);
        if (o.optimizer != 0 || o.decomposer != 0) return false;
        p.lm = o.lm;
        p.mode = 1u | (o.perturb ? 2u : 0u);
    } else {
        if (lopts) p.lm = *lopts; else fx_lm_opts_default(&p.lm);
        p.mode = 0;
    }
    const fx::DeviceBatch& d = db->d;
    if (p.lm.solver != FX_STEP_CHOLESKY || !d.uniform || db->n_large || !db->classes.empty()) return false;
    return !fx::grouped_s_applies(d, p) && fx::grouped_applies(d, p) && fx::grouped_c_applies(d, p);
}
}  // namespace fxh
}  // extern "C++"

int fx_lm_solve_device(fx_ctx* ctx, fx_dbatch* db, const fx_lm_opts* opts) try {
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
FX_CATCH_CODE

int fx_debug_solve_route(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* route) try {
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
FX_CATCH_CODE

int fx_debug_grouped_build(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, int* build) try {
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
    if (*build == 1 && fx::grouped_tiny_applies(db->d, p)) *build = 4;
    return FX_OK;
}
FX_CATCH_CODE

// Diagnostic (not part of the drop-in surface): runs the stamped build of the fused kernel once and
// returns the shader cycles summed over all wavefronts for {setup, eval, form, factor, solve, tail}.
int fx_debug_phase_cycles(fx_ctx* ctx, fx_dbatch* db, const fx_solving_opts* opts, uint64_t cycles[6]) try {
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
FX_CATCH_CODE

int fx_eval_residual_jacobian_device(fx_ctx* ctx, fx_dbatch* db, int which) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    rc = ensure_csr(ctx, db);
    if (rc) return rc;
    FX_HIP(fx::launch_eval(db->d, which ? db->d.vars : db->d.vars0, true, ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_eval_residual_device(fx_ctx* ctx, fx_dbatch* db, int which) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db) return fail(FX_ERR_INVALID, "batch is NULL");
    rc = ensure_resid(ctx, db);
    if (rc) return rc;
    FX_HIP(fx::launch_eval(db->d, which ? db->d.vars : db->d.vars0, false, ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_batch_get_residuals(fx_ctx* ctx, fx_dbatch* db, double* r) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !r) return fail(FX_ERR_INVALID, "bad argument");
    if (!db->d.resid) return fail(FX_ERR_INVALID, "no residuals have been evaluated on this batch yet");
    FX_HIP(hipMemcpyAsync(r, db->d.resid, (size_t)db->d.n_exprs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_batch_get_jacobian_values(fx_ctx* ctx, fx_dbatch* db, double* jvals) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!db || !jvals) return fail(FX_ERR_INVALID, "bad argument");
    if (!db->d.jvals) return fail(FX_ERR_INVALID, "no Jacobian has been evaluated on this batch yet");
    FX_HIP(hipMemcpyAsync(jvals, db->d.jvals, (size_t)db->d.nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_timer_begin(fx_ctx* ctx) try {
    int rc = bind(ctx);
    if (rc) return rc;
    FX_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    return FX_OK;
}
FX_CATCH_CODE

int fx_timer_end(fx_ctx* ctx, float* milliseconds) try {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!milliseconds) return fail(FX_ERR_INVALID, "milliseconds is NULL");
    FX_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    FX_HIP(hipEventSynchronize(ctx->ev_end));
    FX_HIP(hipEventElapsedTime(milliseconds, ctx->ev_begin, ctx->ev_end));
    return FX_OK;
}
FX_CATCH_CODE


int fx_system_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* opts, fx_result* results) try {
    return solve_host(ctx, batch, opts, nullptr, true, results);
}
FX_CATCH_CODE

int fx_lm_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results) try {
    return solve_host(ctx, batch, nullptr, opts, false, results);
}
FX_CATCH_CODE

// One batch over several devices (Systems are independent — contiguous shards, no data-path collective): one
// host thread per context, each solving its shard with fx_system_solve_batch on its own device and stream; the
// throughput counters are summed on the host. Shard r of n = Systems [r N / n, (r + 1) N / n) — the rule of
// fiksi_amd/workloads.py: shard, so a result never depends on how many devices took part.
int fx_system_solve_batch_multi(fx_ctx* const* ctxs, uint32_t n_ctx, const fx_batch* batch, const fx_solving_opts* opts, fx_result* results,
                                fx_throughput* total) try {
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
    struct ShardStatus {
        int code = FX_OK;
        char msg[256] = {0};
    };
    std::vector<ShardStatus> status(n_ctx);
    const int pinned_before = g_wide_routing_pinned;
    fx::run_workers(n_ctx, [&](uint32_t r) {  // (shard 0 on the calling thread; a thread that cannot be had: its shard after that)
        ShardStatus& st = status[r];
        try {
            g_wide_routing_pinned = pinned;  // (thread-local)
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
            st.code = fx_system_solve_batch(ctxs[r], &sub, opts, results + lo);
        } catch (...) {
            st.code = fx::translate_exception();
        }
        if (st.code) snprintf(st.msg, sizeof(st.msg), "%s", fx_last_error());  // (thread-local: carried over to the caller below)
    });
    g_wide_routing_pinned = pinned_before;
    for (uint32_t r = 0; r < n_ctx; ++r)
        if (status[r].code) return fail(status[r].code, "shard %u of %u: %s", r, n_ctx, status[r].msg);
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
FX_CATCH_CODE

// ---- Decomposer::RecursiveAssembly: the device work around the host plan of fx_recursive.h -------------------

int fx_system_prepare_batch(fx_ctx* ctx, const fx_batch* batch, uint32_t perturb, double* out_vars, double* out_params,
                            double* out_scale) try {
    if (!batch || !out_vars || !out_params || !out_scale) return fail(FX_ERR_INVALID, "bad argument");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    BatchHolder hold(ctx, db);  // freed on every way out
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
    return run();
}
FX_CATCH_CODE

int fx_cluster_solve_batch(fx_ctx* ctx, const fx_batch* batch, const fx_lm_opts* opts, fx_result* results) try {
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
    BatchHolder hold(ctx, db);
    tr.stamp("analysis + upload", batch->n_systems);
    db->resident = false;
    db->d.has_pose = 1u;
    rc = fx_lm_solve_device(ctx, db, &o);
    tr.stamp("solve (launches)", batch->n_systems);
    rc = read_back_and_free(ctx, hold.release(), batch, results, rc);
    tr.stamp("wait + read back", batch->n_systems);
    return rc;
}
FX_CATCH_CODE

int fx_pose_transform_points(fx_ctx* ctx, const double* poses, uint32_t n_poses, const uint32_t* pose_of, const uint32_t* var_idx,
                             uint32_t n_points, double* vars, uint32_t n_vars) try {
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
FX_CATCH_CODE

int fx_unscale_vars(fx_ctx* ctx, double scale, const double* scaled, const uint8_t* mask, double* vars, uint32_t n) try {
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
FX_CATCH_CODE

int fx_unscale_vars_strided(fx_ctx* ctx, const double* scales, uint32_t n_systems, uint32_t nvars, const double* scaled, const uint8_t* mask,
                            double* vars) try {
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
FX_CATCH_CODE

int fx_eval_residual_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jvals) try {
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    BatchHolder hold(ctx, db);  // freed on every way out
    rc = jvals ? fx_eval_residual_jacobian_device(ctx, db, 0) : fx_eval_residual_device(ctx, db, 0);
    if (!rc && r && db->d.n_exprs) rc = fx_batch_get_residuals(ctx, db, r);
    if (!rc && jvals && db->d.nnz) rc = fx_batch_get_jacobian_values(ctx, db, jvals);
    return rc;
}
FX_CATCH_CODE

int fx_analyze_batch(fx_ctx* ctx, const fx_batch* batch, uint8_t* dependent) try {
    if (!dependent) return fail(FX_ERR_INVALID, "dependent is NULL");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    BatchHolder hold(ctx, db);  // freed on every way out
    const uint32_t ne = db->d.n_exprs;
    if (fx::analyze_lds_bytes(db->d.max_vars_all, db->d.max_exprs_all) > 150u * 1024u) {
        return fail(FX_ERR_TOO_LARGE, "analyze keeps the dense expressions x variables Jacobian of a System in LDS (limit 150 KB)");
    }
    uint8_t* d_dep = nullptr;
    hipError_t e = hipMalloc((void**)&d_dep, std::max<uint32_t>(ne, 1));
    if (e == hipSuccess) e = fx::launch_analyze(db->d, db->d.vars0, db->d.max_vars_all, db->d.max_exprs_all, d_dep, ctx->stream);
    if (e == hipSuccess && ne) e = hipMemcpyAsync(dependent, d_dep, ne, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_dep) (void)hipFree(d_dep);
    if (e != hipSuccess) return fail(FX_ERR_HIP, "analyze failed: %s", hipGetErrorString(e));
    return FX_OK;
}
FX_CATCH_CODE

int fx_single_pass_blocks(const fx_batch* batch, uint32_t system, uint32_t* n_blocks, uint32_t* block_comp,
                          uint32_t* row_off, uint32_t* rows, uint32_t* var_off, uint32_t* vars) try {
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
FX_CATCH_CODE

void fx_atan2_cr_batch(uint64_t n, const double* y, const double* x, double* out) try {
    for (uint64_t i = 0; i < n; ++i) out[i] = fx::atan2_cr(y[i], x[i]);
}
FX_CATCH_VOID

int fx_qr_symbolic(int32_t nrows, int32_t ncols, const int32_t* colptr, const int32_t* rowidx, int use_colamd,
                   int32_t* col_perm, int32_t* row_perm, int32_t* h_ptr, int32_t* h_rows, int32_t h_cap, int32_t* r_ptr,
                   int32_t* r_rows, int32_t r_cap) try {
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
FX_CATCH_CODE

int fx_eval_residual_dense_jacobian(fx_ctx* ctx, const fx_batch* batch, double* r, double* jac, uint64_t* jac_off,
                                    uint64_t* total) try {
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
    BatchHolder hold(ctx, db);  // freed on every way out
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
    return rc;
}
FX_CATCH_CODE

int fx_constraint_residuals(fx_ctx* ctx, const fx_batch* batch, double* r) try {
    if (!r) return fail(FX_ERR_INVALID, "r is NULL");
    fx_dbatch* db = nullptr;
    int rc = fx_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    BatchHolder hold(ctx, db);  // freed on every way out
    rc = ensure_resid(ctx, db);
    hipError_t e = rc ? hipSuccess : fx::launch_identity_residuals(db->d, db->d.vars0, db->d.resid, ctx->stream);
    if (e != hipSuccess) rc = fail(FX_ERR_HIP, "launch failed: %s", hipGetErrorString(e));
    if (!rc && db->d.n_exprs) rc = fx_batch_get_residuals(ctx, db, r);
    return rc;
}
FX_CATCH_CODE

}  // extern "C"
