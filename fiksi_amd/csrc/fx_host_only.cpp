// `make asan`: the library's host-side logic as a plain g++ translation unit with -fsanitize=address,undefined.
// The host sources (fx_analyze / fx_programs / fx_upload / fx_solve / fx_entry .cpp) are compiled as they are against the stubs of fx_hip_shim.h; the kernel launchers it calls are defined
// here as "no device" stubs, and the sparse path's entry points likewise. Test infrastructure, not a product path.
#define FX_HOST_ONLY 1
#include "fx_analyze.cpp"
#include "fx_programs.cpp"
#include "fx_upload.cpp"
#include "fx_solve.cpp"
#include "fx_entry.cpp"

namespace fx {
hipError_t launch_solve(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_eval(const DeviceBatch&, const double*, bool, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_identity_residuals(const DeviceBatch&, const double*, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_dense_jacobian(const DeviceBatch&, const double*, const uint16_t*, const uint32_t*, const uint16_t*, const uint64_t*,
                                 double*, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_solve_wide(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_solve_wide_qr(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
size_t wide_qr_lds_bytes(uint32_t, uint32_t, uint32_t, uint32_t) { return 0; }
bool grouped_applies(const DeviceBatch&, const LmParams&) { return false; }
hipError_t launch_solve_grouped(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_solve_grouped_general(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
size_t grouped_lds_bytes(const DeviceBatch&, uint32_t, bool) { return 0; }
bool grouped_c_applies(const DeviceBatch&, const LmParams&) { return false; }
hipError_t launch_solve_grouped_c(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
size_t grouped_c_lds_bytes(const DeviceBatch&, uint32_t) { return 0; }
bool grouped_s_applies(const DeviceBatch&, const LmParams&) { return false; }
bool grouped_qr_class_applies(const DeviceBatch&, const LmParams&) { return false; }
hipError_t launch_grouped_qr_class(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
bool grouped_tiny_applies(const DeviceBatch&, const LmParams&) { return false; }
hipError_t launch_solve_grouped_s(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_solve_walk(const DeviceBatch&, const LmParams&, hipStream_t) { return hipErrorNoDevice; }
size_t presort_temp_bytes(uint32_t) { return 0; }
hipError_t launch_pull(void*, const void*, size_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_replicate(void*, size_t, size_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_presort(const DeviceBatch&, float*, uint32_t*, void*, size_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_presort_lists(const DeviceBatch&, float*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_prepare(const DeviceBatch&, uint32_t, double*, double*, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_unscale(double, const double*, const uint8_t*, double*, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_unscale_strided(const double*, uint32_t, uint32_t, const double*, const uint8_t*, double*, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_pose_transform(const double*, const uint32_t*, const uint32_t*, uint32_t, double*, hipStream_t) { return hipErrorNoDevice; }
size_t wide_lds_bytes(const DeviceBatch&) { return 0; }
size_t solve_lds_bytes(const DeviceBatch&) { return 0; }
size_t solve_lds_bytes_units(const DeviceBatch&) { return 0; }
size_t solve_lds_bytes_qr(const DeviceBatch&, bool) { return 0; }
size_t analyze_lds_bytes(uint32_t, uint32_t) { return 0; }
hipError_t launch_analyze(const DeviceBatch&, const double*, uint32_t, uint32_t, uint8_t*, hipStream_t) { return hipErrorNoDevice; }
SparsePlanCache* sparse_cache_new() { return nullptr; }
void sparse_cache_free(SparsePlanCache*) {}
bool sparse_cache_ready(const SparsePlanCache*) { return false; }
void sparse_cache_keep_slab(SparsePlanCache*, size_t) {}
hipError_t sparse_solve_group(const fx_batch*, const DeviceBatch&, const uint32_t*, uint32_t, const LmParams&, hipStream_t, SparsePlanCache*, bool) {
    return hipErrorNoDevice;
}
}  // namespace fx

// ---- allocation-failure injection (tools/alloc_fail_sweep.py, tests/test_host_sanitizers.py) ----------------------------
// This build's own operator new / delete (bound inside the library: -Wl,-Bsymbolic-functions): they count, and the n-th
// allocation after fx_test_fail_alloc_at(n) fails — std::bad_alloc from the throwing forms, NULL from the nothrow ones —
// so that a test can walk a failure through every allocation of an entry point and see an error code come back, the process
// alive, and nothing kept: no block of host memory (fx_test_live_allocations), no block of the make-believe device
// (fx_test_live_device_blocks). The memory still comes from malloc, so AddressSanitizer sees every byte.
namespace {
long g_fail_countdown = 0;  // 0: off
long g_alloc_count = 0;
long g_live = 0;
void* counted_alloc(size_t n, bool nothrow) {
    __atomic_add_fetch(&g_alloc_count, 1, __ATOMIC_RELAXED);
    if (__atomic_load_n(&g_fail_countdown, __ATOMIC_RELAXED) > 0 && __atomic_sub_fetch(&g_fail_countdown, 1, __ATOMIC_RELAXED) == 0) {
        if (nothrow) return nullptr;
        throw std::bad_alloc();
    }
    void* p = std::malloc(n ? n : 1);
    if (!p) {
        if (nothrow) return nullptr;
        throw std::bad_alloc();
    }
    __atomic_add_fetch(&g_live, 1, __ATOMIC_RELAXED);
    return p;
}
void counted_free(void* p) noexcept {
    if (!p) return;
    __atomic_sub_fetch(&g_live, 1, __ATOMIC_RELAXED);
    std::free(p);
}
}  // namespace
void* operator new(size_t n) { return counted_alloc(n, false); }
void* operator new[](size_t n) { return counted_alloc(n, false); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { return counted_alloc(n, true); }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { return counted_alloc(n, true); }
void operator delete(void* p) noexcept { counted_free(p); }
void operator delete[](void* p) noexcept { counted_free(p); }
void operator delete(void* p, size_t) noexcept { counted_free(p); }
void operator delete[](void* p, size_t) noexcept { counted_free(p); }

extern "C" {
void fx_test_fail_alloc_at(long n) { __atomic_store_n(&g_fail_countdown, n, __ATOMIC_RELAXED); }
long fx_test_alloc_count(void) { return __atomic_load_n(&g_alloc_count, __ATOMIC_RELAXED); }
long fx_test_live_allocations(void) { return __atomic_load_n(&g_live, __ATOMIC_RELAXED); }
long fx_test_live_device_blocks(void) { return __atomic_load_n(&fx_shim_live_blocks(), __ATOMIC_RELAXED); }
// The one-structure hint (fx_ctx_set_batch_hints) without a device: 1 when the verification accepts the batch ...
int fx_test_verify_one_structure(const fx_batch* b) { return fxh::verify_one_structure(b) ? 1 : 0; }
// ... and, for a batch that is of one structure, whether the plan made from System 0 alone says what the full analysis says:
// 0 the same, > 0 the first thing that differs, < 0 an error code of the analysis, -100 the hint was not taken
int fx_test_hinted_plan_differs(const fx_batch* b) try {
    fxh::HostPlan full, hint;
    int rc = fxh::analyze(b, &full);
    if (rc) return rc;
    fxh::g_hint_one_structure = true;
    rc = fxh::analyze(b, &hint);
    fxh::g_hint_one_structure = false;
    if (rc) return rc;
    if (!hint.hinted) return -100;
    if (!full.uniform || !hint.uniform) return 1;
    if (full.nnz != hint.nnz || full.max_free != hint.max_free || full.max_rows != hint.max_rows || full.max_vars != hint.max_vars ||
        full.max_exprs != hint.max_exprs || full.max_vars_all != hint.max_vars_all || full.max_exprs_all != hint.max_exprs_all ||
        full.max_pairs != hint.max_pairs || full.max_ents != hint.max_ents || full.max_pairs_tri != hint.max_pairs_tri ||
        full.n_large != hint.n_large || full.max_pairs_large != hint.max_pairs_large || full.max_ents_large != hint.max_ents_large)
        return 2;
    if (full.sys_ncomp != hint.sys_ncomp || full.sys_large != hint.sys_large || full.wide_list != hint.wide_list ||
        full.wide_decision != hint.wide_decision || !hint.sys_class.empty())
        return 3;
    const uint32_t nv0 = b->var_off[1], ne0 = b->expr_off[1];
    if (hint.var_info.size() != nv0 || hint.expr_comp.size() != ne0 || hint.expr_idx16.size() != 4 * (size_t)ne0 || hint.expr_tagx.size() != ne0) return 4;
    for (uint32_t s = 0; s < b->n_systems; ++s) {  // (the device fills the periods in: every System's analysed arrays are the first one's)
        if (memcmp(full.var_info.data() + (size_t)s * nv0, hint.var_info.data(), nv0 * sizeof(uint16_t))) return 5;
        if (memcmp(full.expr_comp.data() + (size_t)s * ne0, hint.expr_comp.data(), ne0 * sizeof(uint16_t))) return 6;
        if (memcmp(full.expr_idx16.data() + 4 * (size_t)s * ne0, hint.expr_idx16.data(), 4 * (size_t)ne0 * sizeof(uint16_t))) return 7;
        if (memcmp(full.expr_tagx.data() + (size_t)s * ne0, hint.expr_tagx.data(), ne0)) return 8;
    }
    return 0;
} catch (...) {
    fxh::g_hint_one_structure = false;
    return fx::translate_exception();
}
}
