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
hipError_t sparse_solve_group(const fx_batch*, const DeviceBatch&, const uint32_t*, uint32_t, const LmParams&, hipStream_t, SparsePlanCache*) {
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
}
