// `make asan`: the library's host-side logic as a plain g++ translation unit with -fsanitize=address,undefined.
// fx_abi.cpp is compiled as it is against the stubs of fx_hip_shim.h; the kernel launchers it calls are defined
// here as "no device" stubs, and the sparse path's entry points likewise. Test infrastructure, not a product path.
#define FX_HOST_ONLY 1
#include "fx_abi.cpp"

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
