// What HBM sustains for the READ : WRITE mix of the Jacobian-assembly kernel (eval_rows_kernel, fx_eval.hip): per ring16
// System it reads 896 B and writes 1 408 B (61 % writes). Streaming kernels with nothing to compute, buffers far past the
// 256 MiB Infinity Cache, 16-byte accesses, every byte touched once per launch:
//   copy 1:1 / read only / write only / K1's 0.64:1 mix, plain and non-temporal stores.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/rw_mix_probe.bin tools/probes/rw_mix_probe.hip
// Prints one JSON line: GB/s moved (read + written) per variant, HIP events over 20 launches.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));

// each thread: R 16-byte reads, W 16-byte writes, all coalesced across the block; NT = non-temporal stores
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const v2d* __restrict__ src, v2d* __restrict__ dst, size_t n_units) {
    const size_t u = (size_t)blockIdx.x;
    if (u >= n_units) return;
    v2d acc = {0.0, 0.0};
    v2d in[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) in[r] = src[(u * R + r) * 256 + threadIdx.x];
#pragma unroll
    for (int r = 0; r < R; ++r) acc += in[r];
    if (W == 0) {
        if (acc.x == 1.2345e300) dst[0] = acc;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        v2d t = acc + (double)w;
        if (NT) __builtin_nontemporal_store(t, &dst[(u * W + w) * 256 + threadIdx.x]);
        else dst[(u * W + w) * 256 + threadIdx.x] = t;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int R, int W, bool NT>
static int run(const char* name, const v2d* src, v2d* dst, size_t bytes, bool last) {
    // units so that the larger of the two streams covers `bytes`
    const size_t per_unit = (size_t)(R > W ? R : W) * 256 * 16;
    const size_t n_units = bytes / per_unit;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_kernel<R, W, NT>), dim3((unsigned)n_units), dim3(256), 0, 0, src, dst, n_units);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix_kernel<R, W, NT>), dim3((unsigned)n_units), dim3(256), 0, 0, src, dst, n_units);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double moved = (double)n_units * (R + W) * 256 * 16;
    printf("\"%s\": {\"read_bytes\": %.0f, \"written_bytes\": %.0f, \"us\": %.1f, \"GBps\": %.0f}%s", name, (double)n_units * R * 256 * 16,
           (double)n_units * W * 256 * 16, ms / reps * 1e3, moved / (ms / reps * 1e-3) / 1e9, last ? "" : ", ");
    return 0;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    v2d *src = nullptr, *dst = nullptr;
    CK(hipMalloc((void**)&src, bytes));
    CK(hipMalloc((void**)&dst, bytes));
    CK(hipMemset(src, 0, bytes));
    CK(hipMemset(dst, 0, bytes));
    CK(hipDeviceSynchronize());
    printf("{");
    if (run<4, 4, false>("copy_1_1", src, dst, bytes, false)) return 1;
    if (run<4, 4, true>("copy_1_1_nt", src, dst, bytes, false)) return 1;
    if (run<4, 0, false>("read_only", src, dst, bytes, false)) return 1;
    if (run<0, 4, false>("write_only", src, dst, bytes, false)) return 1;
    if (run<0, 4, true>("write_only_nt", src, dst, bytes, false)) return 1;
    if (run<7, 11, false>("k1_mix_7_11", src, dst, bytes, false)) return 1;
    if (run<7, 11, true>("k1_mix_7_11_nt", src, dst, bytes, true)) return 1;
    printf("}\n");
    return 0;
}
