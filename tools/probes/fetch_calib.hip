// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of the Jacobian-assembly kernel
// (eval_rows_kernel, fx_kernels.hip): the guide documents the factor 2 for 16 B / lane streaming reads only. Each
// kernel below reads a KNOWN number of distinct bytes, once, from buffers far larger than the 256 MiB Infinity
// Cache; the per-kernel FETCH_SIZE of a `rocprofv3 --pmc FETCH_SIZE` pass over this program, set against the
// byte counts it prints, gives the factor per width (tools/fetch_calib_summary.py -> profiles/).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_calib.bin tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

// one element per thread, coalesced: 1 B (expression tags), 8 B (element fields / parameters), 16 B (the documented case)
__global__ void calib_read_u8(const uint8_t* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = i < n ? a[i] : 0u;
    if (v == 0xEEu) sink[0] = v;  // never true (the buffer holds 1s), but the compiler cannot know: keeps the load
}
__global__ void calib_read_8b(const ushort4* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ushort4 v = i < n ? a[i] : make_ushort4(0, 0, 0, 0);
    if (v.x == 0xFFFF && v.y == 0xFFFF && v.z == 0xFFFF && v.w == 0x1234) sink[0] = v.x;
}
__global__ void calib_read_f64(const double* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = i < n ? a[i] : 0.0;
    if (v == 1.2345e300) sink[0] = 1;
}
__global__ void calib_read_16b(const double2* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 v = i < n ? a[i] : make_double2(0, 0);
    if (v.x == 1.2345e300 && v.y == 7.0) sink[0] = 1;
}
// K1's gather: thread t of a "System" of 32 rows reads 8 doubles of that System's 32-variable slice (256 B), at the
// positions a ring16 row reads; every slice is read completely, each of its bytes by several threads
__global__ void calib_gather(const double* __restrict__ x, size_t n_sys, uint32_t* __restrict__ sink) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t s = row >> 5;
    uint32_t r = (uint32_t)row & 31u;
    double acc = 0.0;
    if (s < n_sys) {
        const double* xs = x + 32 * s;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += xs[(2u * r + (uint32_t)e * 5u) & 31u];
    }
    if (acc == 1.2345e300) sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer: four times the Infinity Cache
    void* buf = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipMemset(sink, 0, 64));
    CK(hipDeviceSynchronize());
    const int reps = 3;
    for (int r = 0; r < reps; ++r) {
        size_t n;
        n = bytes;       hipLaunchKernelGGL(calib_read_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint8_t*)buf, n, sink);
        n = bytes / 8;   hipLaunchKernelGGL(calib_read_8b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const ushort4*)buf, n, sink);
        n = bytes / 8;   hipLaunchKernelGGL(calib_read_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const double*)buf, n, sink);
        n = bytes / 16;  hipLaunchKernelGGL(calib_read_16b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const double2*)buf, n, sink);
        n = bytes / 256; hipLaunchKernelGGL(calib_gather, dim3((unsigned)((n * 32 + 255) / 256)), dim3(256), 0, 0, (const double*)buf, n, sink);
    }
    CK(hipDeviceSynchronize());
    printf("{\"bytes_read_per_launch\": {\"calib_read_u8\": %zu, \"calib_read_8b\": %zu, \"calib_read_f64\": %zu, \"calib_read_16b\": %zu, \"calib_gather\": %zu}}\n",
           bytes, bytes, bytes, bytes, bytes);
    (void)hipFree(buf);
    (void)hipFree(sink);
    return 0;
}
