// What one small solve call can cost at best on this box: wall-clock latencies (host clock, median of 200) of
//   a) one ~10 us kernel launch + hipStreamSynchronize
//   b) the same with its inputs read in place from host-coherent page-locked memory, as a chain of k dependent loads
//   c) the same writing its outputs to host-coherent memory
//   d) two / three dependent kernels + one synchronize (a pull kernel in front, a push kernel behind)
//   e) hipMemcpyAsync H2D + kernel + hipMemcpyAsync D2H + synchronize (the round-3 path)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/oneshot_latency_probe.bin tools/probes/oneshot_latency_probe.hip
// Prints one JSON line (microseconds).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// spins ~`work` dependent FMAs per lane, then CHAIN dependent loads through `in` (each index read from the one before), then
// writes 64 doubles to `out`
__global__ __launch_bounds__(64) void work_kernel(const uint32_t* __restrict__ in, double* __restrict__ out, int chain, int work) {
    uint32_t at = threadIdx.x & 15u;
    for (int c = 0; c < chain; ++c) at = in[at];
    double x = 1.0 + (double)at * 1e-9;
    for (int i = 0; i < work; ++i) x = fma(x, 1.0000001, 1e-9);
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

template <typename F>
static double median_us(F&& f, int reps = 200) {
    std::vector<double> t;
    for (int i = 0; i < reps + 20; ++i) {
        auto a = std::chrono::steady_clock::now();
        f();
        auto b = std::chrono::steady_clock::now();
        if (i >= 20) t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *d_in, *h_in;
    double *d_out, *h_out;
    const size_t bytes = 4096;
    CK(hipMalloc(&d_in, bytes));
    CK(hipMalloc(&d_out, bytes));
    CK(hipHostMalloc(&h_in, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_out, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    for (uint32_t i = 0; i < bytes / 4; ++i) h_in[i] = (i * 7u + 3u) % 16u;
    CK(hipMemcpy(d_in, h_in, bytes, hipMemcpyHostToDevice));
    const int WORK = 4000;  // ~10 us of dependent f64 FMAs
    auto sync = [&] { (void)hipStreamSynchronize(s); };
    // the kernel's own duration by events
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel_back_to_back_us\": %.2f", ms * 1e3 / 50);
    printf(", \"a_launch_sync\": %.2f", median_us([&] { hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK); sync(); }));
    printf(", \"a0_empty_work_launch_sync\": %.2f", median_us([&] { hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, 0); sync(); }));
    for (int chain : {1, 2, 4, 8}) {
        printf(", \"b_host_reads_chain%d\": %.2f", chain,
               median_us([&] { hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, h_in, d_out, chain, WORK); sync(); }));
    }
    printf(", \"c_host_writes\": %.2f", median_us([&] { hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, h_out, 2, WORK); sync(); }));
    printf(", \"d_pull_work\": %.2f", median_us([&] {
               hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, s, (uint4*)d_in, (const uint4*)h_in, (uint32_t)(bytes / 16));
               hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK);
               sync();
           }));
    printf(", \"d_pull_work_push\": %.2f", median_us([&] {
               hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, s, (uint4*)d_in, (const uint4*)h_in, (uint32_t)(bytes / 16));
               hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK);
               hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, s, (uint4*)h_out, (const uint4*)d_out, (uint32_t)(512 / 16));
               sync();
           }));
    printf(", \"e_memcpy_work_memcpy\": %.2f", median_us([&] {
               (void)hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s);
               hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, d_out, 2, WORK);
               (void)hipMemcpyAsync(h_out, d_out, 512, hipMemcpyDeviceToHost, s);
               sync();
           }));
    printf(", \"f_memcpy_work_host_writes\": %.2f", median_us([&] {
               (void)hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s);
               hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d_in, h_out, 2, WORK);
               sync();
           }));
    printf("}\n");
    return 0;
}
