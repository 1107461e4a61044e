// Probe: what an LDS f64 atomic add costs one wavefront, by how many lanes of the instruction hit the same address,
// next to the pieces of a sparse-Cholesky column's dependent chain (LDS read round trip, DPP wave sum, rsqrt + Newton).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe.bin lds_atomic_probe.hip ; prints cycles per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    auto b = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return (b(0) + b(16)) + (b(32) + b(48));
}

__global__ void probe(unsigned long long* out, double* sink) {
    __shared__ double acc[256];
    const int lane = threadIdx.x;
    acc[lane] = 0.0;
    __syncthreads();
    const int REPS = 256;
    int slot = 0;
    for (int ways : {1, 2, 4, 8, 16, 32, 64}) {
        const int tgt = lane / ways;  // `ways` lanes per address
        unsigned long long t0 = clock64();
        double v = 1.0 + lane;
        for (int r = 0; r < REPS; ++r) {
            lds_add_f64(&acc[tgt], v);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            v += acc[lane & 63];  // dependent read back
        }
        unsigned long long t1 = clock64();
        if (lane == 0) out[slot] = (t1 - t0) / REPS;
        sink[lane] = v;
        ++slot;
    }
    {   // plain dependent LDS read chain
        unsigned long long t0 = clock64();
        int idx = lane;
        double v = 0.0;
        for (int r = 0; r < REPS; ++r) { v += acc[idx & 63]; idx = (int)v & 63; }
        unsigned long long t1 = clock64();
        if (lane == 0) out[slot] = (t1 - t0) / REPS;
        sink[64 + lane] = v; ++slot;
    }
    {   // wave_sum chain
        unsigned long long t0 = clock64();
        double v = lane;
        for (int r = 0; r < REPS; ++r) v = wave_sum(v) * 1e-3 + lane;
        unsigned long long t1 = clock64();
        if (lane == 0) out[slot] = (t1 - t0) / REPS;
        sink[128 + lane] = v; ++slot;
    }
    {   // rsqrt + 2 Newton
        unsigned long long t0 = clock64();
        double p = 2.0 + lane;
        for (int r = 0; r < REPS; ++r) { double y = __builtin_amdgcn_rsq(p); y = fma(0.5 * y, fma(-p * y, y, 1.0), y); y = fma(0.5 * y, fma(-p * y, y, 1.0), y); p = y + 2.0; }
        unsigned long long t1 = clock64();
        if (lane == 0) out[slot] = (t1 - t0) / REPS;
        sink[192 + lane] = p; ++slot;
    }
    {   // sqrt + division
        unsigned long long t0 = clock64();
        double p = 2.0 + lane;
        for (int r = 0; r < REPS; ++r) { double d = sqrt(p); p = 1.0 / d + 2.0; }
        unsigned long long t1 = clock64();
        if (lane == 0) out[slot] = (t1 - t0) / REPS;
        sink[lane] += p; ++slot;
    }
}
int main() {
    unsigned long long* d; double* s;
    hipMalloc(&d, 64 * 8); hipMalloc(&s, 512 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"atomic+readback 1-way", "2-way", "4-way", "8-way", "16-way", "32-way", "64-way", "dependent LDS read", "wave_sum (DPP+readlane)", "rsqrt + 2 Newton", "sqrt + 1/x"};
    for (int i = 0; i < 11; ++i) printf("%-28s %llu cycles (clock64 ticks)\n", names[i], h[i]);
    return 0;
}
