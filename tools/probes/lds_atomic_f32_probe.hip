// Probe: throughput of back-to-back LDS atomic adds of one wavefront, f32 against f64, by how many lanes of an instruction
// hit the same address — no read-back between them (the assembly loops of the grouped kernels: 16 ... 40 atomics in a row).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_f32_probe.bin lds_atomic_f32_probe.hip ; prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned long long* out, double* sink) {
    __shared__ double accd[256];
    __shared__ float accf[256];
    const int lane = threadIdx.x;
    accd[lane] = 0.0;
    accf[lane] = 0.f;
    __syncthreads();
    const int REPS = 64;
    int slot = 0;
    for (int ways : {1, 2, 4, 16}) {
        const int tgt = lane / ways;
        {
            unsigned long long t0 = clock64();
            double v = 1.0 + lane;
            for (int r = 0; r < REPS; ++r) {
#pragma unroll
                for (int u = 0; u < 16; ++u) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)&accd[(tgt + u) & 63], v);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            unsigned long long t1 = clock64();
            if (lane == 0) out[slot] = (t1 - t0) * 100 / (REPS * 16);
            ++slot;
        }
        {
            unsigned long long t0 = clock64();
            float v = 1.f + lane;
            for (int r = 0; r < REPS; ++r) {
#pragma unroll
                for (int u = 0; u < 16; ++u) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&accf[(tgt + u) & 63], v, 0, 0, false);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            unsigned long long t1 = clock64();
            if (lane == 0) out[slot] = (t1 - t0) * 100 / (REPS * 16);
            ++slot;
        }
    }
    sink[lane] = accd[lane] + accf[lane];
}
int main() {
    unsigned long long* d; double* s;
    hipMalloc(&d, 64 * 8); hipMalloc(&s, 512 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"f64 1 lane/address", "f32 1 lane/address", "f64 2", "f32 2", "f64 4", "f32 4", "f64 16", "f32 16"};
    printf("{");
    for (int i = 0; i < 8; ++i) printf("\"%s\": %.2f%s", names[i], h[i] / 100.0, i < 7 ? ", " : "");
    printf(", \"unit\": \"clock64 ticks per ds_add instruction of a lone wavefront, 16 in a row\"}\n");
    return 0;
}
