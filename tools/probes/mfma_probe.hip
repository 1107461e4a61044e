// north_star: "MFMA used only where a per-system dense JtJ tile actually beats the memory-bound path (evidence)".
// The one place of the hot path with a dense tile is the trailing update of the 32 x 32 Cholesky factor (and the
// formation of JtJ, which is sparse: 408 products per System). This probe times, on one wavefront, one SIMD:
//   (a) the form the grouped kernel uses (fx_grouped.hip, RStep::factor): for four pivot columns, every row below
//       them, both column halves — `v_fmac_f64_dpp ... row_newbcast`, one instruction per row, serving the FOUR
//       Systems of the wavefront (one per DPP row);
//   (b) the same rank-4 update of the same four 32 x 32 matrices as 16 `v_mfma_f64_16x16x4_f64` (four 16 x 16 tiles
//       per matrix), operands already in MFMA layout — the best case, no data movement charged;
// and the f32 analogues. Both do the full symmetric update (2 x 32 x 32 x 4 flops per matrix).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_probe.bin tools/probes/mfma_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void fnma_dpp(double& acc, double m, double w) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(w), "n"(K));
}
template <int K>
__device__ __forceinline__ void fnma_dpp(float& acc, float m, float w) {
    asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(w), "n"(K));
}

template <typename T>
__global__ __launch_bounds__(64) void dpp_rank4(T* io, long long* cycles, int reps) {
    T a[2][32], mul[2][4];
    for (int q = 0; q < 2; ++q) {
        for (int i = 0; i < 32; ++i) a[q][i] = io[threadIdx.x + 64 * (q * 32 + i)];
        for (int k = 0; k < 4; ++k) mul[q][k] = io[threadIdx.x + k] * T(1e-3);
    }
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 4; i < 32; ++i) {  // four pivot columns (lanes 0..3 of each row), rows 4..31, both halves
            fnma_dpp<0>(a[1][i], a[0][i], mul[1][0]);
            fnma_dpp<0>(a[0][i], a[0][i], mul[0][0]);
            fnma_dpp<1>(a[1][i], a[0][i], mul[1][1]);
            fnma_dpp<1>(a[0][i], a[0][i], mul[0][1]);
            fnma_dpp<2>(a[1][i], a[0][i], mul[1][2]);
            fnma_dpp<2>(a[0][i], a[0][i], mul[0][2]);
            fnma_dpp<3>(a[1][i], a[0][i], mul[1][3]);
            fnma_dpp<3>(a[0][i], a[0][i], mul[0][3]);
        }
    }
    long long t1 = clock64();
    T s = 0;
    for (int q = 0; q < 2; ++q)
        for (int i = 0; i < 32; ++i) s += a[q][i];
    io[threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ __launch_bounds__(64) void mfma_rank4_f64(double* io, long long* cycles, int reps) {
    d4 c[4][4];  // four matrices x four 16 x 16 tiles
    double av[4][2], bv[4][2];
    for (int m = 0; m < 4; ++m) {
        for (int t = 0; t < 4; ++t)
            for (int e = 0; e < 4; ++e) c[m][t][e] = io[threadIdx.x + 64 * (m * 16 + t * 4 + e)];
        for (int h = 0; h < 2; ++h) {
            av[m][h] = io[threadIdx.x + m + h] * 1e-3;
            bv[m][h] = io[threadIdx.x + 7 * m + h] * 1e-3;
        }
    }
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            c[m][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][0], bv[m][0], c[m][0], 0, 0, 0);
            c[m][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][0], bv[m][1], c[m][1], 0, 0, 0);
            c[m][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][1], bv[m][0], c[m][2], 0, 0, 0);
            c[m][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][1], bv[m][1], c[m][3], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int m = 0; m < 4; ++m)
        for (int t = 0; t < 4; ++t)
            for (int e = 0; e < 4; ++e) s += c[m][t][e];
    io[threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ __launch_bounds__(64) void mfma_rank4_f32(float* io, long long* cycles, int reps) {
    f4 c[4][4];
    float av[4][2], bv[4][2];
    for (int m = 0; m < 4; ++m) {
        for (int t = 0; t < 4; ++t)
            for (int e = 0; e < 4; ++e) c[m][t][e] = io[threadIdx.x + 64 * (m * 16 + t * 4 + e)];
        for (int h = 0; h < 2; ++h) {
            av[m][h] = io[threadIdx.x + m + h] * 1e-3f;
            bv[m][h] = io[threadIdx.x + 7 * m + h] * 1e-3f;
        }
    }
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            c[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][0], bv[m][0], c[m][0], 0, 0, 0);
            c[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][0], bv[m][1], c[m][1], 0, 0, 0);
            c[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][1], bv[m][0], c[m][2], 0, 0, 0);
            c[m][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][1], bv[m][1], c[m][3], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int t = 0; t < 4; ++t)
            for (int e = 0; e < 4; ++e) s += c[m][t][e];
    io[threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    void* io;
    long long* cyc;
    CK(hipMalloc(&io, 1 << 20));
    CK(hipMalloc((void**)&cyc, 64));
    CK(hipMemset(io, 0, 1 << 20));
    const int reps = 2000;
    long long h[4] = {0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {  // first pass warms the instruction cache
        hipLaunchKernelGGL(dpp_rank4<double>, dim3(1), dim3(64), 0, 0, (double*)io, cyc, reps);
        CK(hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(mfma_rank4_f64, dim3(1), dim3(64), 0, 0, (double*)io, cyc, reps);
        CK(hipMemcpy(&h[1], cyc, 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(dpp_rank4<float>, dim3(1), dim3(64), 0, 0, (float*)io, cyc, reps);
        CK(hipMemcpy(&h[2], cyc, 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(mfma_rank4_f32, dim3(1), dim3(64), 0, 0, (float*)io, cyc, reps);
        CK(hipMemcpy(&h[3], cyc, 8, hipMemcpyDeviceToHost));
    }
    // clock64() counts at the constant 100 MHz reference clock on gfx9: convert with the shader clock the kernel saw
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double per = 1.0 / reps;
    printf("{\"what\": \"rank-4 trailing update of four 32x32 matrices, one wavefront; ticks of clock64() per update\",\n");
    printf(" \"f64_dpp_fmac_224_instructions\": %.1f, \"f64_mfma_16x16x4_16_instructions\": %.1f,\n", h[0] * per, h[1] * per);
    printf(" \"f32_dpp_fmac_224_instructions\": %.1f, \"f32_mfma_16x16x4_16_instructions\": %.1f, \"clock_rate_khz\": %d}\n", h[2] * per, h[3] * per, khz);
    return 0;
}
