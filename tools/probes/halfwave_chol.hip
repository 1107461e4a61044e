// Probe (the measurement that started fx_grouped.hip; the kernel ended up on DPP row broadcasts instead — in the
// full kernel every ds_swizzle became an LDS-pipe operation the compiler would not pipeline):
// register-resident Cholesky factor+solve of 32x32 SPD systems, one system per wavefront
// (v_readlane broadcasts, as fx_chol.h) against two systems per wavefront, one per 32-lane half
// (ds_swizzle broadcasts inside each half). Prints ns per system for both and the max difference.
//   hipcc -O3 --offload-arch=gfx950 -I fiksi_amd/csrc tools/probes/halfwave_chol.hip -o /tmp/hw && /tmp/hw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "fx_chol.h"

using namespace fx;
constexpr int N = 32;

template <int K>
__device__ __forceinline__ double hbcast(double v) {  // lane K of the caller's 32-lane half
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), K << 5);
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), K << 5);
    return __hiloint2double(hi, lo);
}

template <int K>
struct HStep {
    static __device__ __forceinline__ void factor(double (&a)[N], double& invd, bool& bad, int hl) {
        double piv = hbcast<K>(a[K]);
        bad = bad || !(piv > 0.0) || !(piv < 1e300);
        double rs = rsqrt_refined(piv);
        double ip = rs * rs;
        double ljk = a[K] * rs;
        double mul = (hl > K) ? a[K] * ip : 0.0;
        if (hl >= K) a[K] = ljk;
        if (hl == K) invd = rs;
#pragma unroll
        for (int i = K + 1; i < N; ++i) {
            double aik = hbcast<K>(a[i]);
            a[i] = fma(-aik, mul, a[i]);
        }
        if constexpr (K + 1 < N) HStep<K + 1>::factor(a, invd, bad, hl);
    }
    static __device__ __forceinline__ void fwd(const double (&a)[N], double invd, double& acc, int hl) {
        double yk = hbcast<K>(acc * invd);
        if (hl > K) acc = fma(-a[K], yk, acc);
        if constexpr (K + 1 < N) HStep<K + 1>::fwd(a, invd, acc, hl);
    }
    static __device__ __forceinline__ void bwd(const double (&a)[N], double invd2, double& acc, int hl) {
        double xi = hbcast<K>(acc * invd2);
        if (hl < K) acc = fma(-a[K], xi, acc);
        if constexpr (K > 0) HStep<K - 1>::bwd(a, invd2, acc, hl);
    }
};

__device__ __forceinline__ double entry(uint32_t sys, int i, int j) {
    uint32_t lo = i < j ? i : j, hi = i < j ? j : i;
    uint32_t h = sys * 2654435761u + lo * 40503u + hi * 9176u + 12345u;
    h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    double v = (double)(h & 0xffff) / 65536.0 - 0.5;
    return (i == j) ? 40.0 + v : v;
}

template <int R>
__global__ __launch_bounds__(64) void one_per_wave(double* out, uint32_t nsys) {
    const int lane = threadIdx.x;
    const uint32_t s = blockIdx.x;
    if (s >= nsys) return;
    double x = 1.0 + lane;
    for (int r = 0; r < R; ++r) {
        double a[N];
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = entry(s + r, i, lane & 31);
        double invd = 1.0;
        bool ok = chol_factor<N, double>(a, invd, lane);
        x = chol_solve<N, double>(a, invd, x, lane);
        if (!ok) x = 0.0;
    }
    if (lane < N) out[(size_t)s * N + lane] = x;
}

template <int R>
__global__ __launch_bounds__(64) void two_per_wave(double* out, uint32_t nsys) {
    extern __shared__ double dummy[];
    if (nsys == 0xFFFFFFFFu) dummy[threadIdx.x] = 1.0;  // keeps the dynamic LDS allocation alive
    const int lane = threadIdx.x;
    const int hl = lane & 31;
    const uint32_t s = blockIdx.x * 2 + (lane >> 5);
    double x = 1.0 + hl;
    for (int r = 0; r < R; ++r) {
        double a[N];
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = entry(s + r, i, hl);
        double invd = 1.0;
        bool bad = false;
        HStep<0>::factor(a, invd, bad, hl);
        double acc = x;
        HStep<0>::fwd(a, invd, acc, hl);
        double invd2 = invd * invd;
        HStep<N - 1>::bwd(a, invd2, acc, hl);
        x = acc * invd2;
        if (bad) x = 0.0;
    }
    if (s < nsys) out[(size_t)s * N + hl] = x;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

int main() {
    const uint32_t nsys = 100000;
    constexpr int R = 8;
    double *o1, *o2;
    CK(hipMalloc(&o1, (size_t)nsys * N * 8));
    CK(hipMalloc(&o2, (size_t)nsys * N * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        one_per_wave<R><<<nsys, 64>>>(o1, nsys);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
        printf("rep %d: one/wave %.3f ms (%.1f ns per factor+solve)\n", rep, ms1, 1e6 * ms1 / nsys / R);
        for (int lds : {0, 20 * 1024, 26 * 1024, 32 * 1024, 40 * 1024}) {  // 8+ / 8 / 6 / 5 / 4 wavefronts per CU
            CK(hipFuncSetAttribute((const void*)two_per_wave<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            CK(hipEventRecord(e0));
            two_per_wave<R><<<nsys / 2, 64, lds>>>(o2, nsys);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
            printf("   two/wave, %d KB LDS per wavefront: %.3f ms (%.1f ns)\n", lds / 1024, ms2, 1e6 * ms2 / nsys / R);
        }
    }
    std::vector<double> h1((size_t)nsys * N), h2((size_t)nsys * N);
    CK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
    double md = 0;
    size_t nbits = 0;
    for (size_t i = 0; i < h1.size(); ++i) {
        md = fmax(md, fabs(h1[i] - h2[i]));
        nbits += (h1[i] != h2[i]);
    }
    printf("max |diff| %.3e, %zu of %zu values differ; sample %.17g %.17g\n", md, nbits, h1.size(), h1[5], h2[5]);
    return 0;
}
