// wave64 helpers shared by the HIP kernels (gfx950): lane broadcasts, DPP reductions, LDS float atomics,
// refined reciprocal square roots, the reference LCG's jump-ahead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

// ------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double bcast(double v, int src_lane) {  // src_lane must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// Sum over the 64 lanes, result wave-uniform (identical bits in every lane). Within each row of
// 16 lanes a DPP butterfly (quad_perm, row_half_mirror, row_mirror: no LDS traffic), then the four
// row sums are combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}

// per-precision pieces of the linear algebra
__device__ __forceinline__ void lds_add(double* p, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}
__device__ __forceinline__ void lds_add(float* p, float v) {
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, 0, 0, false);
}
// 1/sqrt(p): hardware seed + Newton steps y <- y + y*(1 - p*y*y)/2 (v_rsq_f64 ~23 bits -> two steps;
// v_rsq_f32 ~1 ulp -> one step)
__device__ __forceinline__ double rsqrt_refined(double p) {
    double y = __builtin_amdgcn_rsq(p);
    y = fma(0.5 * y, fma(-p * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-p * y, y, 1.0), y);
    return y;
}
__device__ __forceinline__ float rsqrt_refined(float p) {
    float y = __builtin_amdgcn_rsqf(p);
    return fmaf(0.5f * y, fmaf(-p * y, y, 1.0f), y);
}
template <typename T> struct Lim;
template <> struct Lim<double> { static __device__ __forceinline__ double huge() { return 1.0e300; } };
template <> struct Lim<float> { static __device__ __forceinline__ float huge() { return 1.0e30f; } };
template <typename T> struct Vec16;  // 16-byte LDS vector of T
template <> struct Vec16<double> { using type = double2; static constexpr int n = 2; };
template <> struct Vec16<float> { using type = float4; static constexpr int n = 4; };
// state after k steps of the reference LCG (rand.rs): the 2^i-step maps are composed bit by bit
__device__ __forceinline__ uint32_t lcg_jump(uint32_t st, uint32_t k) {
    uint32_t a = 1664525u, c = 1013904223u;
    while (k) {
        if (k & 1u) st = a * st + c;
        c = (a + 1u) * c;
        a = a * a;
        k >>= 1;
    }
    return st;
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (lane == 0) ? 0ull : (~0ull >> (64 - lane)); }


// calculate_system_scale (fiksi/src/assemble/mod.rs:32-44, utils.rs:11-33): sqrt of the mean square over all
// variables and the distance parameters of PointPointDistance / PointLineDistance, summed strictly in the
// reference's order (variables, then expressions, each ascending) so the result is bit-identical: a
// wavefront squares 64 values at a time and adds them one by one through v_readlane. The accessors return
// variable i / the tag and parameter of expression i of the System. Wave-uniform result.
template <typename VarFn, typename TagFn, typename ParamFn>
__device__ __forceinline__ double system_scale_wave(uint32_t nvt, uint32_t net, int lane, VarFn var_at, TagFn tag_at,
                                                    ParamFn param_at) {
    double sum = 0.0;
    uint32_t count = nvt;
    for (uint32_t base = 0; base < nvt; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        double t = 0.0;
        if (i < nvt) {
            const double v = var_at(i);
            t = v * v;
        }
        const uint32_t cnt = min(64u, nvt - base);
        for (uint32_t k = 0; k < cnt; ++k) sum += bcast(t, (int)k);
    }
    for (uint32_t base = 0; base < net; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        double t = 0.0;
        bool isd = false;
        if (i < net) {
            const int tag = tag_at(i);
            isd = (tag == 1) || (tag == 4);  // FX_TAG_PPD, FX_TAG_PLD
            if (isd) {
                const double d = param_at(i);
                t = d * d;
            }
        }
        count += (uint32_t)__popcll(__ballot(isd));
        const uint32_t cnt = min(64u, net - base);
        // adding the +0.0 of non-distance rows is exact, so the order of the real terms is kept
        for (uint32_t k = 0; k < cnt; ++k) sum += bcast(t, (int)k);
    }
    return ::sqrt(sum / (double)count);
}

}  // namespace fx
