// The grouped kernels' shared pieces (fx_grouped.hip, fx_grouped_c.hip): a System lives in one DPP row of 16 lanes — row-wide
// broadcasts and sums, the register-resident Cholesky with the columns of a matrix spread over a row, the per-row phases and
// the lambda ladder's verdict codes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_wave.h"

namespace fx {

constexpr int RS = 16;  // lanes per System: one DPP row

// ------------------------------------------------------------------------------------------
// row-wide cross-lane helpers
// ------------------------------------------------------------------------------------------
// lane K of the caller's row of 16 (DPP row_newbcast; the lane is an instruction immediate)
template <int K>
__device__ __forceinline__ double rbcast(double v) {
    // one v_mov_b64_dpp (gfx90a+: 64-bit DPP exists for row_newbcast); mov_dpp, not update_dpp: every source lane of
    // a row broadcast is valid, so no `old` value has to be set up first
    return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + K, 0xF, 0xF, false));
}
template <int K>
__device__ __forceinline__ float rbcast(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + K, 0xF, 0xF, false));
}
// acc = fma(-(m's lane K of the row), w, acc) in one VOP2-DPP instruction (gfx90a+: 64-bit DPP takes row_newbcast)
template <int K>
__device__ __forceinline__ void fnma_rbcast(double& acc, double m, double w) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(w), "n"(K));
}
template <int K>
__device__ __forceinline__ void fnma_rbcast(float& acc, float m, float w) {
    asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(w), "n"(K));
}
// two wait states between the VALU instruction that produced `v` and a DPP read of it (inline asm is opaque to the
// compiler's hazard recogniser); the value passes through so that the order is a data dependence
__device__ __forceinline__ void dpp_settle(double& v) { asm volatile("s_nop 1" : "+v"(v)); }
__device__ __forceinline__ void dpp_settle(float& v) { asm volatile("s_nop 1" : "+v"(v)); }
// the same with m == acc (the register is read through DPP before it is written)
template <int K>
__device__ __forceinline__ void fnma_rbcast_self(double& acc, double w) {
    asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "n"(K));
}
template <int K>
__device__ __forceinline__ void fnma_rbcast_self(float& acc, float w) {
    asm volatile("v_fmac_f32_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "n"(K));
}
// sum over the 16 lanes of a row, identical bits in every lane: the DPP butterfly of wave_sum
template <typename T>
__device__ __forceinline__ T row_sum(T v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}
// wave_sum over a vector laid out 16 entries per accumulator (acc[q] of lane r = entry 16 q + r, entries
// 64 apart added up first): (b0 + b1) + (b2 + b3), the order wave_sum adds its four rows in
template <typename T>
__device__ __forceinline__ T block_sum4(const T (&acc)[4]) {
    return (row_sum(acc[0]) + row_sum(acc[1])) + (row_sum(acc[2]) + row_sum(acc[3]));
}
// sum += t(lane 0) + t(lane 1) + ... + t(lane 15) of the row, strictly in that order
template <int... K>
__device__ __forceinline__ void seq_add_impl(double& sum, double t, std::integer_sequence<int, K...>) {
    ((sum += rbcast<K>(t)), ...);
}
__device__ __forceinline__ void seq_add(double& sum, double t) { seq_add_impl(sum, t, std::make_integer_sequence<int, RS>{}); }

// ------------------------------------------------------------------------------------------
// Cholesky of fx_chol.h with the columns of one matrix spread over a row: lane r holds column r + 16 q in
// a[q][.] (same operations on the same operands as chol_factor / chol_solve: bit-identical results)
// ------------------------------------------------------------------------------------------
template <int NC, typename T, int K, bool BOUNDED>
struct RStep {  // one column step of the factorization / of a triangular solve
    static constexpr int N = RS * NC;
    static constexpr int KA = K / RS, KL = K % RS;  // array and lane of column K
    static __device__ __forceinline__ void factor(T (&a)[NC][N], T (&invd)[NC], bool& bad, int hl, int kmax) {
        {
        const T piv = rbcast<KL>(a[KA][K]);
        bad = bad || !(piv > T(0)) || !(piv < Lim<T>::huge());
        const T rs = rsqrt_refined(piv);
        const T ip = rs * rs;  // 1/pivot
        T mul[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            mul[q] = T(0);
            if (q < KA) continue;  // columns below K: untouched
            const T ljk = a[q][K] * rs;
            const bool above = (q > KA) || (hl > KL);  // column > K
            mul[q] = above ? a[q][K] * ip : T(0);
            if (above || hl == KL) a[q][K] = ljk;
            if (q == KA && hl == KL) invd[q] = rs;
        }
        // a[q][i] -= A_iK (from the lane of column K, which still holds A_iK = L_iK * d_K) * mul[q]: ONE instruction
        // each, v_fmac_f64_dpp with the broadcast as its DPP operand. Array KA last: its update leaves the lane of
        // column K untouched (mul = 0 there), and so no DPP read follows a write of the same register.
        if constexpr (NC <= 2) {
#pragma unroll
            for (int i = K + 1; i < N; ++i) {
#pragma unroll
                for (int q = NC - 1; q >= KA; --q) {
                    if (q == KA && KL == RS - 1) continue;  // no column of this array lies above K
                    if (q == KA) fnma_rbcast_self<KL>(a[q][i], mul[q]);
                    else fnma_rbcast<KL>(a[q][i], a[KA][i], mul[q]);
                }
            }
            // (inline asm is opaque to the hazard recogniser: a DPP read needs two wait states after a VALU write of
            // its source, and the next pivot broadcast may read what the last instruction above wrote)
            asm volatile("s_nop 1");
        } else {
            // the 48-column build lives at the register limit, where the compiler has to reload operands right in
            // front of their use — a VALU write the hand-written DPP read could not be protected from. It takes the
            // compiler's own instructions: one v_mov_b64_dpp per broadcast, shared by the three multiply-adds.
#pragma unroll
            for (int i = K + 1; i < N; ++i) {
                const T aik = rbcast<KL>(a[KA][i]);
#pragma unroll
                for (int q = KA; q < NC; ++q) {
                    if (q == KA && KL == RS - 1) continue;
                    a[q][i] = fma(-aik, mul[q], a[q][i]);
                }
            }
        }
        }
    }
    // acc[q] -= a[q][K] * y_K, y_K = (acc * invd) of column K's lane: the broadcast is the DPP operand of the
    // multiply-add; lanes that must not take part get a zero factor.
    static __device__ __forceinline__ void forward(const T (&a)[NC][N], const T (&invd)[NC], T (&acc)[NC], int hl, int kmax) {
        {
        T t = acc[KA] * invd[KA];
        if constexpr (NC <= 2) {
            dpp_settle(t);
#pragma unroll
            for (int q = NC - 1; q >= KA; --q) {
                if (q == KA && KL == RS - 1) continue;
                const T w = (q > KA || hl > KL) ? a[q][K] : T(0);
                fnma_rbcast<KL>(acc[q], t, w);
            }
        } else {
            const T yk = rbcast<KL>(t);
#pragma unroll
            for (int q = KA; q < NC; ++q) {
                if (q == KA && KL == RS - 1) continue;
                if (q > KA || hl > KL) acc[q] = fma(-a[q][K], yk, acc[q]);
            }
        }
        }
    }
    static __device__ __forceinline__ void backward(const T (&a)[NC][N], const T (&invd2)[NC], T (&acc)[NC], int hl, int kmax) {
        {
        T t = acc[KA] * invd2[KA];
        if constexpr (NC <= 2) {
            dpp_settle(t);
#pragma unroll
            for (int q = 0; q <= KA; ++q) {
                if (q == KA && KL == 0) continue;  // no column of this array lies below K
                const T w = (q < KA || hl < KL) ? a[q][K] : T(0);
                fnma_rbcast<KL>(acc[q], t, w);
            }
        } else {
            const T xi = rbcast<KL>(t);
#pragma unroll
            for (int q = 0; q <= KA; ++q) {
                if (q == KA && KL == 0) continue;
                if (q < KA || hl < KL) acc[q] = fma(-a[q][K], xi, acc[q]);
            }
        }
        }
    }
};

// The steps in blocks of eight columns. BOUNDED (the SinglePass build, whose blocks are mostly far smaller than N):
// kmax is wave-uniform, no row of the wavefront has more than kmax columns in use; the columns past a System's free
// variables are identity padding that no other column depends on, so a block of steps at or past kmax is skipped.
// (One test per eight steps, and none in the other builds: the scheduler needs long straight-line regions here —
// a test per step cost the headline shape 25 %.)
template <int NC, typename T, int KB, bool BOUNDED>
struct RBlock {
    static constexpr int N = RS * NC;
    template <int... I>
    static __device__ __forceinline__ void factor8(T (&a)[NC][N], T (&invd)[NC], bool& bad, int hl, int kmax, std::integer_sequence<int, I...>) {
        (RStep<NC, T, 8 * KB + I, BOUNDED>::factor(a, invd, bad, hl, kmax), ...);
    }
    template <int... I>
    static __device__ __forceinline__ void forward8(const T (&a)[NC][N], const T (&invd)[NC], T (&acc)[NC], int hl, int kmax, std::integer_sequence<int, I...>) {
        (RStep<NC, T, 8 * KB + I, BOUNDED>::forward(a, invd, acc, hl, kmax), ...);
    }
    template <int... I>
    static __device__ __forceinline__ void backward8(const T (&a)[NC][N], const T (&invd2)[NC], T (&acc)[NC], int hl, int kmax, std::integer_sequence<int, I...>) {
        (RStep<NC, T, 8 * KB + 7 - I, BOUNDED>::backward(a, invd2, acc, hl, kmax), ...);
    }
    static __device__ __forceinline__ void factor(T (&a)[NC][N], T (&invd)[NC], bool& bad, int hl, int kmax) {
        if (!BOUNDED || 8 * KB < kmax) factor8(a, invd, bad, hl, kmax, std::make_integer_sequence<int, 8>{});
        if constexpr (8 * KB + 8 < N) RBlock<NC, T, KB + 1, BOUNDED>::factor(a, invd, bad, hl, kmax);
    }
    static __device__ __forceinline__ void forward(const T (&a)[NC][N], const T (&invd)[NC], T (&acc)[NC], int hl, int kmax) {
        if (!BOUNDED || 8 * KB < kmax) forward8(a, invd, acc, hl, kmax, std::make_integer_sequence<int, 8>{});
        if constexpr (8 * KB + 8 < N) RBlock<NC, T, KB + 1, BOUNDED>::forward(a, invd, acc, hl, kmax);
    }
    static __device__ __forceinline__ void backward(const T (&a)[NC][N], const T (&invd2)[NC], T (&acc)[NC], int hl, int kmax) {
        if (!BOUNDED || 8 * KB < kmax) backward8(a, invd2, acc, hl, kmax, std::make_integer_sequence<int, 8>{});
        if constexpr (KB > 0) RBlock<NC, T, KB - 1, BOUNDED>::backward(a, invd2, acc, hl, kmax);
    }
};

// LDS accesses of one wavefront execute in order; between phases that exchange data through LDS inside a
// row only the compiler has to be kept from reordering them (no s_barrier: rows diverge).
// (wavefront scope: a workgroup-scope fence would also drain the global loads in flight, vmcnt(0))
__device__ __forceinline__ void group_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

enum GroupPhase : int { GP_NEXT = 0, GP_COMP = 1, GP_RUN = 2, GP_FINISH = 3, GP_EXIT = 4 };

// What one lambda trial found (lm.rs:115-191), before anything of the row's LM state has changed. LC_REJECT is the plain
// reject — lambda x reject_factor and the next trial — the only verdict after which the loop goes on from the same point
// with the same Jacobian: the trials that follow a plain reject are independent of it and of each other (the ladder below).
enum LadderCode : int {
    LC_REJECT = 0,    // lm.rs:187-190
    LC_SINGULAR = 1,  // lm.rs:134-137
    LC_NAN = 2,       // non-finite step (reported as FX_EXIT_NAN; the reference would spin)
    LC_STEP = 3,      // lm.rs:139-142
    LC_ACCEPT = 4,    // lm.rs:151-186
    LC_REJ_NAN = 5,   // rejected with a NaN trial point and lambda past 1e300
    LC_REJ_FTOL = 6,  // f32 only: rejected within round-off of the current SSE
    LC_CAP = 7,       // max_trials reached before this trial
    LC_FRESH = 8      // the component's start point, no trial
};

// any lane's value (ds_bpermute: the source lane must be active)
__device__ __forceinline__ int lane_get(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }
__device__ __forceinline__ float lane_get(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
__device__ __forceinline__ double lane_get(double v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

}  // namespace fx
