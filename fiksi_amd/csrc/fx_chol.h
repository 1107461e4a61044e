// Register-resident Cholesky of an N x N SPD matrix held one column per lane (wave64, gfx950), and the
// triangular solves on the stored factor. Shared by the fused kernel (fx_kernels.hip) and the blocked
// factorization of the wide kernel (fx_wide.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_wave.h"

namespace fx {

// ------------------------------------------------------------------------------------------
// register-resident Cholesky of the N x N SPD matrix held one column per lane
// ------------------------------------------------------------------------------------------
// On entry lane j (< N) holds A[i][j], i = 0..N-1, in a[i] (A symmetric, so this is also row j).
// On exit lane k holds: a[p] = L[k][p] for p < k (row k of L), a[k] = d_k = L[k][k], and
// a[i] = L[i][k] * d_k for i > k (column k of L, scaled) — both triangular solves then need only
// wave-uniform broadcasts (v_readlane), never a per-lane register index. invd = 1 / d_lane.
// Returns false (wave-uniform) when a pivot is not positive and finite.
template <int N, typename T>
__device__ __forceinline__ bool chol_factor(T (&a)[N], T& invd, int lane) {
    bool bad = false;  // wave-uniform; checked once at the end (a bad pivot only produces NaN/Inf junk)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T piv = bcast(a[k], k);
        bad = bad || !(piv > T(0)) || !(piv < Lim<T>::huge());
        T rs = rsqrt_refined(piv);
        T ip = rs * rs;  // 1/pivot
        T ljk = a[k] * rs;
        T mul = (lane > k) ? a[k] * ip : T(0);  // A_jk / pivot; 0 keeps lanes <= k untouched
        if (lane >= k) a[k] = ljk;
        if (lane == k) invd = rs;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            T aik = bcast(a[i], k);  // lane k still holds A_ik = L_ik * d_k
            a[i] = fma(-aik, mul, a[i]);
        }
    }
    return !bad;
}

// Solves L L^T x = b with the factor layout above. b in `rhs` (lane j holds b_j); returns x_j.
template <int N, typename T>
__device__ __forceinline__ T chol_solve(const T (&a)[N], T invd, T rhs, int lane) {
    T acc = rhs;
#pragma unroll
    for (int k = 0; k < N; ++k) {  // forward: L y = b, y_k = acc_k / d_k
        T yk = bcast(acc * invd, k);
        if (lane > k) acc = fma(-a[k], yk, acc);
    }
    // acc_k = y_k d_k. backward: x_k = (y_k d_k - sum_{i>k} (L_ik d_k) x_i) / d_k^2
    T invd2 = invd * invd;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        T xi = bcast(acc * invd2, i);
        if (lane < i) acc = fma(-a[i], xi, acc);
    }
    return acc * invd2;
}

// The two sweeps of chol_solve on their own (the blocked factorization interleaves them with the
// off-diagonal block). Forward: L y = b, returns y_lane. Backward: L^T x = y, returns x_lane.
template <int N, typename T>
__device__ __forceinline__ T chol_forward(const T (&a)[N], T invd, T rhs, int lane) {
    T acc = rhs;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T yk = bcast(acc * invd, k);
        if (lane > k) acc = fma(-a[k], yk, acc);
    }
    return acc * invd;
}
template <int N, typename T>
__device__ __forceinline__ T chol_backward(const T (&a)[N], T invd, T y, int lane) {
    // chol_solve's backward sweep runs on acc_k = y_k d_k with the column entries stored as L_ik d_k
    T acc = y / invd;
    T invd2 = invd * invd;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        T xi = bcast(acc * invd2, i);
        if (lane < i) acc = fma(-a[i], xi, acc);
    }
    return acc * invd2;
}

}  // namespace fx
