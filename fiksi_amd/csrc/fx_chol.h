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
// nb (wave-uniform, default N): the matrix is nb x nb, padded with the identity up to N. Steps k >= nb of any of the
// routines below change nothing then (pivot 1, multipliers 0, right-hand sides 0 on the padding) and are skipped —
// bit for bit the result of running them. The wide kernel's second diagonal block (nfree - 64 columns) uses it.
template <int N, typename T>
__device__ __forceinline__ bool chol_factor(T (&a)[N], T& invd, int lane, int nb = N) {
    bool bad = false;  // wave-uniform; checked once at the end (a bad pivot only produces NaN/Inf junk)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < nb) {  // (a guarded step, not an early exit: the unrolled body keeps its static register indices)
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
    }
    return !bad;
}

// Solves L L^T x = b with the factor layout above. b in `rhs` (lane j holds b_j); returns x_j.
template <int N, typename T>
__device__ __forceinline__ T chol_solve(const T (&a)[N], T invd, T rhs, int lane, int nb = N) {
    T acc = rhs;
#pragma unroll
    for (int k = 0; k < N; ++k) {  // forward: L y = b, y_k = acc_k / d_k
        if (k < nb) {
            T yk = bcast(acc * invd, k);
            if (lane > k) acc = fma(-a[k], yk, acc);
        }
    }
    // acc_k = y_k d_k. backward: x_k = (y_k d_k - sum_{i>k} (L_ik d_k) x_i) / d_k^2
    T invd2 = invd * invd;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        if (i < nb) {
            T xi = bcast(acc * invd2, i);
            if (lane < i) acc = fma(-a[i], xi, acc);
        }
    }
    return acc * invd2;
}

// The two sweeps of chol_solve on their own (the blocked factorization interleaves them with the
// off-diagonal block). Forward: L y = b, returns y_lane. Backward: L^T x = y, returns x_lane.
template <int N, typename T>
__device__ __forceinline__ T chol_forward(const T (&a)[N], T invd, T rhs, int lane, int nb = N) {
    T acc = rhs;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < nb) {
            T yk = bcast(acc * invd, k);
            if (lane > k) acc = fma(-a[k], yk, acc);
        }
    }
    return acc * invd;
}
template <int N, typename T>
__device__ __forceinline__ T chol_backward(const T (&a)[N], T invd, T y, int lane, int nb = N) {
    // chol_solve's backward sweep runs on acc_k = y_k d_k with the column entries stored as L_ik d_k
    T acc = y / invd;
    T invd2 = invd * invd;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        if (i < nb) {
            T xi = bcast(acc * invd2, i);
            if (lane < i) acc = fma(-a[i], xi, acc);
        }
    }
    return acc * invd2;
}

}  // namespace fx
