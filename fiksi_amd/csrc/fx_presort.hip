// Longest-first hand-out for the grouped kernel (fx_grouped.hip) without any history: a scout pass computes, per System,
// the sum of squared residuals at the start values over the mean square of its variables — on the headline batch the
// logarithm of that number correlates 0.89 with the number of LM trials the System is going to take, and 93 % of the
// Systems that take more than 20 trials lie in its top quarter — and the Systems are handed out in (nearly) descending
// order of it. A batch is as slow as its slowest System plus the time before that System was started; the order changes
// WHEN a System is solved, never its result (Systems are independent).
// Nothing here needs to be exact, so there is no global sort (round 3: a library radix sort, eleven launches and 0.1 ms per
// solve — a tenth of a 12 500-System shard's time): the Systems are dealt into chunks of at most 512 with a stride
// (chunk c holds Systems c, c + nc, c + 2 nc, ...: every chunk is a sample of the whole batch), each chunk is ranked by
// one workgroup in LDS, and position p of the order is entry p / nc of chunk p mod nc — the k-th largest keys of all
// chunks side by side. Two launches, no atomics (a histogram over the keys was tried first: 100 000 atomics on the ~60
// buckets the headline batch's keys fall into took 133 us).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_wave.h"

namespace fx {

#ifndef FX_PS_CHUNK
#define FX_PS_CHUNK 512
#endif
constexpr uint32_t PS_CHUNK = FX_PS_CHUNK;

__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}

// one row of 16 lanes per System
__global__ __launch_bounds__(256) void presort_scout_kernel(DeviceBatch b, float* __restrict__ keys) {
    const uint32_t s = (blockIdx.x * 256u + threadIdx.x) >> 4;
    const uint32_t hl = threadIdx.x & 15u;
    if (s >= b.n_systems) return;
    const uint32_t v0 = b.var_off[s], nvt = b.var_off[s + 1] - v0;
    const uint32_t e0 = b.expr_off[s], net = b.expr_off[s + 1] - e0;
    double sv = 0.0, se = 0.0;
    if (!b.sys_large[s]) {
        for (uint32_t i = hl; i < nvt; i += 16u) {
            const double v = b.vars0[v0 + i];
            sv += v * v;
        }
        for (uint32_t i = hl; i < net; i += 16u) {
            const int tag = b.expr_tag[e0 + i] & 0x7F;
            const ushort4 f4 = reinterpret_cast<const ushort4*>(b.expr_idx)[e0 + i];
            uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
            uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            expand_vars(tag, ff, vars8);
            double v[8], g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = b.vars0[v0 + vars8[e]];
            const double r = eval_expression<double, false>(tag, v, b.expr_param[e0 + i], g);
            se += r * r;
        }
    }
    sv = row16_sum(sv);
    se = row16_sum(se);
    if (hl == 0) {
        double key = se / (sv / (double)(nvt ? nvt : 1u) + 1e-300);
        if (!(key == key) || key > 3.0e38) key = 3.0e38;  // non-finite residuals: first
        keys[s] = (float)key;
    }
}

// workgroup c ranks chunk c (Systems c, c + nc, ...; at most 512) by descending key — every thread counts the entries
// in front of its own — and writes the entry of rank i to position i * nc + c of the order. Keys are non-negative floats, so
// their bit patterns order like the numbers: an entry is the 64-bit word (key bits << 32 | 511 - index), unique per entry, and
// "in front of" is one unsigned compare (two entries per 16-byte LDS read, which every lane reads from the same address).
// (`list`: the n Systems to order are list[0 .. n) — the members of one structure class — instead of 0 .. n)
__global__ __launch_bounds__(PS_CHUNK) void presort_chunk_kernel(uint32_t n, uint32_t nc, const float* __restrict__ keys,
                                                                 uint32_t* __restrict__ order, const uint32_t* __restrict__ list) {
    __shared__ ulonglong2 k2[PS_CHUNK / 2];
    unsigned long long* k = reinterpret_cast<unsigned long long*>(k2);
    const uint32_t t = threadIdx.x, c = blockIdx.x;
    const uint32_t e = c + t * nc;
    const uint32_t s = e < n ? (list ? list[e] : e) : 0xFFFFFFFFu;
    // (the padding: zero — nothing is behind it, and no real entry counts it: a real entry's low word is >= 0 and ties cannot
    // be in front)
    const unsigned long long mine = e < n ? ((unsigned long long)__float_as_uint(keys[s]) << 32) | (unsigned long long)(PS_CHUNK - 1u - t) : 0ull;
    k[t] = mine;
    __syncthreads();
    uint32_t rank = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < PS_CHUNK / 2u; ++j) {
        const ulonglong2 v = k2[j];
        rank += v.x > mine ? 1u : 0u;
        rank += v.y > mine ? 1u : 0u;
    }
    if (e < n) order[rank * nc + c] = s;
}

size_t presort_temp_bytes(uint32_t) { return 16; }

// keys: [n] floats; ids: [2][n], on return ids + n holds the order
hipError_t launch_presort(const DeviceBatch& b, float* keys, uint32_t* ids, void*, size_t, hipStream_t stream) {
    const uint32_t n = b.n_systems;
    if (n == 0) return hipSuccess;
    const uint32_t nc = (n + PS_CHUNK - 1u) / PS_CHUNK;
    hipLaunchKernelGGL(presort_scout_kernel, dim3((n * 16u + 255u) / 256u), dim3(256), 0, stream, b, keys);
    hipLaunchKernelGGL(presort_chunk_kernel, dim3(nc), dim3(PS_CHUNK), 0, stream, n, nc, keys, ids + n, (const uint32_t*)nullptr);
    return hipGetLastError();
}

// the same for parts of the batch: the scout pass over all of it, then list i of `lists` (its Systems at lists + offs[i], counts[i]
// of them) ranked on its own into out + offs[i]
hipError_t launch_presort_lists(const DeviceBatch& b, float* keys, const uint32_t* lists, const uint32_t* offs, const uint32_t* counts, uint32_t n_lists,
                                uint32_t* out, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    hipLaunchKernelGGL(presort_scout_kernel, dim3((b.n_systems * 16u + 255u) / 256u), dim3(256), 0, stream, b, keys);
    for (uint32_t i = 0; i < n_lists; ++i) {
        if (!counts[i]) continue;
        const uint32_t nc = (counts[i] + PS_CHUNK - 1u) / PS_CHUNK;
        hipLaunchKernelGGL(presort_chunk_kernel, dim3(nc), dim3(PS_CHUNK), 0, stream, counts[i], nc, keys, out + offs[i], lists + offs[i]);
    }
    return hipGetLastError();
}

}  // namespace fx
