// Longest-first hand-out for the grouped kernel (fx_grouped.hip) without any history: a scout pass computes, per System,
// the sum of squared residuals at the start values over the mean square of its variables — on the headline batch the
// logarithm of that number correlates 0.89 with the number of LM trials the System is going to take, and 93 % of the
// Systems that take more than 20 trials lie in its top quarter — and a radix sort turns it into the order in which
// rows take Systems from the queue. A batch is as slow as its slowest System plus the time before that System was
// started; the order changes WHEN a System is solved, never its result (Systems are independent).
// Nothing here needs to be exact: the key is a float, summed in whatever order is fastest.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "fx_device.h"
#include "fx_expr.h"
#include "fx_wave.h"

namespace fx {

__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}

// one row of 16 lanes per System
__global__ __launch_bounds__(256) void presort_scout_kernel(DeviceBatch b, float* __restrict__ keys, uint32_t* __restrict__ ids) {
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
        ids[s] = s;
    }
}

size_t presort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const uint32_t*)nullptr,
                                                       (uint32_t*)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

// keys / ids: [2][n] each (in, out); on return ids + n holds the order
hipError_t launch_presort(const DeviceBatch& b, float* keys, uint32_t* ids, void* temp, size_t temp_bytes, hipStream_t stream) {
    const uint32_t n = b.n_systems;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(presort_scout_kernel, dim3((n * 16u + 255u) / 256u), dim3(256), 0, stream, b, keys, ids);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, keys, keys + n, ids, ids + n, (int)n, 0, 32, stream);
}

}  // namespace fx
