// Probe (gfx950): DPP row_newbcast and ds_bpermute semantics on wave64, as used by the 2-D Cholesky.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__global__ void probe(int* out) {
    int lane = threadIdx.x;
    out[lane] = dpp<0x150 + 0>(lane * 10);
    out[64 + lane] = dpp<0x150 + 5>(lane * 10);
    out[128 + lane] = dpp<0x150 + 15>(lane * 10);
    out[192 + lane] = __builtin_amdgcn_ds_bpermute((16 * 2 + (lane & 15)) * 4, lane * 10);  // from DPP row 2, same t
    double d = 1.5 * lane;
    int lo = dpp<0x150 + 7>(__double2loint(d)), hi = dpp<0x150 + 7>(__double2hiint(d));
    out[256 + lane] = (int)__hiloint2double(hi, lo);
}
int main() {
    int* d; hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 5; ++s) { for (int i = 0; i < 64; i += 7) printf("%d:%d ", i, h[64 * s + i]); printf("\n"); }
    return 0;
}
