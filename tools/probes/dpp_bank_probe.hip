// Does a 64-bit DPP instruction (v_mov_b64_dpp / v_fmac_f64_dpp, row_newbcast) honour bank_mask on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    double v = (double)lane, r = -1.0, acc = 1000.0 + lane, w = 1.0;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0x3" : "+v"(r) : "v"(v));
    out[lane] = r;
    double r2 = -1.0;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\tv_mov_b64_dpp %0, %1 row_newbcast:10 row_mask:0xf bank_mask:0xc" : "+v"(r2) : "v"(v));
    out[64 + lane] = r2;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(v), "v"(w));
    out[128 + lane] = acc;
    double self = 100.0 + lane;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:10 row_mask:0xf bank_mask:0xc" : "+v"(self) : "v"(w));
    out[192 + lane] = self;
}
int main() {
    double* d;
    hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int k = 0; k < 4; ++k) {
        printf("%s:", k == 0 ? "mov bank 0x3 from lane 2 (old -1)" : k == 1 ? "mov halves 2 / 10" : k == 2 ? "fmac halves: 1000+lane - src" : "fmac self halves: 100+lane - src");
        for (int i = 0; i < 32; ++i) printf(" %g", h[64 * k + i]);
        printf("\n");
    }
    return 0;
}
