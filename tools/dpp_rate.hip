// Micro-benchmark: cost of getting a wave-uniform operand into an FMA on gfx950 (tuning aid for the gather).
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, f = 1.0f + threadIdx.x * 1e-4f, g = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)        // plain VGPR operands
            asm volatile(R8(R8("v_fmac_f32 %0, %2, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g));
        else if (MODE == 1)   // row_newbcast
            asm volatile(R8(R8("v_fmac_f32_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g));
        else if (MODE == 2)   // quad_perm broadcast
            asm volatile(R8(R8("v_fmac_f32_dpp %0, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g));
        else if (MODE == 3)   // row_shr
            asm volatile(R8(R8("v_fmac_f32_dpp %0, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g));
        else if (MODE == 4)   // readlane + SGPR operand
            asm volatile(R8(R8("v_readlane_b32 s40, %2, 5\n\tv_fmac_f32 %0, s40, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g) : "s40");
        else if (MODE == 5)   // SGPR operand only (already uniform)
            asm volatile("v_readlane_b32 s40, %2, 5\n\t" R8(R8("v_fmac_f32 %0, s40, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g) : "s40");
        else if (MODE == 6)   // 1 readlane per 2 fmac
            asm volatile(R8(R8("v_readlane_b32 s40, %2, 5\n\tv_fmac_f32 %0, s40, %3\n\tv_fmac_f32 %1, s40, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g) : "s40");
        else if (MODE == 7)   // readfirstlane
            asm volatile(R8(R8("v_readfirstlane_b32 s40, %2\n\tv_fmac_f32 %0, s40, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g) : "s40");
        else if (MODE == 8)   // dpp mov then plain fma  (2 instr)
            asm volatile(R8(R8("v_mov_b32_dpp %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %1, %3\n\t")) : "+v"(a0), "+v"(a1) : "v"(f), "v"(g));
        else if (MODE == 9)   // SDWA? no; v_pk_fma with sgpr pair
            asm volatile("v_readlane_b32 s40, %2, 5\n\tv_readlane_b32 s41, %2, 6\n\t" R8(R8("v_pk_fma_f32 %0, s[40:41], %1, %0\n\t")) : "+v"(*(double*)&a0) : "v"(*(double*)&a1), "v"(f) : "s40", "s41");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1;
}
template <int MODE>
void run(const char* name, float* d, double fma_per_iter) {
    const int iters = 2048, blocks = 1024;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s %8.3f ms  -> %6.2f lane-FMA/clk/CU @2.4GHz\n", name, ms, fma_per_iter * iters * blocks * 256.0 / (ms * 1e-3) / 2.4e9 / 256);
}
int main() {
    float* d; (void)hipMalloc(&d, 1024 * 256 * 4);
    run<0>("v_fmac vgpr,vgpr (1 chain)", d, 64);
    run<1>("v_fmac_dpp row_newbcast", d, 64);
    run<2>("v_fmac_dpp quad_perm bcast", d, 64);
    run<3>("v_fmac_dpp row_shr:1", d, 64);
    run<4>("readlane + fmac(sgpr)  [1:1]", d, 64);
    run<5>("fmac(sgpr) only", d, 64);
    run<6>("readlane + 2 fmac(sgpr) [1:2]", d, 128);
    run<7>("readfirstlane + fmac(sgpr) [1:1]", d, 64);
    run<8>("v_mov_dpp + fmac [1:1]", d, 64);
    run<9>("v_pk_fma sgpr pair (2 FMA/instr)", d, 128);
    return 0;
}
