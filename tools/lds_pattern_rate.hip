// Micro-benchmark: LDS cycles per ds_read_b128 for the gather's lane = window-pixel pattern
// (lane = wj*8 + wi reads 16 B at ((wj*RW + wi)*64 + swizzle) -- 64-byte pixel stride), with / without the XOR swizzle.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
__global__ __launch_bounds__(1024) void k(float* out, int iters, int RW, int mode) {
    __shared__ float4 sm[4096];                      // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wi = lane & 7, wj = lane >> 3;
    const int w = threadIdx.x >> 6;
    unsigned key = mode == 1 ? (wj & 3) : (mode == 2 ? ((wj + (wi >> 2)) & 3) : 0);
    unsigned a0 = (unsigned)((((wj + (w & 3)) * RW + wi + (w >> 2)) * 64) + (key << 4));
    unsigned a1 = a0 ^ 0x10, a2 = a0 ^ 0x20, a3 = a0 ^ 0x30;
    float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
    for (int it = 0; it < iters; ++it) {
        asm volatile(R8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t")
                     "s_waitcnt lgkmcnt(0)\n\t"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a.x + b.y + c.z + d.w;
}
void run(const char* name, float* d, int RW, int mode) {
    const int iters = 512, blocks = 512;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, iters, RW, mode);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, iters, RW, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = 32.0 * iters * blocks * 16;
    printf("%-44s RW=%2d  %8.3f ms  -> %5.2f clk per wave-instruction per CU @2.4GHz\n", name, RW, ms, (ms * 1e-3) * 2.4e9 * 256 / instr);
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 1024 * 4);
    for (int RW : {23, 17, 13, 11, 16}) {
        run("no swizzle", d, RW, 0);
        run("key = ry & 3", d, RW, 1);
        run("key = (ry + (rx>>2)) & 3", d, RW, 2);
    }
    return 0;
}
