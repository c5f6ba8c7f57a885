// Micro-benchmark: LDS cycles per ds_read_b128 for MFMA fragment reads (lane -> row l31 of a 32-row operand slab, 16 bytes at
// K slot 2*kq + half) under candidate row strides / swizzles of the bf16 GEMM's LDS image.
//   mode 0: 64-byte rows,  phys slot = slot ^ ((row >> 2) & 3)     (gemm_bf16_asm.hip today: 32 K values per stage)
//   mode 1: 128-byte rows, phys slot = slot ^ ((row >> 1) & 7)
//   mode 2: 128-byte rows, phys slot = slot ^ (row & 7)
//   mode 3: 128-byte rows, phys slot = slot ^ ((row >> 2) & 7)
//   mode 4: 128-byte rows, no swizzle
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    __shared__ float4 sm[8192];                      // 128 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, w = threadIdx.x >> 6;
    unsigned a[4];
    for (int kq = 0; kq < 4; ++kq) {
        const int row = w * 32 + l31;
        const int slot = 2 * (kq & (mode == 0 ? 1 : 3)) + half;
        unsigned addr;
        if (mode == 0) addr = row * 64 + (((slot & 3) ^ ((row >> 2) & 3)) * 16) + (kq >> 1) * 24576;
        else {
            const int f = mode == 1 ? (row >> 1) & 7 : mode == 2 ? row & 7 : mode == 3 ? (row >> 2) & 7 : 0;
            addr = row * 128 + ((slot ^ f) * 16);
        }
        a[kq] = addr;
    }
    float4 p = make_float4(0, 0, 0, 0), q = p, r = p, t = p;
    for (int it = 0; it < iters; ++it) {
        asm volatile(R8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t")
                     "s_waitcnt lgkmcnt(0)\n\t"
                     : "=v"(p), "=v"(q), "=v"(r), "=v"(t) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = p.x + q.y + r.z + t.w;
}
void run(const char* name, float* d, int mode) {
    const int iters = 512, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = 32.0 * iters * 8;           // wave-level instructions per CU (one block of 8 waves per CU)
    printf("%-52s %8.3f ms  -> %5.2f clk per wave-instruction per CU @2.4GHz\n", name, ms, (ms * 1e-3) * 2.4e9 / instr);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run("0  64-B rows, slot ^ ((row>>2)&3)   [today]", d, 0);
    run("1 128-B rows, slot ^ ((row>>1)&7)", d, 1);
    run("2 128-B rows, slot ^ (row&7)", d, 2);
    run("3 128-B rows, slot ^ ((row>>2)&7)", d, 3);
    run("4 128-B rows, no swizzle", d, 4);
    return 0;
}
