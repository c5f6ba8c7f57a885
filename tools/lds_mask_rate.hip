// Micro-benchmark: does an exec-masked ds_read_b128 cost fewer LDS cycles than a full one on gfx950?
// (tuning aid for the gather: partial re-loads of a register-resident window)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X X X X X X X X
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long mask, int stride16) {
    __shared__ float4 sm[4096];                      // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // conflict-free: 16 consecutive lanes -> 16 consecutive 16-byte slots
    unsigned addr = (unsigned)(((threadIdx.x >> 6) * 64 + lane * stride16) & 4095) * 16;
    float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iters; ++it) {
            asm volatile(R8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t")
                         "s_waitcnt lgkmcnt(0)\n\t"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a.x + b.y + c.z + d.w;
}
void run(const char* name, float* d, unsigned long long mask, int stride16 = 1) {
    const int iters = 512, blocks = 512;            // 2 blocks of 16 waves per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, iters, mask, stride16);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, d, iters, mask, stride16);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = 32.0 * iters * blocks * 16;             // wave-level ds_read_b128 instructions
    const double clk_per_instr_per_cu = (ms * 1e-3) * 2.4e9 * 256 / instr;
    printf("%-44s %8.3f ms  -> %5.2f clk per wave-instruction per CU @2.4GHz (%d active lanes)\n", name, ms, clk_per_instr_per_cu,
           __builtin_popcountll(mask));
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 1024 * 4);
    run("all 64 lanes", d, ~0ull);
    run("lanes 0-31", d, 0xffffffffull);
    run("lanes 0-15", d, 0xffffull);
    run("lanes 16-31", d, 0xffff0000ull);
    run("lanes 0-7", d, 0xffull);
    run("every 2nd lane (32)", d, 0x5555555555555555ull);
    run("every 4th lane (16)", d, 0x1111111111111111ull);
    run("every 8th lane (8)", d, 0x0101010101010101ull);
    run("2 of every 8 lanes (16)", d, 0x0303030303030303ull);
    run("lanes 0-7 of each 16 (32)", d, 0x00ff00ff00ff00ffull);
    run("rows: lanes 0-7,32-39 (16)", d, 0x000000ff000000ffull);
    run("all 64, same address (broadcast)", d, ~0ull, 0);
    return 0;
}
