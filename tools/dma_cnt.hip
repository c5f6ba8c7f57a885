// Micro-benchmark: is an LDS-DMA load (buffer_load_dwordx4 ... lds) counted by lgkmcnt as well as vmcnt on gfx950?
// If `s_waitcnt lgkmcnt(0)` right behind the DMA issue takes as long as `vmcnt(0)`, every LDS-read wait of a wave
// also waits for that wave's DMA in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ __launch_bounds__(256) void k(const float* src, long long* out, int stride) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(smem + (wave * 4 + i) * 1024), 16,
                                                 (int)(((blockIdx.x * 16 + wave * 4 + i) * 64 + lane) * stride), 0, 0, 0);
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { long long* o = out + (blockIdx.x * 4 + wave) * 4; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = 0; }
}
int main() {
    float* src; long long* out;
    const int blocks = 256;
    (void)hipMalloc(&src, 1ull << 30); (void)hipMemset(src, 0, 1ull << 30);
    (void)hipMalloc(&out, blocks * 16 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, src, out, 512);
        (void)hipDeviceSynchronize();
    }
    static long long h[256 * 16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < blocks * 4; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; }
    printf("mean s_memtime ticks: issue %.0f | s_waitcnt lgkmcnt(0) %.0f | then vmcnt(0) %.0f\n", a / (blocks * 4), b / (blocks * 4), c / (blocks * 4));
    return 0;
}
