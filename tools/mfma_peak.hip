// Micro-benchmark: achievable v_mfma_f32_32x32x2_f32 rate on this box (tuning aid, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sm[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if (LDS) { for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i * 1e-4f; __syncthreads(); }
    for (int it = 0; it < iters; ++it) {
        if (LDS) { float4 v = *reinterpret_cast<float4*>(&sm[((threadIdx.x * 4 + it * 16) & 4092)]); a = v.x; b = v.y; }
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks, float* d) {
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    run<4, false>("4 acc, 1 wave/SIMD", 256, d);
    run<4, false>("4 acc, 2 waves/SIMD", 512, d);
    run<4, false>("4 acc, 4 waves/SIMD", 1024, d);
    run<1, false>("1 acc (dependent), 1 w/SIMD", 256, d);
    run<1, false>("1 acc (dependent), 2 w/SIMD", 512, d);
    run<2, false>("2 acc, 1 w/SIMD", 256, d);
    run<4, true>("4 acc + ds_read, 1 w/SIMD", 256, d);
    run<4, true>("4 acc + ds_read, 2 w/SIMD", 512, d);
    return 0;
}
