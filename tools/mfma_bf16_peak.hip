// Micro-benchmark: v_mfma_f32_32x32x16_bf16 rate alone and fed from LDS at the split-bf16 kernel's
// ratio (12 ds_read_b128 per 24 MFMAs, optionally 6 ds_write_b128 on top).  Tuning aid, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0 pure MFMA, 1 + 12 reads / 24 MFMA, 2 + reads + 6 writes / 24 MFMA
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 61440 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = 1e-3f * (i & 255);
    __syncthreads();
    uint4 f[12];
    for (int i = 0; i < 12; ++i) f[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const char* base = sm + ((wave & 1) * 64 + (lane & 31)) * 80 + (lane >> 5) * 16;
    char* wbase = sm + 30720 + (threadIdx.x >> 2) * 80 + (threadIdx.x & 3) * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) f[i] = *reinterpret_cast<const uint4*>(base + (i % 6) * 5120 + (i / 6) * 2560 + (it & 1) * 32);
        }
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&f[(p % 3) * 2 + (i >> 1)]),
                                                                 *reinterpret_cast<bf16x8*>(&f[6 + (p / 2) * 2 + (i & 1)]), acc[i], 0, 0, 0);
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<uint4*>(wbase + i * 5120) = f[i];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* d) {
    const int iters = 4096;
    auto kern = k<MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 61440, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 61440, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * 24 * 32768.0;
    printf("%-40s blocks=%5d  %8.3f ms  %7.1f TF (bf16)  = %6.1f TF fp32-equivalent at 6 MFMAs per product\n", name, blocks, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 6);
}

int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    run<0>("pure MFMA, 1 wave/SIMD", 256, d);
    run<0>("pure MFMA, 2 waves/SIMD", 512, d);
    run<1>("12 ds_read_b128 / 24 MFMA, 1 w/SIMD", 256, d);
    run<1>("12 ds_read_b128 / 24 MFMA, 2 w/SIMD", 512, d);
    run<2>("+ 6 ds_write_b128, 1 w/SIMD", 256, d);
    run<2>("+ 6 ds_write_b128, 2 w/SIMD", 512, d);
    return 0;
}
