// Micro-benchmark: what clock (and, with rocm-smi sampled beside it, what package power) the matrix cores get when every SIMD issues
// MFMAs back to back -- v_mfma_f32_16x16x32_bf16, v_mfma_f32_32x32x16_bf16 (the two bf16 shapes of the product) and the exact-fp32
// v_mfma_f32_32x32x2_f32.  Each wave stamps s_memtime (shader clocks) and s_memrealtime (100 MHz) around its loop: the ratio is the
// clock it actually ran at.  Tuning aid, not product code.   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_power tools/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 16x16x32 bf16, 1: 32x32x16 bf16, 2: 32x32x2 f32
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u};
    float s = 0.f;
    if (SHAPE == 0) {
        // eight independent accumulators, written out: left to hipcc the loop's accumulators overlapped by two registers (a[24:27] <- a[22:25]),
        // every MFMA depending on its neighbour
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n\t"
                         "v_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %8, %9, %3\n\t"
                         "v_mfma_f32_16x16x32_bf16 %4, %8, %9, %4\n\tv_mfma_f32_16x16x32_bf16 %5, %8, %9, %5\n\t"
                         "v_mfma_f32_16x16x32_bf16 %6, %8, %9, %6\n\tv_mfma_f32_16x16x32_bf16 %7, %8, %9, %7"
                         : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7])
                         : "v"(a), "v"(b));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else if (SHAPE == 1) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a), *reinterpret_cast<bf16x8*>(&b), acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int SHAPE>
void run(const char* name, double flop_per_mfma, int per_iter, float* d, unsigned long long* st, int blocks, double seconds) {
    // size the loop for ~20 ms per launch, then launch back to back for `seconds`
    const int iters = SHAPE == 2 ? 60000 : (SHAPE == 1 ? 120000 : 120000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, st, iters);
    hipDeviceSynchronize();
    int n = 0; float ms = 0.f;
    hipEventRecord(e0);
    do {
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, st, iters);
        n += 4;
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    } while (ms < seconds * 1e3);
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int w = 0; w < blocks * 4; ++w) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)blocks * 4 * iters * per_iter * flop_per_mfma * n;
    printf("%-28s %d waves/SIMD  %6.1f s  %8.1f TFLOP/s   shader clock median %.2f GHz (min %.2f, max %.2f)  -> peak at that clock %.0f TFLOP/s\n", name,
           blocks / 256, ms / 1e3, flops / ms / 1e9, ghz[ghz.size() / 2], ghz.front(), ghz.back(),
           256.0 * 4 * (flop_per_mfma / (SHAPE == 0 ? 16.0 : (SHAPE == 1 ? 32.0 : 64.0))) * ghz[ghz.size() / 2] / 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    float* d; unsigned long long* st;
    hipMalloc(&d, 1024 * 256 * 4); hipMalloc(&st, 1024 * 4 * 2 * 8);
    run<0>("v_mfma_f32_16x16x32_bf16", 16384.0, 8, d, st, 256, seconds);
    run<0>("v_mfma_f32_16x16x32_bf16", 16384.0, 8, d, st, 512, seconds);
    run<1>("v_mfma_f32_32x32x16_bf16", 32768.0, 4, d, st, 256, seconds);
    run<1>("v_mfma_f32_32x32x16_bf16", 32768.0, 4, d, st, 512, seconds);
    run<2>("v_mfma_f32_32x32x2_f32", 4096.0, 4, d, st, 256, seconds);
    run<2>("v_mfma_f32_32x32x2_f32", 4096.0, 4, d, st, 512, seconds);
    return 0;
}
