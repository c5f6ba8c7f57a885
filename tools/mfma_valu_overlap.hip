// Micro-benchmark: do VALU instructions overlap with dense bf16 MFMAs on one SIMD?  512-thread blocks (2 waves per SIMD),
// one block per CU.  Mode A: all 8 waves run MFMAs; V: all run packed-FMA chains; AV: waves 0-3 MFMA, waves 4-7 VALU
// (each SIMD holds one of each); S: every wave alternates 2 MFMAs with 16 VALU ops in ONE instruction stream.
// Tuning aid (tools/experiments/README.md), not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, bool PK>   // 0 A, 1 V, 2 AV, 3 S;  PK: packed or scalar fp32 FMAs
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 fa = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u), fb = fa;
    f2 v[8];
    float w[16];
    for (int i = 0; i < 8; ++i) v[i] = (f2){1.0f + i, 0.5f * threadIdx.x};
    for (int i = 0; i < 16; ++i) w[i] = 1.0f + i + threadIdx.x;
    const f2 c = (f2){0.999f, 0.999f}, d = (f2){1e-3f, 1e-3f};
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acc[(2 * p) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fa), *reinterpret_cast<bf16x8*>(&fb), acc[(2 * p) & 3], 0, 0, 0);
                acc[(2 * p + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fa), *reinterpret_cast<bf16x8*>(&fb), acc[(2 * p + 1) & 3], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (PK) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = v[i] * c + d;
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) w[i] = w[i] * 0.999f + 1e-3f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fa), *reinterpret_cast<bf16x8*>(&fb), acc[p & 3], 0, 0, 0);
            }
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (PK) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = v[i] * c + d;
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) w[i] = w[i] * 0.999f + 1e-3f;
                    }
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + w[i] + w[i + 8];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, bool PK>
float run(const char* name, float* d) {
    const int iters = 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PK>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, PK>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-70s %8.3f ms\n", name, ms);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, true>("A : 8 waves x 8 MFMA / iteration (2 MFMA waves per SIMD)", d);
    run<1, true>("V : 8 waves x 64 v_pk_fma_f32 / iteration (2 VALU waves per SIMD)", d);
    run<2, true>("AV: per SIMD one wave 8 MFMA, one wave 64 v_pk_fma_f32 per iteration", d);
    run<3, true>("S : every wave 8 MFMA + 64 v_pk_fma_f32 interleaved (2 such waves per SIMD)", d);
    run<1, false>("V': 8 waves x 64 v_fma_f32 / iteration", d);
    run<2, false>("AV': per SIMD one wave 8 MFMA, one wave 64 v_fma_f32 per iteration", d);
    run<3, false>("S': every wave 8 MFMA + 64 v_fma_f32 interleaved", d);
    printf("(overlap: AV ~ max(A/2, V/2);  none: AV ~ A/2 + V/2.   S: overlap ~ max(A, V); none ~ A + V)\n");
    return 0;
}
