// Micro-benchmark: VALU fp32 FMA issue rates on this box (tuning aid for the correlation gather, not product code):
// plain v_fma_f32, packed v_pk_fma_f32, v_fmac_f32_dpp row_newbcast, and ds_read_b128 beside them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * 1e-4f;
    __syncthreads();
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i] + 1}; }
    float f = 1.0f + threadIdx.x * 1e-4f, g = 0.5f;
    const float4* lp = reinterpret_cast<const float4*>(sm) + (threadIdx.x & 63) * 5;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], f, g);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (f2){f, f}, (f2){g, g});
        } else if (MODE == 2) {            // one accumulator chain per 16 (as the gather does), dpp source
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile(
                    "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %2, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %2, %3 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %2, %3 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %3 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %2, %3 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %3 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %2, %3 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %2, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                    : "+v"(a[0]), "+v"(a[1]) : "v"(f), "v"(g));
        } else if (MODE == 3) {            // same, ONE dependent chain of 16
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile(
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                    : "+v"(a[0]) : "v"(f), "v"(g));
        } else if (MODE == 4) {            // plain dependent chain of 64 v_fmac (one accumulator)
#pragma unroll
            for (int u = 0; u < 64; ++u) a[0] = __builtin_fmaf(f, g, a[0]);
        } else if (MODE == 5) {            // gather-like: 4 ds_read_b128 + 16 plain FMAs, 4x per iteration
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 v0 = lp[u * 4 + 0], v1 = lp[u * 4 + 1 + (it & 1)], v2 = lp[u * 4 + 2], v3 = lp[u * 4 + 3];
                a[0] = fmaf(v0.x, f, a[0]); a[0] = fmaf(v0.y, f, a[0]); a[0] = fmaf(v0.z, f, a[0]); a[0] = fmaf(v0.w, f, a[0]);
                a[0] = fmaf(v1.x, f, a[0]); a[0] = fmaf(v1.y, f, a[0]); a[0] = fmaf(v1.z, f, a[0]); a[0] = fmaf(v1.w, f, a[0]);
                a[0] = fmaf(v2.x, f, a[0]); a[0] = fmaf(v2.y, f, a[0]); a[0] = fmaf(v2.z, f, a[0]); a[0] = fmaf(v2.w, f, a[0]);
                a[0] = fmaf(v3.x, f, a[0]); a[0] = fmaf(v3.y, f, a[0]); a[0] = fmaf(v3.z, f, a[0]); a[0] = fmaf(v3.w, f, a[0]);
            }
        } else if (MODE == 6) {            // ds_read_b128 only (16 per iteration)
            float4 s = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 16; ++u) { float4 v = lp[u + (it & 1)]; s.x += v.x; }
            a[0] += s.x;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* d, double fma_per_iter_lane, double lds_bytes_per_iter_lane = 0) {
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double lanes = (double)blocks * 256;
    printf("%-44s blocks=%5d %8.3f ms  %7.1f TFLOP/s (FMA=2)  %6.2f lane-FMA/clk/CU @2.4GHz  LDS %7.1f TB/s\n", name, blocks, ms,
           2 * fma_per_iter_lane * iters * lanes / ms / 1e9, fma_per_iter_lane * iters * lanes / (ms * 1e-3) / 2.4e9 / 256,
           lds_bytes_per_iter_lane * iters * lanes / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    for (int b : {1024, 2048}) {
        run<0>("v_fma_f32 x64 (8 chains)", b, d, 64);
        run<1>("v_pk_fma_f32 x64 (8 chains)", b, d, 128);
        run<2>("v_fmac_f32_dpp row_newbcast x64 (2 chains)", b, d, 64);
        run<3>("v_fmac_f32_dpp row_newbcast x64 (1 chain)", b, d, 64);
        run<4>("v_fmac_f32 x64 (1 chain)", b, d, 64);
        run<5>("16 ds_read_b128 + 64 fma (1 chain)", b, d, 64, 256);
        run<6>("16 ds_read_b128 only", b, d, 0, 256);
    }
    return 0;
}
