// Micro-benchmark (tuning aid): how much of the bf16 matrix pipe one wave per SIMD keeps busy when LDS reads / global loads are
// interleaved with its MFMAs, for the two bf16 MFMA shapes (same FLOPs, same memory instructions per iteration):
//   A  8 x v_mfma_f32_16x16x32_bf16 (16 clk each)    B  4 x v_mfma_f32_32x32x16_bf16 (32 clk each)
// variants: 0 = MFMAs only, 1 = + 4 ds_read_b128 per iteration, 2 = + 4 ds_read_b128 + 2 ds_write_b128 + 2 buffer-style global loads
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_shape_issue.hip -o /tmp/mfma_shape_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define RD(n, off) "ds_read_b128 v[" #n ":" #n "+3], %[la] offset:" #off "\n\t"
#define M16(a, s) "v_mfma_f32_16x16x32_bf16 a[" #a ":" #a "+3], v[" #s ":" #s "+3], v[" #s "+4:" #s "+7], a[" #a ":" #a "+3]\n\t"
#define M32(a, s) "v_mfma_f32_32x32x16_bf16 a[" #a ":" #a "+15], v[" #s ":" #s "+3], v[" #s "+4:" #s "+7], a[" #a ":" #a "+15]\n\t"

template <int SHAPE, int VAR>
__global__ __launch_bounds__(256) void k(float* out, const float4* g, int iters) {
    __shared__ float4 sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = make_float4(1e-3f * i, 0.f, 1.f, 2.f);
    __syncthreads();
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)sm + (threadIdx.x & 63) * 16;
    const float4* gp = g + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
            if (VAR == 0)
                asm volatile(M16(0, 40) M16(4, 48) M16(8, 40) M16(12, 48) M16(16, 40) M16(20, 48) M16(24, 40) M16(28, 48)
                             ::[la] "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71");
            else if (VAR == 1)
                asm volatile(M16(0, 40) RD(56, 0) M16(4, 48) M16(8, 40) RD(60, 1024) M16(12, 48) M16(16, 40) RD(64, 2048) M16(20, 48) M16(24, 40) RD(68, 3072) M16(28, 48) "s_waitcnt lgkmcnt(0)\n\t"
                             ::[la] "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71");
            else
                asm volatile(M16(0, 40) RD(56, 0) M16(4, 48) "global_load_dwordx4 v[72:75], %[gp], off\n\t" M16(8, 40) RD(60, 1024) M16(12, 48) "ds_write_b128 %[la], v[40:43] offset:8192\n\t"
                             M16(16, 40) RD(64, 2048) M16(20, 48) "global_load_dwordx4 v[76:79], %[gp], off offset:2048\n\t" M16(24, 40) RD(68, 3072) M16(28, 48) "ds_write_b128 %[la], v[44:47] offset:12288\n\t" "s_waitcnt lgkmcnt(0) vmcnt(0)\n\t"
                             ::[la] "v"(la), [gp] "v"(gp) : "memory","a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
        } else {
            if (VAR == 0)
                asm volatile(M32(0, 40) M32(16, 48) M32(32, 40) M32(48, 48)
                             ::[la] "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71");
            else if (VAR == 1)
                asm volatile(M32(0, 40) RD(56, 0) M32(16, 48) RD(60, 1024) M32(32, 40) RD(64, 2048) M32(48, 48) RD(68, 3072) "s_waitcnt lgkmcnt(0)\n\t"
                             ::[la] "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71");
            else
                asm volatile(M32(0, 40) RD(56, 0) "global_load_dwordx4 v[72:75], %[gp], off\n\t" M32(16, 48) RD(60, 1024) "ds_write_b128 %[la], v[40:43] offset:8192\n\t"
                             M32(32, 40) RD(64, 2048) "global_load_dwordx4 v[76:79], %[gp], off offset:2048\n\t" M32(48, 48) RD(68, 3072) "ds_write_b128 %[la], v[44:47] offset:12288\n\t" "s_waitcnt lgkmcnt(0) vmcnt(0)\n\t"
                             ::[la] "v"(la), [gp] "v"(gp) : "memory","a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
        }
    }
    if (iters < 0) out[threadIdx.x] = sm[threadIdx.x].x;
}

template <int SHAPE, int VAR>
void run(const char* name, float* d, const float4* g) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, VAR>), dim3(blocks), dim3(256), 0, 0, d, g, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, VAR>), dim3(blocks), dim3(256), 0, 0, d, g, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flops = (double)blocks * 4 * iters * 8 * 16384.0;        // 8 x 16x16x32 = 4 x 32x32x16 = 131 072 FLOP per wave and iteration
    printf("%-64s %8.3f ms  %7.1f TF  (%.1f clk per iteration at 2.39 GHz; MFMA time 128)\n", name, ms, flops / ms / 1e9, ms * 1e-3 * 2.39e9 / iters);
}

int main() {
    float* d; float4* g;
    hipMalloc(&d, 1 << 20); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    run<0, 0>("16x16x32 x8, MFMAs only", d, g);
    run<1, 0>("32x32x16 x4, MFMAs only", d, g);
    run<0, 1>("16x16x32 x8 + 4 ds_read_b128", d, g);
    run<1, 1>("32x32x16 x4 + 4 ds_read_b128", d, g);
    run<0, 2>("16x16x32 x8 + 4 ds_read_b128 + 2 ds_write_b128 + 2 global_load", d, g);
    run<1, 2>("32x32x16 x4 + 4 ds_read_b128 + 2 ds_write_b128 + 2 global_load", d, g);
    return 0;
}
