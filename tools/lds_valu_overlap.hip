// Micro-benchmark: do LDS fragment reads (ds_read_b128, the gather's lane = window-pixel pattern) and packed fp32 FMAs overlap on a
// gfx950 SIMD?  16 waves per CU (4 per SIMD), the tiled gather's occupancy.  Per iteration a wave does G groups of
// {4 ds_read_b128, 8 v_pk_fma_f32} -- the gather's work per (slot, level, 16-channel phase).
//   mode 0: reads only            mode 1: FMAs only
//   mode 2: reads, wait, FMAs on the data just read (the gather's order)
//   mode 3: reads for the NEXT group in flight under the FMAs of this one (software pipeline inside the wave)
//   mode 4: even waves reads only, odd waves FMAs only (the pipes fed by different waves of the same SIMD)
// hipcc --offload-arch=gfx950 -O3 -o tools/lds_valu_overlap tools/lds_valu_overlap.hip && tools/lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
// fixed registers: fragment set 1 = v[64:79], set 2 = v[80:95], accumulators v[96:103], multiplier v[104:105]
#define RD4(B) "ds_read_b128 v[" #B ":" #B "+3], %[a0]\n\tds_read_b128 v[" #B "+4:" #B "+7], %[a0] offset:16\n\t" \
               "ds_read_b128 v[" #B "+8:" #B "+11], %[a0] offset:32\n\tds_read_b128 v[" #B "+12:" #B "+15], %[a0] offset:48\n\t"
#define FM8(B)                                                                                                            \
    "v_pk_fma_f32 v[96:97], v[" #B ":" #B "+1], v[104:105], v[96:97]\n\tv_pk_fma_f32 v[98:99], v[" #B "+2:" #B "+3], v[104:105], v[98:99]\n\t"        \
    "v_pk_fma_f32 v[100:101], v[" #B "+4:" #B "+5], v[104:105], v[100:101]\n\tv_pk_fma_f32 v[102:103], v[" #B "+6:" #B "+7], v[104:105], v[102:103]\n\t" \
    "v_pk_fma_f32 v[96:97], v[" #B "+8:" #B "+9], v[104:105], v[96:97]\n\tv_pk_fma_f32 v[98:99], v[" #B "+10:" #B "+11], v[104:105], v[98:99]\n\t"     \
    "v_pk_fma_f32 v[100:101], v[" #B "+12:" #B "+13], v[104:105], v[100:101]\n\tv_pk_fma_f32 v[102:103], v[" #B "+14:" #B "+15], v[104:105], v[102:103]\n\t"
#define CLOB "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",                     \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105"
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    __shared__ float4 sm[4096];                      // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = make_float4(1e-3f * i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wi = lane & 7, wj = lane >> 3, w = threadIdx.x >> 6;
    const unsigned a0 = (unsigned)(((wj + (w & 3)) * 24 + wi + (w >> 2)) * 80);          // padded rows, 64-byte pixels
    const bool reader = !(w & 1);
    float res;
    asm volatile(
        "v_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 1.0\n\t"
        "v_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\tv_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        RD4(64) RD4(80) "s_waitcnt lgkmcnt(0)\n\t" :: [a0] "v"(a0) : CLOB);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 4 && reader))
            asm volatile(RD4(64) RD4(80) "s_waitcnt lgkmcnt(0)\n\t" :: [a0] "v"(a0) : CLOB);
        else if (MODE == 1 || MODE == 4)
            asm volatile(FM8(64) FM8(80) :: [a0] "v"(a0) : CLOB);
        else if (MODE == 2)
            asm volatile(RD4(64) "s_waitcnt lgkmcnt(0)\n\t" FM8(64) RD4(80) "s_waitcnt lgkmcnt(0)\n\t" FM8(80) :: [a0] "v"(a0) : CLOB);
        else   // MODE 3: set 2 was requested last round; request set 1 now, work on set 2, then the other way round
            asm volatile(RD4(64) "s_waitcnt lgkmcnt(4)\n\t" FM8(80) RD4(80) "s_waitcnt lgkmcnt(4)\n\t" FM8(64) :: [a0] "v"(a0) : CLOB);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, v96, v99\n\tv_add_f32 %0, %0, v100\n\tv_add_f32 %0, %0, v103\n\tv_add_f32 %0, %0, v64\n\tv_add_f32 %0, %0, v95"
                 : "=v"(res) :: CLOB);
    out[blockIdx.x * 1024 + threadIdx.x] = res;
}
template <int MODE>
void run(const char* name, float* d) {
    const int iters = 2048, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // per SIMD and iteration: 4 waves x 2 groups (mode 4: 2 reader waves + 2 FMA waves)
    printf("%-64s %8.3f ms  -> %7.1f clk per {4 reads + 8 pk_fma} group per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    run<0>("0 reads only (8 groups per SIMD-iteration)", d);
    run<1>("1 FMAs only", d);
    run<2>("2 reads, wait, FMAs (gather order)", d);
    run<3>("3 next group's reads under this group's FMAs", d);
    run<4>("4 even waves read, odd waves FMA (half the work of 0 + 1)", d);
    return 0;
}
