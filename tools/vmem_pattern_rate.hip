// Micro-benchmark: what a 64-lane 16-byte-per-lane load costs as a function of how many ROWS of a row-major bf16 matrix it
// touches -- the shape of the bf16 GEMMs' operand fetch (gemm_bf16_asm.hip: an LDS-DMA instruction brings 16 rows x 64 B).
// Every block walks its own 256-row x K tile of A (row stride K*2 bytes), one 64-K-value column block (128 B of every row = 32
// instructions of 1 KiB) after the other as the GEMM's K loop does, R rows x (1024/R) bytes per instruction, 8 instructions in flight
// per wave, 4 or 8 waves per block, one block per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/vmem_pattern_rate tools/vmem_pattern_rate.hip && tools/vmem_pattern_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int R>
__global__ __launch_bounds__(512) void walk(const char* __restrict__ A, unsigned* __restrict__ out, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    constexpr int SEG = 1024 / R;                 // bytes of a row one instruction brings (R >= 8: <= 128)
    constexpr int SPR = SEG >= 128 ? 1 : 128 / SEG;      // instructions per row group and column block
    const int lrow = lane / (64 / R), lcol = (lane % (64 / R)) * 16;
    const size_t rs = (size_t)K * 2;
    const char* base = A + (size_t)blockIdx.x * 256 * rs;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int total = (K / 64) * 32;              // instructions of the tile
    for (int i0 = 0; i0 < total; i0 += 8 * nw) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ins = i0 + wave + nw * u;
            const int cb = ins >> 5, i32 = ins & 31;
            size_t off;
            if (R >= 8) off = (size_t)((i32 / SPR) * R + lrow) * rs + (size_t)cb * 128 + (i32 % SPR) * SEG + lcol;
            else off = (size_t)((i32 * R) / (SEG / 128) / 1 % 256 + lrow) * rs + (size_t)cb * 128 + lcol;      // (R < 8: see main)
            v[u] = ins < total ? *reinterpret_cast<const uint4*>(base + off) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    out[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// R = 1: fully contiguous 1 KiB per instruction (a row's K range in 1 KiB steps): the pattern of a K-major walk
__global__ __launch_bounds__(512) void walk_rows(const char* __restrict__ A, unsigned* __restrict__ out, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const size_t rs = (size_t)K * 2;
    const char* base = A + (size_t)blockIdx.x * 256 * rs;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int per_row = (int)(rs / 1024), total = 256 * per_row;
    for (int i0 = 0; i0 < total; i0 += 8 * nw) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ins = i0 + wave + nw * u;
            v[u] = ins < total ? *reinterpret_cast<const uint4*>(base + (size_t)(ins / per_row) * rs + (size_t)(ins % per_row) * 1024 + lane * 16)
                               : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    out[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int R>
static void run(int K, int nw) {
    const int M = 65536;
    const size_t bytes = (size_t)M * K * 2;
    char* A; unsigned* out;
    (void)hipMalloc(&A, bytes); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMemset(A, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&]() {
        if (R == 1) hipLaunchKernelGGL(walk_rows, dim3(256), dim3(64 * nw), 0, 0, (const char*)A, out, K);
        else hipLaunchKernelGGL(walk<(R == 1 ? 8 : R)>, dim3(256), dim3(64 * nw), 0, 0, (const char*)A, out, K);
    };
    for (int r = 0; r < 2; ++r) launch();
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    (void)hipFree(A); (void)hipFree(out);
    const double per_block = 256.0 * K * 2;
    printf("%2d rows x %4d B per instruction, K=%4d, %d waves: %7.1f us per 256-row tile = %6.1f GB/s per CU, %5.1f clk per instruction\n",
           R, 1024 / R, K, nw, ms * 1e3, per_block / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 / (per_block / 1024.0));
}

int main() {
    for (int nw : {4, 8}) { run<16>(2048, nw); run<8>(2048, nw); run<1>(2048, nw); }
    run<16>(512, 8); run<8>(512, 8); run<1>(512, 8);
    return 0;
}
