// Micro-benchmark: how fast can the 256 CUs pull the OPERAND BYTES of the config-3 channel-mix GEMMs out of L2 / Infinity Cache,
// with no arithmetic and no LDS at all?  The bf16 GEMMs of BASELINE configs[2] are short-K problems (up-projection M=16384,
// N=2048, K=512; down-projection N=512, K=2048): a 256x128 tile needs (256+128) x K x 2 bytes of operands per 2 x 256 x 128 x K
// flops = 85 flop/byte, so at the 2.5 PFLOP/s bf16 peak the tiles would have to be fed at 29 TB/s.  This kernel walks exactly
// the tiles of gemm_bf16_asm.hip (persistent blocks, 1024 or 256 tiles, the same tile -> (row block, column block) order), loading
// each tile's A rows and W rows with 16-byte-per-lane loads (8 in flight per lane) and folding them into a checksum.
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_stream_rate tools/l2_stream_rate.hip && tools/l2_stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void stream_tiles(const uint4* __restrict__ A, const uint4* __restrict__ W, unsigned* __restrict__ out,
                                                    int tiles_n, int tiles_total, int K, int order) {
    // a tile: A rows [256*tm, +256) x K bf16 (row-major, K contiguous), W rows [128*tn, +128) x K bf16
    const int tid = threadIdx.x;
    const size_t a_vec = (size_t)256 * K / 8, w_vec = (size_t)128 * K / 8;      // uint4 (8 bf16) per tile operand
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int tiles_m = tiles_total / tiles_n;
    const int per_block = (tiles_total + gridDim.x - 1) / gridDim.x;
    for (int k = 0; k < per_block; ++k) {
        // order 0: the kernels' own -- block b walks tiles per_block*b .. (consecutive), a tile's row block = t % tiles_m (m fastest);
        //          the column tiles of one row block then sit on ONE XCD (64 row blocks, block -> XCD b % 8) at the same time
        // order 1: n fastest, tile t = b + k * grid: the column tiles of a row block on different XCDs
        const int t = order == 0 ? blockIdx.x * per_block + k : blockIdx.x + k * (int)gridDim.x;
        if (t >= tiles_total) break;
        const int tm = order == 0 ? t % tiles_m : t / tiles_n, tn = order == 0 ? t / tiles_m : t % tiles_n;
        const uint4* a = A + (size_t)tm * a_vec;
        const uint4* w = W + (size_t)tn * w_vec;
        for (size_t i = tid; i < a_vec; i += 512 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + 512 * u < a_vec ? a[i + 512 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        for (size_t i = tid; i < w_vec; i += 512 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + 512 * u < w_vec ? w[i + 512 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    out[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

static void run(const char* name, int M, int N, int K, int blocks, int order) {
    const size_t a_bytes = (size_t)M * K * 2, w_bytes = (size_t)N * K * 2;
    void *A, *W; unsigned* out;
    (void)hipMalloc(&A, a_bytes); (void)hipMalloc(&W, w_bytes); (void)hipMalloc(&out, 1024 * 512 * 4);
    (void)hipMemset(A, 1, a_bytes); (void)hipMemset(W, 2, w_bytes);
    const int tiles_n = N / 128, tiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(stream_tiles, dim3(blocks), dim3(512), 0, 0, (const uint4*)A, (const uint4*)W, out, tiles_n, tiles, K, order);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_tiles, dim3(blocks), dim3(512), 0, 0, (const uint4*)A, (const uint4*)W, out, tiles_n, tiles, K, order);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)tiles * (256 + 128) * K * 2;
    printf("%-32s M=%5d N=%4d K=%4d  %4d tiles on %3d blocks: %6.1f MB of tile operands in %6.1f us = %5.2f TB/s   (unique bytes %5.1f MB)\n",
           name, M, N, K, tiles, blocks, bytes / 1e6, ms * 1e3, bytes / (ms * 1e-3) / 1e12, (a_bytes + w_bytes) / 1e6);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(out);
}

int main() {
    run("up-projection, kernel order", 16384, 2048, 512, 256, 0);
    run("up-projection, n fastest", 16384, 2048, 512, 256, 1);
    run("down-projection, kernel order", 16384, 512, 2048, 256, 0);
    run("down-projection, n fastest", 16384, 512, 2048, 256, 1);
    return 0;
}
