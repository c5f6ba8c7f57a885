// bf16 x bf16 up-projection of the large-batch channel mix (BASELINE config 3: M = B*N*8 >= 16384 rows, K = 512,
// GELU, bf16 output) as PERSISTENT blocks whose tile body is one generated assembly statement
// (gemm_bf16_tile_asm.inc <- tools/gen_gemm_bf16_asm.py).
//
// C[M,N] = bf16(gelu(bf16(A W^T + bias))): both operands bf16 in memory; fp32 accumulation on
// v_mfma_f32_32x32x16_bf16; the Linear's output is rounded to bf16 before the GELU, as under autocast.
// Block = 8 waves, 256x128 tile, 64x64 per wave.  Operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction) into a ring of three super-stages of 2 x 32 K values that runs three super-stages ahead and across
// tile boundaries; one s_barrier per 64 K values; one rolling fragment set; the finished tile's accumulators are parked
// as bf16 pairs (32 registers) and its GELU / conversion / 16-byte stores are issued between the MFMA pairs of the next
// tile.  Why assembly: see tools/experiments/README.md (gemm_bf16_dma.hip) -- in C++ the same loop either carries ~25
// scalar branches per 64 K values (as many clocks as the MFMAs) or, written branch-free, spills.
// LDS rows are unpadded (the DMA writes lane-linear), XOR-swizzled: phys slot = slot ^ ((row >> 2) & 3), applied to
// the per-lane global source address and to the fragment reads (conflict-free 16-lane ds_read_b128 groups).
// GELU column order: the W rows are fetched from LDS permuted (gelu_col) so that a lane's registers 8q..8q+7 are eight
// consecutive output columns: 16-byte stores, 32 contiguous bytes per row and instruction.
#include "common.h"

#include <cstdlib>

#include "gemm_bf16_tile_asm.inc"

namespace pips {

typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// wave-uniform values as scalar registers (the compiler cannot always prove uniformity of an "s" asm operand)
__device__ __forceinline__ unsigned sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
template <typename T>
__device__ __forceinline__ const T* sgpr(const T* ptr) {
    const unsigned long long v = (unsigned long long)reinterpret_cast<uintptr_t>(ptr);
    const unsigned lo = sgpr((unsigned)v), hi = sgpr((unsigned)(v >> 32));
    return reinterpret_cast<const T*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ int gelu_col(int j, int rho) {
    const int h = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    return (2 * j + (r >> 3)) * 16 + 8 * h + (r & 7);
}

// bf16-output GELU on two pairs (degree-5 exponent polynomial, relative error 3.5e-5): the C++ twin of the assembly's,
// used for the last tile's epilogue, which has no main loop to hide under
__device__ __forceinline__ uint2 gelu_bf16x4(unsigned d0, unsigned d1) {
    f2 x[2] = {(f2){__uint_as_float(d0 << 16), __uint_as_float(d0 & 0xffff0000u)},
               (f2){__uint_as_float(d1 << 16), __uint_as_float(d1 & 0xffff0000u)}};
    f2 r[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f2 t = __builtin_elementwise_min(__builtin_elementwise_abs(x[k]), (f2){PIPS_GELU_TMAX, PIPS_GELU_TMAX});
        f2 p = t * 2.554670494e-05f + -6.529359078e-04f;
        p = p * t + 7.452824686e-03f;
        p = p * t + -5.192063601e-02f;
        p = p * t + -4.602978599e-01f;
        p = p * t + -1.150685204e+00f;
        p = p * t;
        p = t * (f2){__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
        r[k] = p * -0.5f + __builtin_elementwise_max(x[k], (f2){0.f, 0.f});
    }
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t4 = {r[0].x, r[0].y, r[1].x, r[1].y};
    bf16x4_ ob = __builtin_convertvector(t4, bf16x4_);
    return *reinterpret_cast<uint2*>(&ob);
}

__global__ __launch_bounds__(512) void gemm_bf16_gelu_asm_kernel(GemmArgs p, int tiles_m, int ntiles) {
    constexpr int BM = 256, BN = 128, WGN = 2;
    constexpr int ROWB = 64, STAGE = (BM + BN) * ROWB, SUP = 2 * STAGE, NSUP = 3, LPW = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned short* __restrict__ Ab = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* __restrict__ Wb = reinterpret_cast<const unsigned short*>(p.W);
    unsigned short* __restrict__ Cb = reinterpret_cast<unsigned short*>(p.C);
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;

    // loader: wave w brings rows [(3w + q)*16, +16) of the combined A|W row list; lane -> row lane>>2, physical slot lane&3
    unsigned rowoff[LPW];
    bool q_is_a[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
        const int g0 = (wave * LPW + q) * 16;
        q_is_a[q] = g0 < BM;
        const int row = (g0 < BM ? g0 : g0 - BM) + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        rowoff[q] = (unsigned)row * (unsigned)(g0 < BM ? p.lda : p.K) * 2u + slot * 16;
    }
    auto tile_base = [&](int tile, int q) -> const char* {            // scalar: first row of the tile in A or W, K = 0
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
        return q_is_a[q] ? reinterpret_cast<const char*>(Ab) + (size_t)m0 * p.lda * 2
                         : reinterpret_cast<const char*>(Wb) + (size_t)n0 * p.K * 2;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wvoff = wave * (LPW * 1024), ringend = lds0 + NSUP * SUP;

    // ---- prologue: super-stages 0 and 1 of the first tile in full, the first half of super-stage 2
    const int tile0 = blockIdx.x;
#pragma unroll
    for (int X = 0; X < 3; ++X)
#pragma unroll
        for (int u = 0; u < (X < 2 ? 2 : 1); ++u)
#pragma unroll
            for (int q = 0; q < LPW; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(tile_base(tile0, q) + rowoff[q] + X * 128 + u * 64),
                                                 (lptr_t)(smem + X * SUP + u * STAGE + wave * (LPW * 1024) + q * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment byte offsets inside a stage (kk = 0; the statement derives kk = 1 by ^ 32)
    const unsigned a_off = (wm * 64 + l31) * ROWB + ((half ^ ((l31 >> 2) & 3)) * 16);
    unsigned b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int brow = wn * 64 + gelu_col(j, l31);
        b_off[j] = (BM + brow) * ROWB + ((half ^ ((brow >> 2) & 3)) * 16);
    }
    const unsigned boff = 8 * half * 4;                               // per-lane part of the bias column offset (bytes)

    u32x16 pa = {}, pb = {};                                           // the parked tile (bf16 pairs), v[64:79], v[80:95]
    unsigned rd = lds0;                                                // LDS address of the super-stage a tile starts with
    unsigned stoff = 0;
    const char* cb0 = nullptr;
    const char* cb1 = nullptr;
    int prow0 = 0, pcolh = 0;
    for (int t = 0; t < my_tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const bool last = t + 1 == my_tiles;
        const int ntile = last ? tile : tile + gridDim.x;
        const char* cq0 = tile_base(tile, 0); const char* cq1 = tile_base(tile, 1); const char* cq2 = tile_base(tile, 2);
        const char* nq0 = tile_base(ntile, 0); const char* nq1 = tile_base(ntile, 1); const char* nq2 = tile_base(ntile, 2);
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
        const float* bias = p.bias + n0 + wn * 64;
#define PIPS_TILE_ASM(TEXT_)                                                                                              \
        asm volatile(TEXT_                                                                                                \
                     : [pa] "+{v[64:79]}"(pa), [pb] "+{v[80:95]}"(pb)                                                    \
                     : [ro0] "v"(rowoff[0]), [ro1] "v"(rowoff[1]), [ro2] "v"(rowoff[2]), [aoff] "v"(a_off),               \
                       [b0off] "v"(b_off[0]), [b1off] "v"(b_off[1]), [stoff] "v"(stoff), [boff] "v"(boff), [rd] "s"(sgpr(rd)), \
                       [ringend] "s"(sgpr(ringend)), [lds0] "s"(sgpr(lds0)), [wvoff] "s"(sgpr(wvoff)),                    \
                       [cq0] "s"(sgpr(cq0)), [cq1] "s"(sgpr(cq1)), [cq2] "s"(sgpr(cq2)), [nq0] "s"(sgpr(nq0)),            \
                       [nq1] "s"(sgpr(nq1)), [nq2] "s"(sgpr(nq2)), [bias] "s"(sgpr(bias)), [cb0] "s"(sgpr(cb0)),          \
                       [cb1] "s"(sgpr(cb1))                                                                               \
                     : PIPS_TILE_CLOBBER)
        if (t == 0) {
            if (last) PIPS_TILE_ASM(PIPS_TILE_TEXT_G0_R0); else PIPS_TILE_ASM(PIPS_TILE_TEXT_G0_R1);
        } else {
            if (last) PIPS_TILE_ASM(PIPS_TILE_TEXT_G1_R0); else PIPS_TILE_ASM(PIPS_TILE_TEXT_G1_R1);
        }
#undef PIPS_TILE_ASM
        rd += 2 * SUP; if (rd >= ringend) rd -= NSUP * SUP;           // eight super-stages on: 8 mod 3 = 2 buffers further
        // where the tile just parked goes: per-lane byte offset + scalar bases of its two 32-row halves
        prow0 = m0 + wm * 64 + l31; pcolh = n0 + wn * 64 + 8 * half;
        stoff = (unsigned)(((size_t)(l31)*p.ldc + 8 * half) * 2);
        cb0 = reinterpret_cast<const char*>(Cb) + ((size_t)(m0 + wm * 64) * p.ldc + n0 + wn * 64) * 2;
        cb1 = cb0 + (size_t)32 * p.ldc * 2;
    }
    // ---- the last tile's epilogue
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
        const int i = pc >> 2, jq = pc & 3, j = jq >> 1, q = jq & 1;
        const u32x16& v = (2 * i + j) < 2 ? pa : pb;
        const int b = 8 * ((2 * i + j) & 1) + 4 * q;
        const uint2 lo = gelu_bf16x4(v[b], v[b + 1]), hi = gelu_bf16x4(v[b + 2], v[b + 3]);
        *reinterpret_cast<uint4*>(Cb + (size_t)(prow0 + i * 32) * p.ldc + pcolh + jq * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

// returns PIPS_OK if the problem was taken, 1 if the caller should use the register-staged kernel of gemm_bf16.hip
int launch_gemm_bf16_asm(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st) {
    static int mode = -1;                       // tuning hook PIPS_BF16_ASM: 0 = off, 1 (default) = on
    if (mode < 0) { const char* e = getenv("PIPS_BF16_ASM"); mode = e ? atoi(e) : 1; }
    if (!mode || !a_bf16 || !out_bf16 || (a.epi & 0xff) != EPI_GELU || a.K != 512 || a.lda % 8 != 0 || a.ldc % 8 != 0 ||
        a.bias == nullptr)
        return 1;
    if (a.M % 256 != 0 || a.N % 128 != 0 || (mode != 2 && (long)(a.M / 256) * (a.N / 128) < 256)) return 1;   // (2: debugging)
    const int tiles_m = a.M / 256, ntiles = tiles_m * (a.N / 128);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        set_error("gemm_bf16_asm: cannot query the device");
        return PIPS_E_LAUNCH;
    }
    int grid = ntiles < cus ? ntiles : cus;
    if (const char* e = getenv("PIPS_BF16_ASM_GRID")) grid = atoi(e) < grid ? atoi(e) : grid;          // debugging
    const size_t lds = (size_t)6 * (256 + 128) * 64;
    static std::atomic<unsigned long long> raised{0};
    const int rc = ensure_dynamic_lds(raised, (const void*)gemm_bf16_gelu_asm_kernel, lds);
    if (rc != PIPS_OK) return rc;
    hipLaunchKernelGGL(gemm_bf16_gelu_asm_kernel, dim3(grid), dim3(512), lds, st, a, tiles_m, ntiles);
    PIPS_CHECK_LAUNCH("gemm_bf16_gelu_asm_kernel");
    return PIPS_OK;
}

}  // namespace pips
