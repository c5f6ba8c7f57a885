// bf16 x bf16 up-projection of the large-batch channel mix -- the first Linear + GELU of the MLP-Mixer's channel FeedForward,
// nets/pips.py:104-105 as instantiated at :118 (BASELINE configs[2]: M = B*N*8 >= 8192 rows, K = 512, GELU, bf16 output):
// 256 x 256 tiles, blocks that walk two tiles each, the tile body ONE generated assembly statement (gemm_bf16_tile_asm.inc <-
// tools/gen_gemm_bf16_asm.py).
//
// C[M,N] = bf16(gelu(bf16(A W^T + bias))): both operands bf16 in memory; fp32 accumulation on v_mfma_f32_32x32x16_bf16
// (AccVGPRs); the Linear's output is rounded to bf16 before the GELU, as under autocast.  Operands reach LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip); LDS rows are unpadded (the DMA writes lane-linear),
// XOR-swizzled: phys slot = slot ^ ((row >> 2) & 3), applied to the per-lane global source address and to the fragment reads.
// GELU column order: the W rows are fetched from LDS permuted (gelu_col) so that a lane's registers 8q..8q+7 are eight
// consecutive output columns: 16-byte stores.
// History: rounds 2-3 also carried a 256 x 128 form of this kernel (gemm_bf16_gelu_asm_kernel) and the
// down-projection as a looped 256 x 128 tile (gemm_bf16_res_asm_kernel, + a four-wave experiment); round 4 moved the
// down-projection to gemm_bf16_t4.hip (48 -> 40 us) and dropped all three -- shapes this kernel does not take run on the
// register-staged gemm_bf16_kernel.
#include "common.h"

#include <cstdlib>

#ifndef PIPS_TILE_INC
#define PIPS_TILE_INC "gemm_bf16_tile_asm.inc"      // tuning builds point this at a traced copy (PIPS_GEN_TRACE=1)
#endif
#include PIPS_TILE_INC

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

// GELU of the parked tile by a piecewise-linear table in LDS: 768 intervals of 1/64 on [-6, 6) as {value, slope} pairs
// built from the exact form (common.h) at kernel start; beyond the table the first / last interval extrapolates (slope
// 0 / 1: gelu(x) = 0 / x there to 1e-8).  Interpolation error h^2/8 max|gelu''| = 2.4e-5 -- the result is rounded to
// bf16 (2^-9 relative).  6 VALU instructions + one ds_read_b64 per value where the polynomial-and-exp form needs ~15
// issue slots: the epilogue is pure VALU time (bf16 MFMAs and VALU work do not overlap on a gfx950 SIMD,
// tools/mfma_valu_overlap.hip) and the LDS pipe is idle in it.
constexpr int GELU_TAB_N = 768;

// The up-projection on 256 x 256 tiles (PIPS_TILE_TEXT_UP256_R0 / _R1): eight waves, wave tile 64 x 128.  Per MFMA a third fewer
// operand bytes cross L2 -> LDS than with the 256 x 128 tile and a quarter fewer fragment reads -- the loop's cost is the MFMAs PLUS its
// vector-memory instructions and fragment reads, so fewer of those is what makes a tile faster.  Ring of four 32-K
// stages ((256 + 256) rows x 64 B = 32 KiB each) three stages ahead, one barrier per stage, run-on into the block's next tile; the
// epilogue (bf16 rounding, table GELU, 16-byte stores) is part of the statement and reads the accumulators directly.
__global__ __launch_bounds__(512) void gemm_bf16_gelu256_asm_kernel(GemmArgs p, int tiles_m, int ntiles) {
    constexpr int BM = 256, BN = 256;
    constexpr int ROWB = 64, STAGE = (BM + BN) * ROWB, NST = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    char* Cb = reinterpret_cast<char*>(p.C);
    const int tpb = p.swz;                            // tiles per block (set by the launcher), consecutive in m
    const int tile_first = blockIdx.x * tpb, tile_end = min(tile_first + tpb, ntiles);
    if (tile_first >= ntiles) return;

    // loader: per stage wave w brings A rows [32w, 32w+32) and W rows [32w, 32w+32) as two 16-row pieces each;
    // lane -> row lane >> 2 of the piece, physical 16-byte slot lane & 3 = slot ^ ((row >> 2) & 3)
    unsigned rowoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + (q & 1) * 16 + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        rowoff[q] = (unsigned)row * (unsigned)(q < 2 ? p.lda : p.K) * 2u + slot * 16;
    }
    auto a_base = [&](int tile) { return sgpr(Ab + (size_t)((tile % tiles_m) * BM) * p.lda * 2); };
    auto w_base = [&](int tile) { return sgpr(Wb + (size_t)((tile / tiles_m) * BN) * p.K * 2); };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wva = lds0 + wave * 2048, wvw = lds0 + BM * ROWB + wave * 2048;

    // ---- the GELU table (behind the ring), published by the first statement's barrier
    float2* tab = reinterpret_cast<float2*>(smem + NST * STAGE);
    for (int k = tid; k < GELU_TAB_N; k += 512) {
        const float x = (float)(k - GELU_TAB_N / 2) * (1.0f / 64.0f);
        const float v0 = gelu_exact(x), v1 = gelu_exact(x + 1.0f / 64.0f);
        tab[k] = make_float2(v0, v1 - v0);
    }
    const unsigned tab_lds = lds0 + NST * STAGE;
    // ---- prologue: stages 0 .. 2 of the first tile
    {
        const char* a0 = a_base(tile_first);
        const char* w0 = w_base(tile_first);
#pragma unroll
        for (int X = 0; X < 3; ++X)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)((q < 2 ? a0 : w0) + rowoff[q] + X * 64),
                                                 (lptr_t)(smem + X * STAGE + (q < 2 ? 0 : BM * ROWB) + wave * 2048 + (q & 1) * 1024), 16, 0, 0);
    }
    // fragment byte offsets (LDS base included; K half 1 = ^ 32 inside the statement)
    const unsigned a_off = lds0 + (wm * 64 + l31) * ROWB + ((half ^ ((l31 >> 2) & 3)) * 16);
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int brow = wn * 128 + (j >> 1) * 64 + gelu_col(j & 1, l31);
        b_off[j] = lds0 + (BM + brow) * ROWB + ((half ^ ((brow >> 2) & 3)) * 16);
    }
    const unsigned boff = 8 * half * 4;                               // per-lane part of the bias column offset (bytes)
    const unsigned stoff = (unsigned)(((size_t)l31 * p.ldc + 8 * half) * 2);
#define PIPS_LO(ptr) sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define PIPS_HI(ptr) sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))
    for (int tile = tile_first; tile < tile_end; ++tile) {
        const bool last = tile + 1 >= tile_end;
        const int ntile = last ? tile : tile + 1;
        const char* ca = a_base(tile);  const char* cw = w_base(tile);
        const char* na = a_base(ntile); const char* nw = w_base(ntile);
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
        const float* bias = p.bias + n0 + wn * 128;
        const char* cb0 = Cb + ((size_t)(m0 + wm * 64) * p.ldc + n0 + wn * 128) * 2;
        const char* cb1 = cb0 + (size_t)32 * p.ldc * 2;
#define PIPS_TILE256(TEXT_)                                                                                                 \
        asm volatile(TEXT_                                                                                                  \
                     :                                                                                                      \
                     : [ro0] "v"(rowoff[0]), [ro1] "v"(rowoff[1]), [ro2] "v"(rowoff[2]), [ro3] "v"(rowoff[3]), [aoff] "v"(a_off),  \
                       [b0off] "v"(b_off[0]), [b1off] "v"(b_off[1]), [b2off] "v"(b_off[2]), [b3off] "v"(b_off[3]),           \
                       [stoff] "v"(stoff), [boff] "v"(boff), [wva] "s"(sgpr(wva)), [wvw] "s"(sgpr(wvw)), [tab] "s"(sgpr(tab_lds)), \
                       [cqa] "s"(PIPS_LO(ca)), [cqah] "s"(PIPS_HI(ca)), [cqw] "s"(PIPS_LO(cw)), [cqwh] "s"(PIPS_HI(cw)),     \
                       [nqa] "s"(PIPS_LO(na)), [nqah] "s"(PIPS_HI(na)), [nqw] "s"(PIPS_LO(nw)), [nqwh] "s"(PIPS_HI(nw)),     \
                       [bias] "s"(sgpr(bias)), [cb0] "s"(sgpr(cb0)), [cb1] "s"(sgpr(cb1))                                    \
                     : PIPS_TILE_UP256_CLOBBER)
        if (last) PIPS_TILE256(PIPS_TILE_TEXT_UP256_R0); else PIPS_TILE256(PIPS_TILE_TEXT_UP256_R1);
#undef PIPS_TILE256
    }
#undef PIPS_LO
#undef PIPS_HI
}

// Which kernel a bf16-operand GEMM goes to: 0 = the register-staged gemm_bf16_kernel, 2 = gemm_bf16_gelu256_asm_kernel (this
// file: up-projection + GELU, bf16 output), 3 = gemm_bf16_t4_res_kernel (gemm_bf16_t4.hip: down-projection + residual).  Pure
// function of the problem and the device's CU count -- also behind pips_gemm_bf16_route(), which lets a test assert that a
// forward's geometry reaches the assembly kernels.  (1 was the 256 x 128 down-projection kernel of rounds 2-3.)
int gemm_bf16_asm_route(const GemmArgs& a, int a_bf16, int out_bf16) {
    if (gemm_bf16_t4_takes(a, a_bf16, out_bf16)) return 3;           // (launch_gemm_bf16 asks these kernels first)
    if (gemm_bf16_t4up_takes(a, a_bf16, out_bf16, nullptr)) return 4;
    if (!PIPS_TUNE("PIPS_BF16_ASM", 1)) return 0;                    // tuning hook: 0 = register-staged kernels only
    if (!a_bf16 || !out_bf16 || (a.epi & 0xff) != EPI_GELU || a.bias == nullptr || a.K != 512) return 0;
    if (a.lda % 8 != 0 || a.ldc % 8 != 0 || a.M % 256 != 0 || a.N % 256 != 0) return 0;
    const int cus = device_cus();
    return cus > 0 && (long)(a.M / 256) * (a.N / 256) >= cus ? 2 : 0;
}

// returns PIPS_OK if the problem was taken, 1 if the caller should use the register-staged kernel of gemm_bf16.hip
int launch_gemm_bf16_asm(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st) {
    if (gemm_bf16_asm_route(a, a_bf16, out_bf16) != 2) return 1;
    const int cus = device_cus();
    const int nt = (a.M / 256) * (a.N / 256);
    int t2 = PIPS_TUNE("PIPS_BF16_UP256_TPB", 2);            // tiles per block ... but never fewer blocks than CUs
    while (t2 > 1 && (nt + t2 - 1) / t2 < cus) --t2;
    GemmArgs b2 = a;
    b2.swz = t2;
    const size_t lds2 = (size_t)4 * 512 * 64 + GELU_TAB_N * 8;
    static std::atomic<unsigned long long> raised2{0};
    const int rc2 = ensure_dynamic_lds(raised2, (const void*)gemm_bf16_gelu256_asm_kernel, lds2);
    if (rc2 != PIPS_OK) return rc2;
    hipLaunchKernelGGL(gemm_bf16_gelu256_asm_kernel, dim3((nt + t2 - 1) / t2), dim3(512), lds2, st, b2, a.M / 256, nt);
    PIPS_CHECK_LAUNCH("gemm_bf16_gelu256_asm_kernel");
    return PIPS_OK;
}

}  // namespace pips
