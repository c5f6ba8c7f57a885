// bf16 x bf16 up-projection of the large-batch channel mix -- the first Linear + GELU of the MLP-Mixer's channel
// FeedForward, nets/pips.py:104-105 as instantiated at :118 (BASELINE config 3: M = B*N*8 >= 16384 rows, K = 512, GELU, bf16 output): blocks that walk a few tiles each, the tile body ONE generated assembly statement
// (gemm_bf16_tile_asm.inc <- tools/gen_gemm_bf16_asm.py).
//
// C[M,N] = bf16(gelu(bf16(A W^T + bias))): both operands bf16 in memory; fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (AccVGPRs); the Linear's output is rounded to bf16 before the GELU, as under autocast.
// Block = 8 waves, 256x128 tile, 64x64 per wave.  Operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction, no VGPR round trip) into a ring of three super-stages of 2 x 32 K values that runs three
// super-stages ahead and across the block's tile boundaries; one s_barrier per 64 K values; one rolling fragment set
// (the reads of the next K half go out as soon as the MFMAs that used the registers are issued).  A finished tile is
// parked as bf16 pairs in 32 registers; its GELU (piecewise-linear LDS table) and 16-byte stores follow at once.
// Why assembly: in C++ the same loop either carries ~25 scalar branches per 64 K values (as many clocks as the MFMAs)
// or, written branch-free, spills (tools/experiments/README.md, gemm_bf16_dma.hip).  Why the GELU is NOT interleaved
// with the next tile's MFMAs (the G1 texts of the generator, PIPS_ASM_DEFER=1: measured 22.0 vs 20.7 ms at config 3):
// VALU work and dense bf16 MFMAs do not overlap on a gfx950 SIMD (tools/mfma_valu_overlap.hip: 8 MFMAs + 64 FMAs per
// wave take the SUM of their times, from one wave or from two), so hiding one under the other buys nothing and the
// interleaved form pays arbitration between the two waves of a SIMD on top.
// Measured (M = 16384, N = 2048): 54.8 us isolated against 59.9 us for gemm_bf16_kernel<256,256>; config 3 21.5 ->
// 20.7 ms.  Per 64 K values a wave spends ~1750 clocks (trace: tools/bf16_asm_trace.py) where its 16 MFMAs need 600.
// LDS rows are unpadded (the DMA writes lane-linear), XOR-swizzled: phys slot = slot ^ ((row >> 2) & 3), applied to
// the per-lane global source address and to the fragment reads (conflict-free 16-lane ds_read_b128 groups).
// GELU column order: the W rows are fetched from LDS permuted (gelu_col) so that a lane's registers 8q..8q+7 are eight
// consecutive output columns: 16-byte stores, 32 contiguous bytes per row and instruction.
#include "common.h"

#include <cstdlib>

#ifndef PIPS_ASM_DEFER
#define PIPS_ASM_DEFER 0     // 1: a tile's GELU / stores ride between the next tile's MFMA pairs (the G1 texts: regenerate
#endif                       // the .inc with PIPS_GEN_DEFER=1)
#if !PIPS_ASM_DEFER
#define PIPS_TILE_TEXT_G1_R0 PIPS_TILE_TEXT_G0_R0
#define PIPS_TILE_TEXT_G1_R1 PIPS_TILE_TEXT_G0_R1
#endif
#ifndef PIPS_TILE_INC
#define PIPS_TILE_INC "gemm_bf16_tile_asm.inc"      // tuning builds point this at a traced copy (PIPS_GEN_TRACE=1)
#endif
#include PIPS_TILE_INC

#ifdef PIPS_ASM_TRACE        // tools/bf16_asm_trace.py: s_memtime stamps of one wave of block 0, 36 per tile
namespace pips { __device__ unsigned* g_asm_trace; }
extern "C" int pips_asm_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_asm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#ifndef PIPS_ASM_TRACE_WAVE
#define PIPS_ASM_TRACE_WAVE 0
#endif
#endif

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
__device__ __forceinline__ float gelu_tab(const float2* __restrict__ tab, float x) {
    const float t = fmaf(x, 64.0f, 384.0f);
    const unsigned i = (unsigned)__builtin_amdgcn_fmed3f(t, 0.0f, (float)(GELU_TAB_N - 1));
    const float2 e = tab[i];
    return fmaf(e.y, t - (float)i, e.x);
}
__device__ __forceinline__ uint2 gelu_bf16x4(const float2* __restrict__ tab, unsigned d0, unsigned d1) {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t4 = {gelu_tab(tab, __uint_as_float(d0 << 16)), gelu_tab(tab, __uint_as_float(d0 & 0xffff0000u)),
                       gelu_tab(tab, __uint_as_float(d1 << 16)), gelu_tab(tab, __uint_as_float(d1 & 0xffff0000u))};
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
    // A block walks TPB consecutive tiles (m fastest: they share the W panel).  TPB is small and the grid large: in the
    // forward the kernel starts while the token-mix kernel before it is still draining, and with one long-lived block per CU
    // the CU that frees up last decided the kernel's end (73 us in situ against 55 us isolated); with ntiles / TPB blocks
    // the hardware hands blocks to CUs as they free up.
    const int tpb = p.swz;                            // tiles per block (set by the launcher)
    const int tile_first = blockIdx.x * tpb, tile_end = min(tile_first + tpb, ntiles);
    if (tile_first >= ntiles) return;

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

    // ---- the GELU table (behind the ring), published by the barrier of the prologue
    float2* tab = reinterpret_cast<float2*>(smem + NSUP * SUP);
    for (int k = tid; k < GELU_TAB_N; k += 512) {
        const float x = (float)(k - GELU_TAB_N / 2) * (1.0f / 64.0f);
        const float v0 = gelu_exact(x), v1 = gelu_exact(x + 1.0f / 64.0f);
        tab[k] = make_float2(v0, v1 - v0);
    }
    // ---- prologue: super-stages 0 and 1 of the first tile in full, the first half of super-stage 2
    const int tile0 = tile_first;
#pragma unroll
    for (int X = 0; X < 3; ++X)
#pragma unroll
        for (int u = 0; u < (X < 2 ? 2 : 1); ++u)
#pragma unroll
            for (int q = 0; q < LPW; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(tile_base(tile0, q) + rowoff[q] + X * 128 + u * 64),
                                                 (lptr_t)(smem + X * SUP + u * STAGE + wave * (LPW * 1024) + q * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
    for (int tile = tile_first, t = 0; tile < tile_end; ++tile, ++t) {
        const int nxt = tile + 1;
        const bool last = nxt >= tile_end;
        const int ntile = last ? tile : nxt;
        const char* cq0 = tile_base(tile, 0); const char* cq1 = tile_base(tile, 1); const char* cq2 = tile_base(tile, 2);
        const char* nq0 = tile_base(ntile, 0); const char* nq1 = tile_base(ntile, 1); const char* nq2 = tile_base(ntile, 2);
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
        const float* bias = p.bias + n0 + wn * 64;
#ifdef PIPS_ASM_TRACE
        int trv = 0;
#define PIPS_TR_OPERAND , [tr] "+v"(trv)
#else
#define PIPS_TR_OPERAND
#endif
#define PIPS_TILE_ASM(TEXT_)                                                                                              \
        asm volatile(TEXT_                                                                                                \
                     : [pa] "+{v[64:79]}"(pa), [pb] "+{v[80:95]}"(pb) PIPS_TR_OPERAND                                    \
                     : [ro0] "v"(rowoff[0]), [ro1] "v"(rowoff[1]), [ro2] "v"(rowoff[2]), [aoff] "v"(a_off),               \
                       [b0off] "v"(b_off[0]), [b1off] "v"(b_off[1]), [stoff] "v"(stoff), [boff] "v"(boff), [rd] "s"(sgpr(rd)), \
                       [ringend] "s"(sgpr(ringend)), [lds0] "s"(sgpr(lds0)), [wvoff] "s"(sgpr(wvoff)),                    \
                       [cq0] "s"(sgpr(cq0)), [cq1] "s"(sgpr(cq1)), [cq2] "s"(sgpr(cq2)), [nq0] "s"(sgpr(nq0)),            \
                       [nq1] "s"(sgpr(nq1)), [nq2] "s"(sgpr(nq2)), [bias] "s"(sgpr(bias)), [cb0] "s"(sgpr(cb0)),          \
                       [cb1] "s"(sgpr(cb1))                                                                               \
                     : PIPS_TILE_CLOBBER)
        if (t == 0 || !PIPS_ASM_DEFER) {
            if (last) PIPS_TILE_ASM(PIPS_TILE_TEXT_G0_R0); else PIPS_TILE_ASM(PIPS_TILE_TEXT_G0_R1);
        } else {
            if (last) PIPS_TILE_ASM(PIPS_TILE_TEXT_G1_R0); else PIPS_TILE_ASM(PIPS_TILE_TEXT_G1_R1);
        }
#undef PIPS_TILE_ASM
#ifdef PIPS_ASM_TRACE
        if (blockIdx.x == 0 && wave == PIPS_ASM_TRACE_WAVE && g_asm_trace && t < 8 && lane < 52) g_asm_trace[t * 64 + lane] = (unsigned)trv;
#endif
        rd += 2 * SUP; if (rd >= ringend) rd -= NSUP * SUP;           // eight super-stages on: 8 mod 3 = 2 buffers further
        // where the tile just parked goes: per-lane byte offset + scalar bases of its two 32-row halves
        prow0 = m0 + wm * 64 + l31; pcolh = n0 + wn * 64 + 8 * half;
        stoff = (unsigned)(((size_t)(l31)*p.ldc + 8 * half) * 2);
        cb0 = reinterpret_cast<const char*>(Cb) + ((size_t)(m0 + wm * 64) * p.ldc + n0 + wn * 64) * 2;
        cb1 = cb0 + (size_t)32 * p.ldc * 2;
        if (!PIPS_ASM_DEFER || last) {
            // GELU + stores of the tile just parked.  (Deferred form: only the last tile's -- the others ride between the
            // next tile's MFMA pairs; measured slower: VALU and bf16 MFMAs do not overlap on a gfx950 SIMD,
            // tools/mfma_valu_overlap.hip, and the interleaved form pays arbitration on top.)
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                const int i = pc >> 2, jq = pc & 3, j = jq >> 1, q = jq & 1;
                const u32x16& v = (2 * i + j) < 2 ? pa : pb;
                const int b = 8 * ((2 * i + j) & 1) + 4 * q;
                const uint2 lo = gelu_bf16x4(tab, v[b], v[b + 1]), hi = gelu_bf16x4(tab, v[b + 2], v[b + 3]);
                *reinterpret_cast<uint4*>(Cb + (size_t)(prow0 + i * 32) * p.ldc + pcolh + jq * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

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

// The down-projection -- the FeedForward's second Linear, nets/pips.py:107, and the residual of PreNormResidual :100
// (K = 2048, fp32 output): one 256x128 tile per block, the same ring / fragment / MFMA
// schedule as a LOOP over the super-stages inside one assembly statement (PIPS_TILE_TEXT_RES); the accumulators start
// from the residual tile, the bias is added at the end, 16-byte fp32 stores in the natural column order.
__global__ __launch_bounds__(512) void gemm_bf16_res_asm_kernel(GemmArgs p, int tiles_m, int ntiles) {
    constexpr int BM = 256, BN = 128, WGN = 2;
    constexpr int ROWB = 64, STAGE = (BM + BN) * ROWB, SUP = 2 * STAGE, NSUP = 3, LPW = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);

    unsigned rowoff[LPW];
    const char* qbase[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
        const int g0 = (wave * LPW + q) * 16;
        const int row = (g0 < BM ? g0 : g0 - BM) + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        rowoff[q] = (unsigned)row * (unsigned)(g0 < BM ? p.lda : p.K) * 2u + slot * 16;
        qbase[q] = g0 < BM ? Ab + (size_t)m0 * p.lda * 2 : Wb + (size_t)n0 * p.K * 2;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wvoff = wave * (LPW * 1024), ringend = lds0 + NSUP * SUP;
#pragma unroll
    for (int X = 0; X < 3; ++X)
#pragma unroll
        for (int u = 0; u < (X < 2 ? 2 : 1); ++u)
#pragma unroll
            for (int q = 0; q < LPW; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(qbase[q] + rowoff[q] + X * 128 + u * 64),
                                                 (lptr_t)(smem + X * SUP + u * STAGE + wave * (LPW * 1024) + q * 1024), 16, 0, 0);
    // (no wait here: the statement loads the residual tile and the bias first, then waits for everything and barriers)

    const unsigned a_off = (wm * 64 + l31) * ROWB + ((half ^ ((l31 >> 2) & 3)) * 16);
    unsigned b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int brow = wn * 64 + j * 32 + l31;
        b_off[j] = (BM + brow) * ROWB + ((half ^ ((brow >> 2) & 3)) * 16);
    }
    const unsigned boff = 4 * half * 4;
    const unsigned roff = (unsigned)(((size_t)l31 * p.ldr + 4 * half) * 4), soff = (unsigned)(((size_t)l31 * p.ldc + 4 * half) * 4);
    const float* bias = p.bias + n0 + wn * 64;
    const char* rb0 = reinterpret_cast<const char*>(p.R) + ((size_t)(m0 + wm * 64) * p.ldr + n0 + wn * 64) * 4;
    const char* rb1 = rb0 + (size_t)32 * p.ldr * 4;
    const char* cb0 = reinterpret_cast<const char*>(p.C) + ((size_t)(m0 + wm * 64) * p.ldc + n0 + wn * 64) * 4;
    const char* cb1 = cb0 + (size_t)32 * p.ldc * 4;
    const int nks = p.K / 64;
    asm volatile(PIPS_TILE_TEXT_RES
                 :
                 : [ro0] "v"(rowoff[0]), [ro1] "v"(rowoff[1]), [ro2] "v"(rowoff[2]), [aoff] "v"(a_off), [b0off] "v"(b_off[0]),
                   [b1off] "v"(b_off[1]), [roff] "v"(roff), [soff] "v"(soff), [boff] "v"(boff), [rd] "s"(sgpr(lds0)),
                   [ringend] "s"(sgpr(ringend)), [lds0] "s"(sgpr(lds0)), [wvoff] "s"(sgpr(wvoff)), [nks] "s"(sgpr((unsigned)nks)),
                   [cq0] "s"(sgpr(qbase[0])), [cq1] "s"(sgpr(qbase[1])), [cq2] "s"(sgpr(qbase[2])), [bias] "s"(sgpr(bias)),
                   [rb0] "s"(sgpr(rb0)), [rb1] "s"(sgpr(rb1)), [cb0] "s"(sgpr(cb0)), [cb1] "s"(sgpr(cb1))
                 : PIPS_TILE_RES_CLOBBER);
}

#ifdef PIPS_TUNING        // an experiment kept for the tuning build (PIPS_BF16_RES4=1): measured 49.6 against 46.9 us, DESIGN.md 4b
// The same tile on FOUR waves, one per SIMD (PIPS_TILE_TEXT_RES4): wave tile 128 x 64 -- 6 fragment reads per 8 MFMAs instead of
// 4 per 4, no SIMD partner to wait for at the barrier -- and a ring of three 64-K stages with 128-byte LDS rows: a DMA instruction
// brings 8 rows x 128 B = eight full cache lines (the 32-K stages above: 16 rows x 64 B; the K loop's time is the number of
// vector-memory instructions x ~52 clocks, i.e. the texture addresser's rate per row segment -- see the generator).  LDS image of a
// stage: rows 0..255 = A, 256..383 = W, 128 bytes each, physical 16-byte slot = slot ^ ((row >> 1) & 7), applied on the global
// side (lane -> row lane >> 3 of the piece, physical slot lane & 7).  Wave w brings A pieces 8w..8w+7 and W pieces 4w..4w+3.
__global__ __launch_bounds__(256) void gemm_bf16_res4_asm_kernel(GemmArgs p, int tiles_m, int ntiles) {
    constexpr int BM = 256, BN = 128, WGN = 2;
    constexpr int ROWB = 128, SUP = (BM + BN) * ROWB, NSUP = 3, NA = 8, NW = 4, NP = NA + NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
    const char* Abase = sgpr(reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2);
    const char* Wbase = sgpr(reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.K * 2);

    unsigned rowoff[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const bool is_a = q < NA;
        const int row = (is_a ? (wave * NA + q) : (wave * NW + q - NA)) * 8 + (lane >> 3);     // row of the A tile / of the W tile
        const int lrow = is_a ? row : BM + row;                                                 // row of the LDS image
        const int slot = (lane & 7) ^ ((lrow >> 1) & 7);
        rowoff[q] = (unsigned)row * (unsigned)(is_a ? p.lda : p.K) * 2u + slot * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wvoffa = wave * (NA * 1024), wvoffw = BM * ROWB + wave * (NW * 1024), ringend = lds0 + NSUP * SUP;
    // prologue: stages 0 and 1 in full, the first six pieces of stage 2 (the statement issues the rest)
#pragma unroll
    for (int X = 0; X < 3; ++X)
#pragma unroll
        for (int q = 0; q < (X < 2 ? NP : 6); ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)((q < NA ? Abase : Wbase) + rowoff[q] + X * 128),
                                             (lptr_t)(smem + X * SUP + (q < NA ? wvoffa + q * 1024 : wvoffw + (q - NA) * 1024)), 16, 0, 0);

    const unsigned a_off = (wm * 128 + l31) * ROWB + ((half ^ ((l31 >> 1) & 7)) * 16);
    unsigned b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int brow = BM + wn * 64 + j * 32 + l31;
        b_off[j] = brow * ROWB + ((half ^ ((brow >> 1) & 7)) * 16);
    }
    const unsigned boff = 4 * half * 4;
    const unsigned roff = (unsigned)(((size_t)l31 * p.ldr + 4 * half) * 4), soff = (unsigned)(((size_t)l31 * p.ldc + 4 * half) * 4);
    const float* bias = p.bias + n0 + wn * 64;
    const char* rb[4];
    const char* cb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rb[i] = reinterpret_cast<const char*>(p.R) + ((size_t)(m0 + wm * 128 + i * 32) * p.ldr + n0 + wn * 64) * 4;
        cb[i] = reinterpret_cast<const char*>(p.C) + ((size_t)(m0 + wm * 128 + i * 32) * p.ldc + n0 + wn * 64) * 4;
    }
    const int nks = p.K / 64;
#define PIPS_LO(ptr) sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define PIPS_HI(ptr) sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))
    asm volatile(PIPS_TILE_TEXT_RES4
                 :
                 : [ro0] "v"(rowoff[0]), [ro1] "v"(rowoff[1]), [ro2] "v"(rowoff[2]), [ro3] "v"(rowoff[3]), [ro4] "v"(rowoff[4]),
                   [ro5] "v"(rowoff[5]), [ro6] "v"(rowoff[6]), [ro7] "v"(rowoff[7]), [ro8] "v"(rowoff[8]), [ro9] "v"(rowoff[9]),
                   [ro10] "v"(rowoff[10]), [ro11] "v"(rowoff[11]), [aoff] "v"(a_off), [b0off] "v"(b_off[0]), [b1off] "v"(b_off[1]),
                   [roff] "v"(roff), [soff] "v"(soff), [boff] "v"(boff), [rd] "s"(sgpr(lds0)), [ringend] "s"(sgpr(ringend)),
                   [lds0] "s"(sgpr(lds0)), [wvoffa] "s"(sgpr(wvoffa)), [wvoffw] "s"(sgpr(wvoffw)), [nks] "s"(sgpr((unsigned)nks)),
                   [cqa] "s"(PIPS_LO(Abase)), [cqah] "s"(PIPS_HI(Abase)), [cqw] "s"(PIPS_LO(Wbase)), [cqwh] "s"(PIPS_HI(Wbase)),
                   [bias] "s"(sgpr(bias)), [rb0] "s"(sgpr(rb[0])), [rb1] "s"(sgpr(rb[1])), [rb2] "s"(sgpr(rb[2])), [rb3] "s"(sgpr(rb[3])),
                   [cb0] "s"(sgpr(cb[0])), [cb1] "s"(sgpr(cb[1])), [cb2] "s"(sgpr(cb[2])), [cb3] "s"(sgpr(cb[3]))
                 : PIPS_TILE_RES4_CLOBBER);
#undef PIPS_LO
#undef PIPS_HI
}
#endif

// Which kernel a bf16-operand GEMM goes to: 0 = the register-staged gemm_bf16_kernel, 1 = gemm_bf16_res_asm_kernel
// (down-projection + residual), 2 = gemm_bf16_gelu_asm_kernel (up-projection + GELU).  Pure function of the problem --
// also behind pips_gemm_bf16_route(), which lets a test assert that a forward's geometry reaches the assembly kernels.
int gemm_bf16_asm_route(const GemmArgs& a, int a_bf16, int out_bf16) {
    if (gemm_bf16_t4_takes(a, a_bf16, out_bf16)) return 3;           // (launch_gemm_bf16 asks that kernel first)
    const int mode = PIPS_TUNE("PIPS_BF16_ASM", 1);        // tuning hook: 0 = off, 1 (default) = on, 2 = on for any tile count
    const int epi = a.epi & 0xff;
    if (!mode || !a_bf16 || a.lda % 8 != 0 || a.ldc % 8 != 0 || a.bias == nullptr || a.K % 64 != 0) return 0;
    if (a.M % 256 != 0 || a.N % 128 != 0 || (mode != 2 && (long)(a.M / 256) * (a.N / 128) < 256)) return 0;   // (2: debugging)
    if (epi == EPI_RESIDUAL && !out_bf16 && a.R != nullptr && a.ldr % 4 == 0 && a.K >= 256 &&
        (unsigned long long)a.M * a.ldr * 4ull < (1ull << 32) && (unsigned long long)a.M * a.ldc * 4ull < (1ull << 32))
        return PIPS_TUNE("PIPS_BF16_ASM_RES", 1) ? 1 : 0;   // tuning hook =0: down-projection on the register-staged kernel
    if (!out_bf16 || epi != EPI_GELU || a.K != 512) return 0;
    return 2;
}

// returns PIPS_OK if the problem was taken, 1 if the caller should use the register-staged kernel of gemm_bf16.hip
int launch_gemm_bf16_asm(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st) {
    const int route = gemm_bf16_asm_route(a, a_bf16, out_bf16);
    if (route == 0) return 1;
    const int tiles_m = a.M / 256, ntiles = tiles_m * (a.N / 128);
    const size_t ring = (size_t)6 * (256 + 128) * 64;
#ifdef PIPS_TUNING
    if (route == 1 && PIPS_TUNE("PIPS_BF16_RES4", 0)) {
        static std::atomic<unsigned long long> raised_r4{0};
        const int rc = ensure_dynamic_lds(raised_r4, (const void*)gemm_bf16_res4_asm_kernel, ring);
        if (rc != PIPS_OK) return rc;
        hipLaunchKernelGGL(gemm_bf16_res4_asm_kernel, dim3(ntiles), dim3(256), ring, st, a, tiles_m, ntiles);
        PIPS_CHECK_LAUNCH("gemm_bf16_res4_asm_kernel");
        return PIPS_OK;
    }
#endif
    if (route == 1) {
        static std::atomic<unsigned long long> raised_r{0};
        const int rc = ensure_dynamic_lds(raised_r, (const void*)gemm_bf16_res_asm_kernel, ring);
        if (rc != PIPS_OK) return rc;
        hipLaunchKernelGGL(gemm_bf16_res_asm_kernel, dim3(ntiles), dim3(512), ring, st, a, tiles_m, ntiles);
        PIPS_CHECK_LAUNCH("gemm_bf16_res_asm_kernel");
        return PIPS_OK;
    }
    const int cus0 = device_cus();
    if (PIPS_TUNE("PIPS_BF16_UP256", 1) && a.N % 256 == 0 && cus0 > 0 && (long)(a.M / 256) * (a.N / 256) >= cus0) {
        const int nt = (a.M / 256) * (a.N / 256);
        int t2 = PIPS_TUNE("PIPS_BF16_UP256_TPB", 2);
        while (t2 > 1 && (nt + t2 - 1) / t2 < cus0) --t2;
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
    int tpb = PIPS_TUNE("PIPS_BF16_ASM_TPB", 4);  // tuning hook: tiles per block (config 3: 1 / 2 / 4 -> 22.1 / 21.0 / 20.7 ms)
    if (tpb < 1) tpb = 1;
    const int cus = device_cus();
    if (cus <= 0) {
        set_error("gemm_bf16_asm: cannot query the device");
        return PIPS_E_LAUNCH;
    }
    int t = tpb;                                // ... but never fewer blocks than CUs
    while (t > 1 && (ntiles + t - 1) / t < cus) --t;
    GemmArgs b = a;
    b.swz = t;
    const int grid = (ntiles + t - 1) / t;
    const size_t lds = ring + GELU_TAB_N * 8;     // the ring + the GELU table
    static std::atomic<unsigned long long> raised{0};
    const int rc = ensure_dynamic_lds(raised, (const void*)gemm_bf16_gelu_asm_kernel, lds);
    if (rc != PIPS_OK) return rc;
    hipLaunchKernelGGL(gemm_bf16_gelu_asm_kernel, dim3(grid), dim3(512), lds, st, b, tiles_m, ntiles);
    PIPS_CHECK_LAUNCH("gemm_bf16_gelu_asm_kernel");
    return PIPS_OK;
}

}  // namespace pips
