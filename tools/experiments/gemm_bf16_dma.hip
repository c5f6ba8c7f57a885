// bf16 x bf16 GEMM for the large-batch channel-mix Linears (BASELINE config 3, M = B*N*8 >= 16384 rows): both operands
// already bf16 in memory (LayerNorm-2 output / hidden activation, pre-converted weights), so nothing has to pass
// through registers on its way to LDS.
//
// C[M,N] = epi(A[M,K] * W[N,K]^T + bias), fp32 accumulation on v_mfma_f32_32x32x16_bf16, C^T accumulators as in
// gemm_bf16.hip.  That kernel (two LDS stages filled through VGPRs, one tile per block, all blocks in lock step) spends
// 57 us on the config-3 up-projection whose MFMAs need 14: per 64 K values a wave's 16 MFMAs take 512 clocks, the GELU
// of its share of the tile costs as many VALU clocks again, and a traced block (tools/dma_gemm_trace.py) shows the
// phases -- staging, fragment reads, MFMAs, GELU, stores -- one after the other instead of under each other.  Here:
//  * PERSISTENT blocks (one per CU, 8 waves, 256x128 tile, 64x64 per wave) walk the tiles;
//  * operands reach LDS by LDS-DMA (global_load_lds_dwordx4: a wave-instruction lands 1 KiB = 16 rows x 64 B, no VGPR
//    round trip, no ds_write pass) into a ring of three super-stages of 2 x 32 K values, two super-stages ahead of the
//    MFMAs and straight across tile boundaries; a wave waits for ITS part with a counted s_waitcnt vmcnt, one raw
//    s_barrier per super-stage (64 K values) publishes it.  The wave's six DMA instructions per super-stage go out
//    one at a time between MFMA pairs (issued back to back by 8 waves they fill the texture-address queue and every
//    wave sits in the issue stall for ~1000 clocks);
//  * inside a wave the fragment registers are refilled on the fly: the reads of the next K block go out as soon as the
//    MFMAs that used the registers are issued (one fragment set: 32 registers);
//  * DEFERRED, INTERLEAVED EPILOGUE: a finished tile's accumulators are parked in a second register set and its GELU
//    / bf16 conversion / stores are issued in 16 pieces BETWEEN the MFMAs of the next tile (two MFMAs, a DMA
//    instruction, a quarter of a piece's VALU work, ...): VALU work under the matrix pipe, stores spread over the main
//    loop.  The bias (and the residual tile) initialise the accumulators;
//  * rows are unpadded in LDS (the DMA writes lane-linear) with an XOR swizzle of the four 16-byte slots of a 64-byte
//    row, phys = slot ^ ((row >> 2) & 3), applied to the per-lane GLOBAL source address and to the fragment reads: the
//    16 lanes of a ds_read_b128 service group touch 16 distinct 16-byte bank groups;
//  * fragment reads are issued from inline asm (hand-counted lgkmcnt): hipcc sees no LDS load that could alias an
//    in-flight LDS-DMA and puts no vmcnt(0) of its own in front of them (tools/experiments/README.md).
// Tile order: m tiles fastest, so the blocks resident at one time share W panels in every XCD's L2 and each A panel is
// fetched by one XCD (block id mod 8 = XCD = m tile mod 8).
#include "common.h"

#include <cstdlib>

#ifndef PIPS_DMA_FAST
#define PIPS_DMA_FAST 0      // 1: a second, flag-free copy of the loop body for the steady state (hipcc 7.2 spills ~130 registers in it)
#endif
#ifndef PIPS_DMA_ABL
#define PIPS_DMA_ABL 0       // tuning builds (tools/build_variant.sh), timing only: 1 no MFMAs, 2 no fragment reads, 4 no DMA
#endif                       // after the prologue, 32 no GELU arithmetic, 64 no GELU-tile stores

#ifdef PIPS_DMA_TRACE        // tuning builds (tools/dma_gemm_trace.py): s_memtime stamps of one wave of block 0; the value is the
namespace pips { __device__ unsigned long long* g_dma_trace; }      // mask of stamp points (each costs an lgkmcnt(0))
extern "C" int pips_dma_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_dma_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#ifndef PIPS_DMA_TRACE_WAVE
#define PIPS_DMA_TRACE_WAVE 0
#endif
#define PIPS_TR(k)                                                                                       \
    if (((PIPS_DMA_TRACE >> (k)) & 1) && g_tr != nullptr && s < 64) {                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
        if (lane == 0) g_tr[s * 8 + (k)] = t_;                                                           \
    }
#else
#define PIPS_TR(k)
#endif

namespace pips {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Column order of the GELU (bf16) tile: MFMA row rho of tile j -- which lane (half h) holds in register r, rho =
// (r&3) + 8*(r>>2) + 4*h -- carries output column (2j + (r>>3))*16 + 8h + (r&7) of the wave's 64 (the W rows are
// fetched from LDS in that order), so that registers 8q..8q+7 of a lane are 8 CONSECUTIVE bf16 columns (one 16-byte
// store) and the two lanes of a row write 32 contiguous bytes instead of two 8-byte pieces of a 16-byte segment.
__device__ __forceinline__ int gelu_col(int j, int rho) {
    const int h = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    return (2 * j + (r >> 3)) * 16 + 8 * h + (r & 7);
}

// GELU whose result is rounded to bf16 (the hidden activation): the form of gelu_exact2 (common.h) with a degree-5
// exponent polynomial, refitted on [0, 4*sqrt(2)]: relative error 3.5e-5, 1/55 of a bf16 half-ulp, 14 instead of
// 17 VALU instructions per pair.  Evaluated in four steps on TWO pairs at once (independent chains: no dependent-
// issue bubbles) so that the steps can sit between the MFMAs of the main loop.
struct GeluPair { f2 x0, x1, t0, t1, p0, p1; };      // x: the inputs, from step 1 on max(x, 0)
#define PIPS_G5(v) ((f2){v, v})
__device__ __forceinline__ void gelu_step1(GeluPair& g) {
    g.t0 = __builtin_elementwise_min(__builtin_elementwise_abs(g.x0), PIPS_G5(PIPS_GELU_TMAX));
    g.t1 = __builtin_elementwise_min(__builtin_elementwise_abs(g.x1), PIPS_G5(PIPS_GELU_TMAX));
    g.p0 = g.t0 * PIPS_G5(2.554670494e-05f) + PIPS_G5(-6.529359078e-04f);
    g.p1 = g.t1 * PIPS_G5(2.554670494e-05f) + PIPS_G5(-6.529359078e-04f);
    g.p0 = g.p0 * g.t0 + PIPS_G5(7.452824686e-03f);
    g.p1 = g.p1 * g.t1 + PIPS_G5(7.452824686e-03f);
    if (!(PIPS_DMA_ABL & 32)) {
        g.x0 = __builtin_elementwise_max(g.x0, PIPS_G5(0.0f));
        g.x1 = __builtin_elementwise_max(g.x1, PIPS_G5(0.0f));
    }
}
__device__ __forceinline__ void gelu_step2(GeluPair& g) {
    g.p0 = g.p0 * g.t0 + PIPS_G5(-5.192063601e-02f);
    g.p1 = g.p1 * g.t1 + PIPS_G5(-5.192063601e-02f);
    g.p0 = g.p0 * g.t0 + PIPS_G5(-4.602978599e-01f);
    g.p1 = g.p1 * g.t1 + PIPS_G5(-4.602978599e-01f);
    g.p0 = g.p0 * g.t0 + PIPS_G5(-1.150685204e+00f);
    g.p1 = g.p1 * g.t1 + PIPS_G5(-1.150685204e+00f);
    g.p0 = g.p0 * g.t0;
    g.p1 = g.p1 * g.t1;
}
__device__ __forceinline__ void gelu_step3(GeluPair& g) {
    g.p0 = g.t0 * (f2){__builtin_amdgcn_exp2f(g.p0.x), __builtin_amdgcn_exp2f(g.p0.y)};
    g.p1 = g.t1 * (f2){__builtin_amdgcn_exp2f(g.p1.x), __builtin_amdgcn_exp2f(g.p1.y)};
}
__device__ __forceinline__ uint2 gelu_step4(const GeluPair& g) {       // -> 4 bf16 (hardware RNE)
    f2 r0 = g.p0 * -0.5f + g.x0;
    f2 r1 = g.p1 * -0.5f + g.x1;
    if (PIPS_DMA_ABL & 32) { r0 = g.x0; r1 = g.x1; }
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t = {r0.x, r0.y, r1.x, r1.y};
    bf16x4 ob = __builtin_convertvector(t, bf16x4);
    return *reinterpret_cast<uint2*>(&ob);
}
#undef PIPS_G5

template <int E, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_bf16_dma_kernel(GemmArgs p, int tiles_m, int ntiles) {
    constexpr int BM = 256, BN = 128, WGN = 2, NW = 4;
    constexpr int WTM = 128, WTN = 64, TM = 4;
    constexpr int ROWB = 64;                         // bytes of a staged row: 32 bf16
    constexpr int STAGE = (BM + BN) * ROWB;          // bytes per stage: A rows then W rows
    constexpr int LPW = (BM + BN) / 16 / NW;         // DMA wave-instructions per wave per stage (6)
    constexpr int NSUP = 3;                          // super-stages (2 stages each) in the ring
    static_assert(E == EPI_GELU ? OUT_BF16 : !OUT_BF16, "GELU -> bf16 hidden activation, residual -> fp32 stream");
    static_assert(LPW == 6, "the DMA instructions are placed by hand");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned short* __restrict__ Ab = reinterpret_cast<const unsigned short*>(p.A);
    const unsigned short* __restrict__ Wb = reinterpret_cast<const unsigned short*>(p.W);

    const int nks = p.K / 64;                        // super-blocks per tile
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_tiles * nks;                // super-blocks this block walks through, over all its tiles
    if (total == 0) return;
#ifdef PIPS_DMA_TRACE
    unsigned long long* g_tr = (blockIdx.x == 0 && wave == PIPS_DMA_TRACE_WAVE && g_dma_trace) ? g_dma_trace + (E == EPI_GELU ? 0 : 512) : nullptr;
#endif

    // ---- loader: wave w brings rows [(w*6 + q)*16, +16) of the combined A|W row list, q < 6; lane -> row lane>>2 of
    // the group, physical slot lane&3, and fetches the logical slot that the swizzle maps there
    unsigned rowoff[LPW];                            // bytes from the tile's first A (W) row at K = 0
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
        const int g0 = (wave * LPW + q) * 16;                        // wave-uniform
        const int row = (g0 < BM ? g0 : g0 - BM) + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        rowoff[q] = (unsigned)row * (unsigned)(g0 < BM ? p.lda : p.K) * 2u + slot * 16;
    }
    // the super-stage being issued: K values [64*it_ks, +64) of tile it_tile, into ring buffer it_buf
    int it_tile = blockIdx.x, it_ks = 0, it_buf = 0, it_left = total;
    const char* it_a = nullptr;
    const char* it_w = nullptr;
    char* it_lds = nullptr;
    int it_m0 = (it_tile % tiles_m) * BM, it_n0 = (it_tile / tiles_m) * BN;
    auto dma_open = [&]() {                          // scalar bases of the next super-stage
        it_a = reinterpret_cast<const char*>(Ab) + ((size_t)it_m0 * p.lda + it_ks * 64) * 2;
        it_w = reinterpret_cast<const char*>(Wb) + ((size_t)it_n0 * p.K + it_ks * 64) * 2;
        it_lds = smem + it_buf * (2 * STAGE) + wave * (LPW * 1024);
    };
    auto dma_one = [&](int u, int q) {               // u, q compile-time after inlining
        const char* base = (wave * LPW + q) * 16 < BM ? it_a : it_w;                 // scalar
        const unsigned ldsa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(it_lds + u * STAGE + q * 1024);
        // scalar base + 32-bit lane offset (the builtin takes a 64-bit lane address: two more registers per instruction);
        // M0 = LDS address of the wave's 1 KiB piece (one wait state between the M0 write and the LDS-DMA)
        if (u == 0)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(rowoff[q]), "s"(base), "s"(ldsa) : "memory");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:64" ::"v"(rowoff[q]), "s"(base), "s"(ldsa) : "memory");
    };
    auto dma_close = [&]() {
        --it_left;
        it_buf = it_buf + 1 == NSUP ? 0 : it_buf + 1;
        if (++it_ks == nks) {
            it_ks = 0;
            it_tile += gridDim.x;
            it_m0 = (it_tile % tiles_m) * BM; it_n0 = (it_tile / tiles_m) * BN;
        }
    };
#pragma unroll
    for (int s = 0; s < NSUP; ++s)
        if (it_left > 0) {
            dma_open();
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < LPW; ++q) dma_one(u, q);
            dma_close();
        }

    // fragment byte offsets inside a stage: row = tile row + l31, logical slot 2*kk + half, swizzled; kept for kk = 0
    // only -- slot ^ 2 is byte offset ^ 32 (rows are 64 bytes), so the kk = 1 address costs one v_xor, not a register
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_off = (wm * WTM + l31) * ROWB + ((half ^ ((l31 >> 2) & 3)) * 16);
    unsigned b_off[2];                               // [j]: W row of MFMA row l31 of tile j (GELU: permuted columns)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int brow = wn * WTN + (E == EPI_GELU ? gelu_col(j, l31) : j * 32 + l31);
        b_off[j] = (BM + brow) * ROWB + ((half ^ ((brow >> 2) & 3)) * 16);
    }

    // ONE fragment set, refilled on the fly: fa[kk][i], fb[kk][j]; the kk = 0 registers are re-loaded for the next K block
    // as soon as this block's kk = 0 MFMAs are issued, the kk = 1 registers after its kk = 1 MFMAs; the next block then
    // waits lgkmcnt(6) for its kk = 0 fragments and lgkmcnt(0) for the kk = 1 ones
    u32x4 fa[2][TM], fb[2][2];
#define PIPS_READ6(kk_, base_)                                                                                         \
    if (!(PIPS_DMA_ABL & 2)) {                                                                                         \
        const unsigned sb_ = lds0 + (base_);                                         /* wave-uniform */                \
        const unsigned ab_ = ((kk_) ? a_off ^ 32u : a_off) + sb_;                                                      \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[kk_][0]) : "v"(ab_));                                             \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk_][0]) : "v"(((kk_) ? b_off[0] ^ 32u : b_off[0]) + sb_));       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk_][1]) : "v"(ab_), "n"(32 * ROWB));                   \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk_][1]) : "v"(((kk_) ? b_off[1] ^ 32u : b_off[1]) + sb_));       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk_][2]) : "v"(ab_), "n"(64 * ROWB));                   \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk_][3]) : "v"(ab_), "n"(96 * ROWB));                   \
    }
    // MFMA pair (kk, i): both j
#define PIPS_MFMA2(kk_, i_)                                                                                            \
    if (!(PIPS_DMA_ABL & 1)) {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[i_][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fb[kk_][j]),        \
                                                                 *reinterpret_cast<const bf16x8*>(&fa[kk_][i_]),       \
                                                                 acc[i_][j], 0, 0, 0);                                 \
    }
#define PIPS_PIN __builtin_amdgcn_sched_barrier(0);

    f32x16 acc[TM][2];
    unsigned prev[TM][2][8];                          // the parked tile, rounded to bf16 pairs (the Linear's bf16 output
                                                     // under autocast: GELU sees the rounded value there too)
    bool has_prev = false;
    int prow0 = 0, pcolh = 0;
    unsigned short* __restrict__ Cb = reinterpret_cast<unsigned short*>(p.C);

    // accumulators start from the bias (and the residual tile): acc[i][j][4g..4g+3] = C[row0 + i*32][colw + cg(j,g) ..+3],
    // cg = j*32 + 8g + 4*half in the natural order, (2j + (g>>1))*16 + 8*half + 4*(g&1) in the GELU order
    auto init_acc = [&](int row0, int colw) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col0 = colw + (E == EPI_GELU ? (2 * j + (g >> 1)) * 16 + 8 * half + 4 * (g & 1) : j * 32 + 8 * g + 4 * half);
                const float4 b4 = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float4 v = b4;
                    if (E == EPI_RESIDUAL) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.R + (size_t)(row0 + i * 32) * p.ldr + col0);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    }
                    acc[i][j][4 * g] = v.x; acc[i][j][4 * g + 1] = v.y; acc[i][j][4 * g + 2] = v.z; acc[i][j][4 * g + 3] = v.w;
                }
            }
    };
    // half-piece hh (0..31) of the parked tile: registers 4*(hh&3).. of prev[hh>>3][(hh>>2)&1], i.e. piece (i, j, q) =
    // (hh>>3, (hh>>2)&1, (hh>>1)&1), its first (hh even) or second (hh odd) four columns
#define PIPS_LOADGP(hh_)                                                                                               \
    {                                                                                                                  \
        constexpr int i_ = (hh_) >> 3, j_ = ((hh_) >> 2) & 1, d_ = 2 * ((hh_) & 3);                                    \
        unsigned d0_ = prev[i_][j_][d_], d1_ = prev[i_][j_][d_ + 1];                                                   \
        asm volatile("" : "+v"(d0_), "+v"(d1_));     /* in straight-line code the unpacking would be done tiles ahead */ \
        gp.x0 = (f2){__uint_as_float(d0_ << 16), __uint_as_float(d0_ & 0xffff0000u)};                                  \
        gp.x1 = (f2){__uint_as_float(d1_ << 16), __uint_as_float(d1_ & 0xffff0000u)};                                  \
    }
    auto store_piece = [&](int piece, uint2 lo, uint2 hi) {          // 8 consecutive columns of one row per lane
        const int i = piece >> 2, jq = piece & 3;                    // jq = 2j + q
        const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (PIPS_DMA_ABL & 64) { asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w)); return; }
        int prow = prow0;
        asm volatile("" : "+v"(prow));               // keep the 8 store addresses of a tile from being hoisted and held
        *reinterpret_cast<uint4*>(Cb + (size_t)(prow + i * 32) * p.ldc + pcolh + jq * 16) = o;
    };

    // One super-stage (64 K values = K blocks a and b) of the main loop.  GEL_: GELU half-pieces LOADA_ / LOADB_ of the
    // parked tile (piece PIECE_) between the MFMA pairs; DMAA_: the second half of the super-stage opened in the
    // previous iteration goes out in block a; OPEN_: a super-stage is opened behind the barrier (it refills the buffer
    // this one used) and its first half goes out in block b; NEXT_: there is a next super-stage to read fragments of;
    // W6_: a super-stage newer than s+1 is in flight.  In the steady state every flag is the constant true and the body
    // is straight-line code: with run-time flags its ~30 scalar branches cost as much as the MFMAs.
#define PIPS_SUPER(GEL_, HH_, DMAA_, OPEN_, NEXT_, W6_)                                                                \
    {                                                                                                                  \
        const unsigned base = sbuf * (2 * STAGE);                                                                      \
        const bool gel_ = (GEL_), dmaa_ = (DMAA_);                                                                     \
        /* ======== K block a: half-pieces HH_, HH_+1 (piece HH_/2), second half of the open super-stage */           \
        if (gel_) { PIPS_LOADGP(HH_) }                                                                                 \
        PIPS_TR(0)                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");           /* kk = 0 fragments of block a */                \
        PIPS_PIN                                                                                                       \
        PIPS_MFMA2(0, 0) if (dmaa_) dma_one(1, 0); if (gel_) gelu_step1(gp); PIPS_PIN                                  \
        PIPS_MFMA2(0, 1) if (dmaa_) dma_one(1, 1); if (gel_) gelu_step2(gp); PIPS_PIN                                  \
        PIPS_MFMA2(0, 2) if (dmaa_) dma_one(1, 2); if (gel_) gelu_step3(gp); PIPS_PIN                                  \
        PIPS_MFMA2(0, 3) if (gel_) { glo = gelu_step4(gp); PIPS_LOADGP((HH_) + 1) } PIPS_PIN                           \
        wait_lgkm0();                                                /* kk = 1 fragments of block a */                \
        PIPS_READ6(0, base + STAGE)                                  /* block b, kk = 0 */                            \
        PIPS_PIN                                                                                                       \
        PIPS_MFMA2(1, 0) if (dmaa_) dma_one(1, 3); if (gel_) gelu_step1(gp); PIPS_PIN                                  \
        PIPS_MFMA2(1, 1) if (dmaa_) dma_one(1, 4); if (gel_) gelu_step2(gp); PIPS_PIN                                  \
        PIPS_MFMA2(1, 2) if (dmaa_) dma_one(1, 5); if (gel_) gelu_step3(gp); PIPS_PIN                                  \
        PIPS_MFMA2(1, 3) if (gel_) { gout = gelu_step4(gp); store_piece((HH_) / 2, glo, gout); PIPS_LOADGP((HH_) + 2) } PIPS_PIN \
        PIPS_READ6(1, base + STAGE)                                  /* block b, kk = 1 */                            \
        if (dmaa_) dma_close();                                                                                        \
        PIPS_TR(2)                                                                                                     \
        /* ======== K block b: half-pieces HH_+2, HH_+3 (piece HH_/2 + 1) */                                          \
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");           /* kk = 0 fragments of block b */                \
        PIPS_PIN                                                                                                       \
        PIPS_MFMA2(0, 0) if (gel_) gelu_step1(gp); PIPS_PIN                                                            \
        PIPS_MFMA2(0, 1) if (gel_) gelu_step2(gp); PIPS_PIN                                                            \
        PIPS_MFMA2(0, 2) if (gel_) gelu_step3(gp); PIPS_PIN                                                            \
        PIPS_MFMA2(0, 3) if (gel_) { glo = gelu_step4(gp); PIPS_LOADGP((HH_) + 3) } PIPS_PIN                           \
        wait_lgkm0();          /* kk = 1 fragments of block b: this wave is done reading super-stage s */              \
        PIPS_TR(4)                                                                                                     \
        /* super-stage s+1 of this wave has landed once only the super-stage issued after it may be in flight (loads   \
           retire in order; any other outstanding access only makes the count more conservative) */                    \
        if (W6_) wait_vm<2 * LPW>(); else wait_vm<0>();                                                                \
        PIPS_TR(5)                                                                                                     \
        __builtin_amdgcn_s_barrier();      /* everybody's part of s+1 landed; everybody is done reading s */           \
        PIPS_TR(6)                                                                                                     \
        const bool open_ = (OPEN_) && !(PIPS_DMA_ABL & 4);                                                             \
        if (open_) dma_open();                                                                                         \
        dma_on = open_;                                                                                                \
        const int nbuf = sbuf + 1 == NSUP ? 0 : sbuf + 1;                                                              \
        if (NEXT_) PIPS_READ6(0, nbuf * (2 * STAGE))                 /* next block a, kk = 0 */                        \
        PIPS_PIN                                                                                                       \
        PIPS_MFMA2(1, 0) if (open_) { dma_one(0, 0); dma_one(0, 1); } if (gel_) gelu_step1(gp); PIPS_PIN               \
        PIPS_MFMA2(1, 1) if (open_) { dma_one(0, 2); dma_one(0, 3); } if (gel_) gelu_step2(gp); PIPS_PIN               \
        PIPS_TR(3)                                                                                                     \
        PIPS_MFMA2(1, 2) if (open_) { dma_one(0, 4); dma_one(0, 5); } if (gel_) gelu_step3(gp); PIPS_PIN               \
        PIPS_MFMA2(1, 3) if (gel_) { gout = gelu_step4(gp); store_piece((HH_) / 2 + 1, glo, gout); } PIPS_PIN          \
        PIPS_TR(1)                                                                                                     \
        if (NEXT_) PIPS_READ6(1, nbuf * (2 * STAGE))                 /* next block a, kk = 1 */                        \
        PIPS_TR(7)                                                                                                     \
        sbuf = nbuf;                                                                                                   \
    }
    // steady state: a super-stage was opened in the previous iteration (s >= 1) and one is opened in this one (s + 3 < total)
#define PIPS_STEP(KS_)                                                                                                 \
    {                                                                                                                  \
        if (PIPS_DMA_FAST && s >= 1 && s + 4 <= total && has_prev) {                                                   \
            PIPS_SUPER(true, 4 * (KS_), true, true, true, true)                                                        \
        } else {                                                                                                       \
            PIPS_SUPER(has_prev, 4 * (KS_), dma_on, it_left > 0, s + 1 < total, s + 2 < total)                         \
        }                                                                                                              \
        ++s;                                                                                                           \
    }

    int sbuf = 0, s = 0;
    int tile = blockIdx.x;
    int row0 = (tile % tiles_m) * BM + wm * WTM + l31, col0 = (tile / tiles_m) * BN + wn * WTN;
    init_acc(row0, col0);
    if (total > 2) wait_vm<4 * LPW>(); else if (total > 1) wait_vm<2 * LPW>(); else wait_vm<0>();      // super-stage 0 landed
    __builtin_amdgcn_s_barrier();
    PIPS_READ6(0, 0)
    PIPS_READ6(1, 0)
    GeluPair gp;
    uint2 gout = make_uint2(0, 0), glo = make_uint2(0, 0);
    bool dma_on = false;                             // a super-stage is open: its second half goes out in the next K block a

    for (int t = 0; t < my_tiles; ++t) {
        if (E == EPI_GELU) {                         // K = 512: the 8 super-stages of a tile carry its predecessor's 8 pieces
            PIPS_STEP(0) PIPS_STEP(1) PIPS_STEP(2) PIPS_STEP(3) PIPS_STEP(4) PIPS_STEP(5) PIPS_STEP(6) PIPS_STEP(7)
        } else {
            for (int ks = 0; ks < nks; ++ks, ++s) {
                if (PIPS_DMA_FAST && s >= 1 && s + 4 <= total) {
                    PIPS_SUPER(false, 0, true, true, true, true)
                } else {
                    PIPS_SUPER(false, 0, dma_on, it_left > 0, s + 1 < total, s + 2 < total)
                }
            }
        }
        // ---- tile finished
        if (E == EPI_GELU) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        typedef float f32x2_ __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                        const f32x2_ v = {acc[i][j][2 * d], acc[i][j][2 * d + 1]};
                        const bf16x2_ r = __builtin_convertvector(v, bf16x2_);
                        prev[i][j][d] = *reinterpret_cast<const unsigned*>(&r);
                    }
            has_prev = true; prow0 = row0; pcolh = col0 + 8 * half;
        } else {
            float* __restrict__ Cf = p.C;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(Cf + (size_t)(row0 + i * 32) * p.ldc + col0 + j * 32 + 8 * g + 4 * half) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
        if (t + 1 < my_tiles) {
            tile += gridDim.x;
            row0 = (tile % tiles_m) * BM + wm * WTM + l31;
            col0 = (tile / tiles_m) * BN + wn * WTN;
            init_acc(row0, col0);
        }
    }
#undef PIPS_STEP
#undef PIPS_SUPER
    if (E == EPI_GELU) {                             // the last tile's epilogue has nothing to hide under
#define PIPS_TAILPIECE(pc_)                                                                                            \
        PIPS_LOADGP(2 * (pc_)) gelu_step1(gp); gelu_step2(gp); gelu_step3(gp); glo = gelu_step4(gp);                   \
        PIPS_LOADGP(2 * (pc_) + 1) gelu_step1(gp); gelu_step2(gp); gelu_step3(gp); gout = gelu_step4(gp);              \
        store_piece(pc_, glo, gout); __builtin_amdgcn_sched_barrier(0);
        PIPS_TAILPIECE(0) PIPS_TAILPIECE(1) PIPS_TAILPIECE(2) PIPS_TAILPIECE(3)
        PIPS_TAILPIECE(4) PIPS_TAILPIECE(5) PIPS_TAILPIECE(6) PIPS_TAILPIECE(7)
        PIPS_TAILPIECE(8) PIPS_TAILPIECE(9) PIPS_TAILPIECE(10) PIPS_TAILPIECE(11)
        PIPS_TAILPIECE(12) PIPS_TAILPIECE(13) PIPS_TAILPIECE(14) PIPS_TAILPIECE(15)
#undef PIPS_TAILPIECE
    }
#undef PIPS_READ6
#undef PIPS_MFMA2
#undef PIPS_PIN
#undef PIPS_LOADGP
}

template <int E, bool OUT_BF16>
static int launch_dma_tile(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = a.M / 256, ntiles = tiles_m * (a.N / 128);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        set_error("gemm_bf16_dma: cannot query the device");
        return PIPS_E_LAUNCH;
    }
    const int grid = ntiles < cus ? ntiles : cus;
    const size_t lds = (size_t)6 * (256 + 128) * 64;
    auto kern = gemm_bf16_dma_kernel<E, OUT_BF16>;
    static std::atomic<unsigned long long> raised{0};          // per instantiation, one bit per device
    const int rc = ensure_dynamic_lds(raised, (const void*)kern, lds);
    if (rc != PIPS_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tiles_m, ntiles);
    PIPS_CHECK_LAUNCH("gemm_bf16_dma_kernel");
    return PIPS_OK;
}

// The config-3 channel-mix shapes; returns PIPS_OK if the problem was taken, 1 if the caller should use the
// register-staged kernel of gemm_bf16.hip (ragged tiles, fp32 A, other epilogues, too few tiles to fill the GPU).
int launch_gemm_bf16_dma(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st) {
    static int mode = -1;                       // tuning hook PIPS_BF16_DMA: 0 = off, 1 (default) = on
    if (mode < 0) { const char* e = getenv("PIPS_BF16_DMA"); mode = e ? atoi(e) : 1; }
    if (!mode || !a_bf16 || a.K % 64 != 0 || a.K < 512 || a.lda % 8 != 0 || a.ldc % 4 != 0) return 1;
    if (a.M % 256 != 0 || a.N % 128 != 0 || (long)(a.M / 256) * (a.N / 128) < 256) return 1;
    const int epi = a.epi & 0xff;
    if (epi == EPI_GELU && out_bf16 && a.K == 512) return launch_dma_tile<EPI_GELU, true>(a, st);   // 8 super-stages = 8 pieces
    if (epi == EPI_RESIDUAL && !out_bf16 && a.R != nullptr && a.ldr % 4 == 0) return launch_dma_tile<EPI_RESIDUAL, false>(a, st);
    return 1;
}

}  // namespace pips
