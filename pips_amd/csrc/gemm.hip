// fp32 GEMM / implicit-GEMM convolution on the CDNA4 matrix cores.
//
// One kernel template serves the encoder's 3x3 / 1x1 convolutions (nets/pips.py:135-136,
// 169-170, 221-223 -- A rows are gathered from an NHWC map, zero padded) and the mixer's
// Linear layers (nets/pips.py:115-122 -- A is a plain row-major matrix).  The arithmetic
// is v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation (bitwise an fmaf
// chain), so the result differs from the reference's fp32 conv/addmm only by summation
// order.  Tiling is for 64-lane waves: each wave owns TM x TN tiles of 32x32, a lane
// feeds A[row = lane&31][k = lane>>5] / B[k = lane>>5][col = lane&31] per MFMA.
//
// A block is WGM x WGN x KS waves: WGM x WGN tile the BM x BN output, the KS wave groups
// split every staged K block (32*KS wide) between them and are summed through LDS at the
// end -- small problems (M = 2048 rows at B=1) get two waves per SIMD without shrinking
// the tile.  LDS image per stage: As[BM][32*KS+4], Bs[BN][32*KS+4] floats; the 16-byte row
// pad makes the ds_read_b128 fragment reads conflict-free for every 16-lane service
// group.  Within a 32-wide K slice the two lane halves take interleaved groups of four K
// values (half h reads k = 8*kk + 4*h + j), the same permutation on A and B, so one
// ds_read_b128 per operand feeds four MFMAs.  Two LDS stages, one barrier per K block:
//   global(kb+1) -> registers  ||  MFMA on stage kb&1 ;  registers -> stage (kb+1)&1.
#include "common.h"

#include <cstdlib>

namespace pips {



template <int BM, int BN, int WGM, int WGN, int KS, bool CONV>
// 128x128 tiles (64 accumulators per lane) must keep two blocks per CU: cap them at 256 registers
__global__ __launch_bounds__(WGM * WGN * KS * 64, (BM * BN >= 128 * 128 && WGM * WGN * KS == 4) ? 2 : 1) void igemm_f32_kernel(GemmArgs p) {
    constexpr int NT = WGM * WGN * KS * 64;
    constexpr int BKB = 32 * KS;                    // K values staged per iteration
    constexpr int LD = BKB + 4;                     // LDS row stride (floats)
    constexpr int TPR = BKB / 4;                    // loader threads per row (float4 each)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWS_PER_PASS = NT / TPR;
    constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;
    constexpr int STAGE = (BM + BN) * LD;           // floats per LDS stage
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/loader mismatch");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be 32-granular");
    static_assert(KS == 1 || (KS - 1) * BM * BN <= 2 * STAGE, "K-split reduction does not fit the stages");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // two stages: [As | Bs] x 2

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (WGM * WGN);
    const int wmn = wave - ks * (WGM * WGN);
    const int wm = wmn / WGN, wn = wmn % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const Tile3 tile_ = xcd_tile_order(p.swz != 0);            // (common.h: XCD-aware order over the whole grid)
    const int bx = tile_.x, by = tile_.y;
    const int m0 = bx * BM, n0 = by * BN;
    const int frame = tile_.z;
#ifdef PIPS_GEMM_TRACE
    // tools/gemm_trace.py: per-block phase timestamps (100 MHz constant clock) for plain GEMMs
    unsigned long long* tr_ = p.trace + 8 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
#define PIPS_T(slot) if (tid == 0) tr_[slot] = wall_clock64();
    if (tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr_[5] = hw; tr_[6] = xcc;
    }
#else
#define PIPS_T(slot)
#endif
    PIPS_T(0)

    const float* __restrict__ Abase = p.A;
    float* __restrict__ Cbase = p.C;
    if (CONV) {
        Abase += (size_t)frame * p.H * p.Win * p.Cin;
        Cbase += (size_t)frame * p.M * p.ldc;
    }

    // loader coordinates: thread -> (row = tid/TPR + pass*ROWS_PER_PASS, 4 floats at cg*4).
    // Rows past M / N are clamped to a valid row instead of predicated: an output row
    // (column) depends only on its own A row (W row) and is never stored when out of range.
    // Per-pass state lives in NAMED scalars (macro-expanded for passes 0..5), not arrays:
    // with the compiler barrier that pins the prefetch, hipcc leaves arrays in scratch.
    const int lrow = tid / TPR, cg = tid % TPR;
    static_assert(PA <= 4 && PB <= 6, "extend the pass macros");
#define PIPS_PASSES_A(X) X(0) X(1) X(2) X(3)
#define PIPS_PASSES_B(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define PIPS_DECL_A(i) int a_hi##i = 0, a_wi##i = 0; unsigned a_off##i = 0; float4 ra##i = make_float4(0.f, 0.f, 0.f, 0.f);
#define PIPS_DECL_B(i) unsigned b_off##i = 0; float4 rb##i = make_float4(0.f, 0.f, 0.f, 0.f);
    PIPS_PASSES_A(PIPS_DECL_A)
    PIPS_PASSES_B(PIPS_DECL_B)
#define PIPS_INIT_A(i)                                                        \
    if constexpr (i < PA) {                                                   \
        int m_ = m0 + lrow + i * ROWS_PER_PASS;                               \
        m_ = m_ < p.M ? m_ : p.M - 1;                                         \
        if (CONV) {                                                           \
            const int ho_ = m_ / p.Wo, wo_ = m_ - ho_ * p.Wo;                 \
            a_hi##i = ho_ * p.cstride - p.pad;                                \
            a_wi##i = wo_ * p.cstride - p.pad;                                \
        } else {                                                              \
            a_off##i = (unsigned)m_ * (unsigned)p.lda + cg * 4;               \
        }                                                                     \
    }
#define PIPS_INIT_B(i)                                                        \
    if constexpr (i < PB) {                                                   \
        int n_ = n0 + lrow + i * ROWS_PER_PASS;                               \
        n_ = n_ < p.N ? n_ : p.N - 1;                                         \
        b_off##i = (unsigned)n_ * (unsigned)p.K + cg * 4;                     \
    }
    PIPS_PASSES_A(PIPS_INIT_A)
    PIPS_PASSES_B(PIPS_INIT_B)
    (void)a_hi0; (void)a_wi0; (void)a_off0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define PIPS_LOADC_A(i)                                                                             \
    if constexpr (i < PA) {                                                                         \
        const int hi_ = a_hi##i + kh_, wi_ = a_wi##i + kw_;                                         \
        const bool ok_ = (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.Win;          \
        const int hc_ = ok_ ? hi_ : 0, wc_ = ok_ ? wi_ : 0;                                         \
        const float4 v_ = *reinterpret_cast<const float4*>(                                         \
            Abase + ((size_t)hc_ * p.Win + wc_) * p.Cin + c0_ + cg * 4);                            \
        ra##i = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                         \
    }
#define PIPS_LOADP_A(i) \
    if constexpr (i < PA) ra##i = *reinterpret_cast<const float4*>(Abase + a_off##i + k0_);
#define PIPS_LOAD_B(i) \
    if constexpr (i < PB) rb##i = *reinterpret_cast<const float4*>(p.W + b_off##i + k0_);
#define PIPS_LOAD_TILES(kb_)                                                                        \
    {                                                                                               \
        const int k0_ = (kb_) * BKB;                                                                \
        if (CONV) {                                                                                 \
            /* Cin % BKB == 0: a staged K block never straddles a filter tap */                     \
            const int tap_ = k0_ / p.Cin;                                                           \
            const int c0_ = k0_ - tap_ * p.Cin;                                                     \
            const int kh_ = tap_ / p.KW, kw_ = tap_ - kh_ * p.KW;                                   \
            PIPS_PASSES_A(PIPS_LOADC_A)                                                             \
        } else {                                                                                    \
            PIPS_PASSES_A(PIPS_LOADP_A)                                                             \
        }                                                                                           \
        PIPS_PASSES_B(PIPS_LOAD_B)                                                                  \
    }
#define PIPS_STORE_A(i) \
    if constexpr (i < PA) *reinterpret_cast<float4*>(&As_[(lrow + i * ROWS_PER_PASS) * LD + cg * 4]) = ra##i;
#define PIPS_STORE_B(i) \
    if constexpr (i < PB) *reinterpret_cast<float4*>(&Bs_[(lrow + i * ROWS_PER_PASS) * LD + cg * 4]) = rb##i;
#define PIPS_STORE_TILES(buf_)                                                                      \
    {                                                                                               \
        float* As_ = smem + (buf_) * STAGE;                                                         \
        float* Bs_ = As_ + BM * LD;                                                                 \
        PIPS_PASSES_A(PIPS_STORE_A)                                                                 \
        PIPS_PASSES_B(PIPS_STORE_B)                                                                 \
    }
#define PIPS_FRAGS(dst_a, dst_b, kk_)                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
            dst_a[i] = *reinterpret_cast<const float4*>(a_frag + i * 32 * LD + (kk_) * 8);          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
            dst_b[j] = *reinterpret_cast<const float4*>(b_frag + j * 32 * LD + (kk_) * 8);          \
    }
// Plain GEMMs accumulate C^T (W fragment as the MFMA A operand): a lane then holds four
// CONSECUTIVE output columns per register quad -> 16-byte bias/residual loads and C stores.
// Convolutions keep C (column sums for the instance-norm statistics stay in-lane).
#define PIPS_MFMA1(c_)                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                          \
                acc[i][j] = CONV ? __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].c_, fb_[j].c_, acc[i][j], 0, 0, 0) \
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(fb_[j].c_, fa_[i].c_, acc[i][j], 0, 0, 0);
#define PIPS_MFMA4(fa, fb)                                                                          \
    {                                                                                               \
        const float4* fa_ = fa; const float4* fb_ = fb;                                             \
        PIPS_MFMA1(x) PIPS_MFMA1(y) PIPS_MFMA1(z) PIPS_MFMA1(w)                                     \
    }
    // fragments of K sub-step kk+1 are read from LDS before the MFMAs of sub-step kk issue
#define PIPS_COMPUTE(buf_)                                                                          \
    {                                                                                               \
        const float* a_frag = smem + (buf_) * STAGE + (wm * WTM + l31) * LD + ks * 32 + half * 4;   \
        const float* b_frag = smem + (buf_) * STAGE + BM * LD + (wn * WTN + l31) * LD + ks * 32 + half * 4; \
        float4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];                                                  \
        PIPS_FRAGS(fa0, fb0, 0);                                                                    \
        PIPS_FRAGS(fa1, fb1, 1);                                                                    \
        PIPS_MFMA4(fa0, fb0);                                                                       \
        PIPS_FRAGS(fa0, fb0, 2);                                                                    \
        PIPS_MFMA4(fa1, fb1);                                                                       \
        PIPS_FRAGS(fa1, fb1, 3);                                                                    \
        PIPS_MFMA4(fa0, fb0);                                                                       \
        PIPS_MFMA4(fa1, fb1);                                                                       \
    }

    const int nk = p.K / BKB;
    PIPS_LOAD_TILES(0);
    PIPS_STORE_TILES(0);
    __syncthreads();
    PIPS_T(1)
#ifdef PIPS_GEMM_TRACE
    const long long cyc0_ = clock64();
#endif
    int buf = 0;
#ifdef PIPS_GEMM_ABLATE
    const bool ab_ld = !(p.epi & 0x100), ab_st = !(p.epi & 0x200), ab_bar = !(p.epi & 0x400);
    for (int kb = 0; kb + 1 < nk; ++kb) {
        if (ab_ld) PIPS_LOAD_TILES(kb + 1);
        asm volatile("" ::: "memory");
        PIPS_COMPUTE(buf);
        asm volatile("" ::: "memory");
        if (ab_st) PIPS_STORE_TILES(buf ^ 1);
        if (ab_bar) __syncthreads();
        buf ^= 1;
    }
#else
    for (int kb = 0; kb + 1 < nk; ++kb) {
        PIPS_LOAD_TILES(kb + 1);
        // keep the global loads ahead of the MFMA phase: left alone, the scheduler sinks them
        // next to the ds_writes and exposes the full memory latency every iteration
        // (a compiler memory barrier; __builtin_amdgcn_sched_barrier here leaves ra/rb in scratch)
        asm volatile("" ::: "memory");
        PIPS_COMPUTE(buf);
        asm volatile("" ::: "memory");
        PIPS_STORE_TILES(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#endif
    PIPS_COMPUTE(buf);
    PIPS_T(2)
#ifdef PIPS_GEMM_TRACE
    if (tid == 0) tr_[7] = (unsigned long long)(clock64() - cyc0_);
#endif
#undef PIPS_LOAD_TILES
#undef PIPS_STORE_TILES
#undef PIPS_PASSES_A
#undef PIPS_PASSES_B
#undef PIPS_DECL_A
#undef PIPS_DECL_B
#undef PIPS_INIT_A
#undef PIPS_INIT_B
#undef PIPS_LOADC_A
#undef PIPS_LOADP_A
#undef PIPS_LOAD_B
#undef PIPS_STORE_A
#undef PIPS_STORE_B
#undef PIPS_FRAGS
#undef PIPS_MFMA1
#undef PIPS_MFMA4
#undef PIPS_COMPUTE

    // ---- K-split reduction: groups ks>0 hand their accumulators to group 0 through LDS
    if (KS > 1) {
        __syncthreads();                                  // all waves are done with the stages
        float* red = smem;                                // [(KS-1)][WGM*WGN][TM*TN*16][64]
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((ks - 1) * (WGM * WGN) + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] =
                            acc[i][j][r];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int g = 0; g < KS - 1; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += red[(((g * (WGM * WGN) + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
    }

    PIPS_T(3)
    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float csum[TN], csq[TN], piv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = csq[j] = piv[j] = 0.f;
    const int epi = p.epi & 0xff;

    if (!CONV) {
        // transposed accumulators: MFMA row index = output column n, MFMA column = output row m
        const bool vec_ok = (p.ldc & 3) == 0 && (epi != EPI_RESIDUAL || (p.ldr & 3) == 0);
        if (vec_ok && m0 + BM <= p.M && n0 + BN <= p.N) {
            const int row0 = m0 + wm * WTM + l31, col0 = n0 + wn * WTN + 4 * half;
            if (epi == EPI_GELU) epilogue_full_tile<EPI_GELU, false, TM, TN>(acc, p.bias, p.R, p.ldr, Cbase, p.ldc, row0, col0);
            else if (epi == EPI_RESIDUAL) epilogue_full_tile<EPI_RESIDUAL, false, TM, TN>(acc, p.bias, p.R, p.ldr, Cbase, p.ldc, row0, col0);
            else epilogue_full_tile<EPI_BIAS, false, TM, TN>(acc, p.bias, p.R, p.ldr, Cbase, p.ldc, row0, col0);
            PIPS_T(4)
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * WTM + i * 32 + l31;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wn * WTN + j * 32 + 8 * g + 4 * half;
                    if (col >= p.N) continue;
                    float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (vec_ok && col + 3 < p.N) {
                        if (p.bias != nullptr) {
                            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
                            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                        }
                        if (epi == EPI_GELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
                        } else if (epi == EPI_RESIDUAL) {
                            const float4 r4 = *reinterpret_cast<const float4*>(p.R + (size_t)row * p.ldr + col);
                            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                        }
                        *reinterpret_cast<float4*>(Cbase + (size_t)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e < p.N) {
                                float t = v[e] + (p.bias != nullptr ? p.bias[col + e] : 0.f);
                                if (epi == EPI_GELU) t = gelu_exact(t);
                                else if (epi == EPI_RESIDUAL) t += p.R[(size_t)row * p.ldr + col + e];
                                Cbase[(size_t)row * p.ldc + col + e] = t;
                            }
                        }
                    }
                }
            }
        }
        PIPS_T(4)
        return;
    }

    // Bias is added to every accumulator BEFORE the (predicated) stores: a bias load first used
    // inside a predicated block makes hipcc put s_waitcnt vmcnt(0) in front of each store, which
    // also waits for the previous store's acknowledgement (16 serialized round trips per tile).
    const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        const bool col_ok = col < p.N;
        const float bv = p.bias != nullptr ? p.bias[col_ok ? col : p.N - 1] : 0.f;
        piv[j] = __shfl(acc[0][j][0] + bv, l31);      // the wave's first row in this column (lanes of half 0, r = 0)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            const int rbase = m0 + wm * WTM + i * 32 + 4 * half;
            float* cp = Cbase + (size_t)rbase * p.ldc + col;
            if (full_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
                    const float d = v[r] - piv[j];
                    csum[j] += d;
                    csq[j] += d * d;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M && col_ok) {
                        cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
                        const float d = v[r] - piv[j];
                        csum[j] += d;
                        csq[j] += d * d;
                    }
                }
            }
        }
    }

    if (CONV && p.stats != nullptr) {
        const int left = p.M - (m0 + wm * WTM), nvalid = left < 0 ? 0 : (left > WTM ? WTM : left);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            store_conv_partial(p.stats, frame, (int)gridDim.x * WGM, bx * WGM + wm, p.N, n0 + wn * WTN + j * 32 + l31, half,
                               csum[j], csq[j], piv[j], nvalid);
    }
    PIPS_T(4)
}

#ifdef PIPS_GEMM_TRACE
static unsigned long long* trace_buffer() {
    static void* buf = nullptr;
    if (!buf) { (void)hipMalloc(&buf, 8u << 20); (void)hipMemset(buf, 0, 8u << 20); }
    return reinterpret_cast<unsigned long long*>(buf);
}
extern "C" int pips_trace_read(void* host, size_t bytes) {
    return (int)hipMemcpy(host, trace_buffer(), bytes, hipMemcpyDeviceToHost);
}
#endif

// XCD-aware tile order (common.h).  [measured, profiles/r4_probe_f32_gemm_xcd_order.txt] plain GEMMs: neutral (-0.5 % .. +0.6 %)
// up to M = 65536 rows, +3.4 % .. +5.3 % on the mixer pass at M = 131072 (BASELINE configs[3]), where the 2048-wide operand
// (1 GiB) no longer fits the 256 MiB Infinity Cache and every XCD otherwise pulls all of W and an eighth-interleaved share of
// the rows; convolutions: no effect at 8 x 368x496, 32 x 720x1280 (fp32) or 64 x 368x496 (bf16 maps).  So: plain GEMMs whose
// wide operand exceeds the Infinity Cache.  PIPS_GEMM_SWZ (tuning builds): 0 off, 1 on wherever there are >= 64 tiles.
static int swizzle_on(bool conv, long tiles, const GemmArgs& a) {
    const int force = PIPS_TUNE("PIPS_GEMM_SWZ", -1);
    if (force >= 0) return force != 0 && tiles >= 64;
    return !conv && (long long)a.M * (a.N > a.K ? a.N : a.K) * 4 > (256ll << 20);
}

template <int BM, int BN, int WGM, int WGN, int KS, bool CONV>
static int launch_tile(const GemmArgs& a_in, int frames, hipStream_t st) {
    static_assert(!CONV || KS == 1, "conv statistics assume KS == 1");
    GemmArgs a = a_in;
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), frames);
    dim3 block(WGM * WGN * KS * 64);
    a.swz = swizzle_on(CONV, (long)grid.x * grid.y * grid.z, a);
    size_t lds = (size_t)2 * (BM + BN) * (32 * KS + 4) * sizeof(float);
    auto kern = igemm_f32_kernel<BM, BN, WGM, WGN, KS, CONV>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> raised{0};      // per instantiation, one bit per device
        const int rc = ensure_dynamic_lds(raised, (const void*)kern, lds);
        if (rc != PIPS_OK) return rc;
    }
#ifdef PIPS_GEMM_TRACE
    a.trace = trace_buffer();
#endif
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    PIPS_CHECK_LAUNCH("igemm_f32_kernel");
    return PIPS_OK;
}

// Debug/tuning hooks: PIPS_GEMM_TILE=<id> forces one tile configuration for plain GEMMs;
// PIPS_GEMM_TILE_UP / PIPS_GEMM_TILE_DOWN do so only for N > K / N < K (the mixer's up- and
// down-projections), for in-situ A/B runs of tools/mixer_bench.py.
static int forced_tile(const GemmArgs& a) {
    const int all = PIPS_TUNE("PIPS_GEMM_TILE", -1), up = PIPS_TUNE("PIPS_GEMM_TILE_UP", -1),
              down = PIPS_TUNE("PIPS_GEMM_TILE_DOWN", -1);
    (void)up; (void)down;
    if (all >= 0) return all;
    if (a.N > a.K && up >= 0) return up;
    if (a.N < a.K && down >= 0) return down;
    return -1;
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
    PIPS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    PIPS_CHECK_ARG(a.K % 32 == 0, "gemm: K=%d must be a multiple of 32", a.K);
    PIPS_CHECK_ARG((a.lda % 4) == 0, "gemm: lda must be a multiple of 4 floats");
    PIPS_CHECK_ARG((unsigned long long)a.M * (unsigned long long)a.lda < (1ull << 32) &&
                       (unsigned long long)a.N * (unsigned long long)a.K < (1ull << 32),
                   "gemm: operand exceeds 2^32 elements");
    const bool k64 = a.K % 64 == 0;
    if (forced_tile(a) < 0) {                              // the four-wave assembly kernels of gemm_f32_t4.hip (up- / down-projection forms)
        int tpb = 1;
        const int route = gemm_f32_t4_route(a, &tpb);
        if (route) return launch_gemm_f32_t4(a, route, tpb, st);
    }
    switch (forced_tile(a)) {
        case 0: return launch_tile<128, 128, 2, 2, 1, false>(a, 1, st);
        case 1: return launch_tile<128, 64, 2, 2, 1, false>(a, 1, st);
        case 2: return launch_tile<64, 128, 2, 2, 1, false>(a, 1, st);
        case 3: return launch_tile<64, 64, 2, 2, 1, false>(a, 1, st);
        case 4: if (k64) return launch_tile<128, 128, 2, 2, 2, false>(a, 1, st); break;
        case 5: if (k64) return launch_tile<64, 64, 2, 2, 2, false>(a, 1, st); break;
        case 6: if (k64) return launch_tile<128, 64, 2, 2, 2, false>(a, 1, st); break;
        case 7: return launch_tile<32, 64, 1, 2, 1, false>(a, 1, st);     // measured at M=2048: 57 TF
        case 8: return launch_tile<64, 32, 2, 1, 1, false>(a, 1, st);     // 56 TF
        case 9: if (a.K % 128 == 0) return launch_tile<64, 64, 2, 2, 4, false>(a, 1, st); break;
        case 10: return launch_tile<256, 128, 4, 2, 1, false>(a, 1, st);   // down-proj: within 0.5 % of KS=2
        default: break;
    }
    // Measured on MI355X (tools/gemm_bench.py): with >= ~2 blocks per CU of 128x128 the big
    // tile wins (90-105 TF at M=16384); the M=2048 mixer GEMMs are prologue/epilogue bound
    // and want many small blocks (64x64: 86 TF at N=2048) or, when even 64x64 gives only one
    // block per CU (N=512), two K-split wave groups per block (70 TF vs 59).
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    if (b128 >= 400) return launch_tile<128, 128, 2, 2, 1, false>(a, 1, st);
    if (b64 >= 800 || !k64) return launch_tile<64, 64, 2, 2, 1, false>(a, 1, st);
    return launch_tile<64, 64, 2, 2, 2, false>(a, 1, st);
}

// tile choice of launch_conv (the caller learns the M-tile count through *tiles_m)
static void conv_tile(int rows, int cout, int frames, int* bm, int* bn) {
    int n = (cout % 128 == 0) ? 128 : (cout % 96 == 0 ? 96 : 64);
    long blocks128 = (long)cdiv(rows, 128) * (cout / n) * frames;
    *bm = (blocks128 >= 384 && n != 64) ? 128 : 64;      // Cout=64: 64x64 tiles measured +7 % (85 vs 79 TF)
    // the 46x62 and 23x31 layers launch under 400 blocks of 64x128: halve the tile to fill more CUs
    // (measured: 104.7 -> 85.8 us and 59 -> 39 us; conv2's 714 blocks and the 96-channel layers prefer the big tile)
    if (*bm == 64 && n == 128 && (long)cdiv(rows, 64) * (cout / 128) * frames < 400) n = 64;
    *bn = n;
    const int force_bm = PIPS_TUNE("PIPS_CONV_BM", 0);      // tuning hook: 64|128
    if (force_bm == 64 || force_bm == 128) *bm = force_bm;
}

int launch_conv(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st) {
    PIPS_CHECK_ARG(a.Cin % 32 == 0, "conv: Cin=%d must be a multiple of 32", a.Cin);
    PIPS_CHECK_ARG(a.N % 32 == 0 && (a.N % 64 == 0 || a.N % 96 == 0), "conv: unsupported Cout=%d", a.N);
    PIPS_CHECK_ARG(a.K == a.KH * a.KW * a.Cin, "conv: K mismatch");
    {                                                        // the four-wave assembly kernels of conv_f32_t4.hip (the big 3x3 layers)
        const int cfg = conv_f32_t4_config(a, frames);
        if (cfg >= 0) return launch_conv_f32_t4(a, cfg, frames, tiles_m, st);
    }
    int bm, bn;
    conv_tile(a.M, a.N, frames, &bm, &bn);
    // partials per frame: m tiles x wave rows (WGM = 4 for the 128x96 tile, 2 elsewhere)
    if (tiles_m) *tiles_m = cdiv(a.M, bm) * ((bn == 96 && bm == 128) ? 4 : 2);
    if (bn == 128) {
        return bm == 128 ? launch_tile<128, 128, 2, 2, 1, true>(a, frames, st)
                         : launch_tile<64, 128, 2, 2, 1, true>(a, frames, st);
    } else if (bn == 96) {
        return bm == 128 ? launch_tile<128, 96, 4, 1, 1, true>(a, frames, st)
                         : launch_tile<64, 96, 2, 1, 1, true>(a, frames, st);
    }
    return bm == 128 ? launch_tile<128, 64, 2, 2, 1, true>(a, frames, st)
                     : launch_tile<64, 64, 2, 2, 1, true>(a, frames, st);
}

}  // namespace pips
