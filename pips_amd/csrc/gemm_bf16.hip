// bf16-operand GEMM for the mixer's Linear layers (BASELINE config 3: "bf16 MFMA operands").
//
// C[M,N] = epi(A[M,K] * W[N,K]^T + bias): A is fp32 (converted to bf16 with the hardware
// round-to-nearest-even v_cvt_pk_bf16_f32 while it is staged into LDS) or already bf16 (the
// 2048-wide hidden activation, the one tensor worth storing narrow); W is bf16 (converted once
// at weight-pack time); products are exact, accumulation is fp32 (v_mfma_f32_32x32x16_bf16),
// bias / GELU / residual / LayerNorm inputs and the residual stream stay fp32.  Same block
// structure as gemm.hip (two LDS stages, one barrier per K block, K-split wave groups, C^T
// accumulators for 16-byte epilogue accesses); a staged K block is 64 elements = the same 128
// bytes per row, so the conflict-free 16-byte-padded LDS image and the fragment addressing
// (lane -> row lane&31, 16 bytes at 32*kk + 16*(lane>>5)) carry over unchanged: per MFMA a lane
// supplies 8 consecutive K values.
#include "common.h"

#include <cstdlib>

namespace pips {

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 pack_bf16x8(const float4& lo, const float4& hi) {
    f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    bf16x8 b = __builtin_convertvector(v, bf16x8);
    return *reinterpret_cast<uint4*>(&b);
}

// CONV: implicit-GEMM convolution on an NHWC fp32 map (zero padding, per-frame m-tiling, the
// pivoted instance-norm partials (store_conv_partial) from the fp32 accumulators) -- same contract as the fp32
// conv of gemm.hip, with bf16 operands.  BKE = K elements per staged block per wave group: 64,
// or 32 for Cin = 96 / 416 so that a block never straddles a filter tap.
template <int BM, int BN, int WGM, int WGN, int KS, bool A_BF16, bool OUT_BF16, int BKE = 64, bool CONV = false>
// 128x128 tiles (64 accumulators per lane) must keep two blocks per CU: cap them at 256 registers
__global__ __launch_bounds__(WGM * WGN * KS * 64, (BM * BN >= 128 * 128 && WGM * WGN * KS == 4) ? 2 : 1) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int NT = WGM * WGN * KS * 64;
    constexpr int BKB = BKE * KS;                   // K elements staged per iteration
    static_assert(!CONV || KS == 1, "conv: no K split");
    constexpr int LDB = BKB * 2 + 16;               // LDS row stride in BYTES (16-byte pad)
    constexpr int TPR = BKB / 8;                    // loader threads per row (8 elements each)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWS_PER_PASS = NT / TPR;
    constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;
    constexpr int STAGE = (BM + BN) * LDB;          // bytes per LDS stage
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/loader mismatch");
    static_assert(PA <= 4 && PB <= 4, "extend the pass macros");
    static_assert(KS == 1 || (KS - 1) * BM * BN * 4 <= 2 * STAGE, "K-split reduction does not fit the stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (WGM * WGN);
    const int wmn = wave - ks * (WGM * WGN);
    const int wm = wmn / WGN, wn = wmn % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const Tile3 tile_ = xcd_tile_order(p.swz != 0);            // (common.h: XCD-aware order over the whole grid)
    const int bx = tile_.x;
    const int m0 = bx * BM, n0 = tile_.y * BN;
    const int lrow = tid / TPR, cg = tid % TPR;
    const int frame = tile_.z;

    const float* __restrict__ Af = p.A;                                          // A_BF16 == false
    if (CONV) Af += (size_t)frame * p.H * p.Win * p.Cin;
    const unsigned short* __restrict__ Ab = reinterpret_cast<const unsigned short*>(p.A);   // A_BF16 == true
    if (CONV) Ab += (size_t)frame * p.H * p.Win * p.Cin;
    const unsigned short* __restrict__ Wb = reinterpret_cast<const unsigned short*>(p.W);

#define PIPS_PASSES(X) X(0) X(1) X(2) X(3)
#define PIPS_DECL(i)                                                                     \
    unsigned a_off##i = 0, b_off##i = 0; int a_hi##i = 0, a_wi##i = 0;                   \
    float4 ra##i##l = make_float4(0.f, 0.f, 0.f, 0.f), ra##i##h = ra##i##l;              \
    uint4 rq##i = make_uint4(0, 0, 0, 0), rb##i = rq##i;
    PIPS_PASSES(PIPS_DECL)
#define PIPS_INIT(i)                                                                     \
    if constexpr (i < PA) {                                                              \
        int m_ = m0 + lrow + i * ROWS_PER_PASS;                                          \
        m_ = m_ < p.M ? m_ : p.M - 1;                                                    \
        if (CONV) {                                                                      \
            const int ho_ = m_ / p.Wo, wo_ = m_ - ho_ * p.Wo;                            \
            a_hi##i = ho_ * p.cstride - p.pad;                                           \
            a_wi##i = wo_ * p.cstride - p.pad;                                           \
        } else {                                                                         \
            a_off##i = (unsigned)m_ * (unsigned)p.lda + cg * 8;                          \
        }                                                                                \
    }                                                                                    \
    if constexpr (i < PB) {                                                              \
        int n_ = n0 + lrow + i * ROWS_PER_PASS;                                          \
        n_ = n_ < p.N ? n_ : p.N - 1;                                                    \
        b_off##i = (unsigned)n_ * (unsigned)p.K + cg * 8;                                \
    }
    PIPS_PASSES(PIPS_INIT)
    (void)a_hi0; (void)a_wi0; (void)a_hi1; (void)a_wi1; (void)a_hi2; (void)a_wi2; (void)a_hi3; (void)a_wi3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define PIPS_LOAD(i)                                                                                 \
    if constexpr (i < PA) {                                                                          \
        if constexpr (A_BF16 && CONV) {                                                              \
            const int hi_ = a_hi##i + kh_, wi_ = a_wi##i + kw_;                                      \
            const bool ok_ = (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.Win;       \
            const uint4 q_ = *reinterpret_cast<const uint4*>(                                        \
                Ab + ((size_t)(ok_ ? hi_ : 0) * p.Win + (ok_ ? wi_ : 0)) * p.Cin + c0_ + cg * 8);    \
            rq##i = ok_ ? q_ : make_uint4(0, 0, 0, 0);                                               \
        } else if constexpr (A_BF16) rq##i = *reinterpret_cast<const uint4*>(Ab + a_off##i + k0_);   \
        else if constexpr (CONV) {                                                                   \
            const int hi_ = a_hi##i + kh_, wi_ = a_wi##i + kw_;                                      \
            const bool ok_ = (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.Win;       \
            const float* src_ = Af + ((size_t)(ok_ ? hi_ : 0) * p.Win + (ok_ ? wi_ : 0)) * p.Cin + c0_ + cg * 8; \
            const float4 l_ = *reinterpret_cast<const float4*>(src_);                                \
            const float4 h_ = *reinterpret_cast<const float4*>(src_ + 4);                            \
            ra##i##l = ok_ ? l_ : make_float4(0.f, 0.f, 0.f, 0.f);                                   \
            ra##i##h = ok_ ? h_ : make_float4(0.f, 0.f, 0.f, 0.f);                                   \
        } else {                                                                                     \
            ra##i##l = *reinterpret_cast<const float4*>(Af + a_off##i + k0_);                        \
            ra##i##h = *reinterpret_cast<const float4*>(Af + a_off##i + k0_ + 4);                    \
        }                                                                                            \
    }                                                                                                \
    if constexpr (i < PB) rb##i = *reinterpret_cast<const uint4*>(Wb + b_off##i + k0_);
#define PIPS_LOAD_TILES(kb_)                                                                         \
    {                                                                                                \
        const int k0_ = (kb_) * BKB;                                                                 \
        const int tap_ = CONV ? k0_ / p.Cin : 0;          /* Cin % BKE == 0: one tap per block */   \
        const int c0_ = CONV ? k0_ - tap_ * p.Cin : 0;                                               \
        const int kh_ = CONV ? tap_ / p.KW : 0, kw_ = CONV ? tap_ - kh_ * p.KW : 0;                  \
        (void)c0_; (void)kh_; (void)kw_;                                                             \
        PIPS_PASSES(PIPS_LOAD)                                                                       \
    }
#define PIPS_STORE(i)                                                                                \
    if constexpr (i < PA)                                                                            \
        *reinterpret_cast<uint4*>(As_ + (lrow + i * ROWS_PER_PASS) * LDB + cg * 16) =               \
            A_BF16 ? rq##i : pack_bf16x8(ra##i##l, ra##i##h);                                        \
    if constexpr (i < PB)                                                                            \
        *reinterpret_cast<uint4*>(Bs_ + (lrow + i * ROWS_PER_PASS) * LDB + cg * 16) = rb##i;
#define PIPS_STORE_TILES(buf_) { char* As_ = smem + (buf_) * STAGE; char* Bs_ = As_ + BM * LDB; PIPS_PASSES(PIPS_STORE) }
#define PIPS_COMPUTE(buf_)                                                                           \
    {                                                                                                \
        const char* a_frag = smem + (buf_) * STAGE + (wm * WTM + l31) * LDB + ks * (BKE * 2) + half * 16;  \
        const char* b_frag = smem + (buf_) * STAGE + BM * LDB + (wn * WTN + l31) * LDB + ks * (BKE * 2) + half * 16; \
        _Pragma("unroll") for (int kk = 0; kk < BKE / 16; ++kk) {                                    \
            uint4 fa[TM], fb[TN];                                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                fa[i] = *reinterpret_cast<const uint4*>(a_frag + i * 32 * LDB + kk * 32);            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
                fb[j] = *reinterpret_cast<const uint4*>(b_frag + j * 32 * LDB + kk * 32);            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                       \
                    acc[i][j] = CONV ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(                      \
                                           *reinterpret_cast<const bf16x8*>(&fa[i]),                 \
                                           *reinterpret_cast<const bf16x8*>(&fb[j]), acc[i][j], 0, 0, 0) \
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(                      \
                                           *reinterpret_cast<const bf16x8*>(&fb[j]),                 \
                                           *reinterpret_cast<const bf16x8*>(&fa[i]), acc[i][j], 0, 0, 0); \
        }                                                                                            \
    }

    const int nk = p.K / BKB;
    PIPS_LOAD_TILES(0);
    PIPS_STORE_TILES(0);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb + 1 < nk; ++kb) {
        PIPS_LOAD_TILES(kb + 1);
        asm volatile("" ::: "memory");              // keep the prefetch ahead of the MFMA phase (see gemm.hip)
        PIPS_COMPUTE(buf);
        asm volatile("" ::: "memory");
        PIPS_STORE_TILES(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    PIPS_COMPUTE(buf);
#undef PIPS_PASSES
#undef PIPS_DECL
#undef PIPS_INIT
#undef PIPS_LOAD
#undef PIPS_LOAD_TILES
#undef PIPS_STORE
#undef PIPS_STORE_TILES
#undef PIPS_COMPUTE

    if (KS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
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

    if (CONV) {
        // C orientation: col = lane&31, row = (r&3) + 8*(r>>2) + 4*half; fp32 output + column partials
        float* __restrict__ Cc = p.C + (size_t)frame * p.M * p.ldc;                                     // !OUT_BF16
        unsigned short* __restrict__ Ch = reinterpret_cast<unsigned short*>(p.C) + (size_t)frame * p.M * p.ldc;
        float csum[TN], csq[TN], piv[TN];
        // bias added before the predicated stores, full tiles unpredicated (see gemm.hip)
        const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            csum[j] = csq[j] = 0.f;
            const int col = n0 + wn * WTN + j * 32 + l31;
            const bool col_ok = col < p.N;
            const float bv = p.bias != nullptr ? p.bias[col_ok ? col : p.N - 1] : 0.f;
            piv[j] = __shfl(acc[0][j][0] + bv, l31);  // the wave's first row in this column (lanes of half 0, r = 0)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
                const int rbase = m0 + wm * WTM + i * 32 + 4 * half;
                float* cp = Cc + (size_t)rbase * p.ldc + col;
                if constexpr (OUT_BF16) {
                    // bf16 map out: channel pairs packed across neighbouring lanes; statistics from the fp32 values
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < p.M && col_ok) {
                            const float d = v[r] - piv[j];
                            csum[j] += d;
                            csq[j] += d * d;
                        }
                    }
                    const int cpair = n0 + wn * WTN + j * 32 + (l31 & ~1);              // N is even: a pair is in or out together
                    store_c_tile_bf16(v, l31, half, [&](int px) -> unsigned short* {
                        const int row = m0 + wm * WTM + i * 32 + px;
                        return (row < p.M && cpair < p.N) ? Ch + (size_t)row * p.ldc + (n0 + wn * WTN + j * 32) : nullptr;
                    });
                } else if (full_tile) {
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
        if (p.stats != nullptr) {
            const int left = p.M - (m0 + wm * WTM), nvalid = left < 0 ? 0 : (left > WTM ? WTM : left);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                store_conv_partial(p.stats, frame, (int)gridDim.x * WGM, bx * WGM + wm, p.N, n0 + wn * WTN + j * 32 + l31,
                                   half, csum[j], csq[j], piv[j], nvalid);
        }
        return;
    }

    // epilogue on C^T accumulators: MFMA row index = output column n, MFMA column = output row m
    const int epi = p.epi & 0xff;
    float* __restrict__ Cf = p.C;
    unsigned short* __restrict__ Cb = reinterpret_cast<unsigned short*>(p.C);
    if (m0 + BM <= p.M && n0 + BN <= p.N && (epi != EPI_RESIDUAL || (p.ldr & 3) == 0)) {
        const int row0 = m0 + wm * WTM + l31, col0 = n0 + wn * WTN + 4 * half;
        if (epi == EPI_GELU) epilogue_full_tile<EPI_GELU, OUT_BF16, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        else if (epi == EPI_RESIDUAL && OUT_BF16 && (p.epi & EPI_RES_BF16))
            epilogue_full_tile<EPI_RESIDUAL, OUT_BF16, TM, TN, true>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        else if (epi == EPI_RESIDUAL) epilogue_full_tile<EPI_RESIDUAL, OUT_BF16, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        else epilogue_full_tile<EPI_BIAS, OUT_BF16, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
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
                if (col + 3 >= p.N) continue;                      // N % 4 == 0 (checked by the launcher)
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.bias != nullptr) {
                    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
                if (epi == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
                } else if (epi == EPI_RESIDUAL) {
                    if (OUT_BF16 && (p.epi & EPI_RES_BF16)) {
                        const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.R) + (size_t)row * p.ldr + col);
                        v[0] += __uint_as_float(rb.x << 16); v[1] += __uint_as_float(rb.x & 0xffff0000u);
                        v[2] += __uint_as_float(rb.y << 16); v[3] += __uint_as_float(rb.y & 0xffff0000u);
                    } else {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.R + (size_t)row * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                }
                if (OUT_BF16) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 t = {v[0], v[1], v[2], v[3]};
                    bf16x4 o = __builtin_convertvector(t, bf16x4);
                    *reinterpret_cast<uint2*>(Cb + (size_t)row * p.ldc + col) = *reinterpret_cast<uint2*>(&o);
                } else {
                    *reinterpret_cast<float4*>(Cf + (size_t)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// XCD-aware tile order (common.h).  PIPS_BF16_SWZ (tuning builds): 0 off, 1 on from 64 tiles; default -1 = by kind and size.
static int bf16_swizzle_on(bool conv, long tiles) {
    const int force = PIPS_TUNE("PIPS_BF16_SWZ", -1);
    if (force >= 0) return force != 0 && tiles >= 64;
    return 0;
}

template <int BM, int BN, int WGM, int WGN, int KS, bool A_BF16, bool OUT_BF16, int BKE = 64>
static int launch_bf16_tile(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), 1);
    a.swz = bf16_swizzle_on(false, (long)grid.x * grid.y);
    dim3 block(WGM * WGN * KS * 64);
    const size_t lds = (size_t)2 * (BM + BN) * (BKE * KS * 2 + 16);
    auto kern = gemm_bf16_kernel<BM, BN, WGM, WGN, KS, A_BF16, OUT_BF16, BKE>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> raised{0};      // per instantiation, one bit per device
        const int rc = ensure_dynamic_lds(raised, (const void*)kern, lds);
        if (rc != PIPS_OK) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    PIPS_CHECK_LAUNCH("gemm_bf16_kernel");
    return PIPS_OK;
}

template <bool A_BF16, bool OUT_BF16>
static int pick_tile(const GemmArgs& a, hipStream_t st) {
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    if (a.K % 64 != 0) {                        // K % 32 == 0: the 544-wide input projection, 32-element K blocks
        if constexpr (!A_BF16) {                // (bf16 C: the input projection of a mixer with a bf16 residual stream)
            if (b128 >= 400) return launch_bf16_tile<128, 128, 2, 2, 1, false, OUT_BF16, 32>(a, st);
            return launch_bf16_tile<64, 64, 2, 2, 1, false, OUT_BF16, 32>(a, st);
        } else {
            set_error("gemm_bf16: K=%d needs fp32 A (32-element K blocks)", a.K);
            return PIPS_E_ARG;
        }
    }
    // tuning hook PIPS_BF16_BIG: 0 = no tile above 128x128, 1 = 256x128 only, 2 (default) = 256x256 where it still
    // gives every CU a tile (config 3 up-projection: 60.7 vs 67.7 us), 256x128 below that
    const int big = PIPS_TUNE("PIPS_BF16_BIG", 2);
    if (big >= 2 && (long)cdiv(a.M, 256) * cdiv(a.N, 256) >= 256) return launch_bf16_tile<256, 256, 4, 2, 1, A_BF16, OUT_BF16>(a, st);
    if (big && (long)cdiv(a.M, 256) * cdiv(a.N, 128) >= 256) return launch_bf16_tile<256, 128, 4, 2, 1, A_BF16, OUT_BF16>(a, st);
    if (b128 >= 400) return launch_bf16_tile<128, 128, 2, 2, 1, A_BF16, OUT_BF16>(a, st);
    if (b64 >= 800 || a.K % 128 != 0) return launch_bf16_tile<64, 64, 2, 2, 1, A_BF16, OUT_BF16>(a, st);
    return launch_bf16_tile<64, 64, 2, 2, 2, A_BF16, OUT_BF16>(a, st);
}

template <int BM, int BN, int BKE, bool A_BF16, bool OUT_BF16>
static int launch_conv_tile_t(const GemmArgs& a_in, int frames, hipStream_t st) {
    GemmArgs a = a_in;
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), frames);
    a.swz = bf16_swizzle_on(true, (long)grid.x * grid.y * grid.z);
    const size_t lds = (size_t)2 * (BM + BN) * (BKE * 2 + 16);
    auto kern = gemm_bf16_kernel<BM, BN, 2, 2, 1, A_BF16, OUT_BF16, BKE, true>;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    PIPS_CHECK_LAUNCH("gemm_bf16_kernel<conv>");
    return PIPS_OK;
}
// map types: 0 = fp32 in / fp32 out (pips_conv_nhwc_bf16), 1 = bf16 in / bf16 out, 2 = bf16 in / fp32 out (conv3 -> pyramid)
template <int BM, int BN, int BKE>
static int launch_conv_tile(const GemmArgs& a, int frames, hipStream_t st, int types) {
    if (types == 1) return launch_conv_tile_t<BM, BN, BKE, true, true>(a, frames, st);
    if (types == 2) return launch_conv_tile_t<BM, BN, BKE, true, false>(a, frames, st);
    return launch_conv_tile_t<BM, BN, BKE, false, false>(a, frames, st);
}

// bf16-operand convolution: NHWC fp32 map, weights bf16 [Cout][kh][kw][Cin]; raw fp32 output
// + instance-norm partials exactly like launch_conv.  Cout = 96 rides a 128-wide tile (the
// idle quarter costs nothing at the bf16 matrix rate).
int launch_conv_bf16(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st, int in_bf16, int out_bf16) {
    PIPS_CHECK_ARG(a.Cin % 32 == 0 && a.K == a.KH * a.KW * a.Cin, "conv_bf16: Cin %% 32, K = kh*kw*Cin");
    PIPS_CHECK_ARG(in_bf16 || !out_bf16, "conv_bf16: fp32 map in, bf16 map out is not built");
    PIPS_CHECK_ARG(a.N % 2 == 0, "conv_bf16: Cout must be even");
    if (conv_c96_t4_takes(a, frames, in_bf16, out_bf16)) return launch_conv_c96_t4(a, frames, tiles_m, st);    // 96 -> 96, 3x3 (conv_bf16_t4c.hip)
    if (in_bf16 == out_bf16) {
        const int rc = launch_conv3x3_c64_bf16(a, frames, tiles_m, st, in_bf16, out_bf16);     // 64 -> 64, 3x3: weights + halo patch in LDS
        if (rc != 1) return rc;
    }
    PIPS_CHECK_ARG(a.in_norm == nullptr, "conv_bf16: only the LDS-resident 64 -> 64 kernel normalises its input on load");
    const int types = in_bf16 ? (out_bf16 ? 1 : 2) : 0;
    const bool k64 = a.Cin % 64 == 0;
    const int bn = a.N <= 64 ? 64 : 128;
    const long blocks128 = (long)cdiv(a.M, 128) * cdiv(a.N, bn) * frames;
    const int bm = blocks128 >= 512 ? 128 : 64;
    if (tiles_m) *tiles_m = cdiv(a.M, bm) * 2;       // partials per frame: m tiles x wave rows (WGM = 2)
    // Cout = 256 (conv2, 416 -> 256): one 256-wide tile per 128 rows -- every A element is fetched once per tap instead of twice
    if (bm == 128 && a.N % 256 == 0 && !k64 && PIPS_TUNE("PIPS_CONV_BN256", 1) &&
        (long)cdiv(a.M, 128) * (a.N / 256) * frames >= 512)
        return launch_conv_tile<128, 256, 32>(a, frames, st, types);
    if (bm == 128) {
        if (bn == 128) return k64 ? launch_conv_tile<128, 128, 64>(a, frames, st, types) : launch_conv_tile<128, 128, 32>(a, frames, st, types);
        return k64 ? launch_conv_tile<128, 64, 64>(a, frames, st, types) : launch_conv_tile<128, 64, 32>(a, frames, st, types);
    }
    if (bn == 128) return k64 ? launch_conv_tile<64, 128, 64>(a, frames, st, types) : launch_conv_tile<64, 128, 32>(a, frames, st, types);
    return k64 ? launch_conv_tile<64, 64, 64>(a, frames, st, types) : launch_conv_tile<64, 64, 32>(a, frames, st, types);
}

// A: fp32 [M][lda] (a_bf16 = 0) or bf16 [M][lda]; W: bf16 [N][K]; C: fp32 or bf16 [M][ldc]
int launch_gemm_bf16(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st) {
    PIPS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16: empty problem");
    PIPS_CHECK_ARG(a.K % 32 == 0, "gemm_bf16: K=%d must be a multiple of 32", a.K);
    PIPS_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0 && a.lda % 8 == 0, "gemm_bf16: N, ldc %% 4 and lda %% 8 required");
    PIPS_CHECK_ARG((unsigned long long)a.M * (unsigned long long)a.lda < (1ull << 32) &&
                       (unsigned long long)a.N * (unsigned long long)a.K < (1ull << 32),
                   "gemm_bf16: operand exceeds 2^32 elements");
    // a bf16 residual exists only beside a bf16 output of the residual epilogue: anything else would read a bf16 R as fp32
    PIPS_CHECK_ARG(!(a.epi & EPI_RES_BF16) || (out_bf16 && (a.epi & 0xff) == EPI_RESIDUAL),
                   "gemm_bf16: PIPS_EPI_RES_BF16 needs out_bf16 and the residual epilogue (epi = PIPS_EPI_RESIDUAL | PIPS_EPI_RES_BF16)");
    if (gemm_bf16_t4_takes(a, a_bf16, out_bf16)) return launch_gemm_bf16_t4(a, st);      // the config-3 down-projection
    {
        int tpb = 1;
        if (gemm_bf16_t4up_takes(a, a_bf16, out_bf16, &tpb)) return launch_gemm_bf16_t4up(a, tpb, st);   // ... and up-projection
    }
    if (a_bf16) return out_bf16 ? pick_tile<true, true>(a, st) : pick_tile<true, false>(a, st);
    return out_bf16 ? pick_tile<false, true>(a, st) : pick_tile<false, false>(a, st);
}

}  // namespace pips
