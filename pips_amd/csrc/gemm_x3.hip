// fp32-grade GEMM / implicit-GEMM convolution on the bf16 matrix cores ("split-bf16", bf16x3).
//
// Every fp32 operand is split EXACTLY into three bf16 terms (round-to-nearest at each step),
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = x - h - m,
// and a product a*b is formed as six exact bf16 products accumulated in fp32,
//     a*b ~= al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh        (smallest first),
// dropping am*bl + al*bm + al*bl <= 2^-23 |a b|.  Measured against fp64 (tools/x3_check.py) the
// truncation error is ~70x below the rounding error an fp32 FMA chain of the same length carries,
// so the result is as accurate as the exact-fp32 MFMA kernel of gemm.hip -- at 6/16 of its matrix
// time (v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate on CDNA4).
//
// W is split once at weight-pack time into three planes [3][N][K]; A (activations / NHWC maps)
// stays fp32 in memory and is split while it is staged into LDS (6 VALU ops per value).  Block
// structure follows gemm.hip: two LDS stages, one barrier per K block (32 K values per wave
// group), K-split wave groups reduced through LDS, C^T accumulators for plain GEMMs.  The LDS
// image holds six planes per stage, unpadded XOR-swizzled rows (reads and writes conflict-free).  A
// wave should own a 64x64 tile (12 fragment reads per 24 MFMAs; smaller wave tiles are LDS-read
// bound).  Measured (PMC): the bf16 matrix pipe is ~30 % busy -- the global -> VGPR -> LDS staging
// does not hide behind the MFMAs of a wave that is alone on its SIMD.
#include "gemm_tail.h"

#include <cstdlib>

#ifndef PIPS_X3_ABL
#define PIPS_X3_ABL 0        // tools/x3_ablate.sh: timing-only ablations of the main loop (results invalid)
#endif

namespace pips {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Exact split of two values into bf16 terms by round-to-nearest-even (hardware v_cvt_pk_bf16_f32):
// h = bf16(x), m = bf16(x - h), l = x - h - m (8 significant bits or fewer are left: exact).  The
// three results hold the pairs (x0 in the low half).  Rounding rather than truncating keeps the
// dropped cross terms zero-mean, so their sum over K grows like sqrt(K), not K.
// (Remainders that are fp32 subnormals -- |x| below ~2^-110 -- flush to zero: h is kept.)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    const bf16x2_ r = __builtin_convertvector(v, bf16x2_);
    return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ void split3_x8(const float4& a, const float4& b, uint4& h, uint4& m, uint4& l) {
    split3_pair(a.x, a.y, h.x, m.x, l.x);
    split3_pair(a.z, a.w, h.y, m.y, l.y);
    split3_pair(b.x, b.y, h.z, m.z, l.z);
    split3_pair(b.z, b.w, h.w, m.w, l.w);
}

__global__ void split_bf16x3_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, size_t npairs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const float2 v = *reinterpret_cast<const float2*>(src + 2 * i);
    unsigned h, m, l;
    split3_pair(v.x, v.y, h, m, l);
    dst[i] = h;
    dst[npairs + i] = m;
    dst[2 * npairs + i] = l;
}

// fp32 [n] -> bf16 planes [3][n] (n even)
int launch_split_bf16x3(const float* src, size_t n, void* dst, hipStream_t st) {
    PIPS_CHECK_ARG((n & 1) == 0, "split_bf16x3: n must be even");
    const size_t npairs = n / 2;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, src,
                       reinterpret_cast<unsigned*>(dst), npairs);
    PIPS_CHECK_LAUNCH("split_bf16x3_kernel");
    return PIPS_OK;
}

// BKE = K values per wave group per staged block: 32 (two 16-wide MFMA steps per wave and stage), or
// 16 with KS = 2 -- the two wave groups take one K step each of a 32-wide stage, which puts two
// waves on every SIMD for the 128x128 tile at the LDS footprint of the one-group version.
template <int BM, int BN, int WGM, int WGN, int KS, bool CONV, int BKE = 32>
__global__ __launch_bounds__(WGM * WGN * KS * 64) void gemm_x3_kernel(GemmArgs p) {
    constexpr int NT = WGM * WGN * KS * 64;
    static_assert(BKE == 32 || (BKE == 16 && KS == 2), "BKE: 32, or 16 with two wave groups");
    constexpr int BKB = BKE * KS;
    constexpr int LDB = BKB * 2;                    // LDS row stride in bytes (one plane), unpadded:
    // 16-byte chunk c of row r lives at chunk c ^ swz(r) (XOR swizzle) -- conflict-free for the
    // fragment reads (8 consecutive rows, one logical chunk) AND for the staging writes (rows of
    // consecutive chunks); a padded row costs the ds_write_b128s a 2-way bank conflict
    constexpr int CH = LDB / 16;                    // chunks per row: 4 (KS=1) or 8 (KS=2)
    static_assert(CH == 4 || CH == 8, "swizzle assumes 64- or 128-byte rows");
    constexpr int TPR = BKB / 8;                    // loader threads per row (8 K values each)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int RPP = NT / TPR;                   // rows per loader pass
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int PLANE_A = BM * LDB, PLANE_B = BN * LDB;
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
    static_assert(BM % RPP == 0 && BN % RPP == 0 && PA >= 1 && PA <= 2 && PB >= 1 && PB <= 2, "tile/loader mismatch");
    static_assert(!CONV || KS == 1, "conv: no K split");
    static_assert(KS == 1 || (KS - 1) * BM * BN * 4 <= 2 * STAGE, "K-split reduction does not fit the stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (WGM * WGN);
    const int wmn = wave - ks * (WGM * WGN);
    const int wm = wmn / WGN, wn = wmn % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const Tile3 tile_ = xcd_tile_order(p.swz != 0);            // (common.h: XCD-aware order over the whole grid)
    const int m0 = tile_.x * BM, n0 = tile_.y * BN;
    const int lrow = tid / TPR, cg = tid % TPR;
    const int frame = tile_.z;
    // swizzle of a row: (r / rows-per-256-bytes) mod CH.  ds_read_b128 is served in 16-lane groups
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) over 64 banks: within a group the rows of equal
    // (r mod rows-per-256-B) must land on distinct chunks -- (r>>2)&3 / (r>>1)&7 does that; the
    // staging writes (8 contiguous lanes = whole rows, 32 banks) stay conflict-free.
    const int swz_w = (CH == 4 ? (lrow >> 2) : (lrow >> 1)) & (CH - 1);   // staging thread's row
    const int swz_r = (CH == 4 ? (l31 >> 2) : (l31 >> 1)) & (CH - 1);     // fragment lane's row
    const int wchunk = (cg ^ swz_w) * 16;
    const int rchunk0 = ((ks * (BKE / 8) + half) ^ swz_r) * 16;           // K step 0
    const int rchunk1 = ((ks * (BKE / 8) + 2 + half) ^ swz_r) * 16;       // K step 1 (BKE = 32 only)

    const float* __restrict__ Af = p.A;
    if (CONV) Af += (size_t)frame * p.H * p.Win * p.Cin;
    const unsigned short* __restrict__ Wb = reinterpret_cast<const unsigned short*>(p.W);
    const size_t wplane = (size_t)p.N * p.K;        // elements between the h / m / l planes of W

    // loader state in named scalars (arrays end up in scratch next to the compiler barriers below)
#define PIPS_PASSES(X) X(0) X(1)
#define PIPS_DECL(i)                                                                     \
    unsigned a_off##i = 0, b_off##i = 0; int a_hi##i = 0, a_wi##i = 0;                   \
    float4 ra##i##l = make_float4(0.f, 0.f, 0.f, 0.f), ra##i##h = ra##i##l;              \
    float4 na##i##l = ra##i##l, na##i##h = ra##i##l;                                     \
    uint4 rb##i##0 = make_uint4(0, 0, 0, 0), rb##i##1 = rb##i##0, rb##i##2 = rb##i##0;   \
    uint4 nb##i##0 = rb##i##0, nb##i##1 = rb##i##0, nb##i##2 = rb##i##0;
    PIPS_PASSES(PIPS_DECL)
#define PIPS_INIT(i)                                                                     \
    if constexpr (i < PA) {                                                              \
        int m_ = m0 + lrow + i * RPP;                                                    \
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
        int n_ = n0 + lrow + i * RPP;                                                    \
        n_ = n_ < p.N ? n_ : p.N - 1;                                                    \
        b_off##i = (unsigned)n_ * (unsigned)p.K + cg * 8;                                \
    }
    PIPS_PASSES(PIPS_INIT)
    (void)a_hi0; (void)a_wi0; (void)a_hi1; (void)a_wi1; (void)a_off0; (void)a_off1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define PIPS_LOAD_A(i)                                                                               \
    if constexpr (i < PA) {                                                                          \
        if constexpr (CONV) {                                                                        \
            const int hi_ = a_hi##i + kh_, wi_ = a_wi##i + kw_;                                      \
            const bool ok_ = (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.Win;       \
            const float* src_ = Af + ((size_t)(ok_ ? hi_ : 0) * p.Win + (ok_ ? wi_ : 0)) * p.Cin + c0_ + cg * 8; \
            const float4 l_ = *reinterpret_cast<const float4*>(src_);                                \
            const float4 h_ = *reinterpret_cast<const float4*>(src_ + 4);                            \
            na##i##l = ok_ ? l_ : make_float4(0.f, 0.f, 0.f, 0.f);                                   \
            na##i##h = ok_ ? h_ : make_float4(0.f, 0.f, 0.f, 0.f);                                   \
        } else {                                                                                     \
            na##i##l = *reinterpret_cast<const float4*>(Af + a_off##i + k0_);                        \
            na##i##h = *reinterpret_cast<const float4*>(Af + a_off##i + k0_ + 4);                    \
        }                                                                                            \
    }
#define PIPS_LOAD_B(i, pl)                                                                           \
    if constexpr (i < PB) nb##i##pl = *reinterpret_cast<const uint4*>(Wb + pl * wplane + b_off##i + k0_);
#define PIPS_LOAD(i) PIPS_LOAD_A(i) PIPS_LOAD_B(i, 0) PIPS_LOAD_B(i, 1) PIPS_LOAD_B(i, 2)
#define PIPS_ROTATE(i)                                                                               \
    ra##i##l = na##i##l; ra##i##h = na##i##h; rb##i##0 = nb##i##0; rb##i##1 = nb##i##1; rb##i##2 = nb##i##2;
#define PIPS_KCOORDS(kb_)                                                                            \
        const int k0_ = (PIPS_X3_ABL & 16) ? 0 : (kb_) * BKB;    /* ABL 16: always block 0 (cache-hot) */ \
        const int tap_ = CONV ? k0_ / p.Cin : 0;          /* Cin % 32 == 0: one tap per block */    \
        const int c0_ = CONV ? k0_ - tap_ * p.Cin : 0;                                               \
        const int kh_ = CONV ? tap_ / p.KW : 0, kw_ = CONV ? tap_ - kh_ * p.KW : 0;                  \
        (void)c0_; (void)kh_; (void)kw_;
#define PIPS_LOAD_TILES(kb_) { PIPS_KCOORDS(kb_) PIPS_PASSES(PIPS_LOAD) }
#define PIPS_STORE(i)                                                                                \
    if constexpr (i < PA) {                                                                          \
        uint4 h_, m_, l_;                                                                            \
        if (PIPS_X3_ABL & 1) { h_ = *reinterpret_cast<uint4*>(&ra##i##l); m_ = *reinterpret_cast<uint4*>(&ra##i##h); l_ = h_; } \
        else split3_x8(ra##i##l, ra##i##h, h_, m_, l_);                                              \
        char* dst_ = As_ + (lrow + i * RPP) * LDB + wchunk;                                         \
        *reinterpret_cast<uint4*>(dst_) = h_;                                                        \
        *reinterpret_cast<uint4*>(dst_ + PLANE_A) = m_;                                              \
        *reinterpret_cast<uint4*>(dst_ + 2 * PLANE_A) = l_;                                          \
    }                                                                                                \
    if constexpr (i < PB) {                                                                          \
        char* dst_ = Bs_ + (lrow + i * RPP) * LDB + wchunk;                                         \
        *reinterpret_cast<uint4*>(dst_) = rb##i##0;                                                  \
        *reinterpret_cast<uint4*>(dst_ + PLANE_B) = rb##i##1;                                        \
        *reinterpret_cast<uint4*>(dst_ + 2 * PLANE_B) = rb##i##2;                                    \
    }
#define PIPS_STORE_TILES(buf_) { char* As_ = smem + (buf_) * STAGE; char* Bs_ = As_ + 3 * PLANE_A; PIPS_PASSES(PIPS_STORE) }
    // one 16-wide K step: 3 A and 3 B fragments per 32x32 tile, six MFMAs per accumulator
#define PIPS_MFMA(fa, fb, pa_, pb_)                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
            acc[i][j] = CONV ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(                              \
                                   *reinterpret_cast<const bf16x8*>(&fa[pa_][i]),                    \
                                   *reinterpret_cast<const bf16x8*>(&fb[pb_][j]), acc[i][j], 0, 0, 0) \
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(                              \
                                   *reinterpret_cast<const bf16x8*>(&fb[pb_][j]),                    \
                                   *reinterpret_cast<const bf16x8*>(&fa[pa_][i]), acc[i][j], 0, 0, 0);
#define PIPS_FRAGS(fa, fb, buf_, kk_)                                                                \
    {                                                                                                \
        const char* a_frag = smem + (buf_) * STAGE + (wm * WTM + l31) * LDB + ((kk_) ? rchunk1 : rchunk0);     \
        const char* b_frag = smem + (buf_) * STAGE + 3 * PLANE_A + (wn * WTN + l31) * LDB + ((kk_) ? rchunk1 : rchunk0); \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                           \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                fa[pl][i] = *reinterpret_cast<const uint4*>(a_frag + pl * PLANE_A + i * 32 * LDB);   \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
                fb[pl][j] = *reinterpret_cast<const uint4*>(b_frag + pl * PLANE_B + j * 32 * LDB);   \
        }                                                                                            \
    }
#define PIPS_MFMA3A(fa, fb) PIPS_MFMA(fa, fb, 2, 0) PIPS_MFMA(fa, fb, 0, 2) PIPS_MFMA(fa, fb, 1, 1)
#define PIPS_MFMA3B(fa, fb) PIPS_MFMA(fa, fb, 1, 0) PIPS_MFMA(fa, fb, 0, 1) PIPS_MFMA(fa, fb, 0, 0)
#define PIPS_SB __builtin_amdgcn_sched_barrier(0);

    const int nk = p.K / BKB;
    if constexpr (BKE == 16) {
        // one K step per wave and stage; the other wave on the SIMD covers this wave's staging
        uint4 fa0[3][TM], fb0[3][TN];
        PIPS_LOAD_TILES(0);
        PIPS_PASSES(PIPS_ROTATE)
        PIPS_STORE_TILES(0);
        PIPS_LOAD_TILES(nk > 1 ? 1 : 0);
        __syncthreads();
        int buf = 0;
        for (int kb = 0; kb + 1 < nk; ++kb) {
            PIPS_PASSES(PIPS_ROTATE)
            PIPS_KCOORDS(kb + 2 < nk ? kb + 2 : nk - 1)
            PIPS_FRAGS(fa0, fb0, buf, 0);
            PIPS_SB
            PIPS_MFMA(fa0, fb0, 2, 0) PIPS_LOAD_A(0) PIPS_LOAD_B(0, 0) PIPS_SB
            PIPS_MFMA(fa0, fb0, 0, 2) PIPS_LOAD_A(1) PIPS_LOAD_B(0, 1) PIPS_SB
            PIPS_MFMA(fa0, fb0, 1, 1) PIPS_LOAD_B(0, 2) PIPS_LOAD_B(1, 0) PIPS_SB
            PIPS_MFMA(fa0, fb0, 1, 0) PIPS_LOAD_B(1, 1) PIPS_LOAD_B(1, 2) PIPS_SB
            PIPS_MFMA(fa0, fb0, 0, 1) PIPS_MFMA(fa0, fb0, 0, 0)
            PIPS_STORE_TILES(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        PIPS_FRAGS(fa0, fb0, buf, 0);
        PIPS_MFMA3A(fa0, fb0) PIPS_MFMA3B(fa0, fb0)
    } else {
    // Software pipeline of one iteration (a wave is alone on its SIMD for the 128x128 tile, so
    // every latency has to be covered by its own MFMAs; phases are fenced with sched_barrier):
    //   global loads of block kb+2 -> registers (a whole iteration ahead of their first use)
    //   LDS reads: fragments of K step 1            | hidden behind the MFMAs of K step 0
    //   MFMA step 0  +  split / ds_write of block kb+1 into the other stage (VALU + LDS in the shadow)
    //   MFMA step 1, first half ; barrier ; LDS reads: step-0 fragments of block kb+1
    //   MFMA step 1, second half                    | hides the post-barrier read latency
    uint4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
    PIPS_LOAD_TILES(0);
    PIPS_PASSES(PIPS_ROTATE)
    PIPS_STORE_TILES(0);
    PIPS_LOAD_TILES(nk > 1 ? 1 : 0);
    __syncthreads();
    int buf = 0;
    PIPS_FRAGS(fa0, fb0, 0, 0);
    for (int kb = 0; kb + 1 < nk; ++kb) {
        PIPS_PASSES(PIPS_ROTATE)                    // block kb+1, loaded one iteration ago
        PIPS_KCOORDS(kb + 2 < nk ? kb + 2 : nk - 1) // block to prefetch (the last one is a harmless reload)
        PIPS_SB
        PIPS_FRAGS(fa1, fb1, buf, 1);
        PIPS_SB
        // the ten global loads are dealt out between the MFMA groups of K step 0: issued in one
        // burst by four lock-stepped waves they queue behind the CU's single address unit and
        // hold up every MFMA behind them in program order
        PIPS_MFMA(fa0, fb0, 2, 0) if (!(PIPS_X3_ABL & 2)) { PIPS_LOAD_A(0) } PIPS_SB
        PIPS_MFMA(fa0, fb0, 0, 2) if (!(PIPS_X3_ABL & 2)) { PIPS_LOAD_A(1) PIPS_LOAD_B(0, 0) } PIPS_SB
        PIPS_MFMA(fa0, fb0, 1, 1) if (!(PIPS_X3_ABL & 2)) { PIPS_LOAD_B(0, 1) PIPS_LOAD_B(0, 2) } PIPS_SB
        PIPS_MFMA(fa0, fb0, 1, 0) if (!(PIPS_X3_ABL & 2)) { PIPS_LOAD_B(1, 0) PIPS_LOAD_B(1, 1) } PIPS_SB
        PIPS_MFMA(fa0, fb0, 0, 1) if (!(PIPS_X3_ABL & 2)) { PIPS_LOAD_B(1, 2) } PIPS_SB
        PIPS_MFMA(fa0, fb0, 0, 0)
        PIPS_MFMA3A(fa1, fb1)
        if (!(PIPS_X3_ABL & 4)) PIPS_STORE_TILES(buf ^ 1);   // split + ds_write in the shadow of 16 MFMAs
        if (!(PIPS_X3_ABL & 8)) __syncthreads();
        PIPS_FRAGS(fa0, fb0, buf ^ 1, 0);
        PIPS_SB
        PIPS_MFMA3B(fa1, fb1)
        buf ^= 1;
    }
    PIPS_FRAGS(fa1, fb1, buf, 1);
    PIPS_MFMA3A(fa0, fb0) PIPS_MFMA3B(fa0, fb0)
    PIPS_MFMA3A(fa1, fb1) PIPS_MFMA3B(fa1, fb1)
    }
#undef PIPS_PASSES
#undef PIPS_DECL
#undef PIPS_INIT
#undef PIPS_LOAD
#undef PIPS_LOAD_A
#undef PIPS_LOAD_B
#undef PIPS_KCOORDS
#undef PIPS_ROTATE
#undef PIPS_LOAD_TILES
#undef PIPS_STORE
#undef PIPS_STORE_TILES
#undef PIPS_MFMA
#undef PIPS_FRAGS
#undef PIPS_MFMA3A
#undef PIPS_MFMA3B
#undef PIPS_SB

    // The epilogue reads the accumulators (v_accvgpr_read).  hipcc pads that MFMA -> VALU read itself, but on the branchy path into the
    // epilogue of the KS = 1 forms its padding came out 3 wait states short of its own table (tools/asm_hazard_lint.py: 9 of 12, across two
    // taken branches): state the distance explicitly, once per tile.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 11");
    __builtin_amdgcn_sched_barrier(0);

    if (!ksplit_reduce<KS, WGM * WGN, TM, TN>(acc, reinterpret_cast<float*>(smem), ks, wmn, lane)) return;

    if (CONV) {
        conv_epilogue<BM, BN, WGM, WTM, WTN, NT, TM, TN>(acc, p, p.C + (size_t)frame * p.M * p.ldc,
                                                        reinterpret_cast<float*>(smem), frame, tile_.x, m0, n0,
                                                        wm, wn, l31, half, tid);
    } else {
        gemm_epilogue<TM, TN>(acc, p, m0 + BM <= p.M && n0 + BN <= p.N, m0 + wm * WTM + l31,
                              n0 + wn * WTN + 4 * half);
    }
}

template <int BM, int BN, int WGM, int WGN, int KS, bool CONV, int BKE = 32>
static int launch_x3_tile(const GemmArgs& a_in, int frames, hipStream_t st) {
    GemmArgs a = a_in;
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), frames);
    {   // XCD-aware tile order (common.h).  PIPS_X3_SWZ (tuning builds): 0 off, 1 on from 64 tiles; default -1 = by kind and size
        const int force = PIPS_TUNE("PIPS_X3_SWZ", -1);
        const long tiles = (long)grid.x * grid.y * grid.z;
        a.swz = force >= 0 ? (force != 0 && tiles >= 64) : 0;
    }
    dim3 block(WGM * WGN * KS * 64);
    const size_t lds = (size_t)2 * 3 * (BM + BN) * (BKE * KS * 2);
    auto kern = gemm_x3_kernel<BM, BN, WGM, WGN, KS, CONV, BKE>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> raised{0};      // per instantiation, one bit per device
        const int rc = ensure_dynamic_lds(raised, (const void*)kern, lds);
        if (rc != PIPS_OK) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    PIPS_CHECK_LAUNCH("gemm_x3_kernel");
    return PIPS_OK;
}

// tuning hooks: PIPS_X3_TILE=<id> for every GEMM, PIPS_X3_TILE_UP / _DOWN for N > K / N < K only
static int x3_forced_tile(const GemmArgs& a) {
    const int all = PIPS_TUNE("PIPS_X3_TILE", -1), up = PIPS_TUNE("PIPS_X3_TILE_UP", -1),
              down = PIPS_TUNE("PIPS_X3_TILE_DOWN", -1);
    (void)up; (void)down;
    if (all >= 0) return all;
    if (a.N > a.K && up >= 0) return up;
    if (a.N < a.K && down >= 0) return down;
    return -1;
}

// A fp32 [M][lda]; W: split planes [3][N][K] bf16; C fp32 [M][ldc]
int launch_gemm_x3(const GemmArgs& a, hipStream_t st) {
    PIPS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_x3: empty problem");
    PIPS_CHECK_ARG(a.K % 32 == 0 && a.lda % 4 == 0, "gemm_x3: K %% 32 and lda %% 4 required");
    PIPS_CHECK_ARG((unsigned long long)a.M * (unsigned long long)a.lda < (1ull << 32) &&
                       (unsigned long long)a.N * (unsigned long long)a.K < (1ull << 32),
                   "gemm_x3: operand exceeds 2^32 elements");
    const bool k64 = a.K % 64 == 0;
    switch (x3_forced_tile(a)) {
        case 0: return launch_x3_tile<128, 128, 2, 2, 1, false>(a, 1, st);
        case 1: return launch_x3_tile<128, 64, 2, 2, 1, false>(a, 1, st);
        case 2: return launch_x3_tile<64, 128, 2, 2, 1, false>(a, 1, st);
        case 3: return launch_x3_tile<64, 64, 2, 2, 1, false>(a, 1, st);
        case 5: if (k64) return launch_x3_tile<64, 64, 2, 2, 2, false>(a, 1, st); break;
        case 7: return launch_x3_tile<128, 128, 2, 2, 2, false, 16>(a, 1, st);
        case 8: return launch_x3_tile<256, 128, 4, 2, 1, false>(a, 1, st);
        case 9: return launch_x3_tile<128, 256, 2, 4, 1, false>(a, 1, st);
        default: break;
    }
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    // one tile per CU or more of 256x128 (8 waves of 64x64, 25 % fewer staged bytes per MFMA): measured
    // 233 -> 194 us (up-projection) and 204 -> 172 us (down-projection) at M = 16384
    if ((long)cdiv(a.M, 256) * cdiv(a.N, 128) >= 256) return launch_x3_tile<256, 128, 4, 2, 1, false>(a, 1, st);
    // 128x128; with two or more tiles per CU the two-wave-group form (2 waves per SIMD) measured 3 % ahead
    if (b128 >= 512) return launch_x3_tile<128, 128, 2, 2, 2, false, 16>(a, 1, st);
    if (b128 >= 200) return launch_x3_tile<128, 128, 2, 2, 1, false>(a, 1, st);
    if (k64) return launch_x3_tile<64, 64, 2, 2, 2, false>(a, 1, st);
    return launch_x3_tile<64, 64, 2, 2, 1, false>(a, 1, st);
}

// NHWC fp32 map, weights split planes [3][Cout][kh][kw][Cin]; raw fp32 output + instance-norm partials
int launch_conv_x3(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st) {
    PIPS_CHECK_ARG(a.Cin % 32 == 0 && a.K == a.KH * a.KW * a.Cin, "conv_x3: Cin %% 32, K = kh*kw*Cin");
    const int bn = a.N <= 64 ? 64 : 128;            // Cout = 96 rides a 128-wide tile
    const int force_bm = PIPS_TUNE("PIPS_X3_CONV_BM", 0);       // tuning hook: 64|128|256
    int bm = (long)cdiv(a.M, 128) * cdiv(a.N, bn) * frames >= 256 ? 128 : 64;
    // 256-row tiles (8 waves of 64x64, a quarter fewer staged bytes per MFMA) once they fill 5/8 of the
    // CUs: measured 96->96 164 -> 150 us, 64->96/s2 118 -> 103 us, 416->256 (192 tiles) 435 -> 304 us;
    // the 96-tile layers (46x62, Cout 128) lose with them (69 -> 91 us)
    if (bn == 128 && (long)cdiv(a.M, 256) * cdiv(a.N, 128) * frames >= 160) bm = 256;
    if (force_bm == 64 || force_bm == 128 || (force_bm == 256 && bn == 128)) bm = force_bm;
    if (tiles_m) *tiles_m = cdiv(a.M, bm) * (bm == 256 ? 4 : 2);     // partials per frame: m tiles x wave rows
    if (bm == 256) return launch_x3_tile<256, 128, 4, 2, 1, true>(a, frames, st);
    if (bm == 128) {
        if (bn == 128) return launch_x3_tile<128, 128, 2, 2, 1, true>(a, frames, st);
        return launch_x3_tile<128, 64, 2, 2, 1, true>(a, frames, st);
    }
    if (bn == 128) return launch_x3_tile<64, 128, 2, 2, 1, true>(a, frames, st);
    return launch_x3_tile<64, 64, 2, 2, 1, true>(a, frames, st);
}

}  // namespace pips
