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
// LDS image: As[BM][36], Bs[BN][36] floats (32 K-values + 4 pad: the 144-byte row stride
// makes the ds_read_b128 fragment reads conflict-free for every 16-lane service group).
// Within a 32-wide K block the two lane halves take interleaved groups of four K values
// (half h reads k = 8*kk + 4*h + j), the same permutation on A and B, so one
// ds_read_b128 per operand feeds four MFMAs.
#include "common.h"

namespace pips {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf(float x) {
    // nn.GELU() default = exact erf form (nets/pips.py:105)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

constexpr int BK = 32;
constexpr int LDS_LD = 36;

template <int BM, int BN, int WGM, int WGN, bool CONV>
__global__ __launch_bounds__(WGM * WGN * 64) void igemm_f32_kernel(GemmArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWS_PER_PASS = NT / 8;
    constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/loader mismatch");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be 32-granular");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int frame = blockIdx.z;

    const float* __restrict__ Abase = p.A;
    float* __restrict__ Cbase = p.C;
    if (CONV) {
        Abase += (size_t)frame * p.H * p.Win * p.Cin;
        Cbase += (size_t)frame * p.M * p.ldc;
    }

    // loader coordinates: thread -> (row = tid/8 + pass*ROWS_PER_PASS, 4 floats at cg*4)
    const int lrow = tid >> 3, cg = tid & 7;
    int a_hi0[PA], a_wi0[PA];
    bool a_ok[PA];
    const float* a_ptr[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + i * ROWS_PER_PASS;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int ho = m / p.Wo, wo = m - ho * p.Wo;
            a_hi0[i] = ho * p.cstride - p.pad;
            a_wi0[i] = wo * p.cstride - p.pad;
            a_ptr[i] = nullptr;
        } else {
            a_hi0[i] = a_wi0[i] = 0;
            a_ptr[i] = Abase + (size_t)(a_ok[i] ? m : 0) * p.lda + cg * 4;
        }
    }
    const float* b_ptr[PB];
    bool b_ok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + i * ROWS_PER_PASS;
        b_ok[i] = n < p.N;
        b_ptr[i] = p.W + (size_t)(b_ok[i] ? n : 0) * p.K + cg * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[PA], rb[PB];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tiles = [&](int kb) {
        const int k0 = kb * BK;
        if (CONV) {
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                const bool ok = a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.Win;
                ra[i] = ok ? *reinterpret_cast<const float4*>(
                                 Abase + ((size_t)hi * p.Win + wi) * p.Cin + c0 + cg * 4)
                           : zero4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                ra[i] = a_ok[i] ? *reinterpret_cast<const float4*>(a_ptr[i] + k0) : zero4;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i)
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(b_ptr[i] + k0) : zero4;
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<float4*>(&As[(lrow + i * ROWS_PER_PASS) * LDS_LD + cg * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i)
            *reinterpret_cast<float4*>(&Bs[(lrow + i * ROWS_PER_PASS) * LDS_LD + cg * 4]) = rb[i];
    };

    const int nk = p.K / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();

    const float* a_frag = &As[(wm * WTM + l31) * LDS_LD + half * 4];
    const float* b_frag = &Bs[(wn * WTN + l31) * LDS_LD + half * 4];

    for (int kb = 0; kb < nk; ++kb) {
        if (kb + 1 < nk) load_tiles(kb + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float4*>(a_frag + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const float4*>(b_frag + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kb + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = csq[j] = 0.f;

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        const bool col_ok = col < p.N;
        const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < p.M && col_ok) {
                    float v = acc[i][j][r] + bv;
                    if (p.epi == EPI_GELU) v = gelu_erf(v);
                    else if (p.epi == EPI_RESIDUAL) v += p.R[(size_t)row * p.ldr + col];
                    Cbase[(size_t)row * p.ldc + col] = v;
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
        }
    }

    if (CONV && p.stats != nullptr) {
        // per-column partial sums of this m-tile: lanes l and l+32 hold the same column
        __syncthreads();                      // As is free now
        float* red = As;                      // [WGM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32);
            float q = csq[j] + __shfl_xor(csq[j], 32);
            if (half == 0) {
                const int c = wn * WTN + j * 32 + l31;
                red[(wm * BN + c) * 2 + 0] = s;
                red[(wm * BN + c) * 2 + 1] = q;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                s += red[(w * BN + c) * 2 + 0];
                q += red[(w * BN + c) * 2 + 1];
            }
            const int col = n0 + c;
            if (col < p.N) {
                float* dst = p.stats + (((size_t)frame * gridDim.x + blockIdx.x) * p.N + col) * 2;
                dst[0] = s;
                dst[1] = q;
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, bool CONV>
static int launch_tile(const GemmArgs& a, int frames, hipStream_t st) {
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), frames);
    dim3 block(WGM * WGN * 64);
    size_t lds = (size_t)(BM + BN) * LDS_LD * sizeof(float);
    hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, CONV>), grid, block, lds, st, a);
    PIPS_CHECK_LAUNCH("igemm_f32_kernel");
    return PIPS_OK;
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
    PIPS_CHECK_ARG(a.K % BK == 0 && a.K > 0, "gemm: K=%d must be a positive multiple of 32", a.K);
    PIPS_CHECK_ARG(a.M > 0 && a.N > 0, "gemm: empty problem");
    PIPS_CHECK_ARG((a.lda % 4) == 0, "gemm: lda must be a multiple of 4 floats");
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b64x128 = (long)cdiv(a.M, 64) * cdiv(a.N, 128);
    if (b128 >= 200) return launch_tile<128, 128, 2, 2, false>(a, 1, st);
    if (b64x128 >= 200) return launch_tile<64, 128, 2, 2, false>(a, 1, st);
    return launch_tile<64, 64, 2, 2, false>(a, 1, st);
}

// tile choice of launch_conv, shared with the stats consumer
static void conv_tile(int rows, int cout, int frames, int* bm, int* bn) {
    int n = (cout % 128 == 0) ? 128 : (cout % 96 == 0 ? 96 : 64);
    long blocks128 = (long)cdiv(rows, 128) * (cout / n) * frames;
    *bn = n;
    *bm = blocks128 >= 384 ? 128 : 64;
}

int conv_tiles_m(int rows_per_frame, int Cout, int frames) {
    int bm, bn;
    conv_tile(rows_per_frame, Cout, frames, &bm, &bn);
    return cdiv(rows_per_frame, bm);
}

int launch_conv(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st) {
    PIPS_CHECK_ARG(a.Cin % BK == 0, "conv: Cin=%d must be a multiple of 32", a.Cin);
    PIPS_CHECK_ARG(a.N % 32 == 0 && (a.N % 64 == 0 || a.N % 96 == 0), "conv: unsupported Cout=%d", a.N);
    PIPS_CHECK_ARG(a.K == a.KH * a.KW * a.Cin, "conv: K mismatch");
    int bm, bn;
    conv_tile(a.M, a.N, frames, &bm, &bn);
    if (tiles_m) *tiles_m = cdiv(a.M, bm);
    if (bn == 128) {
        return bm == 128 ? launch_tile<128, 128, 2, 2, true>(a, frames, st)
                         : launch_tile<64, 128, 2, 2, true>(a, frames, st);
    } else if (bn == 96) {
        return bm == 128 ? launch_tile<128, 96, 4, 1, true>(a, frames, st)
                         : launch_tile<64, 96, 2, 1, true>(a, frames, st);
    }
    return bm == 128 ? launch_tile<128, 64, 2, 2, true>(a, frames, st)
                     : launch_tile<64, 64, 2, 2, true>(a, frames, st);
}

}  // namespace pips
