// Encoder-side kernels that are not GEMM-shaped: the 7x7 stem on the NCHW input, the
// instance-norm statistics/apply passes, the align_corners bilinear resize into the
// concatenated map and the 2x2 average pool of the correlation pyramid.
// Reference: BasicEncoder.forward nets/pips.py:247-281, ResidualBlock.forward :173-181,
// CorrBlock.__init__ :346-352.  Activations are NHWC fp32.
#include "common.h"

namespace pips {

// ------------------------------------------------------------------------------ stem
// conv1 7x7 stride 2 pad 3, 3 -> 64 (nets/pips.py:206,251) reading the caller's NCHW 0..255 frames with the
// 2*(x/255)-1 scaling of :436 applied on load (zero padding is in the scaled domain), on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact products, fp32 accumulation).  K = (ci, kh, kw) with kw padded 7 -> 8 (zero
// weights): an MFMA step takes the tap pair (kw = 2j, 2j+1), so a lane's two K values are NEIGHBOURING input
// pixels and every LDS address of the loop is lane base + immediate.  A block (4 waves) = 4 output rows x 64
// output columns x 64 channels; wave w owns row w: two 32-pixel M tiles x two 32-channel N tiles.  LDS: the scaled
// input tile [3][13][136] fp32 (stride-2 reads: even banks for the first tap of a pair, odd for the second) and the
// weights [84 steps][64 ch][2 taps] (pair-interleaved: conflict-free for both lane halves), 64 KiB: two persistent
// blocks per CU, each stages the weights once and walks tiles.  Also emits per-(frame, tile, wave, channel)
// statistics partials about a pivot (see the epilogue).  (The VALU form this replaces ran at 31 TFLOP/s.)
constexpr int STEM_ROWS = 4, STEM_COLS = 64;
constexpr int STEM_TH = 2 * STEM_ROWS + 5, STEM_TW = 136;          // input tile: 13 x (2*64 + 5 -> 136) per channel
constexpr int STEM_STEPS = 3 * 7 * 4;                              // (ci, kh, kw pair)
constexpr int STEM_LDS = (3 * STEM_TH * STEM_TW + STEM_STEPS * 64 * 2) * 4;

// RGB = float (0..255 values, what the reference's callers pass after .float()) or unsigned char
// (the decoded frames themselves: a quarter of the bytes, bit-identical results).
// V4 (round 6; image width a multiple of 4, frames 16-byte aligned): the padded filter's zero tap sits in FRONT (kw' = kw + 1), which moves the
// tile's first column from 2 col0 - 3 to 2 col0 - 4 -- a multiple of four pixels -- and the tile is staged with SIX aligned 16-byte loads per
// thread, all in flight before the first is used: one memory round trip per tile.  The row-by-row form below issues a (channel, row) per wave
// and step, ten dependent round trips per tile that only the compute unit's other block hides (128 us per launch at the headline's 8 frames
// where the MFMAs need 50).  [measured] profiles/r6_probe_stem_f32_v4.txt
template <typename RGB, bool V4>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const RGB* __restrict__ rgbs,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ out,
                                                           float* __restrict__ stats, int H, int W,
                                                           int Ho, int Wo, int tiles_x, int tiles, int F) {
    extern __shared__ __attribute__((aligned(16))) float stem_sm[];
    float* tile = stem_sm;                                  // [3][STEM_TH][STEM_TW]
    float* wl = stem_sm + 3 * STEM_TH * STEM_TW;            // [STEM_STEPS][64][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // ---- the weights once per (persistent) block
    for (int i = tid; i < STEM_STEPS * 64 * 2; i += 256) {
        const int h = i & 1, n = (i >> 1) & 63, s = i >> 7;          // step s = (ci*7 + kh)*4 + j
        const int j = s & 3, ck = s >> 2, kw = 2 * j + h - (V4 ? 1 : 0);   // (V4: K value 0 of the eight is the zero tap)
        wl[i] = (kw >= 0 && kw < 7) ? w[(ck * 7 + kw) * 64 + n] : 0.f;
    }
  for (int work = blockIdx.x; work < tiles * F; work += gridDim.x) {
    const int frame = work / tiles, tile_id = work - frame * tiles;
    const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
    const int row0 = ty * STEM_ROWS, col0 = tx * STEM_COLS;
    // ---- stage the scaled input tile
    const RGB* src = rgbs + (size_t)frame * 3 * H * W;
    const int hi0 = 2 * row0 - 3, wi0 = 2 * col0 - (V4 ? 4 : 3);
    if constexpr (V4) {
        constexpr int NQ = 3 * STEM_TH * (STEM_TW / 4), NIT4 = (NQ + 255) / 256;
        float4 v[NIT4];
        unsigned in_mask = 0;
#pragma unroll
        for (int it = 0; it < NIT4; ++it) {
            const int q = tid + it * 256, qc = q < NQ ? q : 0;               // (the last step is partial)
            const int r = qc / (STEM_TW / 4), j = qc - r * (STEM_TW / 4);
            const int c = r / STEM_TH, y = r - c * STEM_TH;
            const int hi = hi0 + y, wi = wi0 + 4 * j;
            const bool in = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;     // W % 4 == 0: a quad is inside or outside as a whole
            const RGB* q4 = src + ((size_t)c * H + (in ? hi : 0)) * W + (in ? wi : 0);
            if constexpr (sizeof(RGB) == 1) {
                const uchar4 u = *reinterpret_cast<const uchar4*>(q4);
                v[it] = make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
            } else {
                v[it] = *reinterpret_cast<const float4*>(q4);
            }
            in_mask |= (in ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < NIT4; ++it) {
            const int q = tid + it * 256;
            const bool in = in_mask >> it & 1;
            const float4 o = in ? make_float4(2.0f * (v[it].x / 255.0f) - 1.0f, 2.0f * (v[it].y / 255.0f) - 1.0f, 2.0f * (v[it].z / 255.0f) - 1.0f,
                                              2.0f * (v[it].w / 255.0f) - 1.0f)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < NQ) reinterpret_cast<float4*>(tile)[q] = o;
        }
    } else
    for (int r = wave; r < 3 * STEM_TH; r += 4) {                     // one (channel, row) of the tile per wave and step
        const int c = r / STEM_TH, y = r - c * STEM_TH;
        const int hi = hi0 + y;
        const bool hok = (unsigned)hi < (unsigned)H;
        const RGB* row = src + ((size_t)c * H + (hok ? hi : 0)) * W;
#pragma unroll
        for (int x = lane; x < STEM_TW; x += 64) {
            const int wi = wi0 + x;
            const bool in = hok && (unsigned)wi < (unsigned)W;
            const float v = (float)row[in ? wi : 0];
            tile[r * STEM_TW + x] = in ? 2.0f * (v / 255.0f) - 1.0f : 0.f;
        }
    }
    __syncthreads();
    // ---- 84 steps x (2 x 2) MFMAs per wave
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* a_base = tile + (2 * wave) * STEM_TW + 2 * l31 + half;        // + mt*64 + ((ci*TH + kh)*TW + 2j)
    const float* b_base = wl + l31 * 2 + half;                                  // + s*128 + nt*64
#pragma unroll 1
    for (int ck = 0; ck < 21; ++ck) {
        const int ci = ck / 7, kh = ck - ci * 7;
        const float* ap = a_base + (ci * STEM_TH + kh) * STEM_TW;
        const float* bp = b_base + ck * 4 * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = ap[2 * j], a1 = ap[2 * j + 64];
            const float b0 = bp[j * 128], b1 = bp[j * 128 + 64];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // ---- epilogue.  C/D map of the 32x32 MFMA: col (channel) = lane&31, row (pixel) = (r&3) + 8*(r>>2) + 4*(lane>>5):
    //      a lane holds ONE channel of 16 pixels per tile, so the column sums of the statistics stay in-lane and a
    //      store instruction writes two 128-byte runs (32 channels of two pixels)
    const int orow = row0 + wave;
    const bool row_ok = orow < Ho;
    // Statistics about a PIVOT (the wave's first pixel, per channel): on flat frames a channel is the same number at
    // every pixel and E[x^2] - mean^2 of the raw values would be all cancellation; sums of (x - pivot) are exact zeros
    // there.  Partial = {sum(x-p), sum((x-p)^2), p, count}, combined Chan-style in inorm_finalize_pivot_kernel.
    const int nvalid = row_ok ? min(STEM_COLS, Wo - col0) : 0;            // the wave's valid pixels (col0 < Wo always)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 32 + l31;
        const float bv = bias[n];
        const float pivot = __shfl(acc[0][nt][0] + bv, l31);              // pixel 0 of the row segment: lanes of half 0, r = 0
        float cs = 0.f, cq = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = col0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = acc[mt][nt][r] + bv;
                if (row_ok && col < Wo) {
                    out[(((size_t)frame * Ho + orow) * Wo + col) * 64 + n] = v;
                    const float d = v - pivot;
                    cs += d;
                    cq += d * d;
                }
            }
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (half == 0)
            reinterpret_cast<float4*>(stats)[(((size_t)frame * tiles + tile_id) * 4 + wave) * 64 + n] =
                make_float4(cs, cq, pivot, (float)nvalid);
    }
    __syncthreads();                                      // before the next tile is staged over this one
  }
}

int stem_tiles_m(int Ho, int Wo) { return cdiv(Ho, STEM_ROWS) * cdiv(Wo, STEM_COLS) * 4; }   // partials per frame

int launch_stem(const void* rgbs, int rgb_u8, const float* w, const float* bias, float* out, float* stats,
                int F, int H, int W, int Ho, int Wo, int* tiles_m, hipStream_t st) {
    const int tiles_x = cdiv(Wo, STEM_COLS), tiles = cdiv(Ho, STEM_ROWS) * tiles_x;
    if (tiles_m) *tiles_m = stem_tiles_m(Ho, Wo);
    // persistent blocks (two per CU: 64 KiB of LDS each), each stages the weights once and walks tiles
    const int grid = tiles * F < 512 ? tiles * F : 512;
    static std::atomic<unsigned long long> raised_f{0}, raised_u{0};
    static std::atomic<unsigned long long> raised_f4{0}, raised_u4{0};
    // quads of four pixels: rows and frames start on a quad boundary when W % 4 == 0 and the first frame does
    const bool v4 = W % 4 == 0 && reinterpret_cast<uintptr_t>(rgbs) % (rgb_u8 ? 4 : 16) == 0 && PIPS_TUNE("PIPS_STEM_V4", 1);
#define PIPS_STEM32(T_, V_, RAISED_)                                                                                             \
    {                                                                                                                            \
        const int rc = ensure_dynamic_lds(RAISED_, (const void*)stem_conv_kernel<T_, V_>, STEM_LDS);                            \
        if (rc != PIPS_OK) return rc;                                                                                            \
        hipLaunchKernelGGL((stem_conv_kernel<T_, V_>), dim3(grid), dim3(256), STEM_LDS, st, (const T_*)rgbs, w, bias, out, stats, \
                           H, W, Ho, Wo, tiles_x, tiles, F);                                                                     \
    }
    if (rgb_u8) { if (v4) PIPS_STEM32(unsigned char, true, raised_u4) else PIPS_STEM32(unsigned char, false, raised_u) }
    else { if (v4) PIPS_STEM32(float, true, raised_f4) else PIPS_STEM32(float, false, raised_f) }
#undef PIPS_STEM32
    PIPS_CHECK_LAUNCH("stem_conv_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------ instance-norm stats
// InstanceNorm2d(affine=False, eps=1e-5), biased variance (nets/pips.py:153-157,199-201).
// Pivoted partials {sum(x-p), sum((x-p)^2), p, n} [F][parts][C] (the stem's, and store_conv_partial's of every conv
// kernel) -> mean_rstd [F][C][2].  All partials of a channel are re-based in fp64 on ONE reference r (the first partial's
// pivot, a value of the data): with d = p - r,  sum(x-r) = sum_t [s_t + n_t d_t],  sum(x-r)^2 = sum_t [q_t + 2 d_t s_t +
// n_t d_t^2];  mean = r + S1/N,  var = S2/N - (S1/N)^2 -- one pass, and what cancellation is left happens in fp64 about
// a value within the data's range.
// Block = (frame, 4 channels) x SUBS partial subsets: F x C/4 blocks instead of F x C/16, eight loads per thread in flight at
// once -- the kernel is one memory round trip plus the launch (round 3; the (frame, 16 channels) x 64 subsets form took 6.2 us
// per launch at 8 frames, 21 launches per forward: 131 -> 107 us per forward).  Taken up to 16 frames.
template <int SUBS>
__global__ __launch_bounds__(4 * SUBS) void inorm_finalize_pivot_kernel(const float4* __restrict__ partial, int parts, int C,
                                                                        float* __restrict__ mean_rstd) {
    constexpr int NW = 4 * SUBS / 64;
    __shared__ double red[3][NW][4];                   // per wave and channel
    const int f = blockIdx.x, cl = threadIdx.x & 3, c = blockIdx.y * 4 + cl, sub = threadIdx.x >> 2;
    const float4* p = partial + (size_t)f * parts * C + c;
    const double r = (double)p[0].z;
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t0 = sub; t0 < parts; t0 += 8 * SUBS) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + SUBS * u;
            v[u] = t < parts ? p[(size_t)t * C] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u].w > 0.f) {
                const double nt = v[u].w, d = (double)v[u].z - r;
                n += nt;
                s1 += (double)v[u].x + nt * d;
                s2 += (double)v[u].y + d * (2.0 * (double)v[u].x + nt * d);
            }
    }
    // lanes of a wave: 16 subsets x 4 channels; fold the subsets (lane bits 2..5), then the waves through LDS
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        n += __shfl_xor(n, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 4) { red[0][wave][cl] = n; red[1][wave][cl] = s1; red[2][wave][cl] = s2; }
    __syncthreads();
    if (threadIdx.x < 4) {
        n = 0.0; s1 = 0.0; s2 = 0.0;
        for (int i = 0; i < NW; ++i) { n += red[0][i][cl]; s1 += red[1][i][cl]; s2 += red[2][i][cl]; }
        const double m1 = n > 0.0 ? s1 / n : 0.0;
        double var = n > 0.0 ? s2 / n - m1 * m1 : 0.0;
        if (var < 0.0) var = 0.0;
        mean_rstd[((size_t)f * C + c) * 2 + 0] = (float)(r + m1);
        mean_rstd[((size_t)f * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// Many frames (BASELINE configs[2]: 64): block = (frame, 16 channels) x 64 partial subsets, four loads in flight per thread -- with
// 64 frames there are enough blocks, and the small-block form above costs more in block launches than it saves (222 against 198 us
// over the 21 layers of a forward).
__global__ __launch_bounds__(1024) void inorm_finalize_pivot16_kernel(const float4* __restrict__ partial, int parts, int C,
                                                                    float* __restrict__ mean_rstd) {
    __shared__ double red[3][64][16];
    const int f = blockIdx.x, cl = threadIdx.x & 15, c = blockIdx.y * 16 + cl, sub = threadIdx.x >> 4;
    const float4* p = partial + (size_t)f * parts * C + c;
    const double r = (double)p[0].z;
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t0 = sub; t0 < parts; t0 += 256) {        // four loads in flight per thread (the loop is latency-bound)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 64 * u;
            v[u] = t < parts ? p[(size_t)t * C] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u].w > 0.f) {
                const double nt = v[u].w, d = (double)v[u].z - r;
                n += nt;
                s1 += (double)v[u].x + nt * d;
                s2 += (double)v[u].y + d * (2.0 * (double)v[u].x + nt * d);
            }
    }
    red[0][sub][cl] = n; red[1][sub][cl] = s1; red[2][sub][cl] = s2;
    __syncthreads();
    if (sub == 0) {
        n = 0.0; s1 = 0.0; s2 = 0.0;
        for (int i = 0; i < 64; ++i) { n += red[0][i][cl]; s1 += red[1][i][cl]; s2 += red[2][i][cl]; }
        const double m1 = n > 0.0 ? s1 / n : 0.0;
        double var = n > 0.0 ? s2 / n - m1 * m1 : 0.0;
        if (var < 0.0) var = 0.0;
        mean_rstd[((size_t)f * C + c) * 2 + 0] = (float)(r + m1);
        mean_rstd[((size_t)f * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

int launch_inorm_finalize_pivot(const float* partial, int F, int parts, int C, float* mean_rstd, hipStream_t st) {
    PIPS_CHECK_ARG(C % 16 == 0, "inorm_finalize: C=%d must be a multiple of 16", C);
    if (F <= 16)
        hipLaunchKernelGGL(inorm_finalize_pivot_kernel<256>, dim3(F, C / 4), dim3(1024), 0, st, reinterpret_cast<const float4*>(partial),
                           parts, C, mean_rstd);
    else
        hipLaunchKernelGGL(inorm_finalize_pivot16_kernel, dim3(F, C / 16), dim3(1024), 0, st, reinterpret_cast<const float4*>(partial),
                           parts, C, mean_rstd);
    PIPS_CHECK_LAUNCH("inorm_finalize_pivot_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------ instance-norm apply
// MODE 0: y = relu((x-m)*r)                              nets/pips.py:175/176/252-253/274-275
// MODE 1: y = relu(res + relu((x-m)*r))                  :176,181 (identity shortcut)
// MODE 2: y = relu((res-m2)*r2 + relu((x-m)*r))          :169-170,179,181 (1x1 s2 shortcut + norm3)
template <int MODE>
__global__ __launch_bounds__(256) void inorm_apply_kernel(const float4* __restrict__ x,
                                                          const float2* __restrict__ stats,
                                                          const float4* __restrict__ res,
                                                          const float2* __restrict__ res_stats,
                                                          float4* __restrict__ y, int HW, int C4,
                                                          size_t total4) {
    const size_t per_frame = (size_t)HW * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / per_frame);
        const int c4 = (int)(i % C4);
        const float2* st = stats + ((size_t)f * C4 + c4) * 4;
        float4 v = x[i];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaxf((o[k] - st[k].x) * st[k].y, 0.f);
        if (MODE == 1) {
            const float4 r = res[i];
            o[0] = fmaxf(r.x + o[0], 0.f); o[1] = fmaxf(r.y + o[1], 0.f);
            o[2] = fmaxf(r.z + o[2], 0.f); o[3] = fmaxf(r.w + o[3], 0.f);
        } else if (MODE == 2) {
            const float4 r = res[i];
            const float2* rs = res_stats + ((size_t)f * C4 + c4) * 4;
            const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaxf((rr[k] - rs[k].x) * rs[k].y + o[k], 0.f);
        }
        y[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

int launch_inorm_apply(const float* x, const float* stats, const float* res, const float* res_stats,
                       float* y, int F, int HW, int C, hipStream_t st) {
    PIPS_CHECK_ARG(C % 4 == 0, "inorm_apply: C %% 4");
    const size_t total4 = (size_t)F * HW * (C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float2* s2 = reinterpret_cast<const float2*>(stats);
    const float4* r4 = reinterpret_cast<const float4*>(res);
    const float2* rs2 = reinterpret_cast<const float2*>(res_stats);
    float4* y4 = reinterpret_cast<float4*>(y);
    if (res == nullptr)
        hipLaunchKernelGGL(inorm_apply_kernel<0>, dim3(blocks), dim3(256), 0, st, x4, s2, r4, rs2, y4, HW, C / 4, total4);
    else if (res_stats == nullptr)
        hipLaunchKernelGGL(inorm_apply_kernel<1>, dim3(blocks), dim3(256), 0, st, x4, s2, r4, rs2, y4, HW, C / 4, total4);
    else
        hipLaunchKernelGGL(inorm_apply_kernel<2>, dim3(blocks), dim3(256), 0, st, x4, s2, r4, rs2, y4, HW, C / 4, total4);
    PIPS_CHECK_LAUNCH("inorm_apply_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------------ resize
// F.interpolate(mode='bilinear', align_corners=True) (nets/pips.py:269-272) of an NHWC map
// into channels [coff, coff+C) of the concatenated NHWC map (the torch.cat of :273).
__global__ __launch_bounds__(256) void resize_into_kernel(const float* __restrict__ src, int Hs, int Ws,
                                                          int C4, float4* __restrict__ dst, int Hd,
                                                          int Wd, int Cdst4, int coff4, float sh,
                                                          float sw, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int x = (int)(p % Wd); p /= Wd;
        const int y = (int)(p % Hd);
        const int f = (int)(p / Hd);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        if (Hd == Hs) { y0 = y1 = y; ly0 = 1.f; ly1 = 0.f; }
        else {
            const float r = sh * (float)y;
            y0 = min((int)floorf(r), Hs - 1);
            ly1 = fminf(fmaxf(r - (float)y0, 0.f), 1.f);
            y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
            ly0 = 1.f - ly1;
        }
        if (Wd == Ws) { x0 = x1 = x; lx0 = 1.f; lx1 = 0.f; }
        else {
            const float r = sw * (float)x;
            x0 = min((int)floorf(r), Ws - 1);
            lx1 = fminf(fmaxf(r - (float)x0, 0.f), 1.f);
            x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
            lx0 = 1.f - lx1;
        }
        const float4* s4 = reinterpret_cast<const float4*>(src) + (size_t)f * Hs * Ws * C4 + c4;
        const float4 v00 = s4[((size_t)y0 * Ws + x0) * C4], v01 = s4[((size_t)y0 * Ws + x1) * C4];
        const float4 v10 = s4[((size_t)y1 * Ws + x0) * C4], v11 = s4[((size_t)y1 * Ws + x1) * C4];
        float4 o;
        o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        dst[(((size_t)f * Hd + y) * Wd + x) * Cdst4 + coff4 + c4] = o;
    }
}

int launch_resize_into(const float* src, int F, int Hs, int Ws, int C, float* dst, int Hd, int Wd,
                       int Cdst, int coff, hipStream_t st) {
    PIPS_CHECK_ARG(C % 4 == 0 && Cdst % 4 == 0 && coff % 4 == 0, "resize: channel alignment");
    const size_t total4 = (size_t)F * Hd * Wd * (C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    // area_pixel_compute_scale(align_corners=True): (in-1)/(out-1), 0 when out == 1
    const float sh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
    const float sw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
    hipLaunchKernelGGL(resize_into_kernel, dim3(blocks), dim3(256), 0, st, src, Hs, Ws, C / 4,
                       reinterpret_cast<float4*>(dst), Hd, Wd, Cdst / 4, coff / 4, sh, sw, total4);
    PIPS_CHECK_LAUNCH("resize_into_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------------ avg pool
// F.avg_pool2d(x, 2, stride=2) (nets/pips.py:349), floor output size, NHWC.
__global__ __launch_bounds__(256) void avgpool2_kernel(const float4* __restrict__ src, int H, int W,
                                                       int C4, float4* __restrict__ dst, int Ho, int Wo,
                                                       size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho);
        const int f = (int)(p / Ho);
        const float4* s = src + (((size_t)f * H + 2 * y) * W + 2 * x) * C4 + c4;
        const float4 a = s[0], b = s[C4], c = s[(size_t)W * C4], d = s[(size_t)W * C4 + C4];
        float4 o;
        o.x = (((a.x + b.x) + c.x) + d.x) * 0.25f;
        o.y = (((a.y + b.y) + c.y) + d.y) * 0.25f;
        o.z = (((a.z + b.z) + c.z) + d.z) * 0.25f;
        o.w = (((a.w + b.w) + c.w) + d.w) * 0.25f;
        dst[i] = o;
    }
}

int launch_avgpool2(const float* src, int F, int H, int W, int C, float* dst, hipStream_t st) {
    const int Ho = H / 2, Wo = W / 2;
    PIPS_CHECK_ARG(Ho >= 1 && Wo >= 1, "avgpool: map %dx%d too small", H, W);
    const size_t total4 = (size_t)F * Ho * Wo * (C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(src), H, W, C / 4,
                       reinterpret_cast<float4*>(dst), Ho, Wo, total4);
    PIPS_CHECK_LAUNCH("avgpool2_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------ input resize
// The callers' pre-processing (demo.py:22-28, chain_demo.py:26-28): decoded uint8 frames (F,3,h,w) ->
// F.interpolate(..., (H,W), mode='bilinear') (align_corners=False, no antialias) -> float 0..255 in the
// (F,3,H,W) layout the stem reads.  Source index = scale*(dst+0.5)-0.5 clamped at 0, second tap clamped at the
// border, weights and blend in fp32 in ATen's order (upsample_bilinear2d: h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)).
template <typename T>
__global__ __launch_bounds__(256) void resize_frames_kernel(const T* __restrict__ src, int planes, int h, int w,
                                                            float* __restrict__ dst, int H, int W, float sh, float sw) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)planes * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t pl = i / ((size_t)W * H);
    const float fy = fmaxf(__fsub_rn(__fmul_rn(sh, (float)y + 0.5f), 0.5f), 0.f);
    const float fx = fmaxf(__fsub_rn(__fmul_rn(sw, (float)x + 0.5f), 0.5f), 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const T* p = src + pl * (size_t)h * w;
    const float p00 = (float)p[(size_t)y0 * w + x0], p01 = (float)p[(size_t)y0 * w + x1];
    const float p10 = (float)p[(size_t)y1 * w + x0], p11 = (float)p[(size_t)y1 * w + x1];
    const float top = __fadd_rn(__fmul_rn(lx0, p00), __fmul_rn(lx1, p01));
    const float bot = __fadd_rn(__fmul_rn(lx0, p10), __fmul_rn(lx1, p11));
    dst[i] = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
}

int launch_resize_frames(const void* src, int src_u8, int planes, int h, int w, float* dst, int H, int W, hipStream_t st) {
    const size_t total = (size_t)planes * H * W;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (src_u8)
        hipLaunchKernelGGL(resize_frames_kernel<unsigned char>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)src,
                           planes, h, w, dst, H, W, sh, sw);
    else
        hipLaunchKernelGGL(resize_frames_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)src, planes, h, w, dst,
                           H, W, sh, sw);
    PIPS_CHECK_LAUNCH("resize_frames_kernel");
    return PIPS_OK;
}

}  // namespace pips
