// bf16-activation forms of the encoder's non-GEMM kernels (PIPS_FLAG_BF16_ENCODER, BASELINE configs[2]).
//
// Under torch.autocast(bfloat16) -- the way the reference itself would be run in bf16 -- every convolution of
// BasicEncoder (nets/pips.py:247-281) takes bf16 operands and RETURNS bf16, InstanceNorm / ReLU / the residual adds
// / the bilinear resizes produce bf16 from fp32 arithmetic.  The bf16 encoder mode keeps the same rounding points:
// every activation map lives in HBM as bf16 (half the bytes of the passes that bound the encoder at B = 8 per GPU),
// statistics come from the fp32 accumulators, normalisation / ReLU / adds / interpolation are fp32 arithmetic rounded
// once (hardware round-to-nearest-even) when the result is stored.
#include "common.h"

namespace pips {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------ stem
// conv1 7x7 stride 2 pad 3, 3 -> 64 (nets/pips.py:206,251) on v_mfma_f32_32x32x16_bf16, reading the caller's NCHW 0..255
// frames with the 2*(x/255)-1 of :436 applied on load and rounded to bf16 (what autocast's cast of the conv input does).
// K = (ci, kh) pair x kw padded 7 -> 8: a lane's 8 K values are 8 NEIGHBOURING input pixels of one row, so an A fragment
// is 16 contiguous bytes of the bf16 input tile; an MFMA takes two (ci, kh) pairs (one per lane half): 21 pairs -> 11
// steps (the 22nd pair and the 8th tap carry zero weights).  Block (4 waves) = 4 output rows x 64 output columns x 64
// channels, wave w = row w; LDS: input tile [3][13][136] bf16 + weights [11][2][64][8] bf16 = 33 KB; 168 registers: three blocks per CU.
constexpr int SB_ROWS = 4, SB_COLS = 64;
constexpr int SB_TH = 2 * SB_ROWS + 5, SB_TW = 136;
constexpr int SB_STEPS = 11;
constexpr int SB_TILE_BYTES = 3 * SB_TH * SB_TW * 2;
constexpr int SB_W_BYTES = SB_STEPS * 2 * 64 * 16;

// V4 (round 6; image width a multiple of 4, frames 16-byte aligned): the zero tap of the padded 8 sits in FRONT (kw' = kw + 1), which moves
// the tile's first column from 2 col0 - 3 to 2 col0 - 4 -- a multiple of four pixels: the tile is staged with aligned 16-byte loads (4-byte
// for 8-bit frames), whole quads in or out of the image, 6 loads and one 8-byte LDS store per thread and tile instead of 22 four-byte loads
// and 11 stores: 232-240 -> 198-205 us at BASELINE configs[2].  [measured] profiles/r6_probe_stem_v4.txt
template <typename RGB, bool V4>
__global__ __launch_bounds__(256, 3) void stem_conv_bf16_kernel(const RGB* __restrict__ rgbs, const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                unsigned short* __restrict__ out, float* __restrict__ stats,
                                                                int H, int W, int Ho, int Wo, int tiles_x, int tiles, int F) {
    __shared__ __attribute__((aligned(16))) char sb_tile[SB_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) char sb_w[SB_W_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // ---- weights once per (persistent) block: arena layout [(ci*7+kh)*7 + kw][64] fp32 -> [step][half][n][8 kw] bf16
    for (int i = tid; i < SB_STEPS * 2 * 64 * 4; i += 256) {                 // one dword (two taps) per item
        const int kp = i & 3, n = (i >> 2) & 63, sh = i >> 8;                // sh = step*2 + half = (ci, kh) pair index
        const int k0 = V4 ? 2 * kp - 1 : 2 * kp;                            // filter tap of the pair's first K value (V4: K value 0 is the zero tap)
        const float a = (sh < 21 && k0 >= 0) ? w[(sh * 7 + k0) * 64 + n] : 0.f;
        const float b = (sh < 21 && k0 + 1 < 7) ? w[(sh * 7 + k0 + 1) * 64 + n] : 0.f;
        reinterpret_cast<unsigned*>(sb_w)[i] = pack2_bf16(a, b);
    }
    for (int work = blockIdx.x; work < tiles * F; work += gridDim.x) {
        const int frame = work / tiles, tile_id = work - frame * tiles;
        const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
        const int row0 = ty * SB_ROWS, col0 = tx * SB_COLS;
        // ---- stage the scaled input tile: pixel pairs, every load issued before the first use
        const RGB* src = rgbs + (size_t)frame * 3 * H * W;
        const int hi0 = 2 * row0 - 3, wi0 = 2 * col0 - (V4 ? 4 : 3);
        int tq = tid;
        asm volatile("" : "+v"(tq));                          // opaque per tile: the 11 steps' index arithmetic is recomputed here
                                                              // instead of living in ~60 registers across the persistent loop
        if constexpr (V4) {
            constexpr int NQ = 3 * SB_TH * (SB_TW / 4), NIT4 = (NQ + 255) / 256;
            float v[NIT4][4];                                 // raw pixel values of a quad; scaled when they are written to LDS
            unsigned in_mask = 0;
#pragma unroll
            for (int it = 0; it < NIT4; ++it) {
                const int q = tq + it * 256, qc = q < NQ ? q : 0;              // (the last step is partial)
                const int r = qc / (SB_TW / 4), j = qc - r * (SB_TW / 4);
                const int c = r / SB_TH, y = r - c * SB_TH;
                const int hi = hi0 + y, wi = wi0 + 4 * j;
                const bool in = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;     // W % 4 == 0: the quad is inside or outside as a whole
                const RGB* q4 = src + (c * H + (in ? hi : 0)) * W + (in ? wi : 0);               // < 2^31: one frame
                if constexpr (sizeof(RGB) == 1) {
                    const uchar4 u = *reinterpret_cast<const uchar4*>(q4);
                    v[it][0] = (float)u.x; v[it][1] = (float)u.y; v[it][2] = (float)u.z; v[it][3] = (float)u.w;
                } else {
                    const float4 f = *reinterpret_cast<const float4*>(q4);
                    v[it][0] = f.x; v[it][1] = f.y; v[it][2] = f.z; v[it][3] = f.w;
                }
                in_mask |= (in ? 1u : 0u) << it;
            }
            __syncthreads();                                  // the previous tile's fragment reads are done (and sb_w is written)
#pragma unroll
            for (int it = 0; it < NIT4; ++it) {
                const int q = tid + it * 256;
                const bool in = in_mask >> it & 1;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = in ? 2.0f * (v[it][e] / 255.0f) - 1.0f : 0.f;   // 2*(x/255)-1 (nets/pips.py:436); zero padding in the scaled domain
                if (q < NQ) reinterpret_cast<uint2*>(sb_tile)[q] = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
            }
        } else {
        constexpr int NPAIR = 3 * SB_TH * (SB_TW / 2), NIT = (NPAIR + 255) / 256;
        float v0[NIT], v1[NIT];                               // raw pixel values; scaled when they are written to LDS
        unsigned in_mask = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = tq + it * 256, qc = q < NPAIR ? q : 0;           // (the last step is partial)
            const int r = qc / (SB_TW / 2), xp = qc - r * (SB_TW / 2);
            const int c = r / SB_TH, y = r - c * SB_TH;
            const int hi = hi0 + y, wi = wi0 + 2 * xp;
            const bool hok = (unsigned)hi < (unsigned)H;
            const bool in0 = hok && (unsigned)wi < (unsigned)W, in1 = hok && (unsigned)(wi + 1) < (unsigned)W;
            const int base = (c * H + (hok ? hi : 0)) * W;                   // < 2^31: one frame
            v0[it] = (float)src[base + (in0 ? wi : 0)];
            v1[it] = (float)src[base + (in1 ? wi + 1 : 0)];
            in_mask |= (in0 ? 1u : 0u) << (2 * it) | (in1 ? 1u : 0u) << (2 * it + 1);
        }
        __syncthreads();                                      // the previous tile's fragment reads are done (and sb_w is written)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = tid + it * 256;
            // 2*(x/255)-1 (nets/pips.py:436); zero padding is in the scaled domain
            const float a = (in_mask >> (2 * it) & 1) ? 2.0f * (v0[it] / 255.0f) - 1.0f : 0.f;
            const float b = (in_mask >> (2 * it + 1) & 1) ? 2.0f * (v1[it] / 255.0f) - 1.0f : 0.f;
            if (q < NPAIR) reinterpret_cast<unsigned*>(sb_tile)[q] = pack2_bf16(a, b);
        }
        }
        __syncthreads();
        // ---- 11 steps x (2 x 2) MFMAs per wave
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const char* a_base = sb_tile + (2 * wave) * (SB_TW * 2) + 4 * l31;     // + row(ci, kh) * 272 + mt * 128
        const char* b_base = sb_w + (half * 64 + l31) * 16;                     // + step * 2048 + nt * 512
        // the next step's fragments are requested before this step's MFMAs (and nothing is hoisted further: 11 steps of
        // fragments in flight would cost the registers of two more waves per SIMD)
        uint4 fa[2][2], fb[2][2];
        auto load_frags = [&](int s, int buf) {
            int p = 2 * s + half;                                               // this lane half's (ci, kh) pair
            p = p < 21 ? p : 20;                                                // the 22nd pair has zero weights: any address
            const int ci = p / 7, kh = p - ci * 7;
            const char* ap = a_base + (ci * SB_TH + kh) * (SB_TW * 2);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned* a4 = reinterpret_cast<const unsigned*>(ap + mt * 128);
                fa[buf][mt] = make_uint4(a4[0], a4[1], a4[2], a4[3]);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[buf][nt] = *reinterpret_cast<const uint4*>(b_base + s * 2048 + nt * 512);
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < SB_STEPS; ++s) {
            if (s + 1 < SB_STEPS) load_frags(s + 1, (s + 1) & 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[s & 1][mt]),
                                                                          *reinterpret_cast<const bf16x8*>(&fb[s & 1][nt]),
                                                                          acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: C orientation (lane = channel, register = pixel); pivoted statistics from the fp32 accumulators
        const int orow = row0 + wave;
        const bool row_ok = orow < Ho;
        const int nvalid = row_ok ? min(SB_COLS, Wo - col0) : 0;
        unsigned short* orow_p = out + (((size_t)frame * Ho + (row_ok ? orow : 0)) * Wo + col0) * 64;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 32 + l31;
            const float bv = bias[n];
            const float pivot = __shfl(acc[0][nt][0] + bv, l31);
            float cs = 0.f, cq = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = acc[mt][nt][r] + bv;
                    if (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < nvalid) {
                        const float d = v[r] - pivot;
                        cs += d;
                        cq += d * d;
                    }
                }
                store_c_tile_bf16(v, l31, half, [&](int px) -> unsigned short* {
                    return mt * 32 + px < nvalid ? orow_p + (size_t)(mt * 32 + px) * 64 + nt * 32 : nullptr;
                });
            }
            cs += __shfl_xor(cs, 32);
            cq += __shfl_xor(cq, 32);
            if (half == 0)
                reinterpret_cast<float4*>(stats)[(((size_t)frame * tiles + tile_id) * 4 + wave) * 64 + n] =
                    make_float4(cs, cq, pivot, (float)nvalid);
        }
    }
}

int launch_stem_bf16(const void* rgbs, int rgb_u8, const float* w, const float* bias, void* out, float* stats,
                     int F, int H, int W, int Ho, int Wo, int* tiles_m, hipStream_t st) {
    const int tiles_x = cdiv(Wo, SB_COLS), tiles = cdiv(Ho, SB_ROWS) * tiles_x;
    if (tiles_m) *tiles_m = stem_tiles_m(Ho, Wo);                      // same tiling as the fp32 stem: one partial per wave
    const int cus = device_cus();
    if (cus <= 0) { set_error("stem_bf16: cannot query the device"); return PIPS_E_LAUNCH; }
    const long total = (long)tiles * F;
    // persistent (the weights are converted once per block), ONE round of resident blocks: 168 registers = three blocks per compute unit (a grid of
    // 4 x CUs ran a quarter of the blocks in a second round at one block per unit: 204.6 -> 191.8 us at BASELINE configs[2])
    const int grid = total < 3L * cus ? (int)total : 3 * cus;
    // quads of four pixels: rows and frames start on a quad boundary when W % 4 == 0 and the first frame does
    const bool v4 = W % 4 == 0 && reinterpret_cast<uintptr_t>(rgbs) % (rgb_u8 ? 4 : 16) == 0 && PIPS_TUNE("PIPS_STEM_V4", 1);
#define PIPS_STEM(T_, V_) hipLaunchKernelGGL((stem_conv_bf16_kernel<T_, V_>), dim3(grid), dim3(256), 0, st, (const T_*)rgbs, w, bias, \
                                             (unsigned short*)out, stats, H, W, Ho, Wo, tiles_x, tiles, F)
    if (rgb_u8) { if (v4) PIPS_STEM(unsigned char, true); else PIPS_STEM(unsigned char, false); }
    else { if (v4) PIPS_STEM(float, true); else PIPS_STEM(float, false); }
#undef PIPS_STEM
    PIPS_CHECK_LAUNCH("stem_conv_bf16_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------ instance-norm apply, bf16 maps
// 8 channels (16 bytes) per thread and step.  n(x) = (x - mean) * rstd with the layer's {mean, rstd} per (frame, channel).
// MODE 0: y = relu(n(x))                              nets/pips.py:175/176/252-253/274-275
// MODE 1: y = relu(res + relu(n(x)))                  :176,181 (identity shortcut, res a finished activation)
// MODE 2: y = relu(n2(res) + relu(n(x)))              :169-170,179,181 (1x1 stride-2 shortcut + norm3)
// MODE 3: y = relu(relu(n2(res)) + relu(n(x)))        the first block's identity shortcut is the stem's relu(norm1(.)),
//                                                     which is never materialised: it is recomputed from the stem's raw map
// Block = a run of APPLY_ITERS x (threads / C8) pixels of ONE frame; thread = one 8-channel group c8 (the block has a multiple of C8 threads):
// the 8 x {mean, rstd} of the thread's channels (and the shortcut's) are loaded ONCE into registers.  The grid-stride form of round 4 fetched
// them per element -- 64 (128 with a normalised shortcut) bytes through the L1 for every 48 bytes of map traffic.
// [measured] profiles/r6_probe_inorm_apply.txt
constexpr int APPLY_ITERS = 8;
template <int MODE>
__global__ __launch_bounds__(256) void inorm_apply_bf16_kernel(const uint4* __restrict__ x, const float4* __restrict__ stats,
                                                               const uint4* __restrict__ res,
                                                               const float4* __restrict__ res_stats, uint4* __restrict__ y,
                                                               int HW, int C8, int chunks) {
    const int f = blockIdx.x / chunks, chunk = blockIdx.x - f * chunks;
    const int ppb = blockDim.x / C8;                                         // pixels per step of the block
    const int c8 = threadIdx.x % C8, pl = threadIdx.x / C8;
    float4 sx[4], sr[4];
    {
        const float4* st = stats + ((size_t)f * C8 + c8) * 4;                // 8 x {mean, rstd}
#pragma unroll
        for (int k = 0; k < 4; ++k) sx[k] = st[k];
        if (MODE >= 2) {
            const float4* rs = res_stats + ((size_t)f * C8 + c8) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) sr[k] = rs[k];
        }
    }
    const int p0 = chunk * (APPLY_ITERS * ppb) + pl;
    const size_t base = (size_t)f * HW * C8 + c8;
#pragma unroll
    for (int it = 0; it < APPLY_ITERS; ++it) {
        const int pp = p0 + it * ppb;
        const bool live = pp < HW;
        const int p = live ? pp : HW - 1;                                    // (loads stay unconditional: every step's are in flight together)
        const size_t i = base + (size_t)p * C8;
        const uint4 xv = x[i];
        const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 s = sx[k];
            o[2 * k] = fmaxf((bf16_lo(xw[k]) - s.x) * s.y, 0.f);
            o[2 * k + 1] = fmaxf((bf16_hi(xw[k]) - s.z) * s.w, 0.f);
        }
        if (MODE != 0) {
            const uint4 rv = res[i];
            const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
            float r[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { r[2 * k] = bf16_lo(rw[k]); r[2 * k + 1] = bf16_hi(rw[k]); }
            if (MODE >= 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 s = sr[k];
                    r[2 * k] = (r[2 * k] - s.x) * s.y;
                    r[2 * k + 1] = (r[2 * k + 1] - s.z) * s.w;
                    if (MODE == 3) { r[2 * k] = fmaxf(r[2 * k], 0.f); r[2 * k + 1] = fmaxf(r[2 * k + 1], 0.f); }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaxf(r[k] + o[k], 0.f);
        }
        if (live) y[i] = make_uint4(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7]));
    }
}

int launch_inorm_apply_bf16(const void* x, const float* stats, const void* res, const float* res_stats, int mode, void* y,
                            int F, int HW, int C, hipStream_t st) {
    PIPS_CHECK_ARG(C % 8 == 0 && C <= 2048 && mode >= 0 && mode <= 3, "inorm_apply_bf16: C %% 8, C <= 2048, mode 0..3");
    PIPS_CHECK_ARG(mode == 0 || res != nullptr, "inorm_apply_bf16: mode %d needs a shortcut map", mode);
    PIPS_CHECK_ARG(mode < 2 || res_stats != nullptr, "inorm_apply_bf16: mode %d needs the shortcut's statistics", mode);
    const int C8 = C / 8;
    const int threads = (256 / C8) * C8;                                     // a multiple of the channel groups: 256, 252 (C = 96), ...
    const int chunks = cdiv(HW, APPLY_ITERS * (threads / C8));
    PIPS_CHECK_ARG((long)F * chunks < (1L << 31), "inorm_apply_bf16: %d frames x %d chunks", F, chunks);
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    const float4* s4 = reinterpret_cast<const float4*>(stats);
    const uint4* r4 = reinterpret_cast<const uint4*>(res);
    const float4* rs4 = reinterpret_cast<const float4*>(res_stats);
    uint4* y4 = reinterpret_cast<uint4*>(y);
#define PIPS_APPLY(M_) hipLaunchKernelGGL(inorm_apply_bf16_kernel<M_>, dim3(F * chunks), dim3(threads), 0, st, x4, s4, r4, rs4, y4, HW, C8, chunks)
    if (mode == 0) PIPS_APPLY(0); else if (mode == 1) PIPS_APPLY(1); else if (mode == 2) PIPS_APPLY(2); else PIPS_APPLY(3);
#undef PIPS_APPLY
    PIPS_CHECK_LAUNCH("inorm_apply_bf16_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------------ resize, bf16 maps
// F.interpolate(mode='bilinear', align_corners=True) (nets/pips.py:269-272) of a bf16 NHWC map into channels
// [coff, coff+C) of the concatenated bf16 NHWC map (the torch.cat of :273); fp32 arithmetic, 8 channels per thread.
__global__ __launch_bounds__(256) void resize_into_bf16_kernel(const uint4* __restrict__ src, int Hs, int Ws, int C8,
                                                               uint4* __restrict__ dst, int Hd, int Wd, int Cdst8, int coff8,
                                                               float sh, float sw, size_t total8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        size_t p = i / C8;
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
        const uint4* s8 = src + (size_t)f * Hs * Ws * C8 + c8;
        const uint4 q00 = s8[((size_t)y0 * Ws + x0) * C8], q01 = s8[((size_t)y0 * Ws + x1) * C8];
        const uint4 q10 = s8[((size_t)y1 * Ws + x0) * C8], q11 = s8[((size_t)y1 * Ws + x1) * C8];
        const unsigned a[4] = {q00.x, q00.y, q00.z, q00.w}, b[4] = {q01.x, q01.y, q01.z, q01.w};
        const unsigned c[4] = {q10.x, q10.y, q10.z, q10.w}, d[4] = {q11.x, q11.y, q11.z, q11.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = ly0 * (lx0 * bf16_lo(a[k]) + lx1 * bf16_lo(b[k])) + ly1 * (lx0 * bf16_lo(c[k]) + lx1 * bf16_lo(d[k]));
            const float hi = ly0 * (lx0 * bf16_hi(a[k]) + lx1 * bf16_hi(b[k])) + ly1 * (lx0 * bf16_hi(c[k]) + lx1 * bf16_hi(d[k]));
            o[k] = pack2_bf16(lo, hi);
        }
        dst[(((size_t)f * Hd + y) * Wd + x) * Cdst8 + coff8 + c8] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int launch_resize_into_bf16(const void* src, int F, int Hs, int Ws, int C, void* dst, int Hd, int Wd, int Cdst, int coff,
                            hipStream_t st) {
    PIPS_CHECK_ARG(C % 8 == 0 && Cdst % 8 == 0 && coff % 8 == 0, "resize_bf16: channel alignment");
    const size_t total8 = (size_t)F * Hd * Wd * (C / 8);
    const int blocks = (int)((total8 + 255) / 256 < 4096 ? (total8 + 255) / 256 : 4096);
    const float sh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
    const float sw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
    hipLaunchKernelGGL(resize_into_bf16_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), Hs, Ws,
                       C / 8, reinterpret_cast<uint4*>(dst), Hd, Wd, Cdst / 8, coff / 8, sh, sw, total8);
    PIPS_CHECK_LAUNCH("resize_into_bf16_kernel");
    return PIPS_OK;
}

}  // namespace pips
