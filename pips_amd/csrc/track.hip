// Tracker-side kernels: initial point sample, the fused local-correlation gather that
// builds the mixer input, the token-mixing half of each mixer block, the final
// LayerNorm+mean, and the state update.  None of them is GEMM-shaped; they are laid out
// for coalesced 512-byte channel-last reads and 64-lane waves.
// State tensors are particle-major: row m = (b*N + n)*8 + s.
#include "common.h"

namespace pips {

constexpr int S = PIPS_S;
constexpr int C = PIPS_C;


// ------------------------------------------------------------------------ point sample
// utils.samp.bilinear_sample2d (utils/samp.py:5-78) on frame 0 of each clip: neighbour
// indices clamped to the border, weights unclamped, products and sums rounded one by one
// in the reference's order (samp.py:59-65).  128 threads = 128 channels of one point.
__global__ __launch_bounds__(128) void point_sample_kernel(const float* __restrict__ level0, int S_,
                                                           int H, int W, const float* __restrict__ xy,
                                                           int xy_stride, int N,
                                                           const int* __restrict__ win_start,
                                                           float* __restrict__ out) {
    const int pn = blockIdx.x;            // b*N + n
    const int b = pn / N;
    const float x = xy[(size_t)pn * xy_stride + 0], y = xy[(size_t)pn * xy_stride + 1];
    const float x0f = floorf(x), y0f = floorf(y);
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const int x0 = min(max((int)x0f, 0), W - 1), x1 = min(max((int)x0f + 1, 0), W - 1);
    const int y0 = min(max((int)y0f, 0), H - 1), y1 = min(max((int)y0f + 1, 0), H - 1);
    const float w00 = __fmul_rn(x1f - x, y1f - y), w01 = __fmul_rn(x - x0f, y1f - y);
    const float w10 = __fmul_rn(x1f - x, y - y0f), w11 = __fmul_rn(x - x0f, y - y0f);
    const int f_first = win_start != nullptr ? min(max(win_start[pn], 0), S_ - 1) : 0;
    const float* f0 = level0 + ((size_t)b * S_ + f_first) * H * W * C;   // first frame of the window
    const int c = threadIdx.x;
    const float v00 = f0[((size_t)y0 * W + x0) * C + c], v01 = f0[((size_t)y0 * W + x1) * C + c];
    const float v10 = f0[((size_t)y1 * W + x0) * C + c], v11 = f0[((size_t)y1 * W + x1) * C + c];
    float o = __fadd_rn(__fmul_rn(w00, v00), __fmul_rn(w01, v01));
    o = __fadd_rn(o, __fmul_rn(w10, v10));
    o = __fadd_rn(o, __fmul_rn(w11, v11));
    out[(size_t)pn * C + c] = o;
}

int launch_point_sample_strided(const float* level0, int B, int S_, int H8, int W8, const float* xy,
                                int xy_stride, int N, const int* win_start, float* out, hipStream_t st) {
    hipLaunchKernelGGL(point_sample_kernel, dim3(B * N), dim3(128), 0, st, level0, S_, H8, W8, xy,
                       xy_stride, N, win_start, out);
    PIPS_CHECK_LAUNCH("point_sample_kernel");
    return PIPS_OK;
}

int launch_point_sample(const float* level0, int B, int S_, int H8, int W8, const float* xy, int N,
                        float* out, hipStream_t st) {
    return launch_point_sample_strided(level0, B, S_, H8, W8, xy, 2, N, nullptr, out, st);
}

// ------------------------------------------------------------------------ state init
// nets/pips.py:450-455 (coords), :466 (ffeats repeat), :468 (coords_bak), :474 (first entry
// of coord_predictions2).  Phase 0 writes coords; phase 1 (after the point sample) ffeats.
__global__ void init_coords_kernel(const float* __restrict__ xys, const float* __restrict__ coords_init,
                                   int B, int N, int Sw, float stride, float* __restrict__ coords,
                                   float* __restrict__ coords0, float* __restrict__ out_traj0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // (b*N + n)*Sw + s
    if (i >= B * N * Sw) return;
    const int s = i % Sw, pn = i / Sw, n = pn % N, b = pn / N;
    float x, y;
    if (coords_init != nullptr) {
        const float* p = coords_init + (((size_t)b * Sw + s) * N + n) * 2;
        x = p[0] / stride; y = p[1] / stride;
    } else {
        x = xys[(size_t)pn * 2 + 0] / stride; y = xys[(size_t)pn * 2 + 1] / stride;
    }
    coords[(size_t)i * 2 + 0] = x; coords[(size_t)i * 2 + 1] = y;
    coords0[(size_t)i * 2 + 0] = x; coords0[(size_t)i * 2 + 1] = y;
    float* o = out_traj0 + (((size_t)b * Sw + s) * N + n) * 2;
    o[0] = x * stride; o[1] = y * stride;
}

__global__ void init_ffeats_kernel(const float4* __restrict__ ffeat0, int BN, int Sw, float4* __restrict__ ffeats) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over BN*Sw*32 float4
    if (i >= (size_t)BN * Sw * (C / 4)) return;
    const size_t pn = i / ((size_t)Sw * (C / 4));
    ffeats[i] = ffeat0[pn * (C / 4) + (i % (C / 4))];
}

int launch_init_coords(const float* xys, const float* coords_init, int B, int N, float stride,
                       float* coords, float* coords0, float* out_traj0, hipStream_t st, int Sw) {
    const int total = B * N * Sw;
    hipLaunchKernelGGL(init_coords_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, xys, coords_init, B,
                       N, Sw, stride, coords, coords0, out_traj0);
    PIPS_CHECK_LAUNCH("init_coords_kernel");
    return PIPS_OK;
}

int launch_init_ffeats(const float* ffeat0, int BN, float* ffeats, hipStream_t st, int Sw) {
    const size_t total = (size_t)BN * Sw * (C / 4);
    hipLaunchKernelGGL(init_ffeats_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(ffeat0), BN, Sw, reinterpret_cast<float4*>(ffeats));
    PIPS_CHECK_LAUNCH("init_ffeats_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------- mixer input build
// CorrBlock.corr + CorrBlock.sample (nets/pips.py:384-398, 355-382) without ever forming
// the (B,S,N,H,W) correlation volume, fused with get_3d_embedding (utils/misc.py:44-69)
// and the concat of DeltaBlock.forward (:304-308).
//
// One block per mixer row m=(b,n,s); wave l handles pyramid level l.  All 49 bilinear
// taps of a level share one fractional offset, so they live on an 8x8 integer pixel
// window: the wave reads the window as 8 contiguous 4 KiB row segments (2 pixels x 128
// channels per 1 KiB wave load), forms 64 dot products <ffeat, pixel> (32 lanes x float4
// per pixel, transpose-reduced over the 32-lane halves), scales by 1/sqrt(C), and blends
// 2x2 -> 49 taps written in the reference's transposed order (k = ix*7 + iy).
// Out-of-map pixels contribute 0 (grid_sample zeros padding, align_corners=True).
struct LevelTable {
    size_t off[PIPS_LEVELS];
    int H[PIPS_LEVELS], W[PIPS_LEVELS];
};

// waves_per_eu(2,4): let the compiler spend up to 128 VGPRs so 16 x 1 KiB loads stay in flight per
// wave (left alone it squeezes into 64 VGPRs for 8 waves/SIMD and issues the loads two at a time)
// SCT: window length (mixer rows per particle) as a compile-time constant, 0 = the run-time argument Srt (Pips(S != 8))
template <int SCT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void mixer_input_kernel(const float* __restrict__ pyramid,
                                                          LevelTable lv, int S_, int Srt,
                                                          const float* __restrict__ ffeats,
                                                          const float* __restrict__ coords,
                                                          const float* __restrict__ times, int N,
                                                          const int* __restrict__ win_start,
                                                          float* __restrict__ X) {
    __shared__ float Dw[PIPS_LEVELS][64];
    const int S = SCT ? SCT : Srt;                       // (shadows the file-scope constant)
    const int m = blockIdx.x;
    const int s = m % S, pn = m / S;
    const int b = pn / N;
    // The map buffer holds S_ frames per clip.  S_ = 8 with win_start == null is the plain
    // forward; a longer cache + per-particle window start gives chained tracking, where frames
    // past the end repeat the last one (chain_demo.py:50-52) = a clamp of the frame index.
    const int fstart = win_start != nullptr ? win_start[pn] : 0;
    const int frame = b * S_ + min(max(fstart + s, 0), S_ - 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int lvl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float cxm = coords[(size_t)m * 2 + 0], cym = coords[(size_t)m * 2 + 1];
    float* xrow = X + (size_t)m * PIPS_KIN_PAD;
    const float* ff = ffeats + (size_t)m * C;

    // ---- correlation window of this wave's level
    {
        const int H = lv.H[lvl], W = lv.W[lvl];
        const float inv = 1.0f / (float)(1 << lvl);
        const float cx = cxm * inv, cy = cym * inv;        // coords / 2**i  (:373)
        // bilinear_sampler normalisation (:318-319) and grid_sample's un-normalisation
        // (align_corners=True, CPU form (g+1)*((size-1)/2)), reproduced op by op
        const float gx = __fsub_rn(__fdiv_rn(2.0f * cx, (float)(W - 1)), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(2.0f * cy, (float)(H - 1)), 1.0f);
        const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), (float)(W - 1) / 2.0f);
        const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), (float)(H - 1) / 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx = ix - fx0, wy = iy - fy0;          // east / south weights
        const int bx = (int)fx0 - PIPS_RADIUS, by = (int)fy0 - PIPS_RADIUS;

        const int hsel = lane >> 5, c4 = lane & 31;
        const float4 f4 = *reinterpret_cast<const float4*>(ff + c4 * 4);
        const float* base = pyramid + lv.off[lvl] + (size_t)frame * H * W * C + c4 * 4;
        // Loads are UNCONDITIONAL (out-of-map pixels read a clamped in-map address and are
        // zeroed afterwards): a branch per load would fence each load behind its own use and
        // serialise 32 L2 round trips per wave.  Two batches of 16 x 1 KiB loads in flight.
        float v[32];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float4 t[16];
            bool ok[16];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int py = by + hb * 4 + jj;
                const bool yok = (unsigned)py < (unsigned)H;
                const int pyc = min(max(py, 0), H - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = bx + 2 * q + hsel;
                    ok[jj * 4 + q] = yok && (unsigned)px < (unsigned)W;
                    const int pxc = min(max(px, 0), W - 1);
                    t[jj * 4 + q] = *reinterpret_cast<const float4*>(base + ((size_t)pyc * W + pxc) * C);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = t[e].x * f4.x + t[e].y * f4.y + t[e].z * f4.z + t[e].w * f4.w;
                v[hb * 16 + e] = ok[e] ? d : 0.f;
            }
        }
        // transpose-reduce over the 32 lanes of each half: lane r ends with sum of v[r]
        // (one macro instance per level: a two-variable loop here was left rolled by hipcc and
        // v[] became a 4000-instruction compare/select emulation of dynamic register indexing)
#define PIPS_TR_STEP(O, NH)                                                    \
        {                                                                      \
            const bool up = (lane & (O)) != 0;                                 \
            _Pragma("unroll") for (int k = 0; k < (NH); ++k) {                 \
                const float send = up ? v[k] : v[k + (NH)];                    \
                const float keep = up ? v[k + (NH)] : v[k];                    \
                v[k] = keep + __shfl_xor(send, (O));                           \
            }                                                                  \
        }
        PIPS_TR_STEP(16, 16) PIPS_TR_STEP(8, 8) PIPS_TR_STEP(4, 4) PIPS_TR_STEP(2, 2) PIPS_TR_STEP(1, 1)
#undef PIPS_TR_STEP
        // lane (hsel, r=c4): window row j = r>>2, column 2*(r&3) + hsel
        const float scale = sqrtf((float)C);
        Dw[lvl][(c4 >> 2) * 8 + 2 * (c4 & 3) + hsel] = v[0] / scale;     // corrs / sqrt(C) (:397)
        __syncthreads();                       // (all four waves take the same path)
        if (lane < 49) {
            const int ti = lane / 7, tj = lane - ti * 7;                 // x index, y index
            const float e = 1.0f - wx, so = 1.0f - wy;
            const float nw = Dw[lvl][tj * 8 + ti], ne = Dw[lvl][tj * 8 + ti + 1];
            const float sw = Dw[lvl][(tj + 1) * 8 + ti], se = Dw[lvl][(tj + 1) * 8 + ti + 1];
            float o = nw * (so * e);
            o += ne * (so * wx);
            o += sw * (wy * e);
            o += se * (wy * wx);
            xrow[C + lvl * 49 + lane] = o;                                // k = ix*7 + iy
        }
    }

    // ---- feature copy, sin/cos embedding of (dx, dy, t), raw flow, zero pad
    if (tid < C / 4)
        reinterpret_cast<float4*>(xrow)[tid] = reinterpret_cast<const float4*>(ff)[tid];
    const float dx = cxm - coords[(size_t)pn * S * 2 + 0];              // coords - coords[:,0:1] (:518)
    const float dy = cym - coords[(size_t)pn * S * 2 + 1];
    const float tt = times[s];                                           // linspace(0,S,S) (:519)
    if (tid < 192) {
        const int a = tid >> 6, i = tid & 63;
        const float val = a == 0 ? dx : (a == 1 ? dy : tt);
        const float freq = (float)(i >> 1) * 31.25f;                     // arange(0,64,2)*(1000/64)
        const float arg = __fmul_rn(val, freq);
        xrow[C + PIPS_NCORR + tid] = (i & 1) ? cosf(arg) : sinf(arg);    // misc.py:56-63
    } else if (tid < 192 + 3) {
        const int a = tid - 192;
        xrow[C + PIPS_NCORR + 192 + a] = a == 0 ? dx : (a == 1 ? dy : tt);
    } else if (tid < 192 + 3 + (PIPS_KIN_PAD - PIPS_KIN)) {
        xrow[PIPS_KIN + (tid - 195)] = 0.f;
    }
}

// The same gather on the bf16 MIRROR of the pyramid (bf16 mode, PIPS_FLAG_BF16_MAPS): under torch.autocast the encoder's output
// is a bf16 tensor and CorrBlock.corr multiplies bf16 operands (nets/pips.py:394-395), so bf16 map values are the reference's own
// rounding point; the track features are rounded to bf16 on load (round 6: one rounding contract with gather_mfma_kernel), products
// (exact in fp32) and sums stay fp32.  What it buys is INSTRUCTIONS, not bytes: the direct gather is bound by
// the rate of its vector-memory instructions (16 384 blocks x 4 waves x 32 loads at ~30 clocks each = the 107 us it takes at
// BASELINE configs[2]); with 8 channels per lane a wave load covers FOUR pixels, 16 loads per level instead of 32.
// Lane = (pixel of the group lane >> 4, channel octet lane & 15); 16 partial dot products per lane, transpose-reduced over the 16
// lanes of a pixel group: lane r of group g ends with window row r >> 1, column 4 * (r & 1) + g.
template <int SCT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void mixer_input_bf16maps_kernel(
    const unsigned short* __restrict__ mirror, LevelTable lv, int S_, int Srt, const float* __restrict__ ffeats,
    const float* __restrict__ coords, const float* __restrict__ times, int N, const int* __restrict__ win_start, float* __restrict__ X) {
    __shared__ float Dw[PIPS_LEVELS][64];
    const int S = SCT ? SCT : Srt;
    const int m = blockIdx.x;
    const int s = m % S, pn = m / S;
    const int b = pn / N;
    const int fstart = win_start != nullptr ? win_start[pn] : 0;
    const int frame = b * S_ + min(max(fstart + s, 0), S_ - 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int lvl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float cxm = coords[(size_t)m * 2 + 0], cym = coords[(size_t)m * 2 + 1];
    float* xrow = X + (size_t)m * PIPS_KIN_PAD;
    const float* ff = ffeats + (size_t)m * C;
    {
        const int H = lv.H[lvl], W = lv.W[lvl];
        const float inv = 1.0f / (float)(1 << lvl);
        const float cx = cxm * inv, cy = cym * inv;
        const float gx = __fsub_rn(__fdiv_rn(2.0f * cx, (float)(W - 1)), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(2.0f * cy, (float)(H - 1)), 1.0f);
        const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), (float)(W - 1) / 2.0f);
        const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), (float)(H - 1) / 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx = ix - fx0, wy = iy - fy0;
        const int bx = (int)fx0 - PIPS_RADIUS, by = (int)fy0 - PIPS_RADIUS;

        const int psel = lane >> 4, c8 = lane & 15;
        // BOTH operands are bf16 (round 6): under autocast torch.matmul casts the track features as well as the maps
        // (nets/pips.py:394-397), and gather_mfma_kernel -- the dense route of the same mode -- multiplies bf16 x bf16.  The
        // features are rounded here (RNE, the hardware convert) so that the two routes differ in summation order only.
        float4 fa = *reinterpret_cast<const float4*>(ff + c8 * 8), fb = *reinterpret_cast<const float4*>(ff + c8 * 8 + 4);
        {
            const unsigned p0 = pack2_bf16(fa.x, fa.y), p1 = pack2_bf16(fa.z, fa.w), p2 = pack2_bf16(fb.x, fb.y), p3 = pack2_bf16(fb.z, fb.w);
            fa = make_float4(bf16_lo(p0), bf16_hi(p0), bf16_lo(p1), bf16_hi(p1));
            fb = make_float4(bf16_lo(p2), bf16_hi(p2), bf16_lo(p3), bf16_hi(p3));
        }
        const unsigned short* base = mirror + lv.off[lvl] + (size_t)frame * H * W * C + c8 * 8;
        uint4 t[16];
        bool ok[16];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int py = by + jj;
            const bool yok = (unsigned)py < (unsigned)H;
            const int pyc = min(max(py, 0), H - 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int px = bx + 4 * q + psel;
                ok[jj * 2 + q] = yok && (unsigned)px < (unsigned)W;
                const int pxc = min(max(px, 0), W - 1);
                t[jj * 2 + q] = *reinterpret_cast<const uint4*>(base + ((size_t)pyc * W + pxc) * C);
            }
        }
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint4 u = t[e];
            float d = bf16_lo(u.x) * fa.x;
            d = fmaf(bf16_hi(u.x), fa.y, d);
            d = fmaf(bf16_lo(u.y), fa.z, d);
            d = fmaf(bf16_hi(u.y), fa.w, d);
            d = fmaf(bf16_lo(u.z), fb.x, d);
            d = fmaf(bf16_hi(u.z), fb.y, d);
            d = fmaf(bf16_lo(u.w), fb.z, d);
            d = fmaf(bf16_hi(u.w), fb.w, d);
            v[e] = ok[e] ? d : 0.f;
        }
#define PIPS_TR_STEP(O, NH)                                                    \
        {                                                                      \
            const bool up = (lane & (O)) != 0;                                 \
            _Pragma("unroll") for (int k = 0; k < (NH); ++k) {                 \
                const float send = up ? v[k] : v[k + (NH)];                    \
                const float keep = up ? v[k + (NH)] : v[k];                    \
                v[k] = keep + __shfl_xor(send, (O));                           \
            }                                                                  \
        }
        PIPS_TR_STEP(8, 8) PIPS_TR_STEP(4, 4) PIPS_TR_STEP(2, 2) PIPS_TR_STEP(1, 1)
#undef PIPS_TR_STEP
        const float scale = sqrtf((float)C);
        Dw[lvl][(c8 >> 1) * 8 + 4 * (c8 & 1) + psel] = v[0] / scale;
        __syncthreads();
        if (lane < 49) {
            const int ti = lane / 7, tj = lane - ti * 7;
            const float e = 1.0f - wx, so = 1.0f - wy;
            const float nw = Dw[lvl][tj * 8 + ti], ne = Dw[lvl][tj * 8 + ti + 1];
            const float sw = Dw[lvl][(tj + 1) * 8 + ti], se = Dw[lvl][(tj + 1) * 8 + ti + 1];
            float o = nw * (so * e);
            o += ne * (so * wx);
            o += sw * (wy * e);
            o += se * (wy * wx);
            xrow[C + lvl * 49 + lane] = o;
        }
    }
    if (tid < C / 4)
        reinterpret_cast<float4*>(xrow)[tid] = reinterpret_cast<const float4*>(ff)[tid];
    const float dx = cxm - coords[(size_t)pn * S * 2 + 0];
    const float dy = cym - coords[(size_t)pn * S * 2 + 1];
    const float tt = times[s];
    if (tid < 192) {
        const int a = tid >> 6, i = tid & 63;
        const float val = a == 0 ? dx : (a == 1 ? dy : tt);
        const float freq = (float)(i >> 1) * 31.25f;
        const float arg = __fmul_rn(val, freq);
        xrow[C + PIPS_NCORR + tid] = (i & 1) ? cosf(arg) : sinf(arg);
    } else if (tid < 192 + 3) {
        const int a = tid - 192;
        xrow[C + PIPS_NCORR + 192 + a] = a == 0 ? dx : (a == 1 ? dy : tt);
    } else if (tid < 192 + 3 + (PIPS_KIN_PAD - PIPS_KIN)) {
        xrow[PIPS_KIN + (tid - 195)] = 0.f;
    }
}

int launch_mixer_input_bf16maps(const void* mirror, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B, int S_,
                                const float* ffeats, const float* coords, const float* times, int N, const int* win_start,
                                float* X, hipStream_t st, int Sw) {
    LevelTable lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) { lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l]; }
    const unsigned short* mp = reinterpret_cast<const unsigned short*>(mirror);
    if (Sw == PIPS_S)
        hipLaunchKernelGGL(mixer_input_bf16maps_kernel<PIPS_S>, dim3(B * N * S), dim3(256), 0, st, mp, lv, S_, Sw, ffeats,
                           coords, times, N, win_start, X);
    else
        hipLaunchKernelGGL(mixer_input_bf16maps_kernel<0>, dim3(B * N * Sw), dim3(256), 0, st, mp, lv, S_, Sw, ffeats,
                           coords, times, N, win_start, X);
    PIPS_CHECK_LAUNCH("mixer_input_bf16maps_kernel");
    return PIPS_OK;
}

// fp32 pyramid -> its bf16 mirror (same element offsets), 8 values per thread
__global__ void pyramid_mirror_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const float4 a = src[2 * i], b = src[2 * i + 1];
    dst[i] = make_uint4(pack2_bf16(a.x, a.y), pack2_bf16(a.z, a.w), pack2_bf16(b.x, b.y), pack2_bf16(b.z, b.w));
}

int launch_pyramid_mirror(const float* pyramid, size_t floats, void* mirror, hipStream_t st) {
    const size_t n8 = floats / 8;                       // the pyramid's sections are multiples of 64 floats
    hipLaunchKernelGGL(pyramid_mirror_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(pyramid), reinterpret_cast<uint4*>(mirror), n8);
    PIPS_CHECK_LAUNCH("pyramid_mirror_kernel");
    return PIPS_OK;
}

int launch_mixer_input(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW,
                       int B, int S_, const float* ffeats, const float* coords, const float* times,
                       int N, const int* win_start, float* X, hipStream_t st, int Sw) {
    LevelTable lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) { lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l]; }
    if (Sw == PIPS_S)
        hipLaunchKernelGGL(mixer_input_kernel<PIPS_S>, dim3(B * N * S), dim3(256), 0, st, pyramid, lv, S_, Sw, ffeats,
                           coords, times, N, win_start, X);
    else
        hipLaunchKernelGGL(mixer_input_kernel<0>, dim3(B * N * Sw), dim3(256), 0, st, pyramid, lv, S_, Sw, ffeats,
                           coords, times, N, win_start, X);
    PIPS_CHECK_LAUNCH("mixer_input_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------ token mixing
// First half of a mixer block + the LayerNorm of the second half (nets/pips.py:117-118):
//   y  = x + Conv1d(32->8)(GELU(Conv1d(8->32)(LN1(x))))     tokens = the S axis
//   xn = LN2(y)                                             (input of the 512->2048 GEMM)
// One block per particle: its (8 tokens x 512 channels) tile lives in registers, two
// channels per thread; LN statistics are two-pass (mean, then centred squares) in fp32.
__device__ __forceinline__ void block_sum8(float (&v)[S], float (*red)[S]) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int t = 0; t < S; ++t) v[t] += __shfl_xor(v[t], o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // previous use of red finished
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < S; ++t) red[wave][t] = v[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < S; ++t) v[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

__device__ __forceinline__ void ln_stats(const float (&x0)[S], const float (&x1)[S], float (&mean)[S],
                                         float (&rstd)[S], float (*red)[S]) {
    float a[S];
#pragma unroll
    for (int t = 0; t < S; ++t) a[t] = x0[t] + x1[t];
    block_sum8(a, red);
#pragma unroll
    for (int t = 0; t < S; ++t) mean[t] = a[t] * (1.0f / PIPS_DMIX);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const float d0 = x0[t] - mean[t], d1 = x1[t] - mean[t];
        a[t] = d0 * d0 + d1 * d1;
    }
    block_sum8(a, red);
#pragma unroll
    for (int t = 0; t < S; ++t) rstd[t] = 1.0f / sqrtf(a[t] * (1.0f / PIPS_DMIX) + 1e-5f);
}

// The kernel is instruction-bound (PMC: 3.2k VALU instructions per wave in the previous
// one-channel-per-thread version), so everything is done to cut instruction count: two
// channels per thread in packed float2 arithmetic (v_pk_fma_f32), the eight per-token wave sums of
// every LayerNorm pass in one transpose-reduce, cross-wave combination through 32 floats of LDS.

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK_MASK, false));
}
// Eight wave sums at once by transpose-reduce: three exchange steps (lane^1, ^2, ^4) in which a lane
// keeps the half of its values that matches its lane bit and adds the partner's copy of it -- 8 -> 4
// -> 2 -> 1 value per lane -- then three plain steps (^8, ^16, ^32).  Lane l returns the sum over the
// wave of v[l & 7]: 26 instructions against ~200 for eight separate wave reductions.  DPP quad_perm /
// row_shl / row_shr / row_ror within rows, ds_swizzle across rows, one bpermute across the halves.
// (Neutral at B=1, where one block per CU leaves the kernel latency-bound; -20 % at 2048+ particles,
// where it is VALU-issue bound.)
__device__ __forceinline__ float wave_sum8(const float (&v)[S]) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w[i] = keep + dpp_mov<0xB1, 0xf>(0.f, send);                   // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b1 ? w[2 * j + 1] : w[2 * j], send = b1 ? w[2 * j] : w[2 * j + 1];
        x[j] = keep + dpp_mov<0x4E, 0xf>(0.f, send);                   // quad_perm [2,3,0,1]
    }
    const float keep = b2 ? x[1] : x[0], send = b2 ? x[0] : x[1];
    float recv = dpp_mov<0x104, 0x5>(0.f, send);                       // row_shl:4 into lanes 0-3, 8-11
    recv = dpp_mov<0x114, 0xa>(recv, send);                            // row_shr:4 into lanes 4-7, 12-15
    float y = keep + recv;
    y += dpp_mov<0x128, 0xf>(0.f, y);                                  // row_ror:8  (lane ^ 8)
    y += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(y), 0x401F));   // lane ^ 16
    y += __shfl_xor(y, 32);
    return y;
}

// sums of 8 per-token values over the 256 threads of the block; red is [S][4 waves].  Lane l returns the block total of
// token l & 7 (one 16-byte LDS read per lane instead of eight; the callers finish the statistic in that lane and hand
// it to the wave with v_readlane, which also leaves means and scales in scalar registers)
__device__ __forceinline__ float block_sum8_dpp(const float (&v)[S], float (*red)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float w = wave_sum8(v);
    __syncthreads();                                   // previous readers of red are done
    if (lane < S) red[lane][wave] = w;
    __syncthreads();
    const float4 r = *reinterpret_cast<const float4*>(red[lane & 7]);
    return (r.x + r.y) + (r.z + r.w);
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// 1/sqrt(v) for v >= eps: v_rsq_f32 (1 ulp) + one Newton step -- 5 instructions where the IEEE 1.0f / sqrtf(v) of the
// compiler is ~30 (denormal scaling, div_scale / div_fmas / div_fixup); every thread needs it 16 times per launch
__device__ __forceinline__ float rsqrt_nr(float v) {
    const float r = __builtin_amdgcn_rsqf(v);
    return r * fmaf(-0.5f * v * r, r, 1.5f);
}

// two-pass LayerNorm statistics (mean, then centred squares) of 8 tokens x 512 channels, 2 channels per thread
__device__ __forceinline__ void ln_stats2(const f2 (&x)[S], float (&mean)[S], float (&rstd)[S], float (*red)[4]) {
    float a[S];
#pragma unroll
    for (int t = 0; t < S; ++t) a[t] = x[t].x + x[t].y;
    const float m = block_sum8_dpp(a, red) * (1.0f / PIPS_DMIX);
#pragma unroll
    for (int t = 0; t < S; ++t) mean[t] = lane_bcast(m, t);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const f2 d = x[t] - (f2){mean[t], mean[t]};
        a[t] = d.x * d.x + d.y * d.y;
    }
    const float r = rsqrt_nr(block_sum8_dpp(a, red) * (1.0f / PIPS_DMIX) + 1e-5f);
#pragma unroll
    for (int t = 0; t < S; ++t) rstd[t] = lane_bcast(r, t);
}

// XN_BF16: the LayerNorm-2 output only feeds the up-projection; in the bf16-operand mode that GEMM rounds it to bf16
// while staging it, so it is stored as bf16 here (same hardware round-to-nearest-even: identical operands, half the
// bytes the 16 column tiles of the GEMM re-read)
#ifdef PIPS_TOKEN_TRACE
// tools/token_trace.py (variant build): shader-clock stamps of block 0 .. 255 at the kernel's phases
__device__ unsigned long long g_token_trace[256 * 8];
// (stamps 0..5: shader clock, s_memtime; 6 / 7: the constant 100 MHz clock, s_memrealtime, at stamps 0 / 5 -- their ratio is the clock the
// kernel actually ran at, and the real-time stamps are comparable across compute units)
#define PIPS_TT(k) if (threadIdx.x == 0 && blockIdx.x < 256) { g_token_trace[blockIdx.x * 8 + (k)] = clock64(); \
    if ((k) == 0) g_token_trace[blockIdx.x * 8 + 6] = wall_clock64(); if ((k) == 5) g_token_trace[blockIdx.x * 8 + 7] = wall_clock64(); }
extern "C" int pips_debug_token_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_token_trace), sizeof(unsigned long long) * 256 * 8);
}
#else
#define PIPS_TT(k)
#endif

// 8-byte store of the token-mix outputs, written through the compute die's L2 (sc1: agent scope).  The kernel leaves 33.5 MB at configs[2]
// (residual stream + LayerNorm-2 output), all of it read next by OTHER dies' L2s -- as dirty lines they are written back when the kernel ends,
// after its last wave; written through they leave while the waves still compute: 21.5 -> 20.6 us per launch, 1.262 -> 1.249 ms per mixer pass
// [measured, profiles/r6_probe_store_policy.txt; the same bit on the two GEMMs' output stores: no change, nt / sc0 sc1 nt: +3 %; in the exact-fp32
// mixer at M = 2048 (8.4 MB per launch) it changes nothing on token_mix_kernel and costs the up-projection 1.7 us: r6_probe_store_policy_fp32.txt].
// PIPS_TM_STORE (variant builds, tools/tm_store_ab.sh): 0 plain, 1 the nontemporal builtin, 2 sc1 (product), 3 sc0 sc1, 4 nt, 5 sc0 sc1 nt.
#ifndef PIPS_TM_STORE
#define PIPS_TM_STORE 2
#endif
__device__ __forceinline__ void tm_store8(void* p, uint2 v) {
    const unsigned long long q = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
#if PIPS_TM_STORE == 0
    *reinterpret_cast<uint2*>(p) = v;
#elif PIPS_TM_STORE == 1
    __builtin_nontemporal_store(q, reinterpret_cast<unsigned long long*>(p));
#elif PIPS_TM_STORE == 2
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
#elif PIPS_TM_STORE == 3
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(q) : "memory");
#elif PIPS_TM_STORE == 4
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(q) : "memory");
#else
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(q) : "memory");
#endif
}

template <bool XN_BF16>
__global__ __launch_bounds__(256) void token_mix_kernel(const float* __restrict__ arena, MixLayerW L,
                                                        float* __restrict__ x, float* __restrict__ xn) {
    __shared__ __attribute__((aligned(16))) float red[S][4];
    __shared__ float wsm[32 * 8 + 32 + 8 * 32 + 8];
    const int tid = threadIdx.x;
    PIPS_TT(0)
    // thread -> channels 2*tid, 2*tid+1 (one 8-byte access per token row).  The particle's tile is requested FIRST: hipcc turns
    // the four small weight copies below into load -> wait -> ds_write one after the other, and with them in front the tile's
    // loads (the long ones) left three L2 round trips late -- in a kernel that is all latency at 256 particles.
    float* xp = x + (size_t)blockIdx.x * S * PIPS_DMIX + 2 * tid;
    float* xnp = xn + (size_t)blockIdx.x * S * PIPS_DMIX + 2 * tid;
    f2 xv[S];
    float mean[S], rstd[S];
#pragma unroll
    for (int t = 0; t < S; ++t) xv[t] = *reinterpret_cast<const f2*>(xp + t * PIPS_DMIX);
    const f2 g1 = *reinterpret_cast<const f2*>(arena + L.ln1g + 2 * tid), be1 = *reinterpret_cast<const f2*>(arena + L.ln1b + 2 * tid);
    const f2 g2 = *reinterpret_cast<const f2*>(arena + L.ln2g + 2 * tid), be2 = *reinterpret_cast<const f2*>(arena + L.ln2b + 2 * tid);
    // stage the tiny token-MLP weights: w0[32][8], b0[32], w3[8][32], b3[8] (all four values requested before the first is stored)
    {
        const float w0v = arena[L.tw0 + tid], w3v = arena[L.tw3 + tid];
        const float b0v = arena[L.tb0 + (tid & 31)], b3v = arena[L.tb3 + (tid & 7)];
        wsm[tid] = w0v;
        wsm[288 + tid] = w3v;
        if (tid < 32) wsm[256 + tid] = b0v;
        if (tid >= 64 && tid < 72) wsm[544 + (tid - 64)] = b3v;
    }
    PIPS_TT(1)
    ln_stats2(xv, mean, rstd, red);          // (its barriers also publish wsm)
    PIPS_TT(2)

    f2 h[S], y[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
        h[t] = (xv[t] - (f2){mean[t], mean[t]}) * (g1 * (f2){rstd[t], rstd[t]}) + be1;
        y[t] = (f2){wsm[544 + t], wsm[544 + t]};
    }
#pragma unroll 4
    for (int j = 0; j < 32; ++j) {
        f2 u = (f2){wsm[256 + j], wsm[256 + j]};
#pragma unroll
        for (int t = 0; t < S; ++t) u = h[t] * wsm[j * 8 + t] + u;
        u = gelu_exact2(u);
#pragma unroll
        for (int t = 0; t < S; ++t) y[t] = u * wsm[288 + t * 32 + j] + y[t];
    }
#pragma unroll
    for (int t = 0; t < S; ++t) y[t] += xv[t];
    PIPS_TT(3)

    ln_stats2(y, mean, rstd, red);
    PIPS_TT(4)
#pragma unroll
    for (int t = 0; t < S; ++t) {
        *reinterpret_cast<f2*>(xp + t * PIPS_DMIX) = y[t];
        const f2 o = (y[t] - (f2){mean[t], mean[t]}) * (g2 * (f2){rstd[t], rstd[t]}) + be2;
        if (XN_BF16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t ob = __builtin_convertvector(o, bf16x2_t);
            reinterpret_cast<unsigned*>(xn)[((size_t)blockIdx.x * S + t) * (PIPS_DMIX / 2) + tid] = *reinterpret_cast<const unsigned*>(&ob);
        } else {
            *reinterpret_cast<f2*>(xnp + t * PIPS_DMIX) = o;
        }
    }
    PIPS_TT(5)
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer step for the bf16-operand mixer (BASELINE configs[2]) with the token MLP on the matrix cores.  Under
// torch.autocast the token-mixing Conv1d layers (nets/pips.py:102-109,117) take bf16 operands like every other Linear, so
// the 8 -> 32 -> 8 MLP per channel runs as three v_mfma_f32_32x32x16_bf16 per 32 channels instead of 512 fp32 FMAs per
// channel (1 456 packed VALU instructions per thread in token_mix_kernel, which is VALU-bound at 2048 particles: 35 us):
//   H[32 hidden][32 ch] = W0[32][8 tok -> K = 16, zero padded] * Xn[tok][ch]        (1 MFMA; N = channels)
//   Y[8 tok -> M = 32][32 ch] = W3[tok][32 hidden] * gelu(H + b0)                    (2 MFMAs of K = 16)
// A lane owns 4 tokens (lanes 0-31: tokens 0-3, lanes 32-63: tokens 4-7) x 4 channels (c0 = 128*wave + 4*(lane & 31)):
// the fp32 loads are float4s, the MFMA for channel slot q takes the lane's four tokens of channel c0+q as its K values, H
// comes back with this lane's channel in all 16 registers (hidden units m(r) = (r&3) + 8(r>>2) + 4*half) -- exactly the K
// values the second product wants from this lane once W3's columns are permuted the same way -- and Y's rows 0-7 are the
// lane's own four tokens again: no cross-lane traffic between the three products.  LayerNorm statistics in one pass
// (sum, sum of squares; the operands are rounded to bf16 anyway), fp32 residual stream.
typedef __bf16 bf16x8_tm __attribute__((ext_vector_type(8)));

// One WAVE per particle (round 3, second cut: a 256-thread block per particle spent its time in four block-wide
// reductions and the memory round trips between them -- 30.5 us at 2048 particles against 35 for the VALU kernel): the
// lane holds 4 tokens x 16 channels (c = 128*g + 4*(lane & 31) + q), the LayerNorm sums stay inside the wave (one DPP
// transpose-reduce, no LDS, no barrier), 16 channel slots x 3 MFMAs.
// Round 4: the hidden units' GELU with the degree-5 exponent polynomial (gelu_bf16out2: the result is rounded to bf16 at once):
// -2.4 % on the bf16 mixer pass.  TWO waves per particle (half the tile and ~120 registers per wave, four waves per SIMD, the
// LayerNorm halves meeting in LDS) was built and measured: +2 % with the 12-register spill of a 128-register budget, the same as
// one wave per particle when held to three waves per SIMD (profiles/r4_probe_token_mix_bf16_variants.txt) -- not kept.
// XB: the residual stream x is bf16 in memory (round 5: what the reference's PreNormResidual holds under autocast, nets/pips.py:93-100;
// 84 -> 50 MB per launch at 2048 particles); all arithmetic stays fp32, the new stream is rounded once (RNE) on the way out.
template <bool XB>
__global__ __launch_bounds__(256, 2) void token_mix_mfma_kernel(const float* __restrict__ arena, MixLayerW L, float* __restrict__ x,
                                                                unsigned* __restrict__ xn, int particles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int p = blockIdx.x * 4 + wave;                                      // (wave-uniform)
    if (p >= particles) return;
    PIPS_TT(0)
    // ---- the particle's tile first: its 16 loads are the long ones (HBM / Infinity Cache), the weights below hit L2
    float* xp = x + ((size_t)p * S + 4 * half) * PIPS_DMIX + 4 * l31;         // + r * 512 + g * 128
    unsigned short* xh = reinterpret_cast<unsigned short*>(x) + ((size_t)p * S + 4 * half) * PIPS_DMIX + 4 * l31;   // the same elements of a bf16 stream
    float xv[4][16];                                                          // [token r][g * 4 + q]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (XB) {
                const uint2 v = *reinterpret_cast<const uint2*>(xh + r * PIPS_DMIX + g * 128);
                xv[r][4 * g] = bf16_lo(v.x); xv[r][4 * g + 1] = bf16_hi(v.x); xv[r][4 * g + 2] = bf16_lo(v.y); xv[r][4 * g + 3] = bf16_hi(v.y);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(xp + r * PIPS_DMIX + g * 128);
                xv[r][4 * g] = v.x; xv[r][4 * g + 1] = v.y; xv[r][4 * g + 2] = v.z; xv[r][4 * g + 3] = v.w;
            }
        }
    // ---- weights as MFMA A fragments
    uint4 a1, a2[2];
    {
        const float* w0 = arena + L.tw0 + l31 * 8 + 4 * half;                 // w0[hidden = l31][token 4*half + i]
        a1 = make_uint4(pack2_bf16(w0[0], w0[1]), pack2_bf16(w0[2], w0[3]), 0u, 0u);
        // w3[token = l31 (< 8)][hidden]: register r of the lane = hidden unit (r & 3) + 8 (r >> 2) + 4 half, i.e. four runs of four
        // consecutive floats.  Four UNCONDITIONAL 16-byte loads, masked afterwards: written as `l31 < 8 ? pack(w3[..]) : 0` hipcc
        // sank every load into its own predicated block -- eight load -> s_waitcnt vmcnt(0) round trips in a row at the head of
        // every wave, before the particle's own tile was even requested
        const float4* w3 = reinterpret_cast<const float4*>(arena + L.tw3 + (l31 & 7) * 32 + 4 * half);
        const float4 wq[4] = {w3[0], w3[2], w3[4], w3[6]};
        const unsigned keep = l31 < 8 ? 0xffffffffu : 0u;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const float4 p0 = wq[2 * kc], p1 = wq[2 * kc + 1];
            a2[kc] = make_uint4(pack2_bf16(p0.x, p0.y) & keep, pack2_bf16(p0.z, p0.w) & keep, pack2_bf16(p1.x, p1.y) & keep,
                                pack2_bf16(p1.z, p1.w) & keep);
        }
    }
    float b0r[16], b3r[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) b0r[r] = arena[L.tb0 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
    for (int r = 0; r < 4; ++r) b3r[r] = arena[L.tb3 + 4 * half + r];

    // per-token mean / rstd of the lane's four tokens: one pass of sums, reduced over the wave
    auto ln_stats = [&](const float (&v)[4][16], float (&mean)[4], float (&rstd)[4]) __attribute__((always_inline)) {
        float s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                a += v[r][k]; b += v[r][k + 1];
                c = fmaf(v[r][k], v[r][k], c); d = fmaf(v[r][k + 1], v[r][k + 1], d);
            }
            s1[r] = a + b; s2[r] = c + d;
        }
        float sa[S], sb[S];
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const bool mine = (t >> 2) == half;
            sa[t] = mine ? s1[t & 3] : 0.f;
            sb[t] = mine ? s2[t & 3] : 0.f;
        }
        const float ta = wave_sum8(sa), tb = wave_sum8(sb);                   // lane l: totals of token l & 7
        const float m = ta * (1.0f / PIPS_DMIX);
        const float var = fmaxf(tb * (1.0f / PIPS_DMIX) - m * m, 0.f);
        const float rs = rsqrt_nr(var + 1e-5f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m_lo = lane_bcast(m, r), m_hi = lane_bcast(m, 4 + r);
            const float r_lo = lane_bcast(rs, r), r_hi = lane_bcast(rs, 4 + r);
            mean[r] = half ? m_hi : m_lo;
            rstd[r] = half ? r_hi : r_lo;
        }
    };
    float mean[4], rstd[4];
    PIPS_TT(1)
    ln_stats(xv, mean, rstd);
    PIPS_TT(2)

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 g1 = *reinterpret_cast<const float4*>(arena + L.ln1g + g * 128 + 4 * l31);
        const float4 be1 = *reinterpret_cast<const float4*>(arena + L.ln1b + g * 128 + 4 * l31);
        const float g1a[4] = {g1.x, g1.y, g1.z, g1.w}, be1a[4] = {be1.x, be1.y, be1.z, be1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            float n[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) n[r] = (xv[r][c] - mean[r]) * (rstd[r] * g1a[q]) + be1a[q];
            const uint4 bx = make_uint4(pack2_bf16(n[0], n[1]), pack2_bf16(n[2], n[3]), 0u, 0u);
            f32x16 h;
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = b0r[r];                       // the MFMA accumulates onto the bias
            h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&a1), *reinterpret_cast<const bf16x8_tm*>(&bx), h, 0, 0, 0);
            unsigned hb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f2 gq = gelu_bf16out2((f2){h[2 * i], h[2 * i + 1]});      // (rounded to bf16 in the next line)
                hb[i] = pack2_bf16(gq.x, gq.y);
            }
            const uint4 k0 = make_uint4(hb[0], hb[1], hb[2], hb[3]), k1 = make_uint4(hb[4], hb[5], hb[6], hb[7]);
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = r < 4 ? b3r[r] : 0.f;
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&a2[0]), *reinterpret_cast<const bf16x8_tm*>(&k0), o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&a2[1]), *reinterpret_cast<const bf16x8_tm*>(&k1), o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r][c] += o[r];                     // residual stream, in place
        }
        // the new residual stream of these 128 channels goes out while the next group is computed (all waves of the
        // launch run in one round, in lock-step: stores held back to the end would queue behind one another)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (XB) {
                // the stored stream is what every later reader sees: LayerNorm 2 below works on the ROUNDED values too
                const uint2 o = make_uint2(pack2_bf16(xv[r][4 * g], xv[r][4 * g + 1]), pack2_bf16(xv[r][4 * g + 2], xv[r][4 * g + 3]));
                tm_store8(xh + r * PIPS_DMIX + g * 128, o);
                xv[r][4 * g] = bf16_lo(o.x); xv[r][4 * g + 1] = bf16_hi(o.x); xv[r][4 * g + 2] = bf16_lo(o.y); xv[r][4 * g + 3] = bf16_hi(o.y);
            } else {
                *reinterpret_cast<float4*>(xp + r * PIPS_DMIX + g * 128) = make_float4(xv[r][4 * g], xv[r][4 * g + 1], xv[r][4 * g + 2], xv[r][4 * g + 3]);
            }
        }
    }
    PIPS_TT(3)
    ln_stats(xv, mean, rstd);
    PIPS_TT(4)
    unsigned* xnp = xn + ((size_t)p * S + 4 * half) * (PIPS_DMIX / 2) + 2 * l31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 g2 = *reinterpret_cast<const float4*>(arena + L.ln2g + g * 128 + 4 * l31);
        const float4 be2 = *reinterpret_cast<const float4*>(arena + L.ln2b + g * 128 + 4 * l31);
        const float g2a[4] = {g2.x, g2.y, g2.z, g2.w}, be2a[4] = {be2.x, be2.y, be2.z, be2.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float n[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) n[q] = (xv[r][4 * g + q] - mean[r]) * (rstd[r] * g2a[q]) + be2a[q];
            tm_store8(xnp + r * (PIPS_DMIX / 2) + g * 64, make_uint2(pack2_bf16(n[0], n[1]), pack2_bf16(n[2], n[3])));
        }
    }
    PIPS_TT(5)
}

// ---------------------------------------------------------------------------------------------------------------------
// Any window length (Pips(S != 8), nets/pips.py:295-301: token MLP S -> 4S -> S).  The same arithmetic as token_mix_kernel with
// the token axis as guarded compile-time-unrolled loops over SMAX >= Sw registers (the guards are wave-uniform), weights of
// run-time size in dynamic LDS, plain shuffle reductions.  Not tuned: no shipped checkpoint has S != 8.
template <int SMAX>
__device__ __forceinline__ void block_sum_any(float (&v)[SMAX], int Sw, float* red /* [4][SMAX] */) {
#pragma unroll
    for (int t = 0; t < SMAX; ++t)
        if (t < Sw) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v[t] += __shfl_xor(v[t], o);
        }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       // previous use of red finished
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < SMAX; ++t)
            if (t < Sw) red[wave * SMAX + t] = v[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < SMAX; ++t)
        if (t < Sw) v[t] = (red[t] + red[SMAX + t]) + (red[2 * SMAX + t] + red[3 * SMAX + t]);
}

// two-pass LayerNorm statistics of Sw tokens x 512 channels, 2 channels per thread
template <int SMAX>
__device__ __forceinline__ void ln_stats_any(const f2 (&x)[SMAX], int Sw, float (&mean)[SMAX], float (&rstd)[SMAX], float* red) {
    float a[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) a[t] = x[t].x + x[t].y;
    block_sum_any<SMAX>(a, Sw, red);
#pragma unroll
    for (int t = 0; t < SMAX; ++t) mean[t] = a[t] * (1.0f / PIPS_DMIX);
#pragma unroll
    for (int t = 0; t < SMAX; ++t) {
        const f2 d = x[t] - (f2){mean[t], mean[t]};
        a[t] = d.x * d.x + d.y * d.y;
    }
    block_sum_any<SMAX>(a, Sw, red);
#pragma unroll
    for (int t = 0; t < SMAX; ++t) rstd[t] = rsqrt_nr(a[t] * (1.0f / PIPS_DMIX) + 1e-5f);
}

template <int SMAX, bool XN_BF16>
__global__ __launch_bounds__(256) void token_mix_any_kernel(const float* __restrict__ arena, MixLayerW L, float* __restrict__ x,
                                                            float* __restrict__ xn, int Sw) {
    extern __shared__ float dyn[];
    const int H4 = 4 * Sw, tid = threadIdx.x;
    float* w0 = dyn;                  // [4S][S]
    float* b0 = w0 + H4 * Sw;         // [4S]
    float* w3 = b0 + H4;              // [S][4S]
    float* b3 = w3 + Sw * H4;         // [S]
    float* red = b3 + Sw;             // [4][SMAX]
    for (int i = tid; i < H4 * Sw; i += 256) { w0[i] = arena[L.tw0 + i]; w3[i] = arena[L.tw3 + i]; }
    for (int i = tid; i < H4; i += 256) b0[i] = arena[L.tb0 + i];
    if (tid < Sw) b3[tid] = arena[L.tb3 + tid];

    float* xp = x + (size_t)blockIdx.x * Sw * PIPS_DMIX + 2 * tid;
    f2 xv[SMAX];
    float mean[SMAX], rstd[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) xv[t] = t < Sw ? *reinterpret_cast<const f2*>(xp + t * PIPS_DMIX) : (f2){0.f, 0.f};
    const f2 g1 = *reinterpret_cast<const f2*>(arena + L.ln1g + 2 * tid), be1 = *reinterpret_cast<const f2*>(arena + L.ln1b + 2 * tid);
    const f2 g2 = *reinterpret_cast<const f2*>(arena + L.ln2g + 2 * tid), be2 = *reinterpret_cast<const f2*>(arena + L.ln2b + 2 * tid);
    ln_stats_any<SMAX>(xv, Sw, mean, rstd, red);          // (its barriers also publish the weights)

    f2 h[SMAX], y[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) {
        h[t] = (xv[t] - (f2){mean[t], mean[t]}) * (g1 * (f2){rstd[t], rstd[t]}) + be1;
        const float bt = t < Sw ? b3[t] : 0.f;
        y[t] = (f2){bt, bt};
    }
    for (int j = 0; j < H4; ++j) {
        f2 u = (f2){b0[j], b0[j]};
#pragma unroll
        for (int t = 0; t < SMAX; ++t)
            if (t < Sw) u = h[t] * w0[j * Sw + t] + u;
        u = gelu_exact2(u);
#pragma unroll
        for (int t = 0; t < SMAX; ++t)
            if (t < Sw) y[t] = u * w3[t * H4 + j] + y[t];
    }
#pragma unroll
    for (int t = 0; t < SMAX; ++t) y[t] += xv[t];

    ln_stats_any<SMAX>(y, Sw, mean, rstd, red);
#pragma unroll
    for (int t = 0; t < SMAX; ++t)
        if (t < Sw) {
            *reinterpret_cast<f2*>(xp + t * PIPS_DMIX) = y[t];
            const f2 o = (y[t] - (f2){mean[t], mean[t]}) * (g2 * (f2){rstd[t], rstd[t]}) + be2;
            if (XN_BF16) {
                reinterpret_cast<unsigned*>(xn)[((size_t)blockIdx.x * Sw + t) * (PIPS_DMIX / 2) + tid] = pack2_bf16(o.x, o.y);
            } else {
                *reinterpret_cast<f2*>(xn + ((size_t)blockIdx.x * Sw + t) * PIPS_DMIX + 2 * tid) = o;
            }
        }
}

template <int SMAX>
static int launch_token_mix_any(const float* arena, const MixLayerW& L, float* x, float* xn, int particles, hipStream_t st,
                                int xn_bf16, int Sw) {
    const size_t lds = (size_t)(8 * Sw * Sw + 5 * Sw + 4 * SMAX) * sizeof(float);
    if (xn_bf16)
        hipLaunchKernelGGL((token_mix_any_kernel<SMAX, true>), dim3(particles), dim3(256), lds, st, arena, L, x, xn, Sw);
    else
        hipLaunchKernelGGL((token_mix_any_kernel<SMAX, false>), dim3(particles), dim3(256), lds, st, arena, L, x, xn, Sw);
    PIPS_CHECK_LAUNCH("token_mix_any_kernel");
    return PIPS_OK;
}

int launch_token_mix(const float* arena, const MixLayerW& L, float* x, float* xn, int particles,
                     hipStream_t st, int xn_bf16, int Sw, int x_bf16) {
    PIPS_CHECK_ARG(!x_bf16 || (xn_bf16 && Sw == PIPS_S), "token_mix: a bf16 residual stream needs the bf16 mixer and S = %d", PIPS_S);
    if (Sw != PIPS_S)                                   // (two instantiations: S <= 16 keeps the register budget it had before PIPS_S_MAX = 32)
        return Sw <= 16 ? launch_token_mix_any<16>(arena, L, x, xn, particles, st, xn_bf16, Sw)
                        : launch_token_mix_any<PIPS_S_MAX>(arena, L, x, xn, particles, st, xn_bf16, Sw);
    if (xn_bf16 && (x_bf16 || PIPS_TUNE("PIPS_TOKEN_MFMA", 1))) {
        // bf16-operand mixer: token MLP on the matrix cores, one wave per particle
        if (x_bf16)
            hipLaunchKernelGGL(token_mix_mfma_kernel<true>, dim3(cdiv(particles, 4)), dim3(256), 0, st, arena, L, x, reinterpret_cast<unsigned*>(xn), particles);
        else
            hipLaunchKernelGGL(token_mix_mfma_kernel<false>, dim3(cdiv(particles, 4)), dim3(256), 0, st, arena, L, x, reinterpret_cast<unsigned*>(xn), particles);
        PIPS_CHECK_LAUNCH("token_mix_mfma_kernel");
        return PIPS_OK;
    }
    if (xn_bf16)
        hipLaunchKernelGGL(token_mix_kernel<true>, dim3(particles), dim3(256), 0, st, arena, L, x, xn);
    else
        hipLaunchKernelGGL(token_mix_kernel<false>, dim3(particles), dim3(256), 0, st, arena, L, x, xn);
    PIPS_CHECK_LAUNCH("token_mix_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------ final LN + mean
// nn.LayerNorm(512) then Reduce('b n c -> b c','mean') (nets/pips.py:120-121).
template <bool XB>           // XB: x is the bf16 residual stream
__global__ __launch_bounds__(256) void ln_mean_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ bta, float* __restrict__ out) {
    __shared__ float red[4][S];
    const float* xp = x + (size_t)blockIdx.x * S * PIPS_DMIX;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + (size_t)blockIdx.x * S * PIPS_DMIX;
    const int c0 = threadIdx.x, c1 = threadIdx.x + 256;
    float x0[S], x1[S], mean[S], rstd[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
        if (XB) { x0[t] = __uint_as_float((unsigned)xh[t * PIPS_DMIX + c0] << 16); x1[t] = __uint_as_float((unsigned)xh[t * PIPS_DMIX + c1] << 16); }
        else { x0[t] = xp[t * PIPS_DMIX + c0]; x1[t] = xp[t * PIPS_DMIX + c1]; }
    }
    ln_stats(x0, x1, mean, rstd, red);
    const float g0 = g[c0], g1 = g[c1], b0 = bta[c0], b1 = bta[c1];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        a0 += (x0[t] - mean[t]) * rstd[t] * g0 + b0;
        a1 += (x1[t] - mean[t]) * rstd[t] * g1 + b1;
    }
    out[(size_t)blockIdx.x * PIPS_DMIX + c0] = a0 * (1.0f / S);
    out[(size_t)blockIdx.x * PIPS_DMIX + c1] = a1 * (1.0f / S);
}

// The same on the bf16 residual stream with one WAVE per particle (BASELINE configs[2]: 2048 particles, where the block form above spent
// 16 us on 2-byte loads and four block-wide reductions): a lane holds 8 tokens x 8 channels -- 16-byte loads, a token row is the wave's 64
// lanes -- and both LayerNorm passes are wave_sum8 transposes: no LDS, no barrier.  [measured] profiles/r6_probe_ln_mean_wave.txt
__global__ __launch_bounds__(256) void ln_mean_wave_kernel(const unsigned short* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ bta, float* __restrict__ out, int particles) {
    const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);      // (wave-uniform)
    if (p >= particles) return;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)p * S * PIPS_DMIX) + lane;
    float v[S][8], a[S], mean[S], rstd[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const uint4 q = xr[t * (PIPS_DMIX / 8)];
        v[t][0] = bf16_lo(q.x); v[t][1] = bf16_hi(q.x); v[t][2] = bf16_lo(q.y); v[t][3] = bf16_hi(q.y);
        v[t][4] = bf16_lo(q.z); v[t][5] = bf16_hi(q.z); v[t][6] = bf16_lo(q.w); v[t][7] = bf16_hi(q.w);
    }
    const float4 g0 = *reinterpret_cast<const float4*>(g + 8 * lane), g1 = *reinterpret_cast<const float4*>(g + 8 * lane + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bta + 8 * lane), b1 = *reinterpret_cast<const float4*>(bta + 8 * lane + 4);
#pragma unroll
    for (int t = 0; t < S; ++t) a[t] = ((v[t][0] + v[t][1]) + (v[t][2] + v[t][3])) + ((v[t][4] + v[t][5]) + (v[t][6] + v[t][7]));
    const float m = wave_sum8(a) * (1.0f / PIPS_DMIX);                                // lane l: token l & 7
#pragma unroll
    for (int t = 0; t < S; ++t) mean[t] = lane_bcast(m, t);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[t][k] - mean[t]; q = fmaf(d, d, q); }
        a[t] = q;
    }
    const float r = rsqrt_nr(wave_sum8(a) * (1.0f / PIPS_DMIX) + 1e-5f);
#pragma unroll
    for (int t = 0; t < S; ++t) rstd[t] = lane_bcast(r, t);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < S; ++t) s = fmaf(v[t][k] - mean[t], rstd[t], s);
        acc[k] = s * (1.0f / S);
    }
    float* op = out + (size_t)p * PIPS_DMIX + 8 * lane;
    *reinterpret_cast<float4*>(op) = make_float4(fmaf(acc[0], g0.x, b0.x), fmaf(acc[1], g0.y, b0.y), fmaf(acc[2], g0.z, b0.z), fmaf(acc[3], g0.w, b0.w));
    *reinterpret_cast<float4*>(op + 4) = make_float4(fmaf(acc[4], g1.x, b1.x), fmaf(acc[5], g1.y, b1.y), fmaf(acc[6], g1.z, b1.z), fmaf(acc[7], g1.w, b1.w));
}

// any window length: nn.LayerNorm(512) per token, mean over the Sw tokens
template <int SMAX>
__global__ __launch_bounds__(256) void ln_mean_any_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ bta, float* __restrict__ out, int Sw) {
    __shared__ float red[4 * SMAX];
    const float* xp = x + (size_t)blockIdx.x * Sw * PIPS_DMIX + 2 * threadIdx.x;
    f2 xv[SMAX];
    float mean[SMAX], rstd[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) xv[t] = t < Sw ? *reinterpret_cast<const f2*>(xp + t * PIPS_DMIX) : (f2){0.f, 0.f};
    ln_stats_any<SMAX>(xv, Sw, mean, rstd, red);
    const f2 gg = *reinterpret_cast<const f2*>(g + 2 * threadIdx.x), bb = *reinterpret_cast<const f2*>(bta + 2 * threadIdx.x);
    f2 a = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < SMAX; ++t)
        if (t < Sw) a += (xv[t] - (f2){mean[t], mean[t]}) * (f2){rstd[t], rstd[t]} * gg + bb;
    *reinterpret_cast<f2*>(out + (size_t)blockIdx.x * PIPS_DMIX + 2 * threadIdx.x) = a * (1.0f / (float)Sw);
}

int launch_ln_mean(const float* x, const float* g, const float* b, float* out, int particles,
                   hipStream_t st, int Sw, int x_bf16) {
    PIPS_CHECK_ARG(!x_bf16 || Sw == PIPS_S, "ln_mean: a bf16 residual stream needs S = %d", PIPS_S);
    if (Sw != PIPS_S) {
        if (Sw <= 16) hipLaunchKernelGGL(ln_mean_any_kernel<16>, dim3(particles), dim3(256), 0, st, x, g, b, out, Sw);
        else hipLaunchKernelGGL(ln_mean_any_kernel<PIPS_S_MAX>, dim3(particles), dim3(256), 0, st, x, g, b, out, Sw);
        PIPS_CHECK_LAUNCH("ln_mean_any_kernel");
        return PIPS_OK;
    }
    if (x_bf16 && PIPS_TUNE("PIPS_LN_MEAN_WAVE", 1))
        hipLaunchKernelGGL(ln_mean_wave_kernel, dim3(cdiv(particles, 4)), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(x), g, b, out, particles);
    else if (x_bf16) hipLaunchKernelGGL(ln_mean_kernel<true>, dim3(particles), dim3(256), 0, st, x, g, b, out);
    else hipLaunchKernelGGL(ln_mean_kernel<false>, dim3(particles), dim3(256), 0, st, x, g, b, out);
    PIPS_CHECK_LAUNCH("ln_mean_kernel");
    return PIPS_OK;
}

// ------------------------------------------------------------------------ state update
// nets/pips.py:525-539: ffeats += GELU(Linear(GroupNorm(1,128)(dfeat))); coords += dcoord;
// frame 0 locked to the query; trajectory written in pixels as (B,S,N,2).  Optionally the
// visibility head Linear(128->1) on the NEW features (:559).  One block per particle.
// Thread = output channel o x ALL 8 rows (128 threads): every weight is fetched once per block (the 256-thread form fetched it in two
// waves, and its 128 four-byte loads per thread were as long as its FMAs: 1 MB through each compute unit's L1 per launch at BASELINE
// configs[2]), two rows per v_pk_fma_f32, the old features requested before the product.  Same operations in the same order as before
// (acc = fma(h, w, acc) over ascending k; LayerNorm by 32 lanes x 4 channels per row): results are bit-identical.
// [measured] profiles/r6_probe_state_update.txt
__global__ __launch_bounds__(128) void state_update_kernel(const float* __restrict__ arena,
                                                           size_t o_ng, size_t o_nb, size_t o_wt, size_t o_b,
                                                           size_t o_wv, size_t o_bv,
                                                           const float* __restrict__ delta,
                                                           float* __restrict__ ffeats, float* __restrict__ coords,
                                                           const float* __restrict__ coords0, int N, float stride,
                                                           float* __restrict__ out_traj, float* __restrict__ out_vis) {
    __shared__ __attribute__((aligned(16))) float hs[C][S];      // normalised dfeat, [k][row]
    __shared__ float vred[2][S];
    const int pn = blockIdx.x, tid = threadIdx.x;
    const int b = pn / N, n = pn - b * N;
    const float* dp = delta + (size_t)pn * PIPS_NOUT;
    const int o = tid;

    // the old features are requested first: their round trip runs under the LayerNorm and the product
    float fold[S];
#pragma unroll
    for (int r = 0; r < S; ++r) fold[r] = ffeats[((size_t)pn * S + r) * C + o];

    // LayerNorm over the 128 delta-feature channels of each of the 8 rows (four rows at a time: 32 lanes x 4 channels per row)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = half * 4 + (tid >> 5), l = tid & 31;
        const float* d = dp + row * (C + 2) + 2 + l * 4;
        float v[4] = {d[0], d[1], d[2], d[3]};
        float sum = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int of = 16; of >= 1; of >>= 1) sum += __shfl_xor(sum, of);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = v[k] - mean; sq += t * t; }
#pragma unroll
        for (int of = 16; of >= 1; of >>= 1) sq += __shfl_xor(sq, of);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = l * 4 + k;
            hs[c][row] = (v[k] - mean) * rstd * arena[o_ng + c] + arena[o_nb + c];
        }
    }
    __syncthreads();

    // Linear 128->128 (weights transposed [k][o]) + GELU + residual
    const float bo = arena[o_b + o];
    f2 a[4] = {{bo, bo}, {bo, bo}, {bo, bo}, {bo, bo}};          // rows (0,1) (2,3) (4,5) (6,7)
    const float* wt = arena + o_wt + o;
#pragma unroll 8
    for (int k = 0; k < C; k += 2) {
        // the two weights of a k pair sit in one register pair; op_sel picks the half that serves BOTH results of a packed FMA (written as
        // (f2){w, w} hipcc copies every weight into a second register first)
        const f2 w2 = {wt[(size_t)k * C], wt[(size_t)(k + 1) * C]};
        const float4 p0 = *reinterpret_cast<const float4*>(&hs[k][0]), p1 = *reinterpret_cast<const float4*>(&hs[k][4]);
        const float4 q0 = *reinterpret_cast<const float4*>(&hs[k + 1][0]), q1 = *reinterpret_cast<const float4*>(&hs[k + 1][4]);
        const f2 hk[4] = {{p0.x, p0.y}, {p0.z, p0.w}, {p1.x, p1.y}, {p1.z, p1.w}};
        const f2 hk1[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(hk[i]), "v"(w2));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(hk1[i]), "v"(w2));
    }
    const float acc[S] = {a[0].x, a[0].y, a[1].x, a[1].y, a[2].x, a[2].y, a[3].x, a[3].y};
    float vis_part[S];
    const float wv = out_vis != nullptr ? arena[o_wv + o] : 0.f;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        const float nf = gelu_exact(acc[r]) + fold[r];
        ffeats[((size_t)pn * S + r) * C + o] = nf;
        vis_part[r] = nf * wv;
    }

    if (tid < S) {
        const int t = tid;
        const size_t ci = ((size_t)pn * S + t) * 2;
        float cx = coords[ci] + dp[t * (C + 2) + 0];
        float cy = coords[ci + 1] + dp[t * (C + 2) + 1];
        if (t == 0) { cx = coords0[ci]; cy = coords0[ci + 1]; }      // lock frame 0 (:535-536)
        coords[ci] = cx; coords[ci + 1] = cy;
        float* ot = out_traj + (((size_t)b * S + t) * N + n) * 2;
        ot[0] = cx * stride; ot[1] = cy * stride;                     // :538
    }

    if (out_vis != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int r = 0; r < S; ++r) vis_part[r] += __shfl_xor(vis_part[r], off);
        if ((tid & 63) == 0)
#pragma unroll
            for (int r = 0; r < S; ++r) vred[tid >> 6][r] = vis_part[r];
        __syncthreads();
        if (tid < S) out_vis[((size_t)b * S + tid) * N + n] = vred[0][tid] + vred[1][tid] + arena[o_bv];
    }
}

// vis_predictor alone (nets/pips.py:421-426,559) on the current features: the iters = 0 forward, where
// the reference returns the visibility logits of the INITIAL features.  One wave per mixer row.
__global__ __launch_bounds__(256) void vis_head_kernel(const float* __restrict__ arena, size_t o_wv, size_t o_bv,
                                                       const float* __restrict__ ffeats, int N, int M, int Sw,
                                                       float* __restrict__ out_vis) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const float* f = ffeats + (size_t)m * C;
    float v = f[lane] * arena[o_wv + lane] + f[lane + 64] * arena[o_wv + lane + 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) {
        const int t = m % Sw, pn = m / Sw, b = pn / N, n = pn - b * N;
        out_vis[((size_t)b * Sw + t) * N + n] = v + arena[o_bv];
    }
}

int launch_vis_head(const float* arena, const float* ffeats, int B, int N, float* out_vis, hipStream_t st, int Sw) {
    const ArenaLayout& A = arena_layout();
    const int M = B * N * Sw;
    hipLaunchKernelGGL(vis_head_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, arena, A.w_vis, A.b_vis, ffeats, N, M, Sw, out_vis);
    PIPS_CHECK_LAUNCH("vis_head_kernel");
    return PIPS_OK;
}

// any window length: the same update, the particle's Sw rows in groups of 8; delta rows are ldd floats apart
__global__ __launch_bounds__(256) void state_update_any_kernel(const float* __restrict__ arena,
                                                               size_t o_ng, size_t o_nb, size_t o_wt, size_t o_b,
                                                               size_t o_wv, size_t o_bv,
                                                               const float* __restrict__ delta, int ldd,
                                                               float* __restrict__ ffeats, float* __restrict__ coords,
                                                               const float* __restrict__ coords0, int N, int Sw, float stride,
                                                               float* __restrict__ out_traj, float* __restrict__ out_vis) {
    __shared__ __attribute__((aligned(16))) float hs[C][8];      // normalised dfeat, [k][row of the group]
    __shared__ float vred[4][4];
    const int pn = blockIdx.x, tid = threadIdx.x;
    const int b = pn / N, n = pn - b * N;
    const float* dp = delta + (size_t)pn * ldd;
    const int o = tid & 127, r0 = (tid >> 7) * 4;
    const float bo = arena[o_b + o];
    const float* wt = arena + o_wt + o;
    const float wv = out_vis != nullptr ? arena[o_wv + o] : 0.f;
    for (int g0 = 0; g0 < Sw; g0 += 8) {
        {   // LayerNorm over the 128 delta-feature channels of rows g0 .. g0+7
            const int row = tid >> 5, l = tid & 31;
            const bool live = g0 + row < Sw;
            const float* d = dp + (g0 + (live ? row : 0)) * (C + 2) + 2 + l * 4;
            float v[4] = {d[0], d[1], d[2], d[3]};
            float sum = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int of = 16; of >= 1; of >>= 1) sum += __shfl_xor(sum, of);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = v[k] - mean; sq += t * t; }
#pragma unroll
            for (int of = 16; of >= 1; of >>= 1) sq += __shfl_xor(sq, of);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = l * 4 + k;
                hs[c][row] = live ? (v[k] - mean) * rstd * arena[o_ng + c] + arena[o_nb + c] : 0.f;
            }
        }
        __syncthreads();
        float acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = bo;
#pragma unroll 8
        for (int k = 0; k < C; ++k) {
            const float w = wt[(size_t)k * C];
            const float4 h = *reinterpret_cast<const float4*>(&hs[k][r0]);
            acc[0] = fmaf(h.x, w, acc[0]); acc[1] = fmaf(h.y, w, acc[1]);
            acc[2] = fmaf(h.z, w, acc[2]); acc[3] = fmaf(h.w, w, acc[3]);
        }
        float vis_part[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vis_part[r] = 0.f;
            if (g0 + r0 + r < Sw) {
                float* fp = ffeats + ((size_t)pn * Sw + g0 + r0 + r) * C + o;
                const float nf = gelu_exact(acc[r]) + *fp;
                *fp = nf;
                vis_part[r] = nf * wv;
            }
        }
        if (out_vis != nullptr) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) vis_part[r] += __shfl_xor(vis_part[r], off);
            if ((tid & 63) == 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) vred[tid >> 6][r] = vis_part[r];
            __syncthreads();
            if (tid < 8 && g0 + tid < Sw) {
                const int grp = tid >> 2, r = tid & 3;                  // rows 0-3: waves 0,1; rows 4-7: waves 2,3
                out_vis[((size_t)b * Sw + g0 + tid) * N + n] = vred[grp * 2][r] + vred[grp * 2 + 1][r] + arena[o_bv];
            }
        }
        __syncthreads();                     // hs / vred are re-used by the next group
    }
    if (tid < Sw) {
        const int t = tid;
        const size_t ci = ((size_t)pn * Sw + t) * 2;
        float cx = coords[ci] + dp[t * (C + 2) + 0];
        float cy = coords[ci + 1] + dp[t * (C + 2) + 1];
        if (t == 0) { cx = coords0[ci]; cy = coords0[ci + 1]; }      // lock frame 0 (:535-536)
        coords[ci] = cx; coords[ci + 1] = cy;
        float* ot = out_traj + (((size_t)b * Sw + t) * N + n) * 2;
        ot[0] = cx * stride; ot[1] = cy * stride;                     // :538
    }
}

int launch_state_update(const float* arena, const float* delta, float* ffeats, float* coords,
                        const float* coords0, int B, int N, float stride, float* out_traj,
                        float* out_vis, hipStream_t st, int Sw) {
    if (Sw != PIPS_S) {
        const ArenaLayout& A = arena_layout(Sw);
        hipLaunchKernelGGL(state_update_any_kernel, dim3(B * N), dim3(256), 0, st, arena,
                           A.norm_g, A.norm_b, A.w_upd_t, A.b_upd, A.w_vis, A.b_vis, delta, A.nout_pad, ffeats, coords,
                           coords0, N, Sw, stride, out_traj, out_vis);
        PIPS_CHECK_LAUNCH("state_update_any_kernel");
        return PIPS_OK;
    }
    const ArenaLayout& A = arena_layout();
    hipLaunchKernelGGL(state_update_kernel, dim3(B * N), dim3(128), 0, st, arena,
                       A.norm_g, A.norm_b, A.w_upd_t, A.b_upd, A.w_vis, A.b_vis, delta, ffeats, coords,
                       coords0, N, stride, out_traj, out_vis);
    PIPS_CHECK_LAUNCH("state_update_kernel");
    return PIPS_OK;
}

}  // namespace pips
