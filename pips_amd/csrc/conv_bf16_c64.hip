// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels, bf16 MFMA operands (BASELINE config 3: the four
// layer-1 convolutions of BasicEncoder, nets/pips.py:135-136,173-181 -- 64 frames of 184x248 per GPU).
//
// The implicit-GEMM kernel of gemm_bf16.hip re-reads the fp32 map once per filter tap (nine times 747 MB through L2 at
// 9 TB/s: 721 us per layer where the MFMAs need 86).  Here a persistent block keeps the weights of ALL nine taps in LDS
// (bf16, 72 KiB), stages the input patch of a 4-row x 64-column output tile with its one-pixel halo ONCE (fp32 -> bf16
// while staging, zeros outside the image) and feeds every tap from LDS: the map is read once (+ halo), the output
// written once.  One wave per output row of the tile: 64 pixels x 64 channels = 2x2 MFMA tiles of
// v_mfma_f32_32x32x16_bf16, 9 taps x 4 K slices.  LDS rows (a pixel's / an output channel's 64 input channels = 128 B)
// are padded to 144 B: the 16 lanes of a ds_read_b128 group (consecutive pixels / channels) hit 16 different 16-byte
// bank groups.  Output: raw fp32 NHWC + bias, and the pivoted InstanceNorm partials of common.h (one per wave).
//
// IN_BF16 / OUT_BF16 (the bf16 encoder mode keeps every activation in bf16): the input map is bf16 -- optionally the RAW
// output of the producing convolution, whose relu((x - mean) * rstd) (InstanceNorm + ReLU, nets/pips.py:175-176) is then
// applied while the patch is staged (in_norm = that layer's {mean, rstd} per frame and channel; taps outside the image
// stay zero): the normalised map is never written to HBM -- and the output map is written as bf16 (statistics still from
// the fp32 accumulators).
#include "common.h"

#include <type_traits>

namespace pips {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int C64_ROWS = 4, C64_COLS = 64, C64_PIXB = 144;
constexpr int C64_PW = C64_COLS + 2, C64_PH = C64_ROWS + 2;
constexpr int C64_WBYTES = 9 * 64 * C64_PIXB, C64_PBYTES = C64_PH * C64_PW * C64_PIXB;
constexpr int C64_LDS = C64_WBYTES + C64_PBYTES;

template <bool IN_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256) void conv3x3_c64_bf16_kernel(const void* __restrict__ in_v, const unsigned short* __restrict__ wgt,
                                                               const float* __restrict__ bias, void* __restrict__ out_v,
                                                               float* __restrict__ stats, const float* __restrict__ in_norm,
                                                               int F, int H, int W, int tiles_x, int tiles_per_frame) {
    const float* __restrict__ in = reinterpret_cast<const float*>(in_v);                     // !IN_BF16
    const unsigned short* __restrict__ in_h = reinterpret_cast<const unsigned short*>(in_v); // IN_BF16
    float* __restrict__ out = reinterpret_cast<float*>(out_v);
    unsigned short* __restrict__ out_h = reinterpret_cast<unsigned short*>(out_v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;
    char* Pl = smem + C64_WBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // weights [cout][kh][kw][cin] bf16 -> LDS [tap][cout][cin], rows of 144 B
    for (int i = tid; i < 9 * 64 * 8; i += 256) {
        const int row = i >> 3, c = i & 7;                 // row = tap * 64 + cout
        const int tap = row >> 6, co = row & 63;
        const uint4 v = *reinterpret_cast<const uint4*>(wgt + ((size_t)co * 9 + tap) * 64 + c * 8);
        *reinterpret_cast<uint4*>(Wl + row * C64_PIXB + c * 16) = v;
    }

    const int total = F * tiles_per_frame;
    // The patch of tile t+1 is fetched into REGISTERS (13 x 32 B per thread) while tile t is computed, converted and
    // written to LDS between the two barriers at the top of the next iteration: one wave per SIMD leaves nothing else
    // to hide the memory round trip under.
    constexpr int NCH = C64_PH * C64_PW * 8, NIT = (NCH + 255) / 256;
    float4 lo[NIT], hi[IN_BF16 ? 1 : NIT];           // IN_BF16: lo[] carries the 8 bf16 of a chunk as raw bits
    unsigned inside = 0;                              // IN_BF16 + in_norm: which chunks lie inside the image (bit per step)
    int fetched_frame = -1;
    float4 nrm[4];                                    // {mean, rstd} of this thread's 8 channels (c = tid & 7), fetched frame
    auto fetch = [&](int t) {
        const int f = t / tiles_per_frame, tt = t - f * tiles_per_frame;
        const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int y0 = ty * C64_ROWS, x0 = tx * C64_COLS;
        const float* __restrict__ src = in + (size_t)f * H * W * 64;
        const unsigned short* __restrict__ src_h = in_h + (size_t)f * H * W * 64;
        if (IN_BF16 && in_norm != nullptr && f != fetched_frame) {
            const float4* np = reinterpret_cast<const float4*>(in_norm + ((size_t)f * 64 + (tid & 7) * 8) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) nrm[k] = np[k];
            fetched_frame = f;
        }
        inside = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int pix = i >> 3, c = i & 7;
            const int py = pix / C64_PW, px = pix - py * C64_PW;
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            lo[it] = make_float4(0.f, 0.f, 0.f, 0.f);                    // zeros outside the image
            if constexpr (!IN_BF16) hi[it] = lo[it];
            if (i < NCH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                if constexpr (IN_BF16) {
                    lo[it] = *reinterpret_cast<const float4*>(src_h + ((size_t)gy * W + gx) * 64 + c * 8);
                    inside |= 1u << it;
                } else {
                    const float* p = src + ((size_t)gy * W + gx) * 64 + c * 8;
                    lo[it] = *reinterpret_cast<const float4*>(p);
                    hi[it] = *reinterpret_cast<const float4*>(p + 4);
                }
            }
        }
    };
    if ((int)blockIdx.x < total) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int f = t / tiles_per_frame, tt = t - f * tiles_per_frame;
        const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int y0 = ty * C64_ROWS, x0 = tx * C64_COLS;
        __syncthreads();                                   // the previous tile's fragment reads are done (and Wl is written)
        // ---- patch: rows y0-1 .. y0+4, columns x0-1 .. x0+64, 64 channels, fp32 -> bf16 (hardware RNE)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < NCH) {
                if constexpr (IN_BF16) {
                    uint4 q = *reinterpret_cast<const uint4*>(&lo[it]);
                    if (in_norm != nullptr && (inside >> it & 1)) {          // relu((x - mean) * rstd) of the producing layer
                        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                        unsigned o4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            o4[k] = pack2_bf16(fmaxf((bf16_lo(w4[k]) - nrm[k].x) * nrm[k].y, 0.f),
                                               fmaxf((bf16_hi(w4[k]) - nrm[k].z) * nrm[k].w, 0.f));
                        q = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                    }
                    *reinterpret_cast<uint4*>(Pl + (i >> 3) * C64_PIXB + (i & 7) * 16) = q;
                } else {
                    const f32x8 v = {lo[it].x, lo[it].y, lo[it].z, lo[it].w, hi[it].x, hi[it].y, hi[it].z, hi[it].w};
                    const bf16x8 b = __builtin_convertvector(v, bf16x8);
                    *reinterpret_cast<uint4*>(Pl + (i >> 3) * C64_PIXB + (i & 7) * 16) = *reinterpret_cast<const uint4*>(&b);
                }
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x);

        // ---- 9 taps x 4 K slices of 16 channels; wave = output row y0 + wave
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const char* pa = Pl + (wave * C64_PW + l31) * C64_PIXB + half * 16;      // patch pixel (wave + dy, l31 + dx), K slot half
        const char* pb = Wl + l31 * C64_PIXB + half * 16;                       // weight row (tap, l31)
        // 36 groups (tap, K slice) of 4 fragment reads + 4 MFMAs; the next group's fragments are requested before this
        // group's MFMAs are issued (one wave per SIMD: nobody else covers the LDS latency)
        uint4 fa[2][2], fb[2][2];
#define PIPS_C64_LOAD(buf_, g_)                                                                                        \
        {                                                                                                              \
            constexpr int tap_ = (g_) / 4, kk_ = (g_) % 4, dy_ = tap_ / 3, dx_ = tap_ % 3;                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                fa[buf_][i] = *reinterpret_cast<const uint4*>(pa + ((dy_ * C64_PW + dx_) + i * 32) * C64_PIXB + kk_ * 32); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                fb[buf_][j] = *reinterpret_cast<const uint4*>(pb + (tap_ * 64 + j * 32) * C64_PIXB + kk_ * 32);        \
        }
#define PIPS_C64_MFMA(buf_)                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[buf_][i]),    \
                                                                    *reinterpret_cast<const bf16x8*>(&fb[buf_][j]), acc[i][j], 0, 0, 0);
#define PIPS_C64_STEP(g_)                                                                                              \
        if constexpr ((g_) + 1 < 36) PIPS_C64_LOAD(((g_) + 1) & 1, (g_) + 1)                                           \
        PIPS_C64_MFMA((g_) & 1)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);
#define PIPS_C64_STEP4(g_) PIPS_C64_STEP(g_) PIPS_C64_STEP((g_) + 1) PIPS_C64_STEP((g_) + 2) PIPS_C64_STEP((g_) + 3)
        PIPS_C64_LOAD(0, 0)
        PIPS_C64_STEP4(0) PIPS_C64_STEP4(4) PIPS_C64_STEP4(8) PIPS_C64_STEP4(12) PIPS_C64_STEP4(16) PIPS_C64_STEP4(20)
        PIPS_C64_STEP4(24) PIPS_C64_STEP4(28) PIPS_C64_STEP4(32)
#undef PIPS_C64_STEP4
#undef PIPS_C64_STEP
#undef PIPS_C64_MFMA
#undef PIPS_C64_LOAD

        // ---- epilogue: C orientation -- lane = output channel j*32 + l31, register r = pixel i*32 + (r&3) + 8*(r>>2) + 4*half
        const int y = y0 + wave;
        const bool row_ok = y < H;
        const int nvalid = row_ok ? min(C64_COLS, W - x0) : 0;
        const size_t obase = (((size_t)f * H + (row_ok ? y : 0)) * W + x0) * 64;
        float* __restrict__ orow = out + obase;
        unsigned short* __restrict__ orow_h = out_h + obase;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + l31;
            const float bv = bias != nullptr ? bias[col] : 0.f;
            const float pivot = __shfl(acc[0][j][0] + bv, l31);              // the wave's first pixel (lanes of half 0, r = 0)
            float cs = 0.f, cq = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float vv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = acc[i][j][r] + bv;
                    vv[r] = v;
                    if (px < nvalid) {
                        if (!OUT_BF16) orow[(size_t)px * 64 + col] = v;
                        const float d = v - pivot;
                        cs += d;
                        cq += d * d;
                    }
                }
                if (OUT_BF16)
                    store_c_tile_bf16(vv, l31, half, [&](int px) -> unsigned short* {
                        return i * 32 + px < nvalid ? orow_h + (size_t)(i * 32 + px) * 64 + j * 32 : nullptr;
                    });
            }
            if (stats != nullptr) store_conv_partial(stats, f, tiles_per_frame * 4, tt * 4 + wave, 64, col, half, cs, cq, pivot, nvalid);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong form for bf16 maps (the bf16 encoder's layer 1: 373 MB in + 373 MB out per layer at config 3, where the
// kernel above spent 21k clocks per tile of which its MFMAs need 4.6k -- one wave per SIMD runs fetch wait, staging,
// MFMAs and epilogue strictly one after the other).  Here a block holds TWO groups of four waves, each with its own patch
// buffer and its own sequence of 4-row x 32-column tiles, half a step apart: while group A runs the 72 MFMAs of its tile
// (and has the next tile's patch in flight into registers), group B -- the other wave of every SIMD -- writes out the
// tile it has just finished and stages the next patch (normalise-on-load, VALU + LDS writes); one block-wide barrier per
// half step, then the roles swap.  MFMA pipe and VALU / memory pipes of a SIMD are busy at the same time; the weights
// (72 KiB) are shared by both groups.  LDS: 81 KiB weights + 2 x 28.7 KiB patches.
// The two groups run separately instantiated, branch-free steady-state loops (a shared state machine made hipcc keep the
// tile state in vector registers and copy ~100 registers at every merge); all tile arithmetic is scalar, per-lane address
// offsets are computed once, and interior tiles take paths without bounds checks (first cut: 3000 clocks for the seven
// loads of a fetch, most of them quarter-rate integer multiplies).
constexpr int PP_ROWS = 4, PP_COLS = 32;
constexpr int PP_PW = PP_COLS + 2, PP_PH = PP_ROWS + 2;
constexpr int PP_PBYTES = PP_PH * PP_PW * C64_PIXB;
constexpr int PP_LDS = C64_WBYTES + 2 * PP_PBYTES;
constexpr int PP_NCH = PP_PH * PP_PW * 8, PP_NIT = (PP_NCH + 255) / 256;

typedef short s16x2 __attribute__((ext_vector_type(2)));

#ifdef PIPS_PP_TRACE
// tools/conv_pp_trace.py: s_memtime stamps of blocks 0..7, per group and half step: {after barrier, mid, end, kind}
__device__ unsigned long long* g_pp_trace = nullptr;
int g_pp_dbg_host = 0;             // ablation bits: 1 no epilogue stores, 2 no staging writes, 4 no fetch loads, 8 no MFMA phase, 16 no epilogue at all,
                                   // 32 no stamps
#define PIPS_PP_DBG(bit_) (dbg & (bit_))
#define PIPS_PP_T(h_, slot_, val_)                                                                      \
    if (!(dbg & 32) && g_pp_trace != nullptr && blockIdx.x < 8 && (h_) < 64 && (threadIdx.x & 255) == 0) \
        g_pp_trace[(((size_t)blockIdx.x * 2 + GRP) * 64 + (h_)) * 4 + (slot_)] = (val_);
#define PIPS_PP_DBG_HOST g_pp_dbg_host
#else
#define PIPS_PP_T(h_, slot_, val_)
#define PIPS_PP_DBG(bit_) false
#define PIPS_PP_DBG_HOST 0
#endif

__global__ __launch_bounds__(512) void conv3x3_c64_pp_kernel(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wgt,
                                                             const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                             float* __restrict__ stats, const float* __restrict__ in_norm,
                                                             int F, int H, int W, int tiles_x, int tiles_per_frame,
                                                             unsigned magic_tpf, unsigned magic_tx, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    (void)dbg;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: all tile arithmetic stays on the SALU
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
    const int l31 = lane & 31, half = lane >> 5;
    char* Wl = smem;
    char* Pl = smem + C64_WBYTES + grp * PP_PBYTES;

    for (int i = tid; i < 9 * 64 * 8; i += 512) {           // weights [cout][kh][kw][cin] -> LDS [tap][cout][cin], rows of 144 B
        const int row = i >> 3, c = i & 7;
        const int tap = row >> 6, co = row & 63;
        *reinterpret_cast<uint4*>(Wl + row * C64_PIXB + c * 16) = *reinterpret_cast<const uint4*>(wgt + ((size_t)co * 9 + tap) * 64 + c * 8);
    }

    const int total = F * tiles_per_frame, G = gridDim.x;
    // (the half-step barriers below order LDS traffic only -- lds_barrier(): the patch prefetch and the epilogue's stores
    // stay in flight across them)
    // per-lane constants of the seven fetch / staging steps: chunk i = gtid + 256*it = (patch pixel, 8-channel chunk)
    unsigned goff[PP_NIT], pyx[PP_NIT];                 // byte offset from the patch's first pixel (y0-1, x0-1); py | px << 8
    const int chunk = gtid & 7;
#pragma unroll
    for (int it = 0; it < PP_NIT; ++it) {
        const int pix = (gtid >> 3) + it * 32;
        const int py = pix / PP_PW, px = pix - py * PP_PW;
        goff[it] = (unsigned)((py * W + px) * 128 + chunk * 16);
        pyx[it] = (unsigned)(py | px << 8);
    }
    char* lds_dst = Pl + (gtid >> 3) * C64_PIXB + chunk * 16;          // + it * 32 * 144
    const bool last_ok = gtid + (PP_NIT - 1) * 256 < PP_NCH;

    uint4 fr[PP_NIT];                                   // the fetched patch: 8 bf16 (one pixel's 8-channel chunk) per step
    unsigned inside = 0;                                // edge tiles: which steps lie inside the image
    bool fr_interior = true;                            // (scalar) the fetched patch lies wholly inside the image
    int norm_frame = -1;
    float4 nraw[4];                                     // {mean, rstd} x 2 of this thread's 8 channels, frame norm_frame
    // t -> (frame, tile in frame, first row, first column); divisions by multiplication with ceil(2^32 / d) (exact for
    // t * d < 2^32, which the launcher checks)
    auto tile_xy = [&](int t, int& f, int& tt, int& y0, int& x0) __attribute__((always_inline)) {
        f = (int)__umulhi((unsigned)t, magic_tpf); tt = t - f * tiles_per_frame;
        const int ty = (int)__umulhi((unsigned)tt, magic_tx), tx = tt - ty * tiles_x;
        y0 = ty * PP_ROWS; x0 = tx * PP_COLS;
    };
    auto fetch = [&](int t) __attribute__((always_inline)) {
        if (PIPS_PP_DBG(128)) return;
        int f, tt, y0, x0;
        tile_xy(t, f, tt, y0, x0);
        if (in_norm != nullptr && f != norm_frame) {
            const float4* np = reinterpret_cast<const float4*>(in_norm + ((size_t)f * 64 + chunk * 8) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) nraw[k] = np[k];
            norm_frame = f;
        }
        // the patch's first pixel (y0-1, x0-1): possibly outside the map -- only lanes inside it are loaded
        const char* sbase = reinterpret_cast<const char*>(in) + (((long)f * H + (y0 - 1)) * W + (x0 - 1)) * 128;
        fr_interior = y0 >= 1 && y0 + PP_ROWS + 1 <= H && x0 >= 1 && x0 + PP_COLS + 1 <= W;
        if (PIPS_PP_DBG(4)) return;
        if (fr_interior) {
#pragma unroll
            for (int it = 0; it < PP_NIT; ++it)
                if (it + 1 < PP_NIT || last_ok) fr[it] = *reinterpret_cast<const uint4*>(sbase + goff[it]);
        } else {
            inside = 0;
#pragma unroll
            for (int it = 0; it < PP_NIT; ++it) {
                const int gy = y0 - 1 + (int)(pyx[it] & 0xff), gx = x0 - 1 + (int)(pyx[it] >> 8);
                fr[it] = make_uint4(0, 0, 0, 0);                             // zeros outside the image
                if ((it + 1 < PP_NIT || last_ok) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                    fr[it] = *reinterpret_cast<const uint4*>(sbase + goff[it]);
                    inside |= 1u << it;
                }
            }
        }
    };
    auto stage = [&]() __attribute__((always_inline)) {
        if (PIPS_PP_DBG(64)) return;
        // Plain (unpacked) fp32 ops on purpose: this code runs beside the other group's MFMAs, where a v_pk_*_f32 costs ~20
        // clocks more than the two scalar ops it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the
        // file is built with -fno-slp-vectorize so that hipcc does not re-pack them.
        float nsc[8], nsh[8];                           // rstd and -mean * rstd
        if (in_norm != nullptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nsc[2 * k] = nraw[k].y; nsc[2 * k + 1] = nraw[k].w;
                nsh[2 * k] = -nraw[k].x * nraw[k].y; nsh[2 * k + 1] = -nraw[k].z * nraw[k].w;
            }
        }
#pragma unroll
        for (int it = 0; it < PP_NIT; ++it) {
            if (it + 1 < PP_NIT || last_ok) {
                uint4 q = fr[it];
                if (in_norm != nullptr) {
                    // relu((x - mean) * rstd) of the producing layer as max(x * rstd + (-mean * rstd), 0), rounded to bf16
                    const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                    unsigned o4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = fmaxf(fmaf(bf16_lo(w4[k]), nsc[2 * k], nsh[2 * k]), 0.f);
                        const float hi = fmaxf(fmaf(bf16_hi(w4[k]), nsc[2 * k + 1], nsh[2 * k + 1]), 0.f);
                        o4[k] = pack2_bf16(lo, hi);
                    }
                    const bool keep = fr_interior || (inside >> it & 1);     // taps outside the image stay zero
                    q = keep ? make_uint4(o4[0], o4[1], o4[2], o4[3]) : make_uint4(0, 0, 0, 0);
                }
                if (!PIPS_PP_DBG(2)) *reinterpret_cast<uint4*>(lds_dst + it * (32 * C64_PIXB)) = q;
            }
        }
    };
    f32x16 acc[2];
    const float bias0 = bias != nullptr ? bias[l31] : 0.f, bias1 = bias != nullptr ? bias[32 + l31] : 0.f;
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = j ? bias1 : bias0;      // the MFMAs accumulate onto the bias
        if (PIPS_PP_DBG(8)) return;
        const char* pa = Pl + (gw * PP_PW + l31) * C64_PIXB + half * 16;        // patch pixel (gw + dy, l31 + dx), K slot half
        const char* pb = Wl + l31 * C64_PIXB + half * 16;                       // weight row (tap, l31)
        // fragments are requested PP_AHEAD steps ahead: a step has only two MFMAs (64 clocks) to cover an LDS round trip
        // that takes 150-300 clocks while the other group's staging writes share the LDS
        constexpr int PP_AHEAD = 3, PP_NBUF = 4;
        uint4 fa[PP_NBUF], fb[PP_NBUF][2];
#define PIPS_PP_LOAD(g_)                                                                                               \
        {                                                                                                              \
            constexpr int tap_ = (g_) / 4, kk_ = (g_) % 4, dy_ = tap_ / 3, dx_ = tap_ % 3, buf_ = (g_) % PP_NBUF;       \
            fa[buf_] = *reinterpret_cast<const uint4*>(pa + (dy_ * PP_PW + dx_) * C64_PIXB + kk_ * 32);                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                fb[buf_][j] = *reinterpret_cast<const uint4*>(pb + (tap_ * 64 + j * 32) * C64_PIXB + kk_ * 32);        \
        }
#define PIPS_PP_STEP(g_)                                                                                               \
        if constexpr ((g_) + PP_AHEAD < 36) PIPS_PP_LOAD((g_) + PP_AHEAD)                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[(g_) % PP_NBUF]),    \
                                                             *reinterpret_cast<const bf16x8*>(&fb[(g_) % PP_NBUF][j]), acc[j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
#define PIPS_PP_STEP4(g_) PIPS_PP_STEP(g_) PIPS_PP_STEP((g_) + 1) PIPS_PP_STEP((g_) + 2) PIPS_PP_STEP((g_) + 3)
        PIPS_PP_LOAD(0) PIPS_PP_LOAD(1) PIPS_PP_LOAD(2)
        __builtin_amdgcn_sched_barrier(0);
        PIPS_PP_STEP4(0) PIPS_PP_STEP4(4) PIPS_PP_STEP4(8) PIPS_PP_STEP4(12) PIPS_PP_STEP4(16) PIPS_PP_STEP4(20)
        PIPS_PP_STEP4(24) PIPS_PP_STEP4(28) PIPS_PP_STEP4(32)
#undef PIPS_PP_STEP4
#undef PIPS_PP_STEP
#undef PIPS_PP_LOAD
    };
    // C orientation: lane = output channel j*32 + l31, register r = pixel (r&3) + 8*(r>>2) + 4*half of the wave's row
    const bool odd = l31 & 1;
    const unsigned perm_sel = odd ? 0x03020706u : 0x05040100u;
    const unsigned lane_out = (unsigned)(((4 * half + (odd ? 1 : 0)) * 64 + (l31 & ~1)) * 2);      // bytes: first pixel / channel pair of the lane
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        int f, tt, y0, x0;
        tile_xy(t, f, tt, y0, x0);
        const int y = y0 + gw;
        const bool row_ok = y < H;
        const int nvalid = row_ok ? min(PP_COLS, W - x0) : 0;
        unsigned short* __restrict__ orow = out + (((size_t)f * H + (row_ok ? y : 0)) * W + x0) * 64;
        if (PIPS_PP_DBG(16)) return;
        if (nvalid == PP_COLS) {
            // full row segment: a lane packs its channel's pixel PAIR (r, r+1), trades the dword with the neighbouring lane
            // (channel n ^ 1) and keeps, by one byte permute, {n, n+1} of pixel r (even lanes) or {n-1, n} of pixel r+1 (odd
            // lanes): cvt + dpp + perm + one dword store at an immediate offset per two values; statistics on packed pairs
            // Statistics: sums of x and x^2 about pivot 0, four independent chains per sum (the issue of a lone wave is
            // latency-bound); the bf16 mode's maps carry 2^-9 rounding anyway, the pivoted form is kept on the edge path.
            char* base = reinterpret_cast<char*>(orow) + lane_out;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float cs4[4] = {0.f, 0.f, 0.f, 0.f}, cq4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r0 = 2 * rr;
                    const float va = acc[j][r0], vb = acc[j][r0 + 1];
                    cs4[rr & 1] += va; cs4[2 + (rr & 1)] += vb;
                    cq4[rr & 1] = fmaf(va, va, cq4[rr & 1]); cq4[2 + (rr & 1)] = fmaf(vb, vb, cq4[2 + (rr & 1)]);
                    const unsigned own = pack2_bf16(va, vb);
                    const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);
                    const unsigned o = __builtin_amdgcn_perm(nbr, own, perm_sel);
                    if (!PIPS_PP_DBG(1)) *reinterpret_cast<unsigned*>(base + j * 64 + ((r0 & 3) + 8 * (r0 >> 2)) * 128) = o;
                    else cs4[0] += __uint_as_float(o) * 1e-30f;
                }
                const f2 cs = {cs4[0] + cs4[1], cs4[2] + cs4[3]}, cq = {cq4[0] + cq4[1], cq4[2] + cq4[3]};
                if (stats != nullptr)
                    store_conv_partial(stats, f, tiles_per_frame * 4, tt * 4 + gw, 64, j * 32 + l31, half, cs.x + cs.y, cq.x + cq.y,
                                       0.f, PP_COLS);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + l31;
            const float pivot = __shfl(acc[j][0], l31);                      // the wave's first pixel (lanes of half 0, r = 0)
            float cs = 0.f, cq = 0.f, vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vv[r] = acc[j][r];
                if ((r & 3) + 8 * (r >> 2) + 4 * half < nvalid) {
                    const float d = vv[r] - pivot;
                    cs += d;
                    cq += d * d;
                }
            }
            store_c_tile_bf16(vv, l31, half, [&](int px) -> unsigned short* {
                return px < nvalid ? orow + (size_t)px * 64 + j * 32 : nullptr;
            });
            if (stats != nullptr) store_conv_partial(stats, f, tiles_per_frame * 4, tt * 4 + gw, 64, col, half, cs, cq, pivot, nvalid);
        }
    };

    // This group's k-th tile is blockIdx.x + (2k + GRP) * G.  Group 1 runs half a step behind group 0; both execute
    // 2 * n0 + 2 barriers (n0 = tiles of group 0 >= tiles of group 1).
    const int n0 = (int)blockIdx.x < total ? (total - 1 - (int)blockIdx.x) / (2 * G) + 1 : 0;
    auto run = [&](auto grp_tag) __attribute__((always_inline)) {
        constexpr int GRP = decltype(grp_tag)::value;
        const int first = (int)blockIdx.x + GRP * G;
        const int n_me = first < total ? (total - 1 - first) / (2 * G) + 1 : 0;
        auto tile = [&](int k) __attribute__((always_inline)) { return first + 2 * k * G; };
        if (n_me > 0) fetch(tile(0));
        if (GRP == 0) {
            for (int k = 0; k < n0; ++k) {
                lds_barrier();                                            // write-out + staging half step
                PIPS_PP_T(2 * k, 0, __builtin_amdgcn_s_memtime())
                if (k > 0) epilogue(tile(k - 1));
                PIPS_PP_T(2 * k, 1, __builtin_amdgcn_s_memtime())
                stage();
                PIPS_PP_T(2 * k, 2, __builtin_amdgcn_s_memtime())
                PIPS_PP_T(2 * k, 3, 1ull)
                lds_barrier();                                            // matrix half step (the next patch is requested first)
                PIPS_PP_T(2 * k + 1, 0, __builtin_amdgcn_s_memtime())
                if (k + 1 < n0) fetch(tile(k + 1));
                PIPS_PP_T(2 * k + 1, 1, __builtin_amdgcn_s_memtime())
                compute();
                PIPS_PP_T(2 * k + 1, 2, __builtin_amdgcn_s_memtime())
                PIPS_PP_T(2 * k + 1, 3, 2ull)
            }
            lds_barrier();
            if (n0 > 0) epilogue(tile(n0 - 1));
            lds_barrier();
        } else {
            for (int k = 0; k <= n0; ++k) {                                 // slot pair k: compute tile k-1, then write it out and stage tile k
                lds_barrier();
                PIPS_PP_T(2 * k, 0, __builtin_amdgcn_s_memtime())
                const bool have_prev = k > 0 && k - 1 < n_me;
                if (have_prev) {
                    if (k < n_me) fetch(tile(k));
                    PIPS_PP_T(2 * k, 1, __builtin_amdgcn_s_memtime())
                    compute();
                    PIPS_PP_T(2 * k, 2, __builtin_amdgcn_s_memtime())
                    PIPS_PP_T(2 * k, 3, 2ull)
                }
                lds_barrier();
                PIPS_PP_T(2 * k + 1, 0, __builtin_amdgcn_s_memtime())
                if (have_prev) epilogue(tile(k - 1));
                PIPS_PP_T(2 * k + 1, 1, __builtin_amdgcn_s_memtime())
                if (k < n_me) stage();
                PIPS_PP_T(2 * k + 1, 2, __builtin_amdgcn_s_memtime())
                PIPS_PP_T(2 * k + 1, 3, 1ull)
            }
        }
    };
    if (grp == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// whether the kernel takes a 64 -> 64, 3x3, stride-1 layer on H x W maps (the encoder plans its fused passes with this)
bool conv3x3_c64_takes(int H, int W, int frames) {
    if (!PIPS_TUNE("PIPS_CONV_C64", 1) || W < 48) return false;            // tuning hook: 0 = off
    const int tpf = cdiv(W, C64_COLS) * cdiv(H, C64_ROWS);
    if ((long)tpf * frames < 512) return false;   // small maps: the implicit-GEMM kernel's many small blocks fill the GPU better
    // the callers size the statistics buffers for 2*ceil(Ho*Wo/64)+4 partials per frame (pips_hip.h); this kernel emits
    // one per wave of every 4x64 tile -- more than that bound on degenerate shapes (H=1, W=129): leave those to the
    // implicit-GEMM kernel
    return tpf * 4 <= 2 * cdiv(H * W, 64) + 4;
}

// returns PIPS_OK if taken, 1 if the caller should use the implicit-GEMM kernel
int launch_conv3x3_c64_bf16(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st, int in_bf16, int out_bf16) {
    if (a.Cin != 64 || a.N != 64 || a.KH != 3 || a.KW != 3 || a.cstride != 1 || a.pad != 1 || a.Ho != a.H ||
        a.Wo != a.Win || a.ldc != 64 || !conv3x3_c64_takes(a.Ho, a.Wo, frames))
        return 1;
    const int tiles_x = cdiv(a.Wo, C64_COLS), tiles_y = cdiv(a.Ho, C64_ROWS), tpf = tiles_x * tiles_y;
    const int cus = device_cus();
    if (cus <= 0) {
        set_error("conv3x3_c64: cannot query the device");
        return PIPS_E_LAUNCH;
    }
    if (tiles_m) *tiles_m = tpf * 4;            // partials per frame: one per wave (output row) of every tile
    const long total = (long)tpf * frames;
    const int grid = total < cus ? (int)total : cus;
    PIPS_CHECK_ARG(in_bf16 || a.in_norm == nullptr, "conv3x3_c64: a fused input normalisation needs a bf16 input map");
#define PIPS_C64_LAUNCH(IN_, OUT_)                                                                                      \
    {                                                                                                                   \
        static std::atomic<unsigned long long> raised{0};                                                               \
        const int rc = ensure_dynamic_lds(raised, (const void*)conv3x3_c64_bf16_kernel<IN_, OUT_>, C64_LDS);           \
        if (rc != PIPS_OK) return rc;                                                                                   \
        hipLaunchKernelGGL((conv3x3_c64_bf16_kernel<IN_, OUT_>), dim3(grid), dim3(256), C64_LDS, st, (const void*)a.A,  \
                           reinterpret_cast<const unsigned short*>(a.W), a.bias, (void*)a.C, a.stats, a.in_norm, frames, \
                           a.H, a.Win, tiles_x, tpf);                                                                   \
    }
    if (in_bf16 && out_bf16 && PIPS_TUNE("PIPS_CONV_C64_PP", 1)) {
        // ping-pong kernel: 4 x 32 tiles, one partial per wave -> tiles_pp * 4 parts per frame
        const int tx = cdiv(a.Wo, PP_COLS), tpf_pp = tx * cdiv(a.Ho, PP_ROWS);
        const int cap = a.stats_parts_cap > 0 ? a.stats_parts_cap : 2 * cdiv(a.Ho * a.Wo, 64) + 4;
        if ((a.stats == nullptr || cap >= tpf_pp * 4) && (unsigned long long)tpf_pp * frames * tpf_pp < (1ull << 32)) {
            const unsigned magic_tpf = (unsigned)(((1ull << 32) + tpf_pp - 1) / tpf_pp), magic_tx = (unsigned)(((1ull << 32) + tx - 1) / tx);
            if (tiles_m) *tiles_m = tpf_pp * 4;
            static std::atomic<unsigned long long> raised_pp{0};
            const int rc = ensure_dynamic_lds(raised_pp, (const void*)conv3x3_c64_pp_kernel, PP_LDS);
            if (rc != PIPS_OK) return rc;
            const long total_pp = (long)tpf_pp * frames;
            const int grid_pp = total_pp < cus ? (int)total_pp : cus;
            hipLaunchKernelGGL(conv3x3_c64_pp_kernel, dim3(grid_pp), dim3(512), PP_LDS, st, (const unsigned short*)a.A,
                               reinterpret_cast<const unsigned short*>(a.W), a.bias, (unsigned short*)a.C, a.stats, a.in_norm,
                               frames, a.H, a.Win, tx, tpf_pp, magic_tpf, magic_tx, PIPS_PP_DBG_HOST);
            PIPS_CHECK_LAUNCH("conv3x3_c64_pp_kernel");
            return PIPS_OK;
        }
    }
    if (in_bf16 && out_bf16) PIPS_C64_LAUNCH(true, true)
    else if (!in_bf16 && !out_bf16) PIPS_C64_LAUNCH(false, false)
    else { set_error("conv3x3_c64: mixed fp32 / bf16 maps are not built"); return PIPS_E_ARG; }
#undef PIPS_C64_LAUNCH
    PIPS_CHECK_LAUNCH("conv3x3_c64_bf16_kernel");
    return PIPS_OK;
}

}  // namespace pips

#ifdef PIPS_PP_TRACE
extern "C" int pips_pp_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_pp_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
extern "C" int pips_pp_dbg(int bits) { pips::g_pp_dbg_host = bits; return 0; }
#endif
