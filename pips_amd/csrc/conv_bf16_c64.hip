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
    if (in_bf16 && out_bf16) PIPS_C64_LAUNCH(true, true)
    else if (!in_bf16 && !out_bf16) PIPS_C64_LAUNCH(false, false)
    else { set_error("conv3x3_c64: mixed fp32 / bf16 maps are not built"); return PIPS_E_ARG; }
#undef PIPS_C64_LAUNCH
    PIPS_CHECK_LAUNCH("conv3x3_c64_bf16_kernel");
    return PIPS_OK;
}

}  // namespace pips
