// LDS-tiled variant of the fused correlation gather (CorrBlock.corr + CorrBlock.sample,
// nets/pips.py:384-398, 355-382) for DENSE query sets (BASELINE config 4: N=4096 on a grid).
//
// The direct kernel (track.hip: mixer_input_kernel) reads every particle's 8x8x128 window from
// L2: 132 KB per particle-update, 17 GB per launch at config 4, i.e. it runs at the L2 roof
// while the compulsory HBM traffic is only the pyramid itself.  Dense queries overlap: a
// 16x16-pixel tile of the level-0 map holds ~68 particles whose windows cover 26x26 pixels, so
// staging that region ONCE per tile in LDS cuts the L2 traffic ~6x and moves the gather from
// the L2 roof towards the HBM roof.
//
//   bin_particles_kernel   one block per frame (b,s): counting sort of the N particles by the
//                          16x16 level-0 tile of their current coordinate -> sorted order +
//                          a work list of (tile, first, count<=64) items.  No global atomics:
//                          the order inside a tile does not influence any output value.
//   gather_tiled_kernel    one block (8 waves) per work item.  For each pyramid level and each
//                          16-channel chunk: the tile's halo region is copied into LDS
//                          (channel-last, 80-byte pixel stride = conflict-free ds_read_b128, two
//                          stages so chunk c+1 is fetched while chunk c is consumed), then every
//                          wave takes particles of the item with LANE = WINDOW PIXEL (64 lanes =
//                          8x8 window): 4 ds_read_b128 + 16 FMAs per chunk against the particle's
//                          feature chunk, staged in LDS once per item and read as a broadcast.  No
//                          cross-lane reduction is needed; the 2x2 blend fetches its four neighbours
//                          with ds_bpermute.  Particles whose coordinate lies outside the map are
//                          binned apart and served straight from global memory by the same kernel.
// Output is identical in meaning to mixer_input_kernel (same taps, same transposed order,
// zeros outside the map); the dot products are summed in a different order (fp32 round-off).
#include "common.h"

#include <cstdlib>

namespace pips {

constexpr int S = PIPS_S;
constexpr int C = PIPS_C;
constexpr int TS = 16;                    // level-0 tile edge in map pixels
constexpr int GMAX = 64;                  // particles per work item
constexpr int HALO_LO = 4, HALO_HI = 5;   // window reach (3 / 4) + 1 px slack for the rounded coordinate
constexpr int RMAX = TS + HALO_LO + HALO_HI;        // 25: largest region edge (level 0)
constexpr int CH = 16;                    // channels per staged chunk
constexpr int PIX_LD = CH + 4;            // floats per staged pixel (16-byte pad)
constexpr int NW = 8;                     // waves per block
constexpr int SLOTS = GMAX / NW;          // particles per wave per item

struct TiledLevels {
    size_t off[PIPS_LEVELS];
    int H[PIPS_LEVELS], W[PIPS_LEVELS];
};

// window geometry of one (particle, level): identical arithmetic to mixer_input_kernel
__device__ __forceinline__ void corr_window(float cxm, float cym, int lvl, int H, int W, int& bx, int& by,
                                            float& wx, float& wy) {
    const float inv = 1.0f / (float)(1 << lvl);
    const float cx = cxm * inv, cy = cym * inv;                                   // coords / 2**i (:373)
    const float gx = __fsub_rn(__fdiv_rn(2.0f * cx, (float)(W - 1)), 1.0f);      // :318
    const float gy = __fsub_rn(__fdiv_rn(2.0f * cy, (float)(H - 1)), 1.0f);      // :319
    const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), (float)(W - 1) / 2.0f);       // grid_sample un-normalise
    const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), (float)(H - 1) / 2.0f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    wx = ix - fx0; wy = iy - fy0;
    // clamp before the int conversion: far-out coordinates must not overflow (their windows are empty anyway)
    bx = (int)fminf(fmaxf(fx0, -1.0e6f), 1.0e6f) - PIPS_RADIUS;
    by = (int)fminf(fmaxf(fy0, -1.0e6f), 1.0e6f) - PIPS_RADIUS;
}

// ---------------------------------------------------------------------------- binning
// order  [F][N]        particle indices n of frame f sorted by tile
// items  [F][max_items] int4 {tile, first, count, kind}   kind 0: staged, 1: direct
// nitems [F]
// A particle whose level-0 coordinate lies inside the map has, at every level, all in-map
// pixels of its window inside its tile's halo region (HALO includes 1 px of slack for the
// rounded coordinate).  Particles outside the map go to one extra bin served without staging.
__global__ __launch_bounds__(256) void bin_particles_kernel(const float* __restrict__ coords, int N, int H0, int W0,
                                                            int tiles_x, int tiles_y, int max_items,
                                                            int* __restrict__ order, int4* __restrict__ items,
                                                            int* __restrict__ nitems) {
    extern __shared__ int sm[];                 // hist[ntiles+1] | cursor[ntiles+1]
    const int ntiles = tiles_x * tiles_y;
    int* hist = sm;
    int* cursor = sm + ntiles + 1;
    const int f = blockIdx.x;                   // frame = b*S + s
    const int b = f / S, s = f - b * S;
    for (int t = threadIdx.x; t <= ntiles; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    auto tile_of = [&](int n) {
        const size_t m = ((size_t)b * N + n) * S + s;
        const float x = coords[m * 2 + 0], y = coords[m * 2 + 1];
        if (!(x >= 0.f && x <= (float)(W0 - 1) && y >= 0.f && y <= (float)(H0 - 1))) return ntiles;   // also NaN
        return min((int)y / TS, tiles_y - 1) * tiles_x + min((int)x / TS, tiles_x - 1);
    };
    for (int n = threadIdx.x; n < N; n += blockDim.x) atomicAdd(&hist[tile_of(n)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0, ni = 0;
        for (int t = 0; t <= ntiles; ++t) {
            const int c = hist[t];
            cursor[t] = off;
            for (int c0 = 0; c0 < c; c0 += GMAX) {
                if (ni < max_items)
                    items[(size_t)f * max_items + ni] = make_int4(t, off + c0, min(GMAX, c - c0), t == ntiles ? 1 : 0);
                ++ni;
            }
            off += c;
        }
        nitems[f] = min(ni, max_items);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int pos = atomicAdd(&cursor[tile_of(n)], 1);
        order[(size_t)f * N + pos] = n;
    }
}

// blend of one particle-level: lane holds the correlation of window pixel (row lane>>3, col lane&7)
__device__ __forceinline__ void blend_store(float dval, float wx, float wy, int lane, float* __restrict__ dst) {
    const int t = lane < 49 ? lane : 0;
    const int ti = t / 7, tj = t - ti * 7;
    const int src = tj * 8 + ti;                                // lane holding D[row tj][col ti]
    const float nw = __shfl(dval, src), ne = __shfl(dval, src + 1);
    const float sw = __shfl(dval, src + 8), se = __shfl(dval, src + 9);
    const float e = 1.0f - wx, so = 1.0f - wy;
    float o = nw * (so * e);
    o += ne * (so * wx);
    o += sw * (wy * e);
    o += se * (wy * wx);
    if (lane < 49) dst[lane] = o;                               // k = ix*7 + iy
}

// feature copy, sin/cos embedding of (dx, dy, t), raw flow, zero pad of one mixer row (one wave)
__device__ __forceinline__ void embed_row(const float* __restrict__ ff, float dx, float dy, float tt, int lane,
                                          float* __restrict__ xrow) {
    if (lane < C / 4) reinterpret_cast<float4*>(xrow)[lane] = reinterpret_cast<const float4*>(ff)[lane];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float val = a == 0 ? dx : (a == 1 ? dy : tt);
        const float freq = (float)(lane >> 1) * 31.25f;
        const float arg = __fmul_rn(val, freq);
        xrow[C + PIPS_NCORR + a * 64 + lane] = (lane & 1) ? cosf(arg) : sinf(arg);   // misc.py:56-63
    }
    if (lane < 3) xrow[C + PIPS_NCORR + 192 + lane] = lane == 0 ? dx : (lane == 1 ? dy : tt);
    else if (lane < 3 + (PIPS_KIN_PAD - PIPS_KIN)) xrow[PIPS_KIN + (lane - 3)] = 0.f;
}

// ---------------------------------------------------------------------------- tiled gather
// LDS: feats[GMAX][C] (the item's particle features, read back as wave-uniform broadcasts)
//      region[2][RMAX*RMAX][PIX_LD] (two stages: chunk c+1 is fetched while chunk c is consumed)
__global__ __launch_bounds__(NW * 64) void gather_tiled_kernel(const float* __restrict__ pyramid, TiledLevels lv,
                                                               int S_, const float* __restrict__ ffeats,
                                                               const float* __restrict__ coords,
                                                               const float* __restrict__ times, int N, int tiles_x,
                                                               int max_items, const int* __restrict__ order,
                                                               const int4* __restrict__ items,
                                                               const int* __restrict__ nitems,
                                                               float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* feats = smem;                                        // [GMAX][C]
    float* region = smem + GMAX * C;                            // [2][RMAX*RMAX*PIX_LD]
    constexpr int RSTAGE = RMAX * RMAX * PIX_LD;
    const int f = blockIdx.y;
    if ((int)blockIdx.x >= nitems[f]) return;
    const int4 it = items[(size_t)f * max_items + blockIdx.x];
    const int tile = it.x, first = it.y, count = it.z;
    const int b = f / S, s = f - b * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = lane & 7, wj = lane >> 3;                    // window column / row of this lane
    const size_t frame_base = (size_t)(b * S_ + s);
    const float scale = sqrtf((float)C);

    if (it.w != 0) {
        // ---- particles outside the map: no staging, lane = window pixel straight from global
        for (int idx = wave; idx < count; idx += NW) {
            const int n = __builtin_amdgcn_readfirstlane(order[(size_t)f * N + first + idx]);
            const size_t m = ((size_t)b * N + n) * S + s;
            const float cx = coords[m * 2 + 0], cy = coords[m * 2 + 1];
            const float* fch = ffeats + m * C;
            for (int lvl = 0; lvl < PIPS_LEVELS; ++lvl) {
                const int H = lv.H[lvl], W = lv.W[lvl];
                int bx, by; float wx, wy;
                corr_window(cx, cy, lvl, H, W, bx, by, wx, wy);
                const int px = bx + wi, py = by + wj;
                const bool inmap = (unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H;
                const float* src = pyramid + lv.off[lvl] +
                                   ((frame_base * H + min(max(py, 0), H - 1)) * W + min(max(px, 0), W - 1)) * C;
                float d = 0.f;
                for (int q = 0; q < C / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(src + q * 4);
                    d = fmaf(v.x, fch[q * 4 + 0], d); d = fmaf(v.y, fch[q * 4 + 1], d);
                    d = fmaf(v.z, fch[q * 4 + 2], d); d = fmaf(v.w, fch[q * 4 + 3], d);
                }
                blend_store((inmap ? d : 0.f) / scale, wx, wy, lane, X + m * PIPS_KIN_PAD + C + lvl * 49);
            }
            embed_row(fch, cx - coords[((size_t)b * N + n) * S * 2 + 0], cy - coords[((size_t)b * N + n) * S * 2 + 1],
                      times[s], lane, X + m * PIPS_KIN_PAD);
        }
        return;
    }

    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    // particles of this wave: slot k <-> item particle wave + k*NW
    int pn_[SLOTS];                                             // b*N + n, or -1
    float cx_[SLOTS], cy_[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int idx = wave + k * NW;
        pn_[k] = -1; cx_[k] = 0.f; cy_[k] = 0.f;
        if (idx < count) {
            const int n = __builtin_amdgcn_readfirstlane(order[(size_t)f * N + first + idx]);   // wave-uniform
            pn_[k] = b * N + n;
            const size_t m = (size_t)pn_[k] * S + s;
            cx_[k] = coords[m * 2 + 0]; cy_[k] = coords[m * 2 + 1];
        }
    }
    // stage the item's particle features once: feats[idx][C]
    for (int e = tid; e < count * (C / 4); e += NW * 64) {
        const int idx = e / (C / 4), c4 = e - idx * (C / 4);
        const int n = order[(size_t)f * N + first + idx];
        reinterpret_cast<float4*>(feats)[e] =
            *reinterpret_cast<const float4*>(ffeats + (((size_t)b * N + n) * S + s) * C + c4 * 4);
    }

    for (int lvl = 0; lvl < PIPS_LEVELS; ++lvl) {
        const int H = lv.H[lvl], W = lv.W[lvl];
        // staged region of this tile at this level (inclusive bounds, clipped to the map)
        const int x0 = max(((tx * TS) >> lvl) - HALO_LO, 0), x1 = min((((tx + 1) * TS - 1) >> lvl) + HALO_HI, W - 1);
        const int y0 = max(((ty * TS) >> lvl) - HALO_LO, 0), y1 = min((((ty + 1) * TS - 1) >> lvl) + HALO_HI, H - 1);
        const int RW = max(x1 - x0 + 1, 1), RH = max(y1 - y0 + 1, 1);
        const bool any = x1 >= x0 && y1 >= y0;
        const int nld = any ? RW * RH * (CH / 4) : 0;           // float4 loads per chunk
        const float* lbase = pyramid + lv.off[lvl] + frame_base * H * W * C;

        int bx_[SLOTS], by_[SLOTS];
        float wx_[SLOTS], wy_[SLOTS], acc[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            corr_window(cx_[k], cy_[k], lvl, H, W, bx_[k], by_[k], wx_[k], wy_[k]);
            acc[k] = 0.f;
        }

        // two-stage pipeline over the 4 channel chunks: global -> registers (chunk c+1) overlaps
        // the LDS reads + FMAs of chunk c; up to RL float4 per thread per chunk
        constexpr int RL = (RMAX * RMAX * (CH / 4) + NW * 64 - 1) / (NW * 64);      // 10
        float4 stg[RL];
#define PIPS_STAGE_LOAD(ch_)                                                                    \
        _Pragma("unroll") for (int r = 0; r < RL; ++r) {                                        \
            const int e = tid + r * NW * 64;                                                    \
            const int ec = min(e, max(nld - 1, 0));                                             \
            const int pix = ec / (CH / 4), cg = ec - pix * (CH / 4);                             \
            const int ry = pix / RW, rx = pix - ry * RW;                                        \
            stg[r] = *reinterpret_cast<const float4*>(                                          \
                lbase + ((size_t)(y0 + ry) * W + (x0 + rx)) * C + (ch_) * CH + cg * 4);         \
        }
#define PIPS_STAGE_STORE(buf_)                                                                  \
        _Pragma("unroll") for (int r = 0; r < RL; ++r) {                                        \
            const int e = tid + r * NW * 64;                                                    \
            if (e < nld) *reinterpret_cast<float4*>(&region[(buf_) * RSTAGE + (e / (CH / 4)) * PIX_LD + (e % (CH / 4)) * 4]) = stg[r]; \
        }
        __syncthreads();                                        // previous level's readers are done (and feats are in)
        PIPS_STAGE_LOAD(0)
        PIPS_STAGE_STORE(0)
        __syncthreads();
        for (int ch = 0; ch < C / CH; ++ch) {
            const int buf = ch & 1;
            if (ch + 1 < C / CH) PIPS_STAGE_LOAD(ch + 1)
            // ---- every wave: its particles, lane = window pixel
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                if (pn_[k] < 0) continue;                       // wave-uniform
                const int px = bx_[k] + wi, py = by_[k] + wj;
                const bool inmap = any && (unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H;
                const int rx = min(max(px - x0, 0), RW - 1), ry = min(max(py - y0, 0), RH - 1);
                const float* pix = &region[buf * RSTAGE + (ry * RW + rx) * PIX_LD];
                const float* fch = &feats[(wave + k * NW) * C + ch * CH];     // same address in every lane: broadcast
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(pix + q * 4);
                    const float4 w = *reinterpret_cast<const float4*>(fch + q * 4);
                    d = fmaf(v.x, w.x, d); d = fmaf(v.y, w.y, d); d = fmaf(v.z, w.z, d); d = fmaf(v.w, w.w, d);
                }
                acc[k] += inmap ? d : 0.f;
            }
            if (ch + 1 < C / CH) PIPS_STAGE_STORE(buf ^ 1)
            __syncthreads();
        }
#undef PIPS_STAGE_LOAD
#undef PIPS_STAGE_STORE

        // ---- blend the 8x8 correlations to the 49 taps (transposed order k = ix*7 + iy)
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (pn_[k] < 0) continue;                           // wave-uniform
            blend_store(acc[k] / scale, wx_[k], wy_[k], lane,                         // corrs / sqrt(C) (:397)
                        X + ((size_t)pn_[k] * S + s) * PIPS_KIN_PAD + C + lvl * 49);
        }
    }

    // ---- per particle: feature copy, sin/cos embedding of (dx, dy, t), raw flow, zero pad
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        if (pn_[k] < 0) continue;
        const size_t m = (size_t)pn_[k] * S + s;
        embed_row(ffeats + m * C, cx_[k] - coords[(size_t)pn_[k] * S * 2 + 0], cy_[k] - coords[(size_t)pn_[k] * S * 2 + 1],
                  times[s], lane, X + m * PIPS_KIN_PAD);
    }
}

// ---------------------------------------------------------------------------- host side
size_t tiled_gather_scratch_bytes(int B, int N, int H8, int W8) {
    const int F = B * S;
    const int ntiles = cdiv(W8, TS) * cdiv(H8, TS);
    const int max_items = ntiles + 1 + N / GMAX + 1;
    return align_up((size_t)F * N * sizeof(int), 256) + align_up((size_t)F * max_items * sizeof(int4), 256) +
           align_up((size_t)F * sizeof(int), 256);
}

// Selection.  Measured on MI355X at BASELINE config 4 (B=4, 720x1280, N=4096 grid): this kernel
// 1.3-1.6 ms per launch against 0.73-0.86 ms for the direct kernel -- its 32 short
// stage->barrier->consume phases per item are bound by global-load latency with one block per
// CU, so the 6x cut in L2 traffic does not show yet.  It therefore stays OPT-IN
// (PIPS_GATHER_TILED=1, or pips_mixer_input_build_tiled) until the chunk pipeline is deepened;
// the direct kernel remains the default everywhere.
bool tiled_gather_wanted(int N, int H8, int W8) {
    static int force = -2;
    if (force == -2) { const char* e = getenv("PIPS_GATHER_TILED"); force = e ? atoi(e) : -1; }
    (void)N; (void)H8; (void)W8;
    return force > 0;
}

int launch_mixer_input_tiled(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B,
                             int S_, const float* ffeats, const float* coords, const float* times, int N,
                             float* X, void* scratch, size_t scratch_bytes, hipStream_t st) {
    const int F = B * S, H8 = lvlH[0], W8 = lvlW[0];
    PIPS_CHECK_ARG(S_ == S, "tiled gather: the map buffer must hold %d frames per clip", S);
    if (scratch_bytes < tiled_gather_scratch_bytes(B, N, H8, W8)) {
        set_error("tiled gather: scratch %zu < %zu bytes", scratch_bytes, tiled_gather_scratch_bytes(B, N, H8, W8));
        return PIPS_E_WORKSPACE;
    }
    const int tiles_x = cdiv(W8, TS), tiles_y = cdiv(H8, TS), ntiles = tiles_x * tiles_y;
    const int max_items = ntiles + 1 + N / GMAX + 1;
    char* p = (char*)scratch;
    int* order = (int*)p; p += align_up((size_t)F * N * sizeof(int), 256);
    int4* items = (int4*)p; p += align_up((size_t)F * max_items * sizeof(int4), 256);
    int* nitems = (int*)p;
    PIPS_CHECK_ARG((size_t)2 * (ntiles + 1) * sizeof(int) <= 64 * 1024, "tiled gather: map too large for the tile histogram");
    hipLaunchKernelGGL(bin_particles_kernel, dim3(F), dim3(256), (size_t)2 * (ntiles + 1) * sizeof(int), st, coords, N, H8,
                       W8, tiles_x, tiles_y, max_items, order, items, nitems);
    PIPS_CHECK_LAUNCH("bin_particles_kernel");
    TiledLevels lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) { lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l]; }
    const size_t lds = ((size_t)GMAX * C + (size_t)2 * RMAX * RMAX * PIX_LD) * sizeof(float);
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute((const void*)gather_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        raised = true;
    }
    hipLaunchKernelGGL(gather_tiled_kernel, dim3(max_items, F), dim3(NW * 64), lds, st, pyramid, lv, S_, ffeats, coords,
                       times, N, tiles_x, max_items, order, items, nitems, X);
    PIPS_CHECK_LAUNCH("gather_tiled_kernel");
    return PIPS_OK;
}

}  // namespace pips
