// LDS-tiled fused correlation gather (CorrBlock.corr + CorrBlock.sample, nets/pips.py:384-398,
// 355-382) for DENSE query sets (BASELINE config 4: N=4096 on a grid over 720x1280).
//
// The direct kernel (track.hip: mixer_input_kernel) reads every particle's 4 x (8x8 px x 128 ch)
// windows through the vector L1: 132 KB per particle-update, 17 GB per launch at config 4 -- it runs
// at the L1/L2 roof although the compulsory HBM traffic is only the pyramid itself (0.31 GB).  Dense
// queries overlap, so here each 16x16-pixel tile of the level-0 map (with its halo, at all four levels)
// is staged in LDS ONCE and serves every particle that lives in it (~68 at config 4).
//
//   bin_particles_kernel   one block per frame (b,s): counting sort of the N particles by the tile of
//                          floor(ix), floor(iy) (the level-0 window anchor, computed with the reference's
//                          own un-normalisation arithmetic so the tile test is exact) -> sorted order + a
//                          work list of (tile, first, count <= GMAX) items.  Particles whose anchor is
//                          outside the map form one extra bin served straight from global memory.
//   embed_rows_kernel      feature copy + sin/cos embedding + raw flow + zero pad of every mixer row
//                          (get_3d_embedding, utils/misc.py:44-69; DeltaBlock concat, nets/pips.py:304-308).
//   gather_tiled_kernel    one block (8 waves) per work item, two blocks per CU.  20 phases =
//                          (level 0: 8 chunks of 16 channels; levels 1-3: 4 chunks of 32 channels).  The
//                          tile's region of a (level, chunk) is copied global -> LDS by the waves' own
//                          LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging, no ds_write), double
//                          buffered: chunk p+1 lands while chunk p is consumed, one barrier per phase.
//                          Consumer: LANE = WINDOW PIXEL (64 lanes = the 8x8 integer window of one
//                          particle-level), the particle's feature chunk comes through the scalar cache
//                          into SGPRs (wave-uniform), so a phase costs a lane Q `ds_read_b128` + 4Q FMAs
//                          and no cross-lane reduction.  The LDS image is dense (DMA writes 1 KiB linear
//                          pieces) and XOR-swizzled on the GLOBAL side -- the lane that fetches LDS quad
//                          position j of pixel (rx,ry) reads channel quad j ^ key(rx,ry) -- which makes
//                          the lane=pixel `ds_read_b128` conflict-free for its true 16-lane service groups.
//                          Particles are Morton-sorted inside the item, so consecutive slots of a wave
//                          often share the same window anchor at the coarse levels and re-use the
//                          fragment registers instead of re-reading LDS.  The 2x2 blend of the 8x8
//                          correlations to the 49 taps uses ds_bpermute; the 196 taps of a row are
//                          written together at the end.
// Output is identical in meaning to mixer_input_kernel (same taps, same transposed order, zeros outside
// the map); the dot products are summed in channel order (fp32 round-off differs from the tree sum).
#include "common.h"

#include <cstdlib>

#ifndef PIPS_TILED_REUSE
#define PIPS_TILED_REUSE 1     // re-use the fragment registers between consecutive slots with the same window anchor
#endif

namespace pips {

constexpr int S = PIPS_S;
constexpr int C = PIPS_C;
constexpr int TS = 16;                    // level-0 tile edge in map pixels
constexpr int GMAX = 88;                  // particles per work item
constexpr int NW = 8;                     // waves per block
constexpr int SLOTS = GMAX / NW;          // particle slots per wave
constexpr int SLOT_BYTES = 37 * 1024;     // one stage: >= 17*17 px * 128 B (level 1), whole 1 KiB DMA pieces
constexpr int MAXPIECES = 5;              // DMA pieces per wave per phase: ceil(37 / 8)
constexpr int LDS_MISC = 2048;            // sort keys / sorted particle table
constexpr int LDS_BYTES = 2 * SLOT_BYTES + LDS_MISC;

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

// staged region of tile coordinate t at level lvl along one axis (inclusive, clipped to [0, n-1]).
// Level 0: the anchor floor(ix) of a binned particle lies in [16t, 16t+15] exactly, its window reaches
// -3..+4.  Coarser levels: floor(ix_l) lies in [T-1, T+(16>>l)] with T = (16t)>>l (one pixel of slack
// each side for the independently rounded coordinate), same reach.
__device__ __forceinline__ void region_axis(int t, int lvl, int n, int& lo, int& hi) {
    const int T = (t * TS) >> lvl, w = TS >> lvl;
    lo = max(lvl == 0 ? T - 3 : T - 4, 0);
    hi = min(lvl == 0 ? T + w + 3 : T + w + 4, n - 1);
}

// ---------------------------------------------------------------------------- binning
// order  [F][N]        particle indices n of frame f sorted by tile
// items  [F][max_items] int4 {tile, first, count, kind}   kind 0: staged, 1: direct
// nitems [F]
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
        int bx, by; float wx, wy;
        corr_window(coords[m * 2 + 0], coords[m * 2 + 1], 0, H0, W0, bx, by, wx, wy);
        const int ax = bx + PIPS_RADIUS, ay = by + PIPS_RADIUS;                    // floor(ix), floor(iy)
        if (!((unsigned)ax < (unsigned)W0 && (unsigned)ay < (unsigned)H0) || !(wx == wx) || !(wy == wy))
            return ntiles;                                                         // outside the map / NaN
        return (ay / TS) * tiles_x + ax / TS;
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

// blend of one particle-level: lane holds the correlation of window pixel (row lane>>3, col lane&7);
// returns the tap k = ix*7 + iy of lanes < 49
__device__ __forceinline__ float blend_taps(float dval, float wx, float wy, int lane) {
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
    return o;
}

// ---------------------------------------------------------------------------- embedding rows
// One wave per mixer row m: X[m] = [ffeat 128 | (corr 196: not touched) | sin/cos 192 | dx dy t | 0 x 25]
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ ffeats,
                                                         const float* __restrict__ coords,
                                                         const float* __restrict__ times, int M,
                                                         float* __restrict__ X) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const int s = m % S;
    const size_t m0 = (size_t)(m - s);
    const float dx = coords[(size_t)m * 2 + 0] - coords[m0 * 2 + 0];        // coords - coords[:,0:1] (:518)
    const float dy = coords[(size_t)m * 2 + 1] - coords[m0 * 2 + 1];
    const float tt = times[s];
    float* xrow = X + (size_t)m * PIPS_KIN_PAD;
    const float* ff = ffeats + (size_t)m * C;
    if (lane < C / 4) reinterpret_cast<float4*>(xrow)[lane] = reinterpret_cast<const float4*>(ff)[lane];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float val = a == 0 ? dx : (a == 1 ? dy : tt);
        const float freq = (float)(lane >> 1) * 31.25f;                     // arange(0,64,2)*(1000/64)
        const float arg = __fmul_rn(val, freq);
        xrow[C + PIPS_NCORR + a * 64 + lane] = (lane & 1) ? cosf(arg) : sinf(arg);   // misc.py:56-63
    }
    if (lane < 3) xrow[C + PIPS_NCORR + 192 + lane] = lane == 0 ? dx : (lane == 1 ? dy : tt);
    else if (lane < 3 + (PIPS_KIN_PAD - PIPS_KIN)) xrow[PIPS_KIN + (lane - 3)] = 0.f;
}

// ---------------------------------------------------------------------------- tiled gather
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct LevelGeom {          // wave-uniform description of the staged region of one level
    int x0, y0, RW, RH, W, H;
    int nquads;             // RW*RH*Q 16-byte LDS positions
    size_t base;            // float offset of the (frame, level) map in the pyramid buffer
};

__device__ __forceinline__ LevelGeom level_geom(const TiledLevels& lv, int l, int tx, int ty, size_t frame_base) {
    LevelGeom g;
    int x1, y1;
    g.W = lv.W[l]; g.H = lv.H[l];
    region_axis(tx, l, g.W, g.x0, x1);
    region_axis(ty, l, g.H, g.y0, y1);
    g.RW = max(x1 - g.x0 + 1, 1); g.RH = max(y1 - g.y0 + 1, 1);
    if (x1 < g.x0 || y1 < g.y0) { g.x0 = g.y0 = 0; g.RW = g.RH = 1; }            // (tile beyond this level's map)
    g.nquads = g.RW * g.RH * (l == 0 ? 4 : 8);
    g.base = lv.off[l] + frame_base * g.H * g.W * C;
    return g;
}

// XOR key of region pixel (rx, ry) for Q quads per pixel: makes (pixel index & (16/Q - 1), quad ^ key)
// a bijection of (rx & 3, ry & 3) -> the 16 lanes of a ds_read_b128 service group (4 consecutive x
// in each of 4 consecutive rows) hit 16 different 16-byte bank groups
template <int Q>
__device__ __forceinline__ int swz_key(int rx, int ry) {
#ifdef PIPS_TILED_DBG_NOSWZ
    return 0;
#endif
    return Q == 4 ? (ry & 3) : (((rx >> 1) & 1) | ((ry & 3) << 1));
}

// per-lane global byte offsets (within the (frame, level) map) of the DMA pieces this wave issues
template <int Q>
__device__ __forceinline__ void dma_setup(const LevelGeom& g, int wave, int lane, unsigned (&doff)[MAXPIECES]) {
    const float inv_rw = 1.0f / (float)g.RW;
#pragma unroll
    for (int r = 0; r < MAXPIECES; ++r) {
        const int L = min((wave + r * NW) * 64 + lane, g.nquads - 1);
        const int p = L / Q, j = L - p * Q;
        const int ry = (int)(((float)p + 0.5f) * inv_rw);         // p < 1024: exact
        const int rx = p - ry * g.RW;
        const int q = j ^ swz_key<Q>(rx, ry);
        doff[r] = (unsigned)(((g.y0 + ry) * g.W + (g.x0 + rx)) * (C * 4) + q * 16);
    }
}

// buffer resource over [ptr, ptr + 2 GiB): raw (stride 0) addressing, offsets = soffset (SGPR) + voffset (VGPR);
// keeps every address of the hot loop out of the vector registers
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

__device__ __forceinline__ void dma_issue(__amdgpu_buffer_rsrc_t src, int soff, const unsigned (&doff)[MAXPIECES],
                                          int npieces, int wave, char* lds_slot) {
#pragma unroll
    for (int r = 0; r < MAXPIECES; ++r) {
        const int piece = wave + r * NW;
        if (piece < npieces)                                                     // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lptr_t)(lds_slot + piece * 1024), 16, (int)doff[r], soff, 0, 0);
    }
}

// The particle's feature chunk sits in VGPRs, channel c0+n in lane n of every 16-lane row; the FMA takes
// it as a DPP row broadcast, so it costs no instruction of its own (and no scalar-cache round trip).
// 16 FMAs per asm statement: hipcc pads every statement with an s_nop.
#define PIPS_FD(n, vn) "v_fmac_f32_dpp %0, %1, %" #vn " row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float fma16_bcast(float acc, float f, const float4& a, const float4& b, const float4& c,
                                             const float4& d) {
#ifdef PIPS_TILED_DBG_NODPP
    const int r0 = (threadIdx.x & 63) & 48;
    const float p[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    for (int n = 0; n < 16; ++n) acc = fmaf(__shfl(f, r0 | n), p[n], acc);
    return acc;
#endif
    asm(PIPS_FD(0, 2) PIPS_FD(1, 3) PIPS_FD(2, 4) PIPS_FD(3, 5) PIPS_FD(4, 6) PIPS_FD(5, 7) PIPS_FD(6, 8) PIPS_FD(7, 9)
        PIPS_FD(8, 10) PIPS_FD(9, 11) PIPS_FD(10, 12) PIPS_FD(11, 13) PIPS_FD(12, 14) PIPS_FD(13, 15) PIPS_FD(14, 16)
        PIPS_FD(15, 17)
        : "+v"(acc)
        : "v"(f), "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y),
          "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
    return acc;
}
#undef PIPS_FD

// feature chunk loads of one phase: lane reads channel choff + 16*h + (lane & 15) of each slot's row
template <int Q, int NS>
__device__ __forceinline__ void feats_issue(__amdgpu_buffer_rsrc_t ff, int geo_row, int choff, int lane,
                                            float (&fb)[NS][2]) {
    const int voff = (lane & 15) * 4;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int row = __builtin_amdgcn_readlane(geo_row, k * 4);
#pragma unroll
        for (int h = 0; h < Q / 4; ++h)
            fb[k][h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ff, voff, (row * C + choff + h * 16) * 4, 0));
    }
}

// one (level, chunk) phase of one wave: Q ds_read_b128 + 4Q FMAs per slot
template <int Q, int NS, int SLOT_OFF>
__device__ __forceinline__ void consume(const char* smem, const unsigned (&A)[NS], float (&acc)[NS], unsigned same,
                                        const float (&fb)[NS][2]) {
    float4 v[Q] = {};
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (PIPS_TILED_REUSE == 0 || !((same >> k) & 1u)) {                      // wave-uniform
            unsigned a = A[k];
            asm volatile("" : "+v"(a));          // keep the Q xors here: hoisted out of the chunk loop they cost Q VGPRs per slot
#pragma unroll
            for (int q = 0; q < Q; ++q)
                v[q] = *reinterpret_cast<const float4*>(smem + SLOT_OFF + (a ^ (unsigned)(q << 4)));
        }
        float d = acc[k];
#pragma unroll
        for (int h = 0; h < Q / 4; ++h) d = fma16_bcast(d, fb[k][h], v[h * 4], v[h * 4 + 1], v[h * 4 + 2], v[h * 4 + 3]);
        acc[k] = d;
        __builtin_amdgcn_sched_barrier(0);       // keep the next slot's reads behind these FMAs (one fragment set live)
    }
}

// per-level lane state: LDS byte address of this lane's window pixel (with the swizzle key folded in),
// in-map mask, zeroed accumulators
template <int Q, int NS>
__device__ __forceinline__ void level_setup(const LevelGeom& g, int lvl, int lane, float geo_bx, float geo_by,
                                            unsigned (&A)[NS], float (&acc)[NS], unsigned& inmask) {
    const int wi = lane & 7, wj = lane >> 3;
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        acc[k] = 0.f;
        const int bx = __builtin_amdgcn_readlane(__float_as_int(geo_bx), k * 4 + lvl);
        const int by = __builtin_amdgcn_readlane(__float_as_int(geo_by), k * 4 + lvl);
        const int px = bx + wi, py = by + wj;
        const bool inmap = (unsigned)px < (unsigned)g.W && (unsigned)py < (unsigned)g.H;
        const int rx = min(max(px - g.x0, 0), g.RW - 1), ry = min(max(py - g.y0, 0), g.RH - 1);
        A[k] = (unsigned)((ry * g.RW + rx) * (Q * 16) + (swz_key<Q>(rx, ry) << 4));
        inmask |= inmap ? (1u << k) : 0u;
    }
}

// One level = NCH phases.  Entering it, phase 0's region and feature chunk are already in flight (slot 0 /
// fbA); phase c issues phase c+1's (the next level's first, with Q = 8, after the last chunk) right behind
// the barrier that frees the other slot, then consumes its own.  One barrier per phase; `s_waitcnt vmcnt(0)`
// in front of it covers exactly the previous phase's prefetch.
template <int Q, int NCH, int NS>
__device__ __forceinline__ void run_level(__amdgpu_buffer_rsrc_t map, __amdgpu_buffer_rsrc_t map_next, bool has_next,
                                          const LevelGeom& g, const LevelGeom& gn, char* smem, int wave, int lane,
                                          unsigned same, __amdgpu_buffer_rsrc_t ffeats, int geo_row,
                                          unsigned (&doff)[MAXPIECES], const unsigned (&A)[NS], float (&acc)[NS],
                                          float (&fa)[NS][2], float (&fb)[NS][2]) {
    static_assert(NCH % 2 == 0, "phases come in pairs (static slot / buffer parity)");
    constexpr int CH = Q * 4;
    for (int c = 0; c < NCH; c += 2) {
        // ---- even phase: data in slot 0 / fa; prefetch phase c+1 into slot 1 / fb
        __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0)
        __syncthreads();
        dma_issue(map, (c + 1) * CH * 4, doff, (g.nquads + 63) >> 6, wave, smem + SLOT_BYTES);
        feats_issue<Q, NS>(ffeats, geo_row, (c + 1) * CH, lane, fb);
        consume<Q, NS, 0>(smem, A, acc, same, fa);
        // ---- odd phase: data in slot 1 / fb; prefetch phase c+2 (or the next level's first) into slot 0
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (c + 2 < NCH) {
            dma_issue(map, (c + 2) * CH * 4, doff, (g.nquads + 63) >> 6, wave, smem);
            feats_issue<Q, NS>(ffeats, geo_row, (c + 2) * CH, lane, fa);
        } else if (has_next) {
            dma_setup<8>(gn, wave, lane, doff);
            dma_issue(map_next, 0, doff, (gn.nquads + 63) >> 6, wave, smem);
            feats_issue<8, NS>(ffeats, geo_row, 0, lane, fa);
        }
        consume<Q, NS, SLOT_BYTES>(smem, A, acc, same, fb);
    }
}

// the staged part of one work item for waves that hold NS particle slots each (a wave with fewer particles
// repeats its last one: same values, same destination)
template <int NS>
__device__ __forceinline__ void tile_body(const float* __restrict__ pyramid, const TiledLevels& lv, size_t frame_base,
                                          int tx, int ty, char* smem, int wave, int lane, __amdgpu_buffer_rsrc_t map,
                                          LevelGeom g, unsigned (&doff)[MAXPIECES], __amdgpu_buffer_rsrc_t ffr,
                                          float geo_bx, float geo_by, float geo_wx, float geo_wy, int geo_row,
                                          unsigned long long samebits, float* __restrict__ X) {
    const float scale = sqrtf((float)C);
    unsigned A[NS];
    float acc[NS];
    float fbA[NS][2], fbB[NS][2];
    unsigned inmask;
    feats_issue<4, NS>(ffr, geo_row, 0, lane, fbA);

    auto same_of = [&](int lvl) {
        unsigned sm_ = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) sm_ |= (unsigned)((samebits >> (k * 4 + lvl)) & 1ull) << k;
        return (unsigned)__builtin_amdgcn_readfirstlane(sm_);
    };
    auto finish_level = [&](int lvl) {
        // blend the 8x8 correlations to the 49 taps, k = level*49 + ix*7 + iy (transposed, :379-381)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo_wx), k * 4 + lvl));
            const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo_wy), k * 4 + lvl));
            const int row = __builtin_amdgcn_readlane(geo_row, k * 4);
            const float o = blend_taps(((inmask >> k) & 1u) ? acc[k] / scale : 0.f, wx, wy, lane);
            if (lane < 49) X[(size_t)row * PIPS_KIN_PAD + C + lvl * 49 + lane] = o;
        }
    };

    {
        const LevelGeom gn = level_geom(lv, 1, tx, ty, frame_base);
        const __amdgpu_buffer_rsrc_t mapn = make_rsrc(pyramid + gn.base);
        level_setup<4, NS>(g, 0, lane, geo_bx, geo_by, A, acc, inmask);
        run_level<4, 8, NS>(map, mapn, true, g, gn, smem, wave, lane, same_of(0), ffr, geo_row, doff, A, acc, fbA, fbB);
        finish_level(0);
        g = gn; map = mapn;
    }
    for (int lvl = 1; lvl < PIPS_LEVELS; ++lvl) {
        const LevelGeom gn = level_geom(lv, min(lvl + 1, PIPS_LEVELS - 1), tx, ty, frame_base);
        const __amdgpu_buffer_rsrc_t mapn = make_rsrc(pyramid + gn.base);
        level_setup<8, NS>(g, lvl, lane, geo_bx, geo_by, A, acc, inmask);
        run_level<8, 4, NS>(map, mapn, lvl + 1 < PIPS_LEVELS, g, gn, smem, wave, lane, same_of(lvl), ffr, geo_row, doff, A,
                            acc, fbA, fbB);
        finish_level(lvl);
        g = gn; map = mapn;
    }
}

__global__ __launch_bounds__(NW * 64, 4) void gather_tiled_kernel(const float* __restrict__ pyramid, TiledLevels lv,
                                                                  int S_, const float* __restrict__ ffeats,
                                                                  const float* __restrict__ coords, int N,
                                                                  int tiles_x, int max_items, int F,
                                                                  const int* __restrict__ order,
                                                                  const int4* __restrict__ items,
                                                                  const int* __restrict__ nitems,
                                                                  float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // block -> (frame, item): block id mod 8 is the XCD (observed dispatch order), so XCD x works through
    // frames x, x+8, ... one after another and a frame's tiles share that XCD's L2 for their halos
    int f, item;
    {
        const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
        f = xcd + 8 * (j / max_items);
        item = j - (j / max_items) * max_items;
        if (f >= F) return;                                       // (F is a multiple of 8; defensive)
    }
    if (item >= nitems[f]) return;
    const int4 it = items[(size_t)f * max_items + item];
    const int tile = it.x, first = it.y, count = it.z;
    const int b = f / S, s = f - b * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t frame_base = (size_t)(b * S_ + s);

    if (it.w != 0) {
        // ---- particles anchored outside the map: no staging, lane = window pixel straight from global
        const float scale = sqrtf((float)C);
        const int wi = lane & 7, wj = lane >> 3;
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
                const float o = blend_taps((inmap ? d : 0.f) / scale, wx, wy, lane);
                if (lane < 49) X[m * PIPS_KIN_PAD + C + lvl * 49 + lane] = o;
            }
        }
        return;
    }

    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;

    // ---- start the first stage right away: level 0, chunk 0 -> slot 0
    unsigned doff[MAXPIECES];
    const LevelGeom g = level_geom(lv, 0, tx, ty, frame_base);
    dma_setup<4>(g, wave, lane, doff);
    const __amdgpu_buffer_rsrc_t map = make_rsrc(pyramid + g.base);
    const __amdgpu_buffer_rsrc_t ffr = make_rsrc(ffeats);
    dma_issue(map, 0, doff, (g.nquads + 63) >> 6, wave, smem);

    // ---- sort the item's particles by the Morton code of their level-0 anchor (equal anchors at the
    //      coarse levels become neighbours), spread them over the waves
    unsigned* skey = reinterpret_cast<unsigned*>(smem + 2 * SLOT_BYTES);            // [GMAX]
    float* sxy = reinterpret_cast<float*>(smem + 2 * SLOT_BYTES + GMAX * 4);        // [GMAX][2]
    int* sn = reinterpret_cast<int*>(smem + 2 * SLOT_BYTES + GMAX * 12);            // [GMAX]
    int my_n = 0; float my_x = 0.f, my_y = 0.f; unsigned my_key = 0;
    if (tid < count) {
        my_n = order[(size_t)f * N + first + tid];
        const size_t m = ((size_t)b * N + my_n) * S + s;
        my_x = coords[m * 2 + 0]; my_y = coords[m * 2 + 1];
        int bx, by; float wx, wy;
        corr_window(my_x, my_y, 0, g.H, g.W, bx, by, wx, wy);
        const unsigned ax = (unsigned)(bx + PIPS_RADIUS) & 15u, ay = (unsigned)(by + PIPS_RADIUS) & 15u;
        unsigned mort = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) mort |= ((ax >> i) & 1u) << (2 * i) | ((ay >> i) & 1u) << (2 * i + 1);
        my_key = (mort << 8) | (unsigned)tid;
        skey[tid] = my_key;
    }
    __syncthreads();
    if (tid < count) {
        int rank = 0;
        for (int j = 0; j < count; ++j) rank += skey[j] < my_key ? 1 : 0;
        sn[rank] = my_n; sxy[rank * 2 + 0] = my_x; sxy[rank * 2 + 1] = my_y;
    }
    __syncthreads();
    const int base_n = count / NW, rem = count - base_n * NW;
    const int nslot = base_n + (wave < rem ? 1 : 0);              // particles of this wave (may be 0)
    const int start = wave * base_n + min(wave, rem);
    const int ns = base_n + (rem ? 1 : 0);                        // slots every wave of the block runs

    // ---- lane-parallel window geometry: lane k*4+l <-> (slot k, level l); slots past the wave's own
    //      particles repeat its last one (or the item's first, for an empty wave)
    float geo_bx, geo_by, geo_wx, geo_wy;                              // ints travel as bit patterns
    int geo_row;
    {
        const int k = min(lane >> 2, SLOTS - 1), l = lane & 3;
        const int idx = nslot > 0 ? start + min(k, nslot - 1) : 0;
        const float cx = sxy[idx * 2 + 0], cy = sxy[idx * 2 + 1];
        int bx, by;
        corr_window(cx, cy, l, lv.H[l], lv.W[l], bx, by, geo_wx, geo_wy);
        geo_bx = __int_as_float(bx); geo_by = __int_as_float(by);
        geo_row = (b * N + sn[idx]) * S + s;                           // mixer row m
    }
    // slot k re-uses slot k-1's fragments at level l when both windows have the same anchor
    unsigned long long samebits;
    {
        const int pbx = __shfl_up(__float_as_int(geo_bx), 4), pby = __shfl_up(__float_as_int(geo_by), 4);
        samebits = __ballot(lane >= 4 && pbx == __float_as_int(geo_bx) && pby == __float_as_int(geo_by));
    }

#define PIPS_TILE_CASE(NS_)                                                                                          \
    tile_body<NS_>(pyramid, lv, frame_base, tx, ty, smem, wave, lane, map, g, doff, ffr, geo_bx, geo_by, geo_wx,    \
                   geo_wy, geo_row, samebits, X)
#ifdef PIPS_TILE_ONLY
    (void)ns;
    PIPS_TILE_CASE(PIPS_TILE_ONLY);
#else
    if (ns <= 4) PIPS_TILE_CASE(4);
    else if (ns <= 8) PIPS_TILE_CASE(8);
    else if (ns == 9) PIPS_TILE_CASE(9);
    else if (ns == 10) PIPS_TILE_CASE(10);
    else PIPS_TILE_CASE(11);
#endif
#undef PIPS_TILE_CASE
}

// ---------------------------------------------------------------------------- host side
static int tiled_max_items(int N, int H8, int W8) { return cdiv(W8, TS) * cdiv(H8, TS) + 1 + N / GMAX + 1; }

size_t tiled_gather_scratch_bytes(int B, int N, int H8, int W8) {
    const int F = B * S;
    const int max_items = tiled_max_items(N, H8, W8);
    return align_up((size_t)F * N * sizeof(int), 256) + align_up((size_t)F * max_items * sizeof(int4), 256) +
           align_up((size_t)F * sizeof(int), 256);
}

// Selection: dense query sets (on average >= 16 particles per 16x16 level-0 tile) take the tiled kernel;
// PIPS_GATHER_TILED=0/1 forces it off/on.
bool tiled_gather_wanted(int N, int H8, int W8) {
    static int force = -2;
    if (force == -2) { const char* e = getenv("PIPS_GATHER_TILED"); force = e ? atoi(e) : -1; }
    if (force >= 0) return force > 0;
    return (long)N >= 16L * cdiv(W8, TS) * cdiv(H8, TS) && N >= 1024;
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
    const int max_items = tiled_max_items(N, H8, W8);
    char* p = (char*)scratch;
    int* order = (int*)p; p += align_up((size_t)F * N * sizeof(int), 256);
    int4* items = (int4*)p; p += align_up((size_t)F * max_items * sizeof(int4), 256);
    int* nitems = (int*)p;
    PIPS_CHECK_ARG((size_t)2 * (ntiles + 1) * sizeof(int) <= 64 * 1024, "tiled gather: map too large for the tile histogram");
    PIPS_CHECK_ARG((size_t)H8 * W8 * C * 4 < (1ull << 31), "tiled gather: level-0 map too large for 32-bit offsets");
    hipLaunchKernelGGL(bin_particles_kernel, dim3(F), dim3(256), (size_t)2 * (ntiles + 1) * sizeof(int), st, coords, N, H8,
                       W8, tiles_x, tiles_y, max_items, order, items, nitems);
    PIPS_CHECK_LAUNCH("bin_particles_kernel");
    const int M = B * N * S;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ffeats, coords, times, M, X);
    PIPS_CHECK_LAUNCH("embed_rows_kernel");
    TiledLevels lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) { lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l]; }
    if (hipFuncSetAttribute((const void*)gather_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess) {
        set_error("tiled gather: cannot raise the dynamic LDS limit to %d bytes", LDS_BYTES);
        return PIPS_E_LAUNCH;
    }
    hipLaunchKernelGGL(gather_tiled_kernel, dim3(max_items * F), dim3(NW * 64), LDS_BYTES, st, pyramid, lv, S_, ffeats,
                       coords, N, tiles_x, max_items, F, order, items, nitems, X);
    PIPS_CHECK_LAUNCH("gather_tiled_kernel");
    return PIPS_OK;
}

}  // namespace pips
