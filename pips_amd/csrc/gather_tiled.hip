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
//                          work list of (tile, first, count <= GMAX) items.  Anchors outside the map are
//                          clamped to the nearest border tile (whose halo regions hold whatever part of
//                          such a window is inside the map).
//   embed_rows_kernel      feature copy + sin/cos embedding + raw flow + zero pad of every mixer row
//                          (get_3d_embedding, utils/misc.py:44-69; DeltaBlock concat, nets/pips.py:304-308).
//   gather_tiled_kernel    one block (16 waves) per work item, two blocks per CU (8 waves per SIMD).  32 phases =
//                          4 levels x 8 chunks of 16 channels.  The tile's region of a (level, chunk) is copied
//                          global -> LDS by the waves' own LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPR staging,
//                          no ds_write), double buffered: chunk p+1 lands while chunk p is consumed, one barrier
//                          per phase.  Consumer: LANE = WINDOW PIXEL (64 lanes = the 8x8 integer window of one
//                          particle-level); the particle's feature chunk comes through the scalar cache into
//                          SGPRs (wave-uniform), so a phase costs a lane 4 `ds_read_b128` + 8 `v_pk_fma_f32` per
//                          particle and no cross-lane reduction.  The LDS image is dense (DMA writes 1 KiB linear
//                          pieces) and XOR-swizzled on the GLOBAL side -- the lane that fetches LDS quad position j
//                          of pixel (rx,ry) reads channel quad j ^ (ry & 3) -- which makes the lane=pixel
//                          `ds_read_b128` conflict-free for its true 16-lane service groups.  Particles arrive
//                          sorted by 4x4-pixel cell, so consecutive slots of a wave often share the same window
//                          anchor at the coarse levels and re-use the fragment registers instead of re-reading
//                          LDS.  The 2x2 blend of the 8x8 correlations to the 49 taps uses ds_bpermute.
// Output is identical in meaning to mixer_input_kernel (same taps, same transposed order, zeros outside
// the map); the dot products are summed as an even- and an odd-channel chain (fp32 round-off differs from the
// direct kernel's tree sum).
#include "common.h"

#include <cstdlib>

#ifndef PIPS_TILED_ABLATE
#define PIPS_TILED_ABLATE 0    // tuning builds only: 4 no LDS-DMA, 16 no barriers (8: regenerate the .inc with PIPS_GEN_ABLATE=feats)
#endif
#ifndef PIPS_TILED_REUSE
#define PIPS_TILED_REUSE 1     // re-use the fragment registers between consecutive slots with the same window anchor
#endif

namespace pips {

constexpr int S = PIPS_S;
constexpr int C = PIPS_C;
constexpr int TS = 16;                    // level-0 tile edge in map pixels
constexpr int NW = 16;                    // waves per block
constexpr int SLOTS = 6;                  // particle slots per wave
constexpr int GMAX = NW * SLOTS;          // particles per work item
constexpr int Q = 4;                      // 16-byte channel quads per pixel per chunk (16 channels)
constexpr int NCH = C / (4 * Q);          // chunks (phases) per level
constexpr int SLOT_BYTES = 34 * 1024;     // one stage: >= 23*23 px * 64 B (level 0), whole 1 KiB DMA pieces
constexpr int MAXPIECES = 3;              // DMA pieces per wave per phase: ceil(34 / 16)
constexpr int LDS_MISC = 4096;            // scratch (L2 warm-up landing zone, 256 B per wave)
constexpr int LDS_BYTES = 2 * SLOT_BYTES + LDS_MISC;

#ifdef PIPS_TILED_TRACE      // tuning builds: per-block timestamps (wave 0) at the stage boundaries of gather_tiled_kernel
__device__ unsigned long long* g_tiled_trace;
#define PIPS_TR(i) do { if (threadIdx.x == 0 && g_tiled_trace) g_tiled_trace[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PIPS_TR(i) do { } while (0)
#endif

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
// order  [F][N]        particle indices n of frame f sorted by (tile, 4x4-pixel cell inside the tile in Morton order):
//                      neighbours in the list mostly share their window anchor at the coarse levels
// items  [F][max_items] int4 {tile, first, count, 0}; a tile with more than GMAX particles is split evenly
// nitems [F]
// LDS: hist[nbins] | cursor[nbins] | tile_off[ntiles + 1],  nbins = 16 * ntiles
__global__ __launch_bounds__(1024) void bin_particles_kernel(const float* __restrict__ coords, int N, int H0, int W0,
                                                             int tiles_x, int tiles_y, int max_items,
                                                             int* __restrict__ order, int4* __restrict__ items,
                                                             int* __restrict__ nitems) {
    extern __shared__ int sm[];
    const int ntiles = tiles_x * tiles_y, nbins = ntiles * 16;
    int* hist = sm;
    int* cursor = sm + nbins;
    int* tile_off = sm + 2 * nbins;
    const int f = blockIdx.x;                   // frame = b*S + s
    const int b = f / S, s = f - b * S;
    for (int t = threadIdx.x; t < nbins; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    auto key_of = [&](int n) {
        const size_t m = ((size_t)b * N + n) * S + s;
        int bx, by; float wx, wy;
        corr_window(coords[m * 2 + 0], coords[m * 2 + 1], 0, H0, W0, bx, by, wx, wy);
        // floor(ix), floor(iy), clamped into the map: a particle anchored outside still has its in-map window
        // pixels (at every level) inside the halo region of the nearest border tile
        const int ax = min(max(bx + PIPS_RADIUS, 0), W0 - 1), ay = min(max(by + PIPS_RADIUS, 0), H0 - 1);
        const int cx = (ax >> 2) & 3, cy = (ay >> 2) & 3;
        return ((ay / TS) * tiles_x + ax / TS) * 16 + ((cx & 1) | ((cy & 1) << 1) | ((cx & 2) << 1) | ((cy & 2) << 2));
    };
    for (int n = threadIdx.x; n < N; n += blockDim.x) atomicAdd(&hist[key_of(n)], 1);
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        int c = 0;
        for (int q = 0; q < 16; ++q) c += hist[t * 16 + q];
        tile_off[t] = c;                                        // count for now
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0, ni = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int c = tile_off[t];
            tile_off[t] = off;
            if (c > 0) {
                const int parts = (c + GMAX - 1) / GMAX, per = (c + parts - 1) / parts;
                for (int c0 = 0; c0 < c; c0 += per) {
                    if (ni < max_items) items[(size_t)f * max_items + ni] = make_int4(t, off + c0, min(per, c - c0), 0);
                    ++ni;
                }
            }
            off += c;
        }
        nitems[f] = min(ni, max_items);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        int off = tile_off[t];
        for (int q = 0; q < 16; ++q) { cursor[t * 16 + q] = off; off += hist[t * 16 + q]; }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int pos = atomicAdd(&cursor[key_of(n)], 1);
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
    g.nquads = g.RW * g.RH * Q;
    g.base = lv.off[l] + frame_base * g.H * g.W * C;
    return g;
}

// XOR key of region pixel (rx, ry): LDS 16-byte slot = 4 * (pixel index & 3) + (quad ^ key) is a bijection of
// (rx & 3, ry & 3) -> the 16 lanes of a ds_read_b128 service group (4 consecutive x in each of 4 consecutive
// rows) hit 16 different 16-byte bank groups
__device__ __forceinline__ int swz_key(int rx, int ry) {
#ifdef PIPS_TILED_DBG_NOSWZ
    return 0;
#endif
    (void)rx;
    return ry & 3;
}

// per-lane global byte offsets (within the (frame, level) map) of the DMA pieces this wave issues
__device__ __forceinline__ void dma_setup(const LevelGeom& g, int wave, int lane, unsigned (&doff)[MAXPIECES]) {
    const float inv_rw = 1.0f / (float)g.RW;
#pragma unroll
    for (int r = 0; r < MAXPIECES; ++r) {
        const int L = min((wave + r * NW) * 64 + lane, g.nquads - 1);
        const int p = L / Q, j = L - p * Q;
        const int ry = (int)(((float)p + 0.5f) * inv_rw);         // p < 1024: exact
        const int rx = p - ry * g.RW;
        const int q = j ^ swz_key(rx, ry);
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
        if (!(PIPS_TILED_ABLATE & 4) && piece < npieces)                         // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lptr_t)(lds_slot + piece * 1024), 16, (int)doff[r], soff, 0, 0);
    }
}

// ---- the phase body in assembly --------------------------------------------------------------------------
// hipcc cannot be made to schedule this loop (it hoists every slot's loads to the top of the phase and spills
// hundreds of registers), so one (level, chunk) phase of one wave is ONE asm statement, unrolled by
// tools/gen_gather_asm.py into gather_phase_asm.inc.  Slots go in pairs.  Per pair:
//     v_readlane row -> s_load_dwordx16 per slot: the particle's 16-channel feature chunk, wave-uniform, through
//                       the scalar cache into s[36:67] -- no vector-memory (TA) cycles, no VGPRs.  The pair's loads
//                       are issued back to back, so their latency (an L2 hit: the rows are touched once at the
//                       start of the item) is paid once per pair and covered by the other 7 waves of the SIMD;
// then per slot:
//     4 x ds_read_b128 (3 v_xor): this lane's window pixel, skipped when the slot has the same window anchor as
//                       the previous one (the fragments are still in v[48:63]);
//     s_waitcnt lgkmcnt(0); 8 x v_pk_fma_f32 acc.xy += s[c:c+1] * v[c:c+1] (even / odd channel partial sums).
// Why packed: a v_fmac_f32 with an SGPR or DPP-broadcast source issues at HALF rate on gfx950 (56 / 54 vs 115
// lane-FMA/clk/CU), v_readlane + v_fmac at a quarter; v_pk_fma_f32 with an SGPR pair keeps the full FMA rate
// (105; tools/dpp_rate.hip, tools/valu_peak.hip).
// Registers private to the statement (declared as clobbers): s[36:67] features, s[68:71] address temporaries,
// v[48:63] fragments, v[45:47] swizzled addresses -- the kernel stays within 64 VGPRs / 80 SGPRs = 8 waves per SIMD
// (MI355X admits 8 waves per SIMD only up to .sgpr_count 80).
// No VALU-written SGPR feeds SMEM directly (the readlane result goes through s_lshl/s_add), so no manual wait
// states are needed.
#include "gather_phase_asm.inc"
#define PIPS_A_OPS(K) [acc##K] "+v"(acc[K])
#define PIPS_A_INS(K) [A##K] "v"(A[K])
#define PIPS_A_COMMON [rows] "v"(geo_row), [fb_lo] "s"((unsigned)(fb & 0xffffffffull)), [fb_hi] "s"((unsigned)(fb >> 32)), \
                      [same] "s"(same), [skip] "s"(skip), [soff] "i"(SLOT_OFF)
#define PIPS_A_CLOBBER                                                                               \
    "memory", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48",     \
    "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63",        \
    "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "v45", "v46", "v47", "v48", "v49", "v50", "v51",        \
    "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"

template <int NS, int SLOT_OFF>
struct ConsumeAsm;
#define PIPS_DEFINE_CONSUME(NS_, OUTS, INS)                                                                        \
    template <int SLOT_OFF>                                                                                        \
    struct ConsumeAsm<NS_, SLOT_OFF> {                                                                              \
        static __device__ __forceinline__ void run(const unsigned (&A)[NS_], f2 (&acc)[NS_], unsigned same,         \
                                                   unsigned skip, unsigned long long fb, int geo_row) {             \
            asm volatile(PIPS_PHASE_TEXT_##NS_ : OUTS : INS, PIPS_A_COMMON : PIPS_A_CLOBBER);                       \
        }                                                                                                          \
    };
#define PIPS_CM ,
#define PIPS_O_2 PIPS_A_OPS(0) PIPS_CM PIPS_A_OPS(1)
#define PIPS_I_2 PIPS_A_INS(0) PIPS_CM PIPS_A_INS(1)
#define PIPS_O_4 PIPS_O_2 PIPS_CM PIPS_A_OPS(2) PIPS_CM PIPS_A_OPS(3)
#define PIPS_I_4 PIPS_I_2 PIPS_CM PIPS_A_INS(2) PIPS_CM PIPS_A_INS(3)
PIPS_DEFINE_CONSUME(2, PIPS_O_2, PIPS_I_2)
PIPS_DEFINE_CONSUME(4, PIPS_O_4, PIPS_I_4)
PIPS_DEFINE_CONSUME(5, PIPS_O_4 PIPS_CM PIPS_A_OPS(4), PIPS_I_4 PIPS_CM PIPS_A_INS(4))
PIPS_DEFINE_CONSUME(6, PIPS_O_4 PIPS_CM PIPS_A_OPS(4) PIPS_CM PIPS_A_OPS(5), PIPS_I_4 PIPS_CM PIPS_A_INS(4) PIPS_CM PIPS_A_INS(5))

template <int NS, int SLOT_OFF>
__device__ __forceinline__ void consume(const unsigned (&A)[NS], f2 (&acc)[NS], unsigned same, unsigned skip,
                                        const float* __restrict__ ffeats, int geo_row, int choff) {
    ConsumeAsm<NS, SLOT_OFF>::run(A, acc, PIPS_TILED_REUSE ? (unsigned)__builtin_amdgcn_readfirstlane(same) : 0u,
                                  (unsigned)__builtin_amdgcn_readfirstlane(skip),
                                  (unsigned long long)reinterpret_cast<uintptr_t>(ffeats + choff), geo_row);
}

// per-level lane state: LDS byte address of this lane's window pixel (with the swizzle key folded in),
// in-map mask, zeroed accumulators
template <int NS>
__device__ __forceinline__ void level_setup(const LevelGeom& g, int lvl, int lane, float geo_bx, float geo_by,
                                            unsigned lds_base, unsigned (&A)[NS], f2 (&acc)[NS], unsigned& inmask) {
    const int wi = lane & 7, wj = lane >> 3;
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        acc[k] = (f2){0.f, 0.f};
        const int bx = __builtin_amdgcn_readlane(__float_as_int(geo_bx), k * 4 + lvl);
        const int by = __builtin_amdgcn_readlane(__float_as_int(geo_by), k * 4 + lvl);
        const int px = bx + wi, py = by + wj;
        const bool inmap = (unsigned)px < (unsigned)g.W && (unsigned)py < (unsigned)g.H;
        const int rx = min(max(px - g.x0, 0), g.RW - 1), ry = min(max(py - g.y0, 0), g.RH - 1);
        A[k] = lds_base + (unsigned)((ry * g.RW + rx) * (Q * 16) + (swz_key(rx, ry) << 4));
        inmask |= inmap ? (1u << k) : 0u;
    }
    asm volatile("" : "+v"(inmask));       // keep it a bit mask: re-derived from px/py it costs 2 spilled VGPRs per slot
}

// One level = NCH phases.  Entering it, phase 0's region is already in flight (slot 0); phase c issues phase
// c+1's (the next level's first after the last chunk) right behind the barrier that frees the other slot, then
// consumes its own.  One barrier per phase; `s_waitcnt vmcnt(0)` in front of it covers exactly the previous
// phase's prefetch.
template <int NS>
__device__ __forceinline__ void run_level(__amdgpu_buffer_rsrc_t map, __amdgpu_buffer_rsrc_t map_next, bool has_next,
                                          const LevelGeom& g, const LevelGeom& gn, char* smem, int wave, int lane,
                                          unsigned same, unsigned skip, const float* __restrict__ ffeats, int geo_row,
                                          unsigned (&doff)[MAXPIECES], const unsigned (&A)[NS], f2 (&acc)[NS]) {
    static_assert(NCH % 2 == 0, "phases come in pairs (static slot parity)");
    constexpr int CH = Q * 4;
    for (int c = 0; c < NCH; c += 2) {
        // ---- even phase: data in slot 0; prefetch phase c+1 into slot 1
        __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0)
        if (!(PIPS_TILED_ABLATE & 16)) __syncthreads();
        dma_issue(map, (c + 1) * CH * 4, doff, (g.nquads + 63) >> 6, wave, smem + SLOT_BYTES);
        consume<NS, 0>(A, acc, same, skip, ffeats, geo_row, c * CH);
        // ---- odd phase: data in slot 1; prefetch phase c+2 (or the next level's first) into slot 0
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (!(PIPS_TILED_ABLATE & 16)) __syncthreads();
        if (c + 2 < NCH) {
            dma_issue(map, (c + 2) * CH * 4, doff, (g.nquads + 63) >> 6, wave, smem);
        } else if (has_next) {
            dma_setup(gn, wave, lane, doff);
            dma_issue(map_next, 0, doff, (gn.nquads + 63) >> 6, wave, smem);
        }
        consume<NS, SLOT_BYTES>(A, acc, same, skip, ffeats, geo_row, (c + 1) * CH);
    }
}

// the staged part of one work item for waves that hold NS particle slots each (a wave with fewer particles
// repeats its last one: same values, same destination)
template <int NS>
__device__ __forceinline__ void tile_body(const float* __restrict__ pyramid, const TiledLevels& lv, size_t frame_base,
                                          int tx, int ty, char* smem, int wave, int lane, __amdgpu_buffer_rsrc_t map,
                                          LevelGeom g, unsigned (&doff)[MAXPIECES], const float* __restrict__ ffeats,
                                          float geo_bx, float geo_by, float geo_wx, float geo_wy, int geo_row,
                                          unsigned long long samebits, unsigned skip, float* __restrict__ X) {
    const float scale = sqrtf((float)C);
    // LDS byte address of the stage buffers
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    unsigned A[NS];
    f2 acc[NS];
    unsigned inmask;
    for (int lvl = 0; lvl < PIPS_LEVELS; ++lvl) {
        const LevelGeom gn = level_geom(lv, min(lvl + 1, PIPS_LEVELS - 1), tx, ty, frame_base);
        const __amdgpu_buffer_rsrc_t mapn = make_rsrc(pyramid + gn.base);
        level_setup<NS>(g, lvl, lane, geo_bx, geo_by, lds_base, A, acc, inmask);
        unsigned same = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) same |= (unsigned)((samebits >> (k * 4 + lvl)) & 1ull) << k;
        same = __builtin_amdgcn_readfirstlane(same);
        PIPS_TR(3 + 3 * lvl);
        run_level<NS>(map, mapn, lvl + 1 < PIPS_LEVELS, g, gn, smem, wave, lane, same, skip, ffeats, geo_row, doff, A, acc);
        PIPS_TR(4 + 3 * lvl);
        // blend the 8x8 correlations to the 49 taps, k = level*49 + ix*7 + iy (transposed, :379-381)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo_wx), k * 4 + lvl));
            const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo_wy), k * 4 + lvl));
            const int row = __builtin_amdgcn_readlane(geo_row, k * 4);
            const float o = blend_taps(((inmask >> k) & 1u) ? (acc[k].x + acc[k].y) / scale : 0.f, wx, wy, lane);   // :397
            // (a wave that skipped its last slot must not store it: the slot aliases the wave's last real particle)
            if (lane < 49 && !(k == NS - 1 && skip)) X[(size_t)row * PIPS_KIN_PAD + C + lvl * 49 + lane] = o;
        }
        PIPS_TR(5 + 3 * lvl);
        g = gn; map = mapn;
    }
}

__global__ __launch_bounds__(NW * 64, 8) void gather_tiled_kernel(const float* __restrict__ pyramid, TiledLevels lv,
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
    PIPS_TR(0);
    const int4 it = items[(size_t)f * max_items + item];
    const int tile = it.x, first = it.y, count = it.z;
#ifdef PIPS_TILED_TRACE
    if (threadIdx.x == 0 && g_tiled_trace) g_tiled_trace[(size_t)blockIdx.x * 16 + 15] = (unsigned long long)count;
#endif
    const int b = f / S, s = f - b * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t frame_base = (size_t)(b * S_ + s);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;

    // ---- start the first stage right away: level 0, chunk 0 -> slot 0
    unsigned doff[MAXPIECES];
    const LevelGeom g = level_geom(lv, 0, tx, ty, frame_base);
    dma_setup(g, wave, lane, doff);
    const __amdgpu_buffer_rsrc_t map = make_rsrc(pyramid + g.base);
    dma_issue(map, 0, doff, (g.nquads + 63) >> 6, wave, smem);

    // ---- warm the L2 with the item's particle features (one dword per 128-byte line, dropped into the LDS
    //      scratch area by the DMA engine: no VGPR, tracked by vmcnt like the stage loads)
    if (tid < count * 4) {                                              // 4 lines per 512-byte feature row
        const int n = order[(size_t)f * N + first + (tid >> 2)];
        const unsigned off = (unsigned)((((size_t)b * N + n) * S + s) * (C * 4) + (tid & 3) * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(ffeats), (lptr_t)(smem + 2 * SLOT_BYTES + wave * 256), 4, (int)off, 0,
                                                 0, 0);
    }
    // ---- the list is already ordered (bin_particles_kernel); spread it over the waves
    const int base_n = count / NW, rem = count - base_n * NW;
    const int nslot = base_n + (wave < rem ? 1 : 0);              // particles of this wave (may be 0)
    const int start = wave * base_n + min(wave, rem);
    const int ns = base_n + (rem ? 1 : 0);                        // slots every wave of the block runs
    PIPS_TR(1);

    // ---- lane-parallel window geometry: lane k*4+l <-> (slot k, level l); slots past the wave's own
    //      particles repeat its last one (or the item's first, for an empty wave)
    float geo_bx, geo_by, geo_wx, geo_wy;                              // ints travel as bit patterns
    int geo_row;
    {
        const int k = min(lane >> 2, SLOTS - 1), l = lane & 3;
        const int idx = nslot > 0 ? start + min(k, nslot - 1) : 0;
        geo_row = (b * N + order[(size_t)f * N + first + idx]) * S + s;     // mixer row m
        const float cx = coords[(size_t)geo_row * 2 + 0], cy = coords[(size_t)geo_row * 2 + 1];
        int bx, by;
        corr_window(cx, cy, l, lv.H[l], lv.W[l], bx, by, geo_wx, geo_wy);
        geo_bx = __int_as_float(bx); geo_by = __int_as_float(by);
    }
    // slot k re-uses slot k-1's fragments at level l when both windows have the same anchor
    unsigned long long samebits;
    {
        const int pbx = __shfl_up(__float_as_int(geo_bx), 4), pby = __shfl_up(__float_as_int(geo_by), 4);
        samebits = __ballot(lane >= 4 && pbx == __float_as_int(geo_bx) && pby == __float_as_int(geo_by));
    }
    PIPS_TR(2);
#define PIPS_TILE_CASE(NS_)                                                                                          \
    tile_body<NS_>(pyramid, lv, frame_base, tx, ty, smem, wave, lane, map, g, doff, ffeats, geo_bx, geo_by, geo_wx,  \
                   geo_wy, geo_row, samebits, (unsigned)__builtin_amdgcn_readfirstlane(nslot < NS_ ? 1 : 0), X)
#ifdef PIPS_TILE_ONLY
    (void)ns;
    PIPS_TILE_CASE(PIPS_TILE_ONLY);
#else
    if (ns <= 2) PIPS_TILE_CASE(2);
    else if (ns <= 4) PIPS_TILE_CASE(4);
    else if (ns == 5) PIPS_TILE_CASE(5);
    else PIPS_TILE_CASE(6);
#endif
#undef PIPS_TILE_CASE
}

// ---------------------------------------------------------------------------- host side
static int tiled_max_items(int N, int H8, int W8) { return cdiv(W8, TS) * cdiv(H8, TS) + N / GMAX + 1; }

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
    return (long)N >= 16L * cdiv(W8, TS) * cdiv(H8, TS) && N >= 1024 &&
           ((size_t)33 * cdiv(W8, TS) * cdiv(H8, TS) + 1) * sizeof(int) <= 64 * 1024;
}

int launch_mixer_input_tiled(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B,
                             int S_, const float* ffeats, const float* coords, const float* times, int N,
                             float* X, void* scratch, size_t scratch_bytes, hipStream_t st, hipEvent_t* ev) {
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
    const size_t bin_lds = ((size_t)2 * 16 * ntiles + ntiles + 1) * sizeof(int);
    PIPS_CHECK_ARG(bin_lds <= 64 * 1024, "tiled gather: map too large for the tile histogram");
    PIPS_CHECK_ARG((size_t)H8 * W8 * C * 4 < (1ull << 31), "tiled gather: level-0 map too large for 32-bit offsets");
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(bin_particles_kernel, dim3(F), dim3(1024), bin_lds, st, coords, N, H8, W8, tiles_x, tiles_y, max_items,
                       order, items, nitems);
    PIPS_CHECK_LAUNCH("bin_particles_kernel");
    const int M = B * N * S;
    if (ev) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ffeats, coords, times, M, X);
    PIPS_CHECK_LAUNCH("embed_rows_kernel");
    TiledLevels lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) { lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l]; }
    {
        static std::atomic<unsigned long long> raised{0};
        const int rc = ensure_dynamic_lds(raised, (const void*)gather_tiled_kernel, LDS_BYTES);
        if (rc != PIPS_OK) return rc;
    }
    if (ev) (void)hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(gather_tiled_kernel, dim3(max_items * F), dim3(NW * 64), LDS_BYTES, st, pyramid, lv, S_, ffeats,
                       coords, N, tiles_x, max_items, F, order, items, nitems, X);
    if (ev) (void)hipEventRecord(ev[3], st);
    PIPS_CHECK_LAUNCH("gather_tiled_kernel");
    return PIPS_OK;
}

}  // namespace pips

#ifdef PIPS_TILED_TRACE
extern "C" int pips_tiled_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_tiled_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif
