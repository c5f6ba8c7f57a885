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
//   gather_tiled_kernel    ONE persistent block (16 waves) per CU; block id mod 8 = XCD, whose 32 CUs work through the
//                          tiles of one frame side by side (halos shared in that XCD's L2).  Per work item 8 phases,
//                          one per 16-channel chunk: the tile's regions of ALL FOUR levels of the chunk (<= 72 KiB) are
//                          copied global -> LDS by the waves' own LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR
//                          staging, no ds_write), double buffered -- chunk c+1 lands while chunk c is consumed, one
//                          barrier per chunk.  Consumer: LANE = WINDOW PIXEL (64 lanes = the 8x8 integer window of one
//                          particle-level); the particle's feature chunk comes through the scalar cache into SGPRs
//                          (wave-uniform), so a chunk costs a lane 4 `ds_read_b128` + 8 `v_pk_fma_f32` per
//                          particle-level and no cross-lane reduction; the 48 accumulators of a wave (4 levels x
//                          6 particles x even/odd chain) stay in registers across the phases.  The LDS image is a
//                          sequence of 1 KiB linear DMA pieces; staged rows are PADDED by one 16-byte slot (filled by a
//                          duplicate fetch): a pixel's four quads sit at consecutive addresses and the b128 reads of a
//                          service group are conflict-free without a swizzle.  Particles arrive
//                          sorted by 4x4-pixel cell, so the two slots of a pair often share the window anchor at the
//                          coarse levels and one fragment serves both.  The 2x2 blend of the 8x8 correlations to the
//                          49 taps uses ds_bpermute.  The whole item body is generated assembly (gather_item_asm.inc,
//                          tools/gen_gather_asm.py): see gather_item below.
// Output is identical in meaning to mixer_input_kernel (same taps, same transposed order, zeros outside
// the map); the dot products are summed as an even- and an odd-channel chain and scaled by 1/sqrt(128) through the
// blend weights (fp32 round-off differs from the direct kernel's tree sum and division: 6e-5 against an fp64 run of the
// reference arithmetic at 160-pixel-wide maps, where its fp32 run is itself 5e-5 off -- tests/test_config45_gpu.py).
#include "common.h"

#include <cstdlib>

#ifndef PIPS_TILED_ABLATE
#define PIPS_TILED_ABLATE 0    // tuning builds only: 4 no LDS-DMA, 128 no item body (PIPS_GEN_ABLATE=feats,reads,fma,dma,warm,epi: parts of it)
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
constexpr int LDS_MISC = 15360;           // scratch behind the stages (item entries, record prefetch, landing zone of the L2 touches)

#ifndef PIPS_TRACE_WAVE
#define PIPS_TRACE_WAVE 0
#endif
#ifdef PIPS_TILED_TRACE      // tuning builds: per-block timestamps (wave 0) at the stage boundaries of gather_tiled_kernel
__device__ unsigned long long* g_tiled_trace;
#define PIPS_TR(i) do { if (threadIdx.x == 0 && g_tiled_trace) g_tiled_trace[((size_t)blockIdx.x * 64 + (t & 63)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
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

// ---------------------------------------------------------------------------- binning
// order  [F][N][4]     one int4 PER LEVEL {bx | by << 16, bits of wx, bits of wy, mixer row m} of frame f's particles sorted by
//                      (tile, 4x4-pixel cell inside the tile in Morton order): neighbours in the list mostly share their
//                      window anchor at the coarse levels.  The window geometry of nets/pips.py:318-319 + grid_sample (two
//                      fp32 divisions per level) is computed HERE, once per particle-level, not per lane of the gather
//                      (anchors far outside the map are clamped to +-20000: their windows are empty either way)
// items  [F][max_items] int4 {tile, first, count, 0}; a tile with more than GMAX particles is split evenly
// nitems [F]
// LDS: hist[nbins] | cursor[nbins] | tile_off[ntiles + 1],  nbins = 16 * ntiles
__global__ __launch_bounds__(1024) void bin_particles_kernel(const float* __restrict__ coords, int N, TiledLevels lv,
                                                             int tiles_x, int tiles_y, int max_items,
                                                             int4* __restrict__ order, int4* __restrict__ items,
                                                             int* __restrict__ nitems, int* __restrict__ slot_of) {
    const int H0 = lv.H[0], W0 = lv.W[0];
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
        const size_t m = ((size_t)b * N + n) * S + s;
        const float cx = coords[m * 2 + 0], cy = coords[m * 2 + 1];
        if (slot_of) slot_of[m] = f * N + pos;                      // (bf16 mode: where embed_rows_kernel puts the row's bf16 features)
#pragma unroll
        for (int l = 0; l < PIPS_LEVELS; ++l) {
            int bx, by; float wx, wy;
            corr_window(cx, cy, l, lv.H[l], lv.W[l], bx, by, wx, wy);
            bx = min(max(bx, -20000), 20000); by = min(max(by, -20000), 20000);
            order[((size_t)f * N + pos) * PIPS_LEVELS + l] =
                make_int4((int)(((unsigned)bx & 0xffffu) | ((unsigned)by << 16)), __float_as_int(wx), __float_as_int(wy), (int)m);
        }
    }
}

// ---------------------------------------------------------------------------- embedding rows
// One wave per mixer row m: X[m] = [ffeat 128 | (corr 196: not touched) | sin/cos 192 | dx dy t | 0 x 25]
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ ffeats,
                                                         const float* __restrict__ coords,
                                                         const float* __restrict__ times, int M,
                                                         float* __restrict__ X, const int* __restrict__ slot_of,
                                                         uint4* __restrict__ featb) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const int s = m % S;
    const size_t m0 = (size_t)(m - s);
    const float dx = coords[(size_t)m * 2 + 0] - coords[m0 * 2 + 0];        // coords - coords[:,0:1] (:518)
    const float dy = coords[(size_t)m * 2 + 1] - coords[m0 * 2 + 1];
    const float tt = times[s];
    float* xrow = X + (size_t)m * PIPS_KIN_PAD;
    const float* ff = ffeats + (size_t)m * C;
    if (lane < C / 4) {
        const float4 v = reinterpret_cast<const float4*>(ff)[lane];
        reinterpret_cast<float4*>(xrow)[lane] = v;
        // bf16 mode: the row's features as bf16 (RNE), in the order of the frame's sorted particle list -- an item's features are
        // then ONE contiguous run that gather_mfma_kernel's loaders stream like a chunk of the maps
        if (featb) reinterpret_cast<uint2*>(featb)[(size_t)slot_of[m] * (C / 4) + lane] = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
    }
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

// Level geometry of a work item, LANE-PARALLEL: lane l (l < 4) of every wave holds level l's staged region; the
// wave-uniform values are pulled out with v_readlane where they are needed (keeps them out of the SGPR file, which
// the phase body fills with feature chunks, and keeps the set-up free of scalar branches)
struct LaneGeom {
    int x0, y0, RW, RH, W, H;
    int nquads;             // RH * (RW*Q + 1) 16-byte LDS positions: a staged row = RW pixels of Q quads + one pad slot
    unsigned base;          // byte offset of the (frame, level) map in the pyramid buffer
    int lbase, lsize;       // LDS byte offset of the level's stage pair, size of one stage
};
#define PIPS_RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))

// buffer resource over the pyramid: raw (stride 0) addressing, byte offset = soffset (SGPR) + voffset (VGPR) < 4 GiB;
// keeps every address of the hot loop out of the vector registers
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffff, 0x00020000);
}

// ---- LDS layout: per level two stages (chunk parity) side by side, so that the parity is a 16-bit immediate
// offset of the ds_reads:  [L0 s0 | L0 s1 | L1 s0 | L1 s1 | L2 .. | L3 .. | scratch]
// stage sizes in 1 KiB DMA pieces: 34, 19, 11, 8 >= (23^2, 17^2, 13^2, 11^2 px) * 64 B
constexpr int SZ0 = 34 * 1024, SZ1 = 19 * 1024, SZ2 = 11 * 1024, SZ3 = 8 * 1024;
constexpr int LB0 = 0, LB1 = 2 * SZ0, LB2 = LB1 + 2 * SZ1, LB3 = LB2 + 2 * SZ2;
constexpr int LDS_MISC_OFF = LB3 + 2 * SZ3;                    // 144 KiB
constexpr int LDS_BYTES_V3 = LDS_MISC_OFF + LDS_MISC;
constexpr int MAXP = 5;                                         // DMA pieces per wave per phase: ceil(72 / 16)
// ---- one work item of one wave in assembly ------------------------------------------------------------------
// gather_item(): everything between the item's index arithmetic and its stores is ONE asm statement, unrolled by
// tools/gen_gather_asm.py into gather_item_asm.inc:
//   set-up     LDS byte address of this lane's window pixel for the 24 units (level, slot) -- LANE = WINDOW PIXEL, 64
//              lanes = the 8x8 integer window of one particle-level -- with the swizzle key folded in, in-map bits,
//              zeroed accumulators;
//   8 phases   one per 16-channel chunk: `s_waitcnt vmcnt(0)` + ONE barrier, then the wave's (up to five) DMA pieces of
//              the next chunk's stage (all four levels) are issued between its first FMA groups -- issued back to
//              back by 16 waves they fill the texture-address queue (64 B/clk: ~1150 clk per 72 KiB stage) and every
//              wave sits in the issue stall.  Work goes by PAIRS of slots: per level 8 ds_read_b128 (the second slot's
//              four are skipped when it has the first one's window anchor -- one fragment serves both), one
//              `s_waitcnt lgkmcnt(0)`, 16 v_pk_fma_f32 with the two particles' feature chunks as SGPR-pair operands
//              (wave-uniform, through the scalar cache: no vector-memory cycles, no VGPRs).  A pair's features are
//              requested while the previous pair works (two SGPR sets of 2 x 16), across phase boundaries too, so a
//              scalar load has ~4 FMA groups to land; the L2 lines behind them are kept warm by one-dword DMA touches
//              two chunks ahead (the map traffic of an XCD's 32 CUs would evict them).
//              Why packed FMAs: a v_fmac_f32 with an SGPR or DPP-broadcast source issues at HALF rate on gfx950,
//              v_pk_fma_f32 with an SGPR pair keeps the full rate (tools/dpp_rate.hip, tools/valu_peak.hip);
//   blend      per slot: 16 ds_bpermute (the 2x2 neighbours of the 49 taps at the four levels) under one wait,
//              weights from lane-parallel registers by v_readlane, 4 stores of 49 lanes.
// Why one statement: the 48 accumulators + 24 addresses of a wave live across the eight barriers.  As C++ around
// per-phase asm statements hipcc spilled ~60 VGPRs around the set-up and the blend (each reload a scratch round
// trip of ~1 us) and those two parts cost as much as the phases; here nothing crosses a statement boundary.
#ifndef PIPS_ITEM_INC
#define PIPS_ITEM_INC "gather_item_asm.inc"      // tuning builds point this at an ablated copy
#endif
#include PIPS_ITEM_INC

typedef unsigned u4v __attribute__((ext_vector_type(4)));

// the DMA pieces of one wave: piece id g = wave + 16 r over the concatenation of the four levels' pieces
struct WavePieces {
    unsigned doff[MAXP];        // per lane: byte offset (from the pyramid start) of the 16 bytes this lane fetches
    int lds[MAXP];              // wave-uniform: LDS byte offset of the piece (stage parity 0); -1 = no piece
    int par[MAXP];              // wave-uniform: stage size of the piece's level (the parity stride)
};

__device__ __forceinline__ LaneGeom lane_geom(const TiledLevels& lv, int lane, int tx, int ty, size_t frame_base) {
    LaneGeom g;
    const int l = lane & 3;
    // (the table entries go through opaque registers: hipcc otherwise turns the selects below into a dynamically
    //  indexed load and copies the kernel argument to scratch memory for it)
    int W0 = lv.W[0], W1 = lv.W[1], W2 = lv.W[2], W3 = lv.W[3], H0 = lv.H[0], H1 = lv.H[1], H2 = lv.H[2], H3 = lv.H[3];
    size_t o0 = lv.off[0], o1 = lv.off[1], o2 = lv.off[2], o3 = lv.off[3];
    asm volatile("" : "+s"(W0), "+s"(W1), "+s"(W2), "+s"(W3), "+s"(H0), "+s"(H1), "+s"(H2), "+s"(H3));
    asm volatile("" : "+s"(o0), "+s"(o1), "+s"(o2), "+s"(o3));
    g.W = l == 0 ? W0 : (l == 1 ? W1 : (l == 2 ? W2 : W3));
    g.H = l == 0 ? H0 : (l == 1 ? H1 : (l == 2 ? H2 : H3));
    const size_t off = l == 0 ? o0 : (l == 1 ? o1 : (l == 2 ? o2 : o3));
    // Staged region along one axis (inclusive, clipped to the map).  Level 0: the anchor floor(ix) of a binned particle
    // lies in [16t, 16t+15] exactly, its window reaches -3..+4.  Coarser levels: floor(ix_l) lies in [T-1, T+(16>>l)] with
    // T = (16t)>>l (one pixel of slack each side for the independently rounded coordinate), same reach.
    const int Tx = (tx * TS) >> l, Ty = (ty * TS) >> l, w = TS >> l, h = l == 0 ? 3 : 4;
    g.x0 = max(Tx - h, 0); g.y0 = max(Ty - h, 0);
    const int x1 = min(Tx + w + h, g.W - 1), y1 = min(Ty + w + h, g.H - 1);
    g.RW = max(x1 - g.x0 + 1, 1); g.RH = max(y1 - g.y0 + 1, 1);
    if (x1 < g.x0 || y1 < g.y0) { g.x0 = g.y0 = 0; g.RW = g.RH = 1; }            // (tile beyond this level's map)
    g.nquads = g.RH * (g.RW * Q + 1);
    g.base = (unsigned)((off + frame_base * g.H * g.W * C) * sizeof(float));
    g.lbase = l == 0 ? LB0 : (l == 1 ? LB1 : (l == 2 ? LB2 : LB3));
    g.lsize = l == 0 ? SZ0 : (l == 1 ? SZ1 : (l == 2 ? SZ2 : SZ3));
    return g;
}

__device__ __forceinline__ void dma_setup(const LaneGeom& g, int wave, int lane, WavePieces& wp) {
    const int np = (g.nquads + 63) >> 6;
    const int c1 = PIPS_RL(np, 0), c2 = c1 + PIPS_RL(np, 1), c3 = c2 + PIPS_RL(np, 2), c4 = c3 + PIPS_RL(np, 3);
#pragma unroll
    for (int r = 0; r < MAXP; ++r) {
        const int gp = wave + r * NW;
        const int l = (gp >= c1) + (gp >= c2) + (gp >= c3);                      // wave-uniform
        const int piece = gp - (l > 0 ? c1 : 0) - (l > 1 ? c2 - c1 : 0) - (l > 2 ? c3 - c2 : 0);
        const int RW = PIPS_RL(g.RW, l), W = PIPS_RL(g.W, l), x0 = PIPS_RL(g.x0, l), y0 = PIPS_RL(g.y0, l);
        // LDS slot L of the level's stage = (row ry, position e in the row); rows are PADDED by one 16-byte slot (a pixel's
        // four quads sit at consecutive addresses and the 16 lanes of a ds_read_b128 service group -- 4 consecutive x in
        // each of 4 consecutive window rows -- hit 16 different bank groups: no swizzle, no address arithmetic per quad);
        // the lane that fills a pad slot fetches the row's last quad again
        const int L = min(piece * 64 + lane, PIPS_RL(g.nquads, l) - 1);
        const int pitch = RW * Q + 1;
        const int ry = L / pitch;
        const int e = min(L - ry * pitch, RW * Q - 1);
        const int rx = e >> 2, q = e & 3;
        wp.doff[r] = (unsigned)PIPS_RL(g.base, l) + (unsigned)(((y0 + ry) * W + (x0 + rx)) * (C * 4) + q * 16);
        wp.lds[r] = gp < c4 ? PIPS_RL(g.lbase, l) + piece * 1024 : -1;
        wp.par[r] = PIPS_RL(g.lsize, l);
    }
}

// ---- per-tile tables (tile_table_kernel, once per launch): what a wave needs for a work item beyond its particles depends
// on the tile and the map sizes only -- staged regions, the wave's DMA pieces and every lane's source offset in them.  The
// first cut recomputed it per item and wave (~700 VALU instructions, 12k of the 70k clocks of an item).
//   gpk_tab  [tile][wave][32]        the asm's lane-parallel input: lanes l / 4+l / 8+l = x0|y0<<16, RW|RH<<16, W|H<<16 of
//                                    level l; lane 16+r = LDS offset of the wave's DMA piece r (-1: none), lane 24+r = its
//                                    stage size (= which level it belongs to)
//   doff_tab [tile][wave][MAXP][64]  byte offset (from the pyramid start, frame 0) of the 16 bytes lane fetches in piece r
__global__ __launch_bounds__(NW * 64) void tile_table_kernel(TiledLevels lv, int tiles_x, int* __restrict__ gpk_tab,
                                                             unsigned* __restrict__ doff_tab) {
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const LaneGeom g = lane_geom(lv, lane, tx, ty, 0);
    WavePieces wp;
    dma_setup(g, wave, lane, wp);
#pragma unroll
    for (int r = 0; r < MAXP; ++r) doff_tab[(((size_t)tile * NW + wave) * MAXP + r) * 64 + lane] = wp.doff[r];
    if (lane < 32) {
        const unsigned v0 = (unsigned)g.x0 | ((unsigned)g.y0 << 16), v1 = (unsigned)g.RW | ((unsigned)g.RH << 16),
                       v2 = (unsigned)g.W | ((unsigned)g.H << 16);                // (g is lane-parallel by lane & 3)
        int v = (int)(lane < 4 ? v0 : (lane < 8 ? v1 : (lane < 12 ? v2 : 0u)));
#pragma unroll
        for (int r = 0; r < MAXP; ++r) {
            if (lane == 16 + r) v = wp.lds[r];
            if (lane == 24 + r) v = wp.par[r];
        }
        gpk_tab[((size_t)tile * NW + wave) * 32 + lane] = v;
    }
}

// One persistent block (16 waves) per CU; block id mod 8 is the XCD (observed dispatch order), so the 32 CUs of
// XCD x work through the tiles of frames x, x+8, ... side by side and a frame's halos are shared in that XCD's L2.
// Per work item (tile, <= 96 particles): the regions of ALL FOUR levels of a 16-channel chunk are copied
// global -> LDS by the waves' own LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR staging, no ds_write), double
// buffered: chunk c+1 lands while chunk c is consumed, one barrier per chunk.  C++ does the item's index arithmetic
// (lane-parallel) and starts the first stage; gather_item_asm.inc does the rest.
struct FrameStrides { unsigned b[PIPS_LEVELS]; };          // bytes of one frame's map per level

__global__ __launch_bounds__(NW * 64, 1) void gather_tiled_kernel(const float* __restrict__ pyramid, FrameStrides fs,
                                                                  int S_, const float* __restrict__ ffeats, int N,
                                                                  int max_items, int F,
                                                                  const int4* __restrict__ order,
                                                                  const int4* __restrict__ items,
                                                                  const int* __restrict__ nitems,
                                                                  const int* __restrict__ gpk_tab,
                                                                  const unsigned* __restrict__ doff_tab,
                                                                  float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int xcd = blockIdx.x & 7, J = gridDim.x >> 3;
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    u4v rs;                                                        // the same descriptor as `map`, as plain SGPR values
    {
        const unsigned long long pa = (unsigned long long)reinterpret_cast<uintptr_t>(pyramid);
        rs = (u4v){(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    }
    const __amdgpu_buffer_rsrc_t recs = make_rsrc(order), frs = make_rsrc(ffeats);
    // scratch behind the stages:  [junk 256 B | entries 64 x 16 B | per wave: records 2 x 24 x 16 B | waves 0-1: rows 2 x 64 x 4 B]
    char* misc = smem + LDS_MISC_OFF;
    int4* ent = reinterpret_cast<int4*>(misc + 256);
    int4* recbuf = reinterpret_cast<int4*>(misc + 256 + 1024) + wave * (2 * 4 * SLOTS);
    unsigned* rowbuf = reinterpret_cast<unsigned*>(misc + 256 + 1024 + NW * 2 * 4 * SLOTS * 16) + min(wave, 1) * 128;
    // Records of item (first, count) this wave needs, by LDS-DMA into buffer p (no VGPRs: the copy is in flight across
    // the asm statement of the item before): the (slot, level) records of the wave's six slots (16 B each, lanes 0-23:
    // lane 4k+l = slot k, level l) and, in waves 0-1, the mixer row of particle tid (4 B) for the L2 touches of the features
    auto prefetch_records = [&](int f_, int first, int count, int p, int lane, int tid) {
        const int base_n = count / NW, rem = count - base_n * NW;
        const int nslot = base_n + (wave < rem ? 1 : 0), start = wave * base_n + min(wave, rem);
        const unsigned o0 = (unsigned)(((size_t)f_ * N + first) * (PIPS_LEVELS * sizeof(int4)));
        if (lane < 4 * SLOTS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(recs, (lptr_t)(recbuf + p * (4 * SLOTS)), 16,
                                                     (int)(o0 + (unsigned)((nslot > 0 ? start + min(lane >> 2, nslot - 1) : 0) * 4 + (lane & 3)) * 16u), 0, 0, 0);
        if (tid < count)                                                         // (count <= 96: waves 0 and 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(recs, (lptr_t)(rowbuf + p * 64), 4, (int)(o0 + (unsigned)tid * 64u + 12u), 0, 0, 0);
    };
    for (int base = 0;; base += 64) {
        // ---- this block's next (up to) 64 work items: item i of the block is entry j + i J of the XCD's list (frames
        //      xcd, xcd + 8, ... one after another); lane-parallel look-up, entries {tile, first, count, frame} in LDS
        __syncthreads();
        if (wave == 0) {
            int gi = (blockIdx.x >> 3) + (base + lane0) * J, fr = xcd;
            int4 e = make_int4(0, 0, 0, -1);
            for (; fr < F; fr += 8) {
                const int n = nitems[fr];
                if (gi < n) break;
                gi -= n;
            }
            if (fr < F) { e = items[(size_t)fr * max_items + gi]; e.w = fr; }
            ent[lane0] = e;
        }
        __syncthreads();
        // (entries travel as scalars: an int4 held in vector registers across the asm statement costs spills)
#define PIPS_ENT(e_, i_) const int4 ev_##e_ = ent[i_]; const int e_##_tile = __builtin_amdgcn_readfirstlane(ev_##e_.x), \
        e_##_first = __builtin_amdgcn_readfirstlane(ev_##e_.y), e_##_count = __builtin_amdgcn_readfirstlane(ev_##e_.z), \
        e_##_f = __builtin_amdgcn_readfirstlane(ev_##e_.w)
        PIPS_ENT(c0, 0);
        if (c0_f < 0) break;
        int tile = c0_tile, count = c0_count, f = c0_f;
        prefetch_records(c0_f, c0_first, c0_count, 0, lane0, tid0);
        __builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0)
        bool more = true;
        for (int i = 0; i < 64; ++i) {
            if (f < 0) { more = false; break; }
            // per-iteration copies the compiler cannot see through: lane-dependent addresses are otherwise hoisted out of
            // the loop and, with only v0-v22 live across the asm statement, spilled -- and a scratch reload travels the
            // vector-memory pipe behind 72 KiB of stage DMA
            int lane = lane0, tid = tid0;
            asm volatile("" : "+v"(lane), "+v"(tid));
#ifdef PIPS_TILED_TRACE
            const int t = base + i;
#endif
            PIPS_TR(0);
            const int b = f / S, s = f - b * S;
            const unsigned fi = (unsigned)(b * S_ + s);                                  // frame index in the map buffer
            const int p = i & 1;
            // ---- the next item's records on their way while this one is worked on (first in the DMA queue: the
            //      stage pieces, issued from the asm, are 72 KiB)
            PIPS_ENT(nx, min(i + 1, 63));
            const int nf = i + 1 < 64 ? nx_f : -1;
            if (nf >= 0) prefetch_records(nf, nx_first, nx_count, p ^ 1, lane, tid);
            PIPS_TR(13);
            // ---- staged regions and the wave's DMA pieces from the tile tables (the asm issues the pieces, the first stage
            //      between its address set-up: no barrier needed -- a wave gets here only after the barrier of the previous
            //      item's last phase, behind which nobody reads parity 0)
            int gpk = gpk_tab[((size_t)tile * NW + wave) * 32 + (lane & 31)];
            WavePieces wp;
            {
                const unsigned* dt = doff_tab + (((size_t)tile * NW + wave) * MAXP) * 64 + lane;
#pragma unroll
                for (int r = 0; r < MAXP; ++r) {
                    const int par = PIPS_RL(gpk, 24 + r);                                   // the piece's level, by its stage size
                    const unsigned fb = fi * (par == SZ0 ? fs.b[0] : (par == SZ1 ? fs.b[1] : (par == SZ2 ? fs.b[2] : fs.b[3])));
                    wp.doff[r] = dt[r * 64] + fb;
                }
            }
            if (PIPS_TILED_ABLATE & 4) gpk = (lane & 31) >= 16 && (lane & 31) < 16 + MAXP ? -1 : gpk;
            PIPS_TR(14);

            // ---- the list is already ordered (bin_particles_kernel); spread it over the waves
            const int base_n = count / NW, rem = count - base_n * NW;
            const int nslot = __builtin_amdgcn_readfirstlane(base_n + (wave < rem ? 1 : 0));   // particles of this wave (may be 0)
            // the item's particle features into the L2 (chunks 0-1 here, the rest two chunks ahead of their use)
            unsigned warm_off = 0;
            if (tid < count) {
                warm_off = rowbuf[p * 64 + lane] * (unsigned)(C * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lptr_t)misc, 4, (int)warm_off, 0, 0, 0);
            }
            // ---- lane-parallel window geometry: lane k*4+l <-> (slot k, level l), straight out of the records (computed by
            //      bin_particles_kernel); slots past the wave's own particles repeat its last one (or the item's first)
            float geo_wx, geo_wy;
            int geo_bx, geo_by, geo_row;
            {
                const int4 r = recbuf[p * (4 * SLOTS) + min(lane, 4 * SLOTS - 1)];
                geo_bx = (int)(short)(r.x & 0xffff);
                geo_by = r.x >> 16;
                geo_wx = __int_as_float(r.y);
                geo_wy = __int_as_float(r.z);
                geo_row = r.w;                                                        // mixer row m
            }
            PIPS_TR(15);
            // slot k re-uses slot k-1's fragment at level l when both windows have the same anchor: bit 4k+l
            unsigned same;
            {
                const int pbx = __shfl_up(geo_bx, 4), pby = __shfl_up(geo_by, 4);
                same = (unsigned)__ballot(lane >= 4 && lane < 4 * SLOTS && pbx == geo_bx && pby == geo_by);
                same = PIPS_TILED_REUSE ? __builtin_amdgcn_readfirstlane(same) : 0u;
            }
            PIPS_TR(1);
#ifdef PIPS_TILED_TRACE
            int trv = 0;
#define PIPS_TR_OPERAND [tr] "+v"(trv)
#else
#define PIPS_TR_OPERAND
#endif
            if (!(PIPS_TILED_ABLATE & 128))
            asm volatile(PIPS_ITEM_TEXT
                         : PIPS_TR_OPERAND
                         : [bx] "v"(geo_bx), [by] "v"(geo_by), [wx] "v"(geo_wx), [wy] "v"(geo_wy), [row] "v"(geo_row),
                           [gpk] "v"(gpk), [doff0] "v"(wp.doff[0]), [doff1] "v"(wp.doff[1]),
                           [doff2] "v"(wp.doff[2]), [doff3] "v"(wp.doff[3]), [doff4] "v"(wp.doff[4]), [warm] "v"(warm_off),
                           [fb] "s"(ffeats), [xp] "s"(X), [rsrc] "s"(rs), [same] "s"(same),
                           [nslot] "s"(nslot), [count] "s"(count), [wave] "s"(wave), [ldsb] "s"(lds_base)
                         : PIPS_ITEM_CLOBBER);
            PIPS_TR(2);
#ifdef PIPS_TILED_TRACE
            if (wave == PIPS_TRACE_WAVE && lane >= 1 && lane < 9 && g_tiled_trace) g_tiled_trace[((size_t)blockIdx.x * 64 + (t & 63)) * 16 + 4 + lane] = (unsigned)trv;
#endif
            tile = nx_tile; count = nx_count; f = nf;
        }
        if (!more) break;
    }
}

// ---------------------------------------------------------------------------- tiled gather on the bf16 mirror (matrix cores)
// The bf16 mode of the same work items (PIPS_FLAG_BF16_MAPS + a dense query set).  Under torch.autocast the reference
// correlates bf16 operands (`torch.matmul(fmap1, fmap2s)`, nets/pips.py:394-397: BOTH the track features and the maps are cast
// to bf16), so the faithful arithmetic is a bf16 x bf16 -> fp32 product -- which is the matrix cores' native form, 16x the
// rate of the vector ALUs' fp32 FMAs.  That changes the right formulation: instead of lane = window pixel with one FMA chain
// per (particle, level, pixel) out of LDS fragments (gather_tiled_kernel: 4 B of LDS traffic per FMA, bound by LDS -> VGPR
// bandwidth), the tile's region is multiplied against ALL of the item's particles,
//        D[region pixel][particle] = sum_c map[pixel][c] * feat[particle][c]        (v_mfma_f32_32x32x16_bf16)
// -- CorrBlock.corr restricted to the tile -- and each particle then keeps the 8 x 8 window it needs.  The products outside the
// windows are redundant (11-33 % of a block's products land in a window) and still cost a fraction of the vector-ALU form.
//   * ONE persistent block of 16 waves per compute unit, the waves SPECIALISED: 12 product waves and 4 loader waves, each role with
//     its own loop over the batches of GM_ENTS work items.  The first cut (every wave loading, multiplying and storing; two blocks
//     per compute unit for overlap) ran 350 us where its parts, timed alone, needed 65 (map loads) + 75 (products) + 40 (stores) +
//     42 (the rest): a wave's vector-memory counter is in order, so a wait for map loads also waited for the tap stores issued
//     before them, and every step exposed a memory round trip.  Now a loader wave issues nothing but loads, a product wave none;
//   * batch head (first loader wave, lane = item): the items' geometry and the loaders' address table (gm_geo_store) into LDS;
//   * a batch's items form ONE stream of 32-KiB elements: per item its run of bf16 feature rows and records (embed_rows_kernel and
//     bin_particles_kernel write them in the sorted order, so the run is contiguous), then its map chunks -- the region of each
//     level cut into pixel blocks of 8 x 4 pixels = the 32 rows of one MFMA, four blocks per chunk.  In the step that consumes
//     element q the loaders write element q + 1 (requested two steps earlier) into LDS (feature / record buffers, or stage buffer
//     (q + 1) & 1) and request element q + 3; ONE barrier per step.  256-byte rows with the 16-byte chunk index XORed with
//     (row & 15): fragment reads and staging writes are conflict-free;
//   * the product waves issue no load: a wave = (particle block pb of 32, block-in-chunk) reads the 8 B fragments of ITS lane's
//     particle from the feature buffer in the item's first step (which also hosts the previous item's last blend) and skips a
//     pixel block that no window of its 32 particles reaches;
//   * the accumulator layout does the window test almost for free: a lane holds, for ITS particle (column), the 4 x 4 pixels
//     x = 4 half + (r & 3), y = r >> 2 of the block, so the window coordinate of register r is (dx0 + (r & 3), dy0 + (r >> 2)) with
//     ONE (dx0, dy0) per lane and block, the target address in the per-level window buffer win[particle][8][8] (+1 float of
//     padding per particle: the 32 lanes of a write are 32 particles) is one base + immediates, and the validity of a value is
//     the AND of an x- and a y-mask: 16 ds_write_b32 under execution masks;
//   * two window buffers (level parity): the 2 x 2 blend of level l's 8 x 8 correlations to the 49 taps, in the reference's
//     transposed order (k = ix * 7 + iy, :379-381) with the weights, scaling and operation order of gather_tiled_kernel's
//     epilogue, by (particle, iy) rows of 7 taps, runs at the top of the step after the level's last chunk (the last level's in the
//     next item's first step).  A window pixel outside the map is never written: the blend tests its neighbours against the map (zeros padding,
//     :324) instead of clearing the buffer.
// DESIGN.md 4f has the measurements that led here (tools/gm_trace.py).
constexpr int GM_PWAVES = 12, GM_LWAVES = 4, GM_WAVES = GM_PWAVES + GM_LWAVES, GM_THREADS = GM_WAVES * 64;
constexpr int GM_PTHREADS = GM_PWAVES * 64, GM_LTHREADS = GM_LWAVES * 64;
constexpr int GM_PB = GMAX / 32;                  // particle blocks per item
constexpr int GM_CHUNK = GM_PWAVES / GM_PB;       // pixel blocks per chunk = product waves per particle block (4)
constexpr int GM_BLK_BYTES = 32 * C * 2;          // 8 KiB: 32 pixels x 128 channels bf16
constexpr int GM_STAGE = GM_CHUNK * GM_BLK_BYTES; // 32 KiB
constexpr int GM_WIN_ROW = 65;                    // floats per particle window: 64 + 1, so that the 32 particles (lanes) of a scatter hit 32 banks
constexpr int GM_WIN_BYTES = GMAX * GM_WIN_ROW * 4;
constexpr int GM_WIN_OFF = 2 * GM_STAGE + 128;    // (the scatter's per-lane base may lie up to 108 bytes below a particle's window)
constexpr int GM_REC_OFF = GM_WIN_OFF + 2 * GM_WIN_BYTES;
constexpr int GM_FEAT_OFF = GM_REC_OFF + 2 * GMAX * PIPS_LEVELS * 16;      // (two record buffers: item parity)
constexpr int GM_ENTS = 16;                       // work items looked up at a time (a batch; BASELINE configs[3] has 8 per block: tests/test_kernels_gpu.py
                                                  // ::test_gather_mfma_batches covers blocks that walk several batches)
constexpr int GM_ENT_OFF = GM_FEAT_OFF + GMAX * C * 2;
constexpr int GM_CHUNKS_MAX = 16;                 // stream elements per item: its features / records + its chunks (TS = 16: 5 + 4 + 2 + 2 at most)
constexpr int GM_CT_OFF = GM_ENT_OFF + GM_ENTS * 64;             // per item and stream element: four 64-bit source addresses (pieces 0, 2, 4, 6)
constexpr int GM_LDS = GM_CT_OFF + GM_ENTS * GM_CHUNKS_MAX * 32;
constexpr int GM_TAPS = 49;                       // (2 r + 1)^2 taps per level (PIPS_NCORR = 4 x 49 is a mixer row's whole correlation block)
constexpr int GM_PIECES = GM_CHUNK * 32 * 16 / GM_LTHREADS;                        // 16-byte pieces per loader thread and chunk (8)
static_assert(GM_PB * GM_CHUNK == GM_PWAVES && GM_CHUNK * 32 * 16 % GM_LTHREADS == 0, "wave <-> (particle block, block of the chunk)");
static_assert(GM_ENTS <= 64 && GM_LDS <= 160 * 1024 && GM_WIN_OFF % 16 == 0 && GM_REC_OFF % 16 == 0 && GM_FEAT_OFF % 16 == 0, "LDS layout");

#ifndef GM_ABLATE
#define GM_ABLATE 0      // debugging builds only (timing, wrong results): 1 no loads in the loaders, 2 no products / scatter, 4 no tap stores,
                         // 8 no blend, 16 batch heads only, 32 no LDS writes in the loaders, 64 every map chunk read from one place, 128 no fragment reads in the products
#endif
typedef __bf16 bf16x8_gm __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_gm __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 GM_LD16(const char* p) {
    if (GM_ABLATE & 1) return make_uint4((unsigned)(uintptr_t)p, 0u, 0u, 0u);
    const u32x4_gm t = *reinterpret_cast<const u32x4_gm*>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
#ifdef GM_TRACE          // tuning builds (tools/gm_trace.py): time stamps of waves 0 (product) and 12 (loader) of blocks 0 and 1
__device__ unsigned long long* g_gm_trace;
constexpr int GM_TRN = 8192;
#define GM_T(tag_) do { if (trp && tn < GM_TRN) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) trp[tn] = (t_ << 8) | (unsigned)(tag_); ++tn; } } while (0)
#else
#define GM_T(tag_) do { } while (0)
#endif
#ifndef GM_ROLE
#define GM_ROLE 0       // compile-time probe of one role's register use: 1 loader only, 2 product only
#endif
#ifndef GM_BU
#define GM_BU 2          // taps per thread and pass of the blend (3: same time, 8 more registers; 4 spills)
#endif
#ifndef GM_NOSKIP
#define GM_NOSKIP 0     // 1: no skipping of (pixel block, particle block) pairs without a window (probe)
#endif

// the level's staged region: the same rectangle as lane_geom() (window reach of every particle binned into the tile), cut into
// pixel blocks of 8 x 4.  P = x0 | y0 << 16 (map coordinates of the region's corner), Q = RW | RH << 8 | nbx << 16 | nblk << 24
__device__ __forceinline__ void gm_level_geom(int l, int tx, int ty, int Wl, int Hl, int& P, int& Q) {
    const int Tx = (tx * TS) >> l, Ty = (ty * TS) >> l, w = TS >> l, h = l == 0 ? 3 : 4;
    int x0 = max(Tx - h, 0), y0 = max(Ty - h, 0);
    const int x1 = min(Tx + w + h, Wl - 1), y1 = min(Ty + w + h, Hl - 1);
    int RW = max(x1 - x0 + 1, 1), RH = max(y1 - y0 + 1, 1);
    if (x1 < x0 || y1 < y0) { x0 = y0 = 0; RW = RH = 1; }   // (tile beyond this level's map)
    const int nbx = (RW + 7) >> 3, nblk = nbx * ((RH + 3) >> 2);
    P = x0 | (y0 << 16);
    Q = RW | (RH << 8) | (nbx << 16) | (nblk << 24);
}

// per-level geometry of a work item (wave-uniform; named scalars and packed fields, selected by ternaries: an array indexed by the
// run-time level goes to scratch -- and so does a lambda's closure); a level's blocks fill whole chunks.  f < 0: no item
struct GmGeo { int first, count, f, P0, P1, P2, P3, Q0, Q1, Q2, Q3, cs1, cs2, cs3, nchunks; };
// one lane's entry {tile, first, count, frame} -> the item's geometry, as 4 x int4 (the batch head: lane i works out item i once;
// worked out per item by every wave instead, the scalar code sat on the loaders' path, tools/gm_trace.py)
__device__ __forceinline__ void gm_geo_store(int4* geo, ulonglong4* ct, int4 ev, int N, int tiles_x, int W0, int W1, int W2, int W3, int H0, int H1,
                                             int H2, int H3, unsigned ob0, unsigned ob1, unsigned ob2, unsigned ob3, unsigned long long mirror_a,
                                             unsigned long long featb_a, unsigned long long order_a) {
    const int ty = ev.x / tiles_x, tx = ev.x - ty * tiles_x;
    int P0, P1, P2, P3, Q0, Q1, Q2, Q3;
    gm_level_geom(0, tx, ty, W0, H0, P0, Q0);
    gm_level_geom(1, tx, ty, W1, H1, P1, Q1);
    gm_level_geom(2, tx, ty, W2, H2, P2, Q2);
    gm_level_geom(3, tx, ty, W3, H3, P3, Q3);
    const int cs1 = (((unsigned)Q0 >> 24) + GM_CHUNK - 1) / GM_CHUNK;                 // first chunk of level 1, 2, 3; number of chunks
    const int cs2 = cs1 + (((unsigned)Q1 >> 24) + GM_CHUNK - 1) / GM_CHUNK;
    const int cs3 = cs2 + (((unsigned)Q2 >> 24) + GM_CHUNK - 1) / GM_CHUNK;
    const int nchunks = min(cs3 + (int)((((unsigned)Q3 >> 24) + GM_CHUNK - 1) / GM_CHUNK), GM_CHUNKS_MAX - 1);
    geo[0] = make_int4(ev.y, ev.z, ev.w, nchunks);
    geo[1] = make_int4(P0, P1, P2, P3);
    geo[2] = make_int4(Q0, Q1, Q2, Q3);
    geo[3] = make_int4(cs1, cs2, cs3, 0);
    // the loaders' table, one entry per element of the item's stream: the source ADDRESSES of the thread pieces 0, 2, 4, 6 (the odd
    // pieces follow from them), the element's kind in the low bits of the first (addresses are multiples of 128): bits 0-1 the level
    // of a map chunk, bit 2 = the item's run of bf16 feature rows and records.  Element 0 = that run (sorted order: slot f N +
    // first; pieces 0-5 = 24 KiB of feature rows, pieces 6-7 = 6 KiB of records); element 1 + ci = map chunk ci, piece 2 b = pixel
    // block b (a chunk's blocks past the level's last repeat it: never used).  Worked out here once per item -- per chunk in the
    // loaders the address arithmetic was 1.4 k clocks of scalar code on their path (tools/gm_trace.py)
    if (ev.w < 0) return;
    {
        const unsigned long long slot = (unsigned long long)(ev.w * N + ev.y), fa = featb_a + slot * (C * 2);
        ct[0] = make_ulonglong4(fa | 4ull, fa + 8192ull, fa + 16384ull, order_a + slot * (PIPS_LEVELS * 16));
    }
    int ci = 1;
#pragma unroll
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        const int P = l == 0 ? P0 : (l == 1 ? P1 : (l == 2 ? P2 : P3)), Q = l == 0 ? Q0 : (l == 1 ? Q1 : (l == 2 ? Q2 : Q3));
        const int Wl = l == 0 ? W0 : (l == 1 ? W1 : (l == 2 ? W2 : W3)), Hl = l == 0 ? H0 : (l == 1 ? H1 : (l == 2 ? H2 : H3));
        const unsigned ob = l == 0 ? ob0 : (l == 1 ? ob1 : (l == 2 ? ob2 : ob3));
        const int x0 = P & 0xffff, y0 = (unsigned)P >> 16, nbx = (Q >> 16) & 0xff, nblk = (unsigned)Q >> 24;
        for (int c0 = 0; c0 < nblk && ci < GM_CHUNKS_MAX; c0 += GM_CHUNK, ++ci) {
            unsigned d[GM_CHUNK];
#pragma unroll
            for (int b = 0; b < GM_CHUNK; ++b) {
                const int gb = min(c0 + b, nblk - 1), byi = gb / nbx, bxi = gb - byi * nbx;
                d[b] = ob + (unsigned)((ev.w * Hl + y0 + byi * 4) * Wl + x0 + bxi * 8) * (unsigned)(C * 2);
            }
            ct[ci] = make_ulonglong4((mirror_a + d[0]) | (unsigned long long)l, mirror_a + d[1], mirror_a + d[2], mirror_a + d[3]);
        }
    }
}
// item `it` of the batch, wave-uniform (scalars)
__device__ __forceinline__ GmGeo gm_geo(const int4* geo, int it) {
    GmGeo G;
    const int4* gp = geo + 4 * min(it, GM_ENTS - 1);
    const int4 a = gp[0], b = gp[1], c = gp[2], d = gp[3];
#define GM_RFL(x_) __builtin_amdgcn_readfirstlane(x_)
    G.first = GM_RFL(a.x); G.count = GM_RFL(a.y); G.f = it < GM_ENTS ? GM_RFL(a.z) : -1; G.nchunks = GM_RFL(a.w);
    G.P0 = GM_RFL(b.x); G.P1 = GM_RFL(b.y); G.P2 = GM_RFL(b.z); G.P3 = GM_RFL(b.w);
    G.Q0 = GM_RFL(c.x); G.Q1 = GM_RFL(c.y); G.Q2 = GM_RFL(c.z); G.Q3 = GM_RFL(c.w);
    G.cs1 = GM_RFL(d.x); G.cs2 = GM_RFL(d.y); G.cs3 = GM_RFL(d.z);
#undef GM_RFL
    return G;
}

__global__ __launch_bounds__(GM_THREADS) void gather_mfma_kernel(const unsigned short* __restrict__ mirror, TiledLevels lv,
                                                                const uint4* __restrict__ featb, int N, int max_items, int F,
                                                                const int4* __restrict__ order, const int4* __restrict__ items,
                                                                const int* __restrict__ nitems, int tiles_x,
                                                                float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= GM_PWAVES;                           // (wave-uniform)
    const int ltid = tid - GM_PTHREADS;                              // loader thread id (0..255)
    const int pb = wave % GM_PB, bl = wave / GM_PB;                  // product wave = (particle block, block of the chunk)
    const int xcd = blockIdx.x & 7, J = gridDim.x >> 3, jb = blockIdx.x >> 3;
    int4* rec = reinterpret_cast<int4*>(smem + GM_REC_OFF);
    int4* ent = reinterpret_cast<int4*>(smem + GM_ENT_OFF);
    ulonglong4* ctab = reinterpret_cast<ulonglong4*>(smem + GM_CT_OFF);
    const int jme = pb * 32 + l31;                                   // a product lane's particle (MFMA column) within the item
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);      // (for ds_write in assembly)
    // the levels' map sizes and offsets as scalars (static indices: a dynamically indexed kernel-argument array goes to scratch)
    const int W0 = lv.W[0], W1 = lv.W[1], W2 = lv.W[2], W3 = lv.W[3], H0 = lv.H[0], H1 = lv.H[1], H2 = lv.H[2], H3 = lv.H[3];
    const size_t o0 = lv.off[0], o1 = lv.off[1], o2 = lv.off[2], o3 = lv.off[3];
#ifdef GM_TRACE
    unsigned long long* trp = (g_gm_trace && blockIdx.x < 2 && (wave == 0 || wave == GM_PWAVES)) ? g_gm_trace + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * GM_TRN : nullptr;
    int tn = 0;
#endif
#define GM_SEL4(l_, a0, a1, a2, a3) ((l_) == 0 ? (a0) : ((l_) == 1 ? (a1) : ((l_) == 2 ? (a2) : (a3))))
#define geo_of(it_) gm_geo(ent, (it_))
    typedef GmGeo Geo;
#define GM_LEVEL_OF(G_, ci_) (((ci_) >= G_.cs1) + ((ci_) >= G_.cs2) + ((ci_) >= G_.cs3))
    // ---- a batch = this block's next (up to) GM_ENTS work items: item i of the block is entry jb + i J of the XCD's list (frames xcd,
    //      xcd + 8, ... one after another); lane-parallel look-up by the first loader wave (the loaders have the registers: in a product wave
    //      it costs spills), the items' geometry and chunk table (gm_geo_store) in LDS.  Each role
    //      runs its own loop over the batches (one loop around both roles keeps either role's values alive through the other: spills)
#define GM_BATCH_HEAD(LOOKUP_)                                                                                                      \
        lds_barrier();                                                                                                          \
        if ((LOOKUP_) && lane < GM_ENTS) {                                                                                      \
            int gi = jb + (base + lane) * J, fr = xcd;                                                                          \
            int4 e = make_int4(0, 0, 0, -1);                                                                                    \
            for (; fr < F; fr += 8) {                                                                                           \
                const int n = nitems[fr];                                                                                       \
                if (gi < n) break;                                                                                              \
                gi -= n;                                                                                                        \
            }                                                                                                                   \
            if (fr < F) { e = items[(size_t)fr * max_items + gi]; e.w = fr; }                                                   \
            gm_geo_store(ent + 4 * lane, ctab + GM_CHUNKS_MAX * lane, e, N, tiles_x, W0, W1, W2, W3, H0, H1, H2, H3,               \
                         (unsigned)(o0 * 2), (unsigned)(o1 * 2), (unsigned)(o2 * 2), (unsigned)(o3 * 2),                         \
                         (unsigned long long)reinterpret_cast<uintptr_t>(mirror), (unsigned long long)reinterpret_cast<uintptr_t>(featb), \
                         (unsigned long long)reinterpret_cast<uintptr_t>(order));                                                \
        }                                                                                                                       \
        __syncthreads();                                                                                                        \
        bool more = true;                                                                                                       \
        if (GM_ABLATE & 16) { if (ent[4 * (GM_ENTS - 1)].z < 0) break; continue; }
    if (loader) { if (GM_ROLE == 2) return;
      for (int base = 0;; base += GM_ENTS) {
        GM_BATCH_HEAD(wave == GM_PWAVES)
        {
            // =================================================================== loader waves: nothing but loads (and LDS writes)
            // A batch's items form ONE stream of elements: per item its run of bf16 feature rows and records (24 + 6 KiB, contiguous in
            // the sorted order embed_rows_kernel / bin_particles_kernel wrote them in), then its map chunks (4 pixel blocks = 32 KiB
            // each).  Element q goes global -> register set q & 1 -> LDS (the feature / record buffers, or stage buffer q & 1): in
            // the step that consumes element q, element q + 1 is delivered and element q + 3 requested -- across item boundaries
            // (an item has >= 5 elements).  A thread moves 8 pieces of 16 bytes per element: source = wave-uniform base (the batch
            // head's table) + a per-lane offset; map rows i = ltid >> 4 and i + 16 of every block, 16-byte chunk c = ltid & 15.
            // No mask and no clamp: a slot outside the region holds whatever lies there in the buffer (a slack behind the mirror
            // keeps the last level's last rows inside it, pips_pyramid_floats) -- a window pixel that falls on such a slot lies
            // outside the map, and the blend tests that; feature rows / records past the item's particles are never used either.
            // Requests and deliveries are UNCONDITIONAL (addresses are selected, not instructions) and the two register sets
            // alternate in straight-line code (two steps per loop iteration): behind a conditional request, or a run-time choice of
            // the set, the compiler's wait-count pass must assume the set being delivered was requested last and waits for
            // vmcnt(0) -- i.e. for the request issued one step ago as well, which makes the pipeline one step deep instead of two.
            // Past the batch's last element the stream repeats a valid chunk (never read).
            const unsigned offF = (unsigned)ltid * 16u, offF7 = (unsigned)min(ltid, GMAX * PIPS_LEVELS - 257) * 16u;   // (records: 6 KiB = pieces 6 and 7's first half)
#define GM_REQUEST(itx_, e_, pre) { const ulonglong4 d_ = ctab[(itx_) * GM_CHUNKS_MAX + (e_)]; GM_REQUEST_D(d_, pre) }
            // (d_: the element's table entry -- one address for the wave, a broadcast; the step loop reads it ahead of its delivery, so that
            //  the read does not queue behind the delivery's eight LDS writes)
#define GM_RFL64(x_) (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((x_) >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(x_)))
#define GM_REQUEST_D(d_, pre)                                                                                                   \
            {                                                                                                                   \
                const unsigned long long a0_ = GM_RFL64(d_.x), a2_ = GM_RFL64(d_.y), a4_ = GM_RFL64(d_.z), a6_ = GM_RFL64(d_.w); \
                const bool isf_ = (a0_ & 4ull) != 0;                                                                            \
                const int l_ = (int)(a0_ & 3ull);                                                                               \
                const unsigned offA_ = GM_SEL4(l_, oA0, oA1, oA2, oA3), offB_ = GM_SEL4(l_, oB0, oB1, oB2, oB3);                \
                const unsigned ve_ = isf_ ? offF : offA_, vo_ = isf_ ? offF : offB_, v7_ = isf_ ? offF7 : offB_;                \
                const unsigned long long odd_ = isf_ ? 4096ull : 0ull;               /* (a run's odd pieces: the next 4 KiB; a block's: its rows i + 16) */ \
                const bool fix_ = (GM_ABLATE & 64) && !isf_;                         /* (timing probe: every map chunk = the buffer's first 32 KiB) */ \
                const char* s0_ = fix_ ? reinterpret_cast<const char*>(mirror) : reinterpret_cast<const char*>(a0_ & ~7ull);    \
                const char* s2_ = fix_ ? s0_ + 8192 : reinterpret_cast<const char*>(a2_);                                       \
                const char* s4_ = fix_ ? s0_ + 16384 : reinterpret_cast<const char*>(a4_);                                      \
                const char* s6_ = fix_ ? s0_ + 24576 : reinterpret_cast<const char*>(a6_);                                      \
                /* (loaded as a native vector and re-packed: a struct copy from the selected pointers keeps `pre` in scratch memory) */ \
                pre[0] = GM_LD16(s0_ + ve_); pre[1] = GM_LD16(s0_ + odd_ + vo_); pre[2] = GM_LD16(s2_ + ve_); pre[3] = GM_LD16(s2_ + odd_ + vo_); \
                pre[4] = GM_LD16(s4_ + ve_); pre[5] = GM_LD16(s4_ + odd_ + vo_); pre[6] = GM_LD16(s6_ + ve_); pre[7] = GM_LD16(s6_ + odd_ + v7_); \
            }
            // set `pre` -> LDS: a map chunk into stage buffer par_ (piece p = 2 block + row half at p * 4 KiB + this thread's swizzled
            // slot), a feature / record element into the feature buffer (pieces 0-5: rows 16 p + (ltid >> 4), the same slot
            // formula) and the record buffer rb_ of its item (pieces 6, 7: plain)
#define GM_DELIVER(par_, pre, isf_, rb_)                                                                                        \
            {                                                                                                                   \
                if (GM_ABLATE & 32) { asm volatile("" :: "v"(pre[0].x), "v"(pre[1].x), "v"(pre[2].x), "v"(pre[3].x), "v"(pre[4].x), "v"(pre[5].x), "v"(pre[6].x), "v"(pre[7].x)); } else { \
                char* t05_ = smem + ((isf_) ? GM_FEAT_OFF : (par_) * GM_STAGE) + ldsA;                                          \
                char* t6_ = (isf_) ? smem + GM_REC_OFF + (rb_) * (GMAX * PIPS_LEVELS * 16) + offF : t05_ + 6 * 4096;            \
                char* t7_ = (isf_) ? smem + GM_REC_OFF + (rb_) * (GMAX * PIPS_LEVELS * 16) + 4096 + offF7 : t05_ + 7 * 4096;    \
                *reinterpret_cast<uint4*>(t05_) = pre[0];             *reinterpret_cast<uint4*>(t05_ + 4096) = pre[1];          \
                *reinterpret_cast<uint4*>(t05_ + 2 * 4096) = pre[2];  *reinterpret_cast<uint4*>(t05_ + 3 * 4096) = pre[3];      \
                *reinterpret_cast<uint4*>(t05_ + 4 * 4096) = pre[4];  *reinterpret_cast<uint4*>(t05_ + 5 * 4096) = pre[5];      \
                *reinterpret_cast<uint4*>(t6_) = pre[6];                                                                        \
                *reinterpret_cast<uint4*>(t7_) = pre[7]; }                                                                      \
            }
            const int ldsA = (ltid >> 4) * 256 + (((ltid & 15) ^ ((ltid >> 4) & 15)) << 4);      // row i (and i + 16: same swizzle), chunk c
            uint4 preA[GM_PIECES], preB[GM_PIECES];      // even / odd elements: two requests in flight (a third set: same time, measured)
            // per-lane offsets of rows i and i + 16 (two image rows further down) in a block of level l
            const unsigned oA0 = (unsigned)(((ltid >> 7) * W0 + ((ltid >> 4) & 7)) * (C * 2) + (ltid & 15) * 16), oB0 = oA0 + (unsigned)(2 * W0 * C * 2),
                           oA1 = (unsigned)(((ltid >> 7) * W1 + ((ltid >> 4) & 7)) * (C * 2) + (ltid & 15) * 16), oB1 = oA1 + (unsigned)(2 * W1 * C * 2),
                           oA2 = (unsigned)(((ltid >> 7) * W2 + ((ltid >> 4) & 7)) * (C * 2) + (ltid & 15) * 16), oB2 = oA2 + (unsigned)(2 * W2 * C * 2),
                           oA3 = (unsigned)(((ltid >> 7) * W3 + ((ltid >> 4) & 7)) * (C * 2) + (ltid & 15) * 16), oB3 = oA3 + (unsigned)(2 * W3 * C * 2);
            // the loaders need an item's frame (< 0: no item) and number of chunks only
#define GM_ITEM_FN(it_, f_, n_)                                                                                                 \
            {                                                                                                                   \
                const int4 a_ = ent[4 * min((it_), GM_ENTS - 1)];                                                               \
                f_ = (it_) < GM_ENTS ? __builtin_amdgcn_readfirstlane(a_.z) : -1;                                               \
                n_ = __builtin_amdgcn_readfirstlane(a_.w);                                                                      \
            }
            int it = 0, s = 0, f0, nch, fn, nchn;
            GM_ITEM_FN(0, f0, nch)
            GM_ITEM_FN(1, fn, nchn)
            bool kA = false, kB = false;                  // what set A / B holds: a feature / record element?  for which record buffer?
            int rA = 0, rB = 0;
#define GM_LSTEP(par_, pre, k_, r_)                                                                                             \
            {   /* the step consumes element s of item `it` (0: its features, the previous item's last blend; 1 + c: chunk c):    \
                   element s + 1 of the stream delivered, element s + 3 requested */                                            \
                GM_T(20);                                                                                                       \
                const int nel = nch + 1;                                                                                        \
                const bool hasnext = fn >= 0, rnext = s + 3 >= nel;                                                             \
                const int itr = (rnext && hasnext) ? it + 1 : it;                                                               \
                const int er = rnext ? (hasnext ? s + 3 - nel : 1) : s + 3;                                                     \
                const ulonglong4 dn_ = ctab[itr * GM_CHUNKS_MAX + er];                                                          \
                GM_DELIVER(par_, pre, k_, r_)                                                                                   \
                GM_T(21);                                                                                                       \
                GM_REQUEST_D(dn_, pre)                                                                                          \
                k_ = er == 0; r_ = itr & 1;                                                                                     \
                GM_T(22);                                                                                                       \
                if (!done) {     /* (the loop is left at its end only: an exit between the two steps merges their wait states) */ \
                    lds_barrier();                                                                                              \
                    GM_T(23);                                                                                                   \
                    if (++s == nel) {                                                                                           \
                        if (!hasnext) done = true;                                                                              \
                        else { ++it; nch = nchn; GM_ITEM_FN(it + 1, fn, nchn) s = 0; GM_T(12); }                                \
                    }                                                                                                           \
                }                                                                                                               \
            }
            if (f0 < 0) {
                lds_barrier();                                           // (A)
                more = false;
            } else {
                GM_T(1);
                GM_REQUEST(0, 0, preA)                                   // the batch's first item: its features / records and first chunks, exposed
                GM_REQUEST(0, 1, preB)
                GM_DELIVER(0, preA, true, 0)
                GM_REQUEST(0, 2, preA)
                lds_barrier();                                           // (A) the first item's features and records are in LDS
                bool done = false;
                do {
                    GM_LSTEP(1, preB, kB, rB)
                    GM_LSTEP(0, preA, kA, rA)
                } while (!done);
                lds_barrier();                                           // (the product waves' last blend)
                more = it + 1 >= GM_ENTS;
            }
#undef GM_ITEM_FN
#undef GM_LSTEP
        }
        if (!more) break;
      }
#undef GM_REQUEST
#undef GM_REQUEST_D
#undef GM_RFL64
#undef GM_DELIVER
    } else { if (GM_ROLE == 1) return;
      for (int base = 0;; base += GM_ENTS) {
        GM_BATCH_HEAD(false)
        {
            // =================================================================== product waves: LDS, MFMA, tap stores -- no load at all
            // (an item's features and records come in through the loaders' stream: the step that consumes that element is the item's
            // first, and hosts the PREVIOUS item's last blend)
            // the step after a level's last chunk: 2x2 blend of its 8x8 correlations to the 49 taps, k = ix*7 + iy (transposed, :379-381);
            // neighbours outside the map count as zero (:324).  The last level's blend runs in the NEXT item's first step (its records are
            // in the other record buffer, its windows in the buffer level 0 does not use)
#define GM_BLEND(l_, recb_, cnt_)                                                                                               \
            {   /* a thread = one (particle, ix) column of 7 taps k = ix * 7 + iy = 28 contiguous bytes of the mixer row (round 6: a 16-    \
                   and a 12-byte store instead of seven 4-byte ones 28 bytes apart); the two window columns it needs are read once (16 \
                   values instead of 4 per tap); windows wholly inside the map -- all of a wave's, as a rule -- skip the per-pixel tests */ \
                const int Wl = GM_SEL4(l_, W0, W1, W2, W3), Hl = GM_SEL4(l_, H0, H1, H2, H3);                                   \
                const float* winf = reinterpret_cast<const float*>(smem + GM_WIN_OFF + ((l_) & 1) * GM_WIN_BYTES);              \
                for (int idx = tid; idx < ((GM_ABLATE & 8) ? 0 : (cnt_) * 7); idx += GM_PTHREADS) {                            \
                    const int j = idx / 7, ti = idx - j * 7;                                                                    \
                    const int4 r = (recb_)[j * PIPS_LEVELS + (l_)];                                                             \
                    const float* wv = winf + j * GM_WIN_ROW + ti;                                                               \
                    float za[8], zb[8];                                       /* window columns ti, ti + 1, rows 0..7 */         \
                    _Pragma("unroll") for (int c = 0; c < 8; ++c) { za[c] = wv[8 * c]; zb[c] = wv[8 * c + 1]; }                 \
                    const int px = (int)(short)(r.x & 0xffff) + ti, py = (r.x >> 16);     /* map pixel of the column's first north-west neighbour */ \
                    if (__builtin_amdgcn_ballot_w64(!(px >= 0 && px + 1 < Wl && py >= 0 && py + 7 < Hl)) != 0ull) {              \
                        const bool x0in = (unsigned)px < (unsigned)Wl, x1in = (unsigned)(px + 1) < (unsigned)Wl;                \
                        _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                         \
                            const bool yin = (unsigned)(py + c) < (unsigned)Hl;                                                 \
                            za[c] = (yin && x0in) ? za[c] : 0.f;                                                                \
                            zb[c] = (yin && x1in) ? zb[c] : 0.f;                                                                \
                        }                                                                                                       \
                    }                                                                                                           \
                    const float wx = __int_as_float(r.y), wy = __int_as_float(r.z);                                             \
                    const float e = 1.0f - wx, so = 1.0f - wy;                                                                  \
                    const float k128 = 0.08838834764831845f;                  /* the 1/sqrt(128) of :397 rides on the weights */ \
                    const float w0 = __fmul_rn(__fmul_rn(so, e), k128), w1 = __fmul_rn(__fmul_rn(so, wx), k128),                \
                                w2 = __fmul_rn(__fmul_rn(wy, e), k128), w3 = __fmul_rn(__fmul_rn(wy, wx), k128);                \
                    float o_[7];                                                                                                \
                    _Pragma("unroll") for (int tj = 0; tj < 7; ++tj) {                                                          \
                        float o = __fmul_rn(w0, za[tj]);                                                                        \
                        o = fmaf(w1, zb[tj], o); o = fmaf(w2, za[tj + 1], o); o = fmaf(w3, zb[tj + 1], o);                      \
                        o_[tj] = o;                                                                                             \
                    }                                                                                                           \
                    if (!(GM_ABLATE & 4)) {                                   /* (the runs are 4-byte aligned) */                 \
                        typedef float f4u_ __attribute__((ext_vector_type(4), aligned(4)));                                     \
                        typedef float f3u_ __attribute__((ext_vector_type(3), aligned(4)));                                     \
                        float* xo = X + ((size_t)r.w * PIPS_KIN_PAD + C + GM_TAPS * (l_) + 7 * ti);                             \
                        *reinterpret_cast<f4u_*>(xo) = (f4u_){o_[0], o_[1], o_[2], o_[3]};                                      \
                        *reinterpret_cast<f3u_*>(xo + 4) = (f3u_){o_[4], o_[5], o_[6]};                                         \
                    }                                                                                                           \
                }                                                                                                               \
            }
            // the window scatter of one (pixel block, particle block) product: the 16 values under execution masks = (x in the window) & (y in
            // the window): four + four ballots, then per value one scalar AND into exec and the write -- one assembly statement, so that
            // nothing else runs under a partial mask (a compare + select + write per value took twice the instructions).  The values
            // may come straight out of the last MFMA: an MFMA result read by a DS instruction needs 12 wait states (v_mfma_f32_32x32x16_bf16,
            // tools/asm_hazard_lint.py), which the compiler inserts for its own instructions but not in front of an assembly statement:
            // s_nop 10 = 11, + s_mov + s_and (rounds 5: s_nop 15 + s_nop 7)
#define GM_SCATTER(acc, l_, dx0, dy0)                                                                                           \
            {                                                                                                                   \
                const unsigned wbo = lds0 + (unsigned)(GM_WIN_OFF + ((l_) & 1) * GM_WIN_BYTES + jme * (GM_WIN_ROW * 4) + (dy0) * 32 + (dx0) * 4); \
                const unsigned long long mx0 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 0) < 8u), mx1 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 1) < 8u), \
                                         mx2 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 2) < 8u), mx3 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 3) < 8u), \
                                         my0 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 0) < 8u), my1 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 1) < 8u), \
                                         my2 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 2) < 8u), my3 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 3) < 8u); \
                unsigned long long sv;                                                                                          \
                asm volatile("s_nop 10\n\ts_mov_b64 %0, exec\n\t"                                                              \
                "s_and_b64 exec, %1, %5\n\tds_write_b32 %9, %10 offset:0\n\t"                                                   \
                "s_and_b64 exec, %2, %5\n\tds_write_b32 %9, %11 offset:4\n\t"                                                   \
                "s_and_b64 exec, %3, %5\n\tds_write_b32 %9, %12 offset:8\n\t"                                                   \
                "s_and_b64 exec, %4, %5\n\tds_write_b32 %9, %13 offset:12\n\t"                                                  \
                "s_and_b64 exec, %1, %6\n\tds_write_b32 %9, %14 offset:32\n\t"                                                  \
                "s_and_b64 exec, %2, %6\n\tds_write_b32 %9, %15 offset:36\n\t"                                                  \
                "s_and_b64 exec, %3, %6\n\tds_write_b32 %9, %16 offset:40\n\t"                                                  \
                "s_and_b64 exec, %4, %6\n\tds_write_b32 %9, %17 offset:44\n\t"                                                  \
                "s_and_b64 exec, %1, %7\n\tds_write_b32 %9, %18 offset:64\n\t"                                                  \
                "s_and_b64 exec, %2, %7\n\tds_write_b32 %9, %19 offset:68\n\t"                                                  \
                "s_and_b64 exec, %3, %7\n\tds_write_b32 %9, %20 offset:72\n\t"                                                  \
                "s_and_b64 exec, %4, %7\n\tds_write_b32 %9, %21 offset:76\n\t"                                                  \
                "s_and_b64 exec, %1, %8\n\tds_write_b32 %9, %22 offset:96\n\t"                                                  \
                "s_and_b64 exec, %2, %8\n\tds_write_b32 %9, %23 offset:100\n\t"                                                 \
                "s_and_b64 exec, %3, %8\n\tds_write_b32 %9, %24 offset:104\n\t"                                                 \
                "s_and_b64 exec, %4, %8\n\tds_write_b32 %9, %25 offset:108\n\t"                                                 \
                "s_mov_b64 exec, %0"                                                                                            \
                : "=&s"(sv)                                                                                                     \
                : "s"(mx0), "s"(mx1), "s"(mx2), "s"(mx3), "s"(my0), "s"(my1), "s"(my2), "s"(my3), "v"(wbo),                     \
                "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]), "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15])\
                : "memory", "scc");                                                                                             \
            }
            GM_T(1);
            lds_barrier();                                               // (A)
            int g = 0, countp = 0;                                       // stream index of the item's element 0; the previous item's particles
            for (int it = 0; it < GM_ENTS; ++it) {
                const Geo G = geo_of(it);
                if (G.f < 0) { more = false; break; }
                const bool hasnext = geo_of(it + 1).f >= 0;
                const int count = G.count, nchunks = G.nchunks, cs1 = G.cs1, cs2 = G.cs2, cs3 = G.cs3;
                const int4* recp = rec + (it & 1) * (GMAX * PIPS_LEVELS);
                const int4* recq = rec + ((it + 1) & 1) * (GMAX * PIPS_LEVELS);  // the previous item's records
                GM_T(30);
                uint4 bfr[8];                                            // B operand: this lane's particle, channels 16 ks + 8 half ... + 8
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    bfr[ks] = *reinterpret_cast<const uint4*>(smem + GM_FEAT_OFF + jme * 256 + (((ks * 2 + half) ^ (jme & 15)) << 4));
                // ---- the item's first step (its features / records element): the previous item's last level
                if (it > 0) { GM_BLEND(3, recq, countp) GM_T(43); }
                lds_barrier();
                GM_T(44);
                const bool active = pb * 32 < count;                     // (wave-uniform) this wave's particle block holds particles
                for (int s = 0; s < nchunks; ++s) {
                    GM_T(40);
                    // the blend of the level that ended with the previous step, AHEAD of this step's products (behind them it cost 7 %: measured)
                    if (s == cs1 || s == cs2 || s == cs3) {
                        const int l = GM_LEVEL_OF(G, s - 1);
                        GM_BLEND(l, recp, count)
                        GM_T(43);
                    }
                    GM_T(45);
                    if (active && !(GM_ABLATE & 2)) {
                        // ---- products of chunk s: this wave's block of the chunk x its particle block, scattered into the windows
                        const int l = GM_LEVEL_OF(G, s);
                        const int c0 = (s - GM_SEL4(l, 0, cs1, cs2, cs3)) * GM_CHUNK;
                        const int P = GM_SEL4(l, G.P0, G.P1, G.P2, G.P3), Q = GM_SEL4(l, G.Q0, G.Q1, G.Q2, G.Q3);
                        const int nbx = (Q >> 16) & 0xff, nblk = (unsigned)Q >> 24;
                        const int gb = c0 + bl;
                        if (gb < nblk) {
                            const int rx_ = recp[jme * PIPS_LEVELS + l].x;           // window anchor in region coordinates
                            const int bxr = (int)(short)(rx_ & 0xffff) - (P & 0xffff), byr = (rx_ >> 16) - (int)((unsigned)P >> 16);
                            const unsigned inv_nbx = (65536u + (unsigned)nbx - 1u) / (unsigned)nbx;
                            const int byi = (int)(((unsigned)gb * inv_nbx) >> 16), bxi = gb - byi * nbx;
                            const int dx0 = bxi * 8 + 4 * half - bxr, dy0 = byi * 4 - byr;
                            // a lane's 4 x 4 pixels touch its particle's window iff dx0, dy0 in [-3, 7]; particles are binned by 4 x 4
                            // cell, a block of 32 consecutive ones covers part of the tile: a pixel block none of them reaches is skipped
                            const bool hit = (unsigned)(dx0 + 3) < 11u && (unsigned)(dy0 + 3) < 11u && jme < count;       // (slots past the item's particles hold the records behind it)
                            GM_T(46);
                            if (__builtin_amdgcn_ballot_w64(hit) != 0ull || (GM_NOSKIP)) {
                            f32x16 acc;
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                            const char* ap = smem + ((g + s + 1) & 1) * GM_STAGE + bl * GM_BLK_BYTES + l31 * 256;
                            // all eight fragment reads in flight, THEN the MFMAs (round 6: left alone hipcc reads one fragment at a time into one
                            // register set, each MFMA behind its own LDS round trip)
                            uint4 afr[8];
#pragma unroll
                            for (int ks = 0; ks < 8; ++ks)
                                afr[ks] = (GM_ABLATE & 128) ? make_uint4((unsigned)ks, (unsigned)lane, 0u, 0u)            // (timing probe: no fragment reads)
                                                            : *reinterpret_cast<const uint4*>(ap + (((ks * 2 + half) ^ (l31 & 15)) << 4));
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int ks = 0; ks < 8; ++ks)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_gm*>(&afr[ks]),
                                                                              *reinterpret_cast<const bf16x8_gm*>(&bfr[ks]), acc, 0, 0, 0);
#ifdef GM_TRACE
                            { int d_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(d_) : "v"(acc[15])); asm volatile("" :: "s"(d_)); }
                            GM_T(41);
#endif
                            GM_SCATTER(acc, l, dx0, dy0)
                            }
                        }
                    }
                    GM_T(42);
                    lds_barrier();
                    GM_T(44);
                }
                if (!hasnext) {                                          // the batch's last item: its last level, exposed
                    GM_BLEND(3, recp, count)
                    lds_barrier();
                }
                g += nchunks + 1;
                countp = count;
            }
#undef GM_BLEND
        }
        if (!more) break;
      }
    }
#undef GM_BATCH_HEAD
#undef GM_LEVEL_OF
#undef GM_SEL4
#undef geo_of
}

#ifndef PIPS_GM_V_DEFAULT
#define PIPS_GM_V_DEFAULT 1     // the bf16 mode's kernel: 1 gather_mfma_kernel, 2 gather_mfma2_kernel (round 6's re-cut: variant builds, A/B and bit comparison)
#endif
#if PIPS_GM_V_DEFAULT == 2 || defined(PIPS_TUNING)
// ---------------------------------------------------------------------------- gather_mfma2_kernel (round 6; variant builds only)
// NOT in the product library (PIPS_GM_V_DEFAULT == 2 or -DPIPS_TUNING builds it): bit-identical to gather_mfma_kernel and 3-4 % SLOWER on the
// same box (149-153 against 143-149 us, profiles/r6_probe_gather_mfma2.txt) -- kept with its probes because they name what the two share.
// The same work items, products, window scatter and blend arithmetic as gather_mfma_kernel above, re-cut along what its trace showed
// (profiles/r5_probe_gather_mfma_trace_final.txt: of the ~3.3 k clocks of a step the MFMAs are 0.3 k; the loaders spend 2 k on address
// arithmetic, eight register loads and eight ds_write_b128 per thread, the product waves 1.4 k on the blend every third step and ~1.5 k
// on per-step geometry):
//   * the map chunks and the records go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no register round trip, no
//     ds_write, no staging registers; a 1 KiB piece = 4 consecutive pixels of 256 B, the XOR swizzle of the 256-byte rows applied on
//     the GLOBAL side (the lane that fills LDS position (row, p) fetches chunk p ^ (row & 15));
//   * THREE stage buffers: at the top of step q the two request waves ask for the chunk of step q + 2 (16 pieces each: two pixel
//     blocks per wave) and then wait for everything older, i.e. the chunk of step q + 1, requested a whole step ago.  With two buffers the request could only go
//     out once the step before had released its buffer and had to land within ONE step: the first cut ran at the loaded memory
//     latency, 107 us for the requests alone (profiles/r6_probe_gather_mfma2_ablation.txt);
//   * the 2 x 2 blend of a finished level has its own two waves (g2_blend_role): it runs beside the NEXT level's products, one slice
//     per step, off everybody's critical path; a thread = one (particle, ix) column of 7 taps, which are 28 contiguous bytes of the
//     mixer row (16 + 12-byte stores instead of seven 4-byte ones);
//   * the product waves take their particle's bf16 feature row (the B operand) straight from memory into registers, behind the previous
//     item's last products: no feature buffer in LDS (that is where the third stage buffer fits) and no step of its own (13 steps per
//     interior item instead of 14);
//   * per-step geometry comes from the batch head's table (level, block coordinates), the per-level anchors are read once per item.
// Arithmetic and summation order are gather_mfma_kernel's: results are bit-identical (tools/gather_dump.py, tests/test_kernels_gpu.py).
constexpr int G2_NSTAGE = 3;
constexpr int G2_WIN_OFF = G2_NSTAGE * GM_STAGE + 128;               // (the scatter's per-lane base may lie up to 108 bytes below a particle's window)
constexpr int G2_REC_OFF = G2_WIN_OFF + 2 * GM_WIN_BYTES;
constexpr int G2_ENT_OFF = G2_REC_OFF + 2 * GMAX * PIPS_LEVELS * 16; // per item 4 x int4: {first, count, f, nchunks}, {cs1, cs2, cs3, -}, {P0..P3}, {base0..base3}
constexpr int G2_CT_OFF = G2_ENT_OFF + GM_ENTS * 64;                 // per item and chunk uint2 {level | nvalid << 2, (bxi | byi << 4) << 8 b}
constexpr int G2_LDS = G2_CT_OFF + GM_ENTS * GM_CHUNKS_MAX * 8;
static_assert(GMAX * 7 <= GM_PTHREADS, "the blend: one (particle, column) per product thread");
static_assert(G2_LDS <= 160 * 1024 && G2_WIN_OFF % 16 == 0 && G2_REC_OFF % 16 == 0 && G2_ENT_OFF % 16 == 0, "LDS layout");

#ifndef G2_ROLE
#define G2_ROLE 0        // compile-time probe of one role's register use: 1 aux only, 2 product only
#endif
#ifndef G2_ABLATE
#define G2_ABLATE 0      // timing probes (wrong results): 1 no DMA, 2 no products, 4 no tap stores, 8 no blend, 16 no window scatter, 32 no fragment reads
#endif

#ifdef G2_TRACE          // tuning builds (tools/g2_trace.py): clocks per phase of a step, summed over the launch, for the product wave 0 and the aux wave 12 of
                         // blocks 0 and 1 -- accumulated in registers, written once at the end (no store inside the pipeline)
__device__ unsigned long long* g_g2_trace;
#define G2_TDECL unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = __builtin_amdgcn_s_memtime(); int tsteps_ = 0;
#define G2_T(i_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc_[i_] += t_ - tprev_; tprev_ = t_; }
#define G2_TSTEP ++tsteps_;
#define G2_TRACE_SYNC(acc_) { int d_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(d_) : "v"(acc_[15])); asm volatile("" :: "s"(d_)); }
#define G2_TDUMP(role_) if (g_g2_trace && blockIdx.x < 2 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == GM_PWAVES)) { \
        unsigned long long* o_ = g_g2_trace + ((size_t)blockIdx.x * 2 + (role_)) * 16;                                          \
        for (int i_ = 0; i_ < 8; ++i_) o_[i_] = tacc_[i_];                                                                      \
        o_[8] = (unsigned long long)tsteps_; }
#else
#define G2_TDECL
#define G2_T(i_)
#define G2_TSTEP
#define G2_TRACE_SYNC(acc_)
#define G2_TDUMP(role_)
#endif

// lane i of the batch head: item i's geometry and chunk table
__device__ __forceinline__ void g2_geo_store(int4* geo, uint2* ct, int4 ev, int tiles_x, int W0, int W1, int W2, int W3, int H0, int H1,
                                             int H2, int H3, unsigned ob0, unsigned ob1, unsigned ob2, unsigned ob3) {
    const int ty = ev.x / tiles_x, tx = ev.x - ty * tiles_x;
    int P0, P1, P2, P3, Q0, Q1, Q2, Q3;
    gm_level_geom(0, tx, ty, W0, H0, P0, Q0);
    gm_level_geom(1, tx, ty, W1, H1, P1, Q1);
    gm_level_geom(2, tx, ty, W2, H2, P2, Q2);
    gm_level_geom(3, tx, ty, W3, H3, P3, Q3);
    const int cs1 = (((unsigned)Q0 >> 24) + GM_CHUNK - 1) / GM_CHUNK;
    const int cs2 = cs1 + (((unsigned)Q1 >> 24) + GM_CHUNK - 1) / GM_CHUNK;
    const int cs3 = cs2 + (((unsigned)Q2 >> 24) + GM_CHUNK - 1) / GM_CHUNK;
    const int nchunks = min(cs3 + (int)((((unsigned)Q3 >> 24) + GM_CHUNK - 1) / GM_CHUNK), GM_CHUNKS_MAX);
    geo[0] = make_int4(ev.y, ev.z, ev.w, nchunks);
    geo[1] = make_int4(cs1, cs2, cs3, 0);
    geo[2] = make_int4(P0, P1, P2, P3);
    if (ev.w < 0) return;
    // byte offset (in the bf16 mirror) of the region's corner pixel per level
#define G2_BASE(P_, W_, H_, ob_) ((ob_) + (unsigned)((ev.w * (H_) + (int)((unsigned)(P_) >> 16)) * (W_) + ((P_) & 0xffff)) * (unsigned)(C * 2))
    geo[3] = make_int4((int)G2_BASE(P0, W0, H0, ob0), (int)G2_BASE(P1, W1, H1, ob1), (int)G2_BASE(P2, W2, H2, ob2), (int)G2_BASE(P3, W3, H3, ob3));
#undef G2_BASE
    int ci = 0;
#pragma unroll
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        const int Q = l == 0 ? Q0 : (l == 1 ? Q1 : (l == 2 ? Q2 : Q3));
        const int nbx = (Q >> 16) & 0xff, nblk = (unsigned)Q >> 24;
        for (int c0 = 0; c0 < nblk && ci < GM_CHUNKS_MAX; c0 += GM_CHUNK, ++ci) {
            unsigned bxy = 0;
#pragma unroll
            for (int b = 0; b < GM_CHUNK; ++b) {
                const int gb = min(c0 + b, nblk - 1), byi = gb / nbx, bxi = gb - byi * nbx;     // (blocks past the level's last repeat it: never used)
                bxy |= (unsigned)(bxi | (byi << 4)) << (8 * b);
            }
            ct[ci] = make_uint2((unsigned)l | ((unsigned)min(nblk - c0, GM_CHUNK) << 2), bxy);
        }
    }
}

// the two roles are separate functions: each gets its own register allocation (one body with both roles keeps either role's values alive
// through the other: 10 spilled vector registers and 142 spilled scalars in the first cut)
#define G2_SEL4(l_, a0, a1, a2, a3) ((l_) == 0 ? (a0) : ((l_) == 1 ? (a1) : ((l_) == 2 ? (a2) : (a3))))
#define G2_RFL(x_) __builtin_amdgcn_readfirstlane((int)(x_))
    // item `it_` of the batch: f < 0 or it_ >= GM_ENTS: no item
#define G2_ITEM(it_, first_, count_, f_, nch_)                                                                                  \
    { const int4 a_ = ent[4 * min((it_), GM_ENTS - 1)];                                                                         \
      first_ = G2_RFL(a_.x); count_ = G2_RFL(a_.y); f_ = (it_) < GM_ENTS ? G2_RFL(a_.z) : -1; nch_ = G2_RFL(a_.w); }
struct G2Args {
    const unsigned short* mirror; const uint4* featb; const int4* order; const int4* items; const int* nitems; float* X;
    int N, max_items, F, tiles_x;
    int W0, W1, W2, W3, H0, H1, H2, H3;
    unsigned ob0, ob1, ob2, ob3;
};

__device__ __attribute__((noinline)) void g2_aux_role(const G2Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, J = gridDim.x >> 3, jb = blockIdx.x >> 3;
    int4* ent = reinterpret_cast<int4*>(smem + G2_ENT_OFF);
    uint2* ctab = reinterpret_cast<uint2*>(smem + G2_CT_OFF);
    // (the argument block lives in the kernel's private memory: its fields arrive in vector registers -- make them scalars again)
    const int N = G2_RFL(A.N), F = G2_RFL(A.F), W0 = G2_RFL(A.W0), W1 = G2_RFL(A.W1), W2 = G2_RFL(A.W2), W3 = G2_RFL(A.W3),
              H0 = G2_RFL(A.H0), H1 = G2_RFL(A.H1), H2 = G2_RFL(A.H2), H3 = G2_RFL(A.H3), max_items = G2_RFL(A.max_items), tiles_x = G2_RFL(A.tiles_x);
    const unsigned ob0 = (unsigned)G2_RFL(A.ob0), ob1 = (unsigned)G2_RFL(A.ob1), ob2 = (unsigned)G2_RFL(A.ob2), ob3 = (unsigned)G2_RFL(A.ob3);
#define G2_RFLP(T_, p_) reinterpret_cast<T_>((unsigned long long)(unsigned)G2_RFL((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(p_)) | \
                                           ((unsigned long long)(unsigned)G2_RFL((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(p_) >> 32)) << 32))
    const int4* __restrict__ items = G2_RFLP(const int4*, A.items);
    const int* __restrict__ nitems = G2_RFLP(const int*, A.nitems);
    const int aw = wave - GM_PWAVES;                                     // request wave 0..1: the chunk's blocks 2 aw, 2 aw + 1
    // the request stream is ~30 instructions per step and everything else waits for what it fetches: it goes first.  (The aux waves are
    // the youngest of their SIMDs: at equal priority the three product waves beside them won the issue arbitration and a step's eight
    // requests took ~1.9 k clocks instead of ~0.8 k, tools/g2_trace.py)
    __builtin_amdgcn_s_setprio(3);
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    // raw buffer descriptors as plain SGPR values (wave-uniform by construction: the requests below are assembly statements)
    u4v rs_map, rs_rec;
    {
        const unsigned long long pm = (unsigned long long)reinterpret_cast<uintptr_t>(A.mirror), pr = (unsigned long long)reinterpret_cast<uintptr_t>(A.order);
        rs_map = (u4v){(unsigned)G2_RFL((unsigned)pm), (unsigned)G2_RFL((unsigned)(pm >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        rs_rec = (u4v){(unsigned)G2_RFL((unsigned)pr), (unsigned)G2_RFL((unsigned)(pr >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
    }
    // per-lane source offset inside a 1 KiB piece whose first row is 4 k (mod 16): row (lane >> 4), chunk (lane & 15) ^ row
    unsigned vo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) vo[k] = (unsigned)((lane >> 4) * 256 + ((((lane & 15) ^ (4 * k + (lane >> 4))) & 15) << 4));
    const unsigned vlin = (unsigned)lane * 16u;
    const unsigned pitch0 = (unsigned)W0 * (C * 2), pitch1 = (unsigned)W1 * (C * 2), pitch2 = (unsigned)W2 * (C * 2), pitch3 = (unsigned)W3 * (C * 2);
    // LDS-DMA requests are ASSEMBLY statements: behind the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read
    // that follows (it cannot tell the blend's window reads from the stage being filled) -- which makes the pipeline one request deep
    // -- and a waterfall loop around every request whose descriptor it cannot prove uniform.  M0 (the LDS address of a piece) is saved
    // and restored; `s_nop 4`: an SGPR fresh from v_readfirstlane needs 5 wait states before a vector-memory instruction reads it;
    // one SALU instruction sits between every write of M0 and the load that uses it (tools/asm_hazard_lint.py checks the built code).
    // 8 pieces of one pixel block: image rows 0..3 (pitch pt_) x pixel quads 0..1 (1 KiB apart) -> 8 KiB at LDS address lb_
#define G2_DMA8(lb_, so_, pt_)                                                                                                  \
    if (!(G2_ABLATE & 1)) {                                                                                                     \
        unsigned t0_, t1_, ms_;                                                                                                 \
        asm volatile("s_nop 4\n\ts_mov_b32 %[ms], m0\n\t"                                                                      \
                     "s_mov_b32 m0, %[lb]\n\ts_add_u32 %[t1], %[so], 0x400\n\tbuffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"   \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t0], %[so], %[pt]\n\tbuffer_load_dwordx4 %[v1], %[rs], %[t1] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t1], %[t0], 0x400\n\tbuffer_load_dwordx4 %[v2], %[rs], %[t0] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t0], %[t0], %[pt]\n\tbuffer_load_dwordx4 %[v3], %[rs], %[t1] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t1], %[t0], 0x400\n\tbuffer_load_dwordx4 %[v0], %[rs], %[t0] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t0], %[t0], %[pt]\n\tbuffer_load_dwordx4 %[v1], %[rs], %[t1] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t1], %[t0], 0x400\n\tbuffer_load_dwordx4 %[v2], %[rs], %[t0] offen lds\n\t" \
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v3], %[rs], %[t1] offen lds\n\t"                \
                     "s_mov_b32 m0, %[ms]"                                                                                      \
                     : [t0] "=&s"(t0_), [t1] "=&s"(t1_), [ms] "=&s"(ms_)                                                        \
                     : [rs] "s"(rs_map), [lb] "s"(G2_RFL(lb_)), [so] "s"(G2_RFL(so_)), [pt] "s"(G2_RFL(pt_)), [v0] "v"(vo[0]), [v1] "v"(vo[1]), [v2] "v"(vo[2]), [v3] "v"(vo[3]) \
                     : "memory", "scc");                                                                                        \
    }
    // one plain 1 KiB piece (records)
#define G2_DMA1(lb_, so_)                                                                                                       \
    if (!(G2_ABLATE & 1)) {                                                                                                     \
        unsigned ms_;                                                                                                           \
        asm volatile("s_nop 4\n\ts_mov_b32 %[ms], m0\n\ts_mov_b32 m0, %[lb]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[rs], %[so] offen lds\n\t" \
                     "s_mov_b32 m0, %[ms]"                                                                                      \
                     : [ms] "=&s"(ms_) : [rs] "s"(rs_rec), [lb] "s"(G2_RFL(lb_)), [so] "s"(G2_RFL(so_)), [v] "v"(vlin) : "memory");             \
    }
    // chunk s_ of item it_, block `aw` -> stage st_
#define G2_REQUEST_CHUNK(it_, s_, st_)                                                                                          \
    {                                                                                                                           \
        const uint2 m_ = ctab[(it_) * GM_CHUNKS_MAX + (s_)];                                                                    \
        const int4 bs_ = ent[4 * (it_) + 3];                                                                                    \
        const int l_ = G2_RFL(m_.x) & 3, bxy_ = (G2_RFL(m_.y) >> (16 * aw)) & 0xffff;                                           \
        const unsigned pt_ = G2_SEL4(l_, pitch0, pitch1, pitch2, pitch3);                                                       \
        const unsigned bse_ = (unsigned)G2_RFL(G2_SEL4(l_, bs_.x, bs_.y, bs_.z, bs_.w));                                        \
        const unsigned soa_ = bse_ + (unsigned)((bxy_ >> 4) & 15) * 4u * pt_ + (unsigned)(bxy_ & 15) * (8u * C * 2);            \
        const unsigned sob_ = bse_ + (unsigned)((bxy_ >> 12) & 15) * 4u * pt_ + (unsigned)((bxy_ >> 8) & 15) * (8u * C * 2);    \
        const unsigned lb_ = lds0 + (unsigned)((st_) * GM_STAGE + 2 * aw * GM_BLK_BYTES);                                       \
        G2_DMA8(lb_, soa_, pt_)                                                                                                 \
        G2_DMA8(lb_ + (unsigned)GM_BLK_BYTES, sob_, pt_)                                                                        \
    }
    // an item's records (6 plain pieces: three per wave) -> REC[rb_]
#define G2_REQUEST_RECORDS(f_, first_, rb_)                                                                                     \
    {                                                                                                                           \
        const unsigned slot_ = (unsigned)((f_) * N + (first_));                                                                 \
        const unsigned lr_ = lds0 + (unsigned)(G2_REC_OFF + (rb_) * (GMAX * PIPS_LEVELS * 16) + aw * 3072);                     \
        const unsigned sr_ = slot_ * (PIPS_LEVELS * 16) + (unsigned)aw * 3072u;                                                 \
        G2_DMA1(lr_, sr_)                                                                                                       \
        G2_DMA1(lr_ + 1024u, sr_ + 1024u)                                                                                       \
        G2_DMA1(lr_ + 2048u, sr_ + 2048u)                                                                                       \
    }
    G2_TDECL
    for (int base = 0;; base += GM_ENTS) {
        // ---- batch head: this block's next (up to) GM_ENTS work items (gather_mfma_kernel's order), lane-parallel in the first aux wave
        lds_barrier();
        if (aw == 0 && lane < GM_ENTS) {
            int gi = jb + (base + lane) * J, fr = xcd;
            int4 e = make_int4(0, 0, 0, -1);
            for (; fr < F; fr += 8) {
                const int n = nitems[fr];
                if (gi < n) break;
                gi -= n;
            }
            if (fr < F) { e = items[(size_t)fr * max_items + gi]; e.w = fr; }
            g2_geo_store(ent + 4 * lane, ctab + GM_CHUNKS_MAX * lane, e, tiles_x, W0, W1, W2, W3, H0, H1, H2, H3, ob0, ob1, ob2, ob3);
        }
        __syncthreads();
        bool more = true;
        int first0, count0, f0, nch0;
        G2_ITEM(0, first0, count0, f0, nch0)
        (void)count0;
        if (f0 < 0) break;
        // ---- prologue of the batch: the first item's records and first two chunks, exposed
        G2_REQUEST_RECORDS(f0, first0, 0)
        G2_REQUEST_CHUNK(0, 0, 0)
        {
            int f1, first1, count1, nch1;
            G2_ITEM(1, first1, count1, f1, nch1)
            (void)first1; (void)count1; (void)nch1;
            if (nch0 > 1) G2_REQUEST_CHUNK(0, 1, 1)
            else if (f1 >= 0) G2_REQUEST_CHUNK(1, 0, 1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                               // (P)
        G2_T(6)
        int st = 0;                                                  // the stage this step consumes (q mod 3)
        for (int it = 0; it < GM_ENTS; ++it) {
            int first, count, f, nch, firstn, countn, fn, nchn;
            G2_ITEM(it, first, count, f, nch)
            if (f < 0) { more = false; break; }
            G2_ITEM(it + 1, firstn, countn, fn, nchn)
            (void)first; (void)countn; (void)count;
            const int cs1 = G2_RFL(ent[4 * it + 1].x);
            for (int s = 0; s < nch; ++s) {
                // step q: [records of the next item, once] [REQUEST chunk q + 2 -- its stage was released by the barrier of step q - 1]
                // [wait: everything older than that request, i.e. chunk q + 1, requested a step ago: two steps of latency cover] [barrier]
                G2_T(0)
                // (the next item's records: once level 0's steps are over -- the blend waves read the PREVIOUS item's records, in the same
                //  buffer, through all of them)
                if (s == cs1 && fn >= 0) G2_REQUEST_RECORDS(fn, firstn, (it + 1) & 1)
                bool requested = false;
                {
                    const int st2 = st == 0 ? 2 : st - 1;            // (q + 2) mod 3
                    if (s + 2 < nch) { G2_REQUEST_CHUNK(it, s + 2, st2) requested = true; }
                    else if (fn >= 0 && s + 2 - nch < nchn) { G2_REQUEST_CHUNK(it + 1, s + 2 - nch, st2) requested = true; }
                }
                G2_T(2)
                if (requested && !(G2_ABLATE & 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                G2_T(4)
                lds_barrier();
                G2_T(5)
                G2_TSTEP
                st = st == 2 ? 0 : st + 1;
            }
            if (fn < 0) { more = it + 1 >= GM_ENTS; break; }
        }
        if (!more) break;
    }
    G2_TDUMP(1)
#undef G2_DMA8
#undef G2_DMA1
#undef G2_RFLP
#undef G2_REQUEST_CHUNK
#undef G2_REQUEST_RECORDS
}

// The two blend waves: the 2 x 2 blend of a finished level's 8 x 8 correlations to the 49 taps (k = ix * 7 + iy, transposed, :379-381) runs
// beside the products of the NEXT level, in slices -- one per step of that level (it reads the window buffer of the other parity; the
// last level's blend runs beside the next item's level 0).  These waves issue nothing but LDS reads and tap stores, so nobody waits for
// their stores: in the product waves the blend sat on the critical path of every third step (30 of 145 us), in the request waves its
// stores sat in front of every `s_waitcnt vmcnt`.
__device__ __attribute__((noinline)) void g2_blend_role(const G2Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int bt = threadIdx.x - (GM_PWAVES + 2) * 64;                    // 0..127
    int4* rec = reinterpret_cast<int4*>(smem + G2_REC_OFF);
    int4* ent = reinterpret_cast<int4*>(smem + G2_ENT_OFF);
    const int W0 = G2_RFL(A.W0), W1 = G2_RFL(A.W1), W2 = G2_RFL(A.W2), W3 = G2_RFL(A.W3),
              H0 = G2_RFL(A.H0), H1 = G2_RFL(A.H1), H2 = G2_RFL(A.H2), H3 = G2_RFL(A.H3);
#define G2_RFLP(T_, p_) reinterpret_cast<T_>((unsigned long long)(unsigned)G2_RFL((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(p_)) | \
                                           ((unsigned long long)(unsigned)G2_RFL((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(p_) >> 32)) << 32))
    float* __restrict__ X = G2_RFLP(float*, A.X);
#undef G2_RFLP
    // one column: idx_ = particle * 7 + ix of level l_ (records recb_): 7 taps = 28 contiguous bytes of the mixer row (16 + 12-byte stores); the two
    // window columns it needs are read once; neighbours outside the map count as zero (:324) -- a window pixel outside the map is never written,
    // the in-map tests replace clearing the buffer; weights, scaling and operation order of gather_tiled_kernel's epilogue
#define G2_BLEND1(l_, recb_, idx_)                                                                                              \
    {                                                                                                                           \
        const int Wl = G2_SEL4(l_, W0, W1, W2, W3), Hl = G2_SEL4(l_, H0, H1, H2, H3);                                           \
        const float* winf = reinterpret_cast<const float*>(smem + G2_WIN_OFF + ((l_) & 1) * GM_WIN_BYTES);                      \
        const int j = (idx_) / 7, ti = (idx_) - j * 7;                                                                          \
        const int4 r = (recb_)[j * PIPS_LEVELS + (l_)];                                                                         \
        const float* wv = winf + j * GM_WIN_ROW + ti;                                                                           \
        float za[8], zb[8];                                                   /* window columns ti, ti + 1, rows 0..7 */         \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) { za[c] = wv[8 * c]; zb[c] = wv[8 * c + 1]; }                             \
        const int px = (int)(short)(r.x & 0xffff) + ti, py = (r.x >> 16);                                                       \
        /* (windows wholly inside the map -- all of a wave's, as a rule -- skip the per-pixel tests) */                          \
        if (__builtin_amdgcn_ballot_w64(!(px >= 0 && px + 1 < Wl && py >= 0 && py + 7 < Hl)) != 0ull) {                          \
            const bool x0in = (unsigned)px < (unsigned)Wl, x1in = (unsigned)(px + 1) < (unsigned)Wl;                            \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                     \
                const bool yin = (unsigned)(py + c) < (unsigned)Hl;                                                             \
                za[c] = (yin && x0in) ? za[c] : 0.f;                                                                            \
                zb[c] = (yin && x1in) ? zb[c] : 0.f;                                                                            \
            }                                                                                                                   \
        }                                                                                                                       \
        const float wx = __int_as_float(r.y), wy = __int_as_float(r.z);                                                         \
        const float e = 1.0f - wx, so = 1.0f - wy;                                                                              \
        const float k128 = 0.08838834764831845f;                              /* the 1/sqrt(128) of :397 rides on the weights */ \
        const float w0 = __fmul_rn(__fmul_rn(so, e), k128), w1 = __fmul_rn(__fmul_rn(so, wx), k128),                            \
                    w2 = __fmul_rn(__fmul_rn(wy, e), k128), w3 = __fmul_rn(__fmul_rn(wy, wx), k128);                            \
        float o_[7];                                                                                                            \
        _Pragma("unroll") for (int tj = 0; tj < 7; ++tj) {                                                                      \
            float o = __fmul_rn(w0, za[tj]);                                                                                    \
            o = fmaf(w1, zb[tj], o); o = fmaf(w2, za[tj + 1], o); o = fmaf(w3, zb[tj + 1], o);                                  \
            o_[tj] = o;                                                                                                         \
        }                                                                                                                       \
        if (!(G2_ABLATE & 4)) {                                                                                                 \
            /* (a GLOBAL pointer: a flat store also counts in lgkmcnt and would stall every LDS wait; the runs are 4-byte aligned) */ \
            typedef float f4u_ __attribute__((ext_vector_type(4), aligned(4)));                                                 \
            typedef float f3u_ __attribute__((ext_vector_type(3), aligned(4)));                                                 \
            __attribute__((address_space(1))) float* xo_ =                                                                      \
                (__attribute__((address_space(1))) float*)(X + ((size_t)r.w * PIPS_KIN_PAD + C + GM_TAPS * (l_) + 7 * ti));     \
            *reinterpret_cast<__attribute__((address_space(1))) f4u_*>(xo_) = (f4u_){o_[0], o_[1], o_[2], o_[3]};               \
            *reinterpret_cast<__attribute__((address_space(1))) f3u_*>(xo_ + 4) = (f3u_){o_[4], o_[5], o_[6]};                  \
        } else { asm volatile("" :: "v"(o_[0]), "v"(o_[1]), "v"(o_[2]), "v"(o_[3]), "v"(o_[4]), "v"(o_[5]), "v"(o_[6])); }       \
    }
    // the steps of level L_ (s0_ <= s < s1_) host the blend of the level before (lb_, records recb_, cnt_ particles): step k takes the
    // columns [k R, (k + 1) R), R = ceil(columns / steps)
#define G2_BLEVEL(s0_, s1_, lb_, recb_, cnt_)                                                                                   \
    {                                                                                                                           \
        const int ns_ = (s1_) - (s0_), rows_ = (cnt_) * 7, R_ = ns_ > 0 ? (rows_ + ns_ - 1) / ns_ : 0;                          \
        for (int s = (s0_); s < (s1_); ++s) {                                                                                   \
            const int r0_ = (s - (s0_)) * R_, r1_ = min(r0_ + R_, rows_);                                                       \
            if (!(G2_ABLATE & 8))                                                                                               \
                for (int idx = r0_ + bt; idx < r1_; idx += 128) G2_BLEND1(lb_, recb_, idx)                                      \
            lds_barrier();                                                                                                      \
        }                                                                                                                       \
    }
    for (;;) {
        lds_barrier();                                                   // (batch head)
        __syncthreads();
        bool more = true;
        int first0, count0, f0, nch0;
        G2_ITEM(0, first0, count0, f0, nch0)
        (void)count0; (void)nch0; (void)first0;
        if (f0 < 0) break;
        lds_barrier();                                                   // (P)
        int countp = 0;
        for (int it = 0; it < GM_ENTS; ++it) {
            int first, count, f, nch, firstn, countn, fn, nchn;
            G2_ITEM(it, first, count, f, nch)
            if (f < 0) { more = false; break; }
            G2_ITEM(it + 1, firstn, countn, fn, nchn)
            (void)first; (void)firstn; (void)countn; (void)nchn;
            const int4 cs = ent[4 * it + 1];
            const int cs1 = G2_RFL(cs.x), cs2 = G2_RFL(cs.y), cs3 = G2_RFL(cs.z);
            const int4* recp = rec + (it & 1) * (GMAX * PIPS_LEVELS);
            const int4* recq = rec + ((it + 1) & 1) * (GMAX * PIPS_LEVELS);      // the previous item's records
            G2_BLEVEL(0, cs1, 3, recq, (it > 0 ? countp : 0))
            G2_BLEVEL(cs1, cs2, 0, recp, count)
            G2_BLEVEL(cs2, cs3, 1, recp, count)
            G2_BLEVEL(cs3, nch, 2, recp, count)
            countp = count;
            if (fn < 0) {                                                // the batch's last item: its last level, exposed
                if (!(G2_ABLATE & 8))
                    for (int idx = bt; idx < count * 7; idx += 128) G2_BLEND1(3, recp, idx)
                more = it + 1 >= GM_ENTS;
                break;
            }
        }
        if (!more) break;
    }
#undef G2_BLEND1
#undef G2_BLEVEL
}

__device__ __attribute__((noinline)) void g2_product_role(const G2Args& A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = wave % GM_PB, bl = wave / GM_PB;                  // product wave = (particle block, block of the chunk)
    int4* rec = reinterpret_cast<int4*>(smem + G2_REC_OFF);
    int4* ent = reinterpret_cast<int4*>(smem + G2_ENT_OFF);
    uint2* ctab = reinterpret_cast<uint2*>(smem + G2_CT_OFF);
    const int jme = pb * 32 + l31;                                   // this lane's particle (MFMA column) within the item
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    // this lane's share of its particle's bf16 feature row: the B operand, channels 16 ks + 8 half ... + 8 (rows past the item's
    // particles hold the list's next rows -- a slack of GMAX rows lies behind the last frame's)
#define G2_RFLP(T_, p_) reinterpret_cast<T_>((unsigned long long)(unsigned)G2_RFL((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(p_)) | \
                                           ((unsigned long long)(unsigned)G2_RFL((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(p_) >> 32)) << 32))
    const int N = G2_RFL(A.N);
    const char* fbase = G2_RFLP(const char*, A.featb) + (size_t)jme * (C * 2) + half * 16;
#undef G2_RFLP
#define G2_LOAD_FEATS(dst, f_, first_)                                                                                          \
    { const char* p_ = fbase + (size_t)((f_) * N + (first_)) * (C * 2);                                                        \
      _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                                          \
          const u32x4_gm t_ = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_gm*>((const __attribute__((address_space(1))) char*)(p_ + ks * 32)); \
          dst[ks] = make_uint4(t_.x, t_.y, t_.z, t_.w); } }
    G2_TDECL
    for (;;) {
        lds_barrier();                                                   // (batch head: the first aux wave looks the items up)
        __syncthreads();
        bool more = true;
        int first0, count0, f0, nch0;
        G2_ITEM(0, first0, count0, f0, nch0)
        (void)count0; (void)nch0;
        if (f0 < 0) break;
        uint4 bfr[8];
        G2_LOAD_FEATS(bfr, f0, first0)                                   // (exposed: the batch's first item)
        lds_barrier();                                                   // (P)
        int st = 0;                                                      // the stage this step consumes
        for (int it = 0; it < GM_ENTS; ++it) {
            int first, count, f, nch, firstn, countn, fn, nchn;
            G2_ITEM(it, first, count, f, nch)
            if (f < 0) { more = false; break; }
            G2_ITEM(it + 1, firstn, countn, fn, nchn)
            (void)first; (void)countn; (void)nchn;
            const int4 cs = ent[4 * it + 1];
            const int cs1 = G2_RFL(cs.x), cs2 = G2_RFL(cs.y), cs3 = G2_RFL(cs.z);
            const int4 pv = ent[4 * it + 2];
            const int P0 = G2_RFL(pv.x), P1 = G2_RFL(pv.y), P2 = G2_RFL(pv.z), P3 = G2_RFL(pv.w);
            const int4* recp = rec + (it & 1) * (GMAX * PIPS_LEVELS);
            const int an0 = recp[jme * PIPS_LEVELS + 0].x, an1 = recp[jme * PIPS_LEVELS + 1].x, an2 = recp[jme * PIPS_LEVELS + 2].x,
                      an3 = recp[jme * PIPS_LEVELS + 3].x;
            const bool active = pb * 32 < count;
            // the steps go level by level (the level a compile-time constant: what depends on it alone -- this lane's anchor relative to the
            // region, its window's address -- is worked out once per level, and the blend of the level before has a fixed place: the level's
            // first step, AHEAD of that step's products; behind them it cost 7 % in gather_mfma_kernel)
#define G2_PLEVEL(L_, an_, P_, s0_, s1_, UNUSED_)                                                                               \
            {                                                                                                                   \
                const int bxo = 4 * half - ((int)(short)((an_) & 0xffff) - ((P_) & 0xffff)), byo = (int)((unsigned)(P_) >> 16) - ((an_) >> 16); \
                const unsigned wb = lds0 + (unsigned)(G2_WIN_OFF + ((L_) & 1) * GM_WIN_BYTES + jme * (GM_WIN_ROW * 4));          \
                for (int s = (s0_); s < (s1_); ++s) {                                                                           \
                    G2_T(0)                                                                                                     \
                    const uint2 m_ = ctab[it * GM_CHUNKS_MAX + s];                                                              \
                    const int nvalid = (G2_RFL(m_.x) >> 2) & 7, bxy = (G2_RFL(m_.y) >> (8 * bl)) & 0xff;                       \
                    if (active && bl < nvalid && !(G2_ABLATE & 2)) {                                                            \
                        const int dx0 = (bxy & 15) * 8 + bxo, dy0 = (bxy >> 4) * 4 + byo;                                       \
                        G2_PRODUCT(wb, dx0, dy0)                                                                                \
                    }                                                                                                           \
                    /* behind the item's last products: the NEXT item's feature rows into the same registers (a second register set held   \
                       across the item cost the fragment reads their double buffering: 128 registers per wave).  They have the barrier and \
                       the next item's first blend to land */                                                                   \
                    if ((L_) == 3 && s == (s1_) - 1 && fn >= 0) G2_LOAD_FEATS(bfr, fn, firstn)                                  \
                    G2_T(4)                                                                                                     \
                    lds_barrier();                                                                                              \
                    G2_T(5)                                                                                                     \
                    G2_TSTEP                                                                                                    \
                    st = st == 2 ? 0 : st + 1;                                                                                  \
                }                                                                                                               \
            }
            // one (pixel block, particle block) product and its window scatter.  A lane's 4 x 4 pixels touch its particle's window iff dx0, dy0
            // in [-3, 7]; a pixel block no window of the wave's 32 particles reaches is skipped (slots past the item's particles hold the
            // records behind it)
#define G2_PRODUCT(wb_, dx0, dy0)                                                                                               \
            {                                                                                                                   \
                const bool hit = (unsigned)((dx0) + 3) < 11u && (unsigned)((dy0) + 3) < 11u && jme < count;                     \
                G2_T(1)                                                                                                         \
                if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {                                                                 \
                    f32x16 acc;                                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                                \
                    const char* ap = smem + st * GM_STAGE + bl * GM_BLK_BYTES + l31 * 256;                                      \
                    /* all eight fragment reads in flight, THEN the MFMAs (left alone hipcc reads one fragment at a time into one register \
                       set, each MFMA behind its own LDS round trip) */                                                         \
                    uint4 afr[8];                                                                                               \
                    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                            \
                        afr[ks] = (G2_ABLATE & 32) ? make_uint4((unsigned)ks, (unsigned)lane, 0u, 0u)                           \
                                                   : *reinterpret_cast<const uint4*>(ap + (((ks * 2 + half) ^ (l31 & 15)) << 4)); \
                    __builtin_amdgcn_sched_barrier(0);                                                                          \
                    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                            \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_gm*>(&afr[ks]),            \
                                                                      *reinterpret_cast<const bf16x8_gm*>(&bfr[ks]), acc, 0, 0, 0); \
                    G2_TRACE_SYNC(acc)                                                                                          \
                    G2_T(2)                                                                                                     \
                    if (G2_ABLATE & 16) { asm volatile("" :: "v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15])); } else {    \
                    const unsigned wbo = (wb_) + (unsigned)((dy0) * 32 + (dx0) * 4);                                            \
                    const unsigned long long mx0 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 0) < 8u), mx1 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 1) < 8u), \
                                             mx2 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 2) < 8u), mx3 = __builtin_amdgcn_ballot_w64((unsigned)((dx0) + 3) < 8u), \
                                             my0 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 0) < 8u), my1 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 1) < 8u), \
                                             my2 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 2) < 8u), my3 = __builtin_amdgcn_ballot_w64((unsigned)((dy0) + 3) < 8u); \
                    unsigned long long sv;                                                                                      \
                    /* the window scatter: the 16 values under execution masks = (x in the window) & (y in the window).  They come straight  \
                       out of the last MFMA: 12 wait states in front of the first DS read of an accumulator (tools/asm_hazard_lint.py:        \
                       s_nop 10 = 11, + s_mov + s_and) */                                                                       \
                    asm volatile("s_nop 10\n\ts_mov_b64 %0, exec\n\t"                                                          \
                    "s_and_b64 exec, %1, %5\n\tds_write_b32 %9, %10 offset:0\n\t"                                             \
                    "s_and_b64 exec, %2, %5\n\tds_write_b32 %9, %11 offset:4\n\t"                                             \
                    "s_and_b64 exec, %3, %5\n\tds_write_b32 %9, %12 offset:8\n\t"                                             \
                    "s_and_b64 exec, %4, %5\n\tds_write_b32 %9, %13 offset:12\n\t"                                            \
                    "s_and_b64 exec, %1, %6\n\tds_write_b32 %9, %14 offset:32\n\t"                                            \
                    "s_and_b64 exec, %2, %6\n\tds_write_b32 %9, %15 offset:36\n\t"                                            \
                    "s_and_b64 exec, %3, %6\n\tds_write_b32 %9, %16 offset:40\n\t"                                            \
                    "s_and_b64 exec, %4, %6\n\tds_write_b32 %9, %17 offset:44\n\t"                                            \
                    "s_and_b64 exec, %1, %7\n\tds_write_b32 %9, %18 offset:64\n\t"                                            \
                    "s_and_b64 exec, %2, %7\n\tds_write_b32 %9, %19 offset:68\n\t"                                            \
                    "s_and_b64 exec, %3, %7\n\tds_write_b32 %9, %20 offset:72\n\t"                                            \
                    "s_and_b64 exec, %4, %7\n\tds_write_b32 %9, %21 offset:76\n\t"                                            \
                    "s_and_b64 exec, %1, %8\n\tds_write_b32 %9, %22 offset:96\n\t"                                            \
                    "s_and_b64 exec, %2, %8\n\tds_write_b32 %9, %23 offset:100\n\t"                                           \
                    "s_and_b64 exec, %3, %8\n\tds_write_b32 %9, %24 offset:104\n\t"                                           \
                    "s_and_b64 exec, %4, %8\n\tds_write_b32 %9, %25 offset:108\n\t"                                           \
                    "s_mov_b64 exec, %0"                                                                                        \
                    : "=&s"(sv)                                                                                                 \
                    : "s"(mx0), "s"(mx1), "s"(mx2), "s"(mx3), "s"(my0), "s"(my1), "s"(my2), "s"(my3), "v"(wbo),                 \
                    "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]), "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]) \
                    : "memory", "scc");                                                                                         \
                    }                                                                                                           \
                    G2_T(3)                                                                                                     \
                }                                                                                                               \
            }
            G2_PLEVEL(0, an0, P0, 0, cs1, )
            G2_PLEVEL(1, an1, P1, cs1, cs2, )
            G2_PLEVEL(2, an2, P2, cs2, cs3, )
            G2_PLEVEL(3, an3, P3, cs3, nch, )
#undef G2_PLEVEL
#undef G2_PRODUCT
            if (fn < 0) { more = it + 1 >= GM_ENTS; break; }
        }
        if (!more) break;
    }
    G2_TDUMP(0)
#undef G2_LOAD_FEATS
}
#undef G2_SEL4
#undef G2_RFL
#undef G2_ITEM

__global__ __launch_bounds__(GM_THREADS) void gather_mfma2_kernel(const unsigned short* __restrict__ mirror, TiledLevels lv,
                                                                 const uint4* __restrict__ featb, int N, int max_items, int F,
                                                                 const int4* __restrict__ order, const int4* __restrict__ items,
                                                                 const int* __restrict__ nitems, int tiles_x,
                                                                 float* __restrict__ X) {
    G2Args A;
    A.mirror = mirror; A.featb = featb; A.order = order; A.items = items; A.nitems = nitems; A.X = X;
    A.N = N; A.max_items = max_items; A.F = F; A.tiles_x = tiles_x;
    A.W0 = lv.W[0]; A.W1 = lv.W[1]; A.W2 = lv.W[2]; A.W3 = lv.W[3]; A.H0 = lv.H[0]; A.H1 = lv.H[1]; A.H2 = lv.H[2]; A.H3 = lv.H[3];
    A.ob0 = (unsigned)(lv.off[0] * 2); A.ob1 = (unsigned)(lv.off[1] * 2); A.ob2 = (unsigned)(lv.off[2] * 2); A.ob3 = (unsigned)(lv.off[3] * 2);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= GM_PWAVES + 2) { if (G2_ROLE != 2) g2_blend_role(A); }
    else if (wave >= GM_PWAVES) { if (G2_ROLE != 2) g2_aux_role(A); }
    else if (G2_ROLE != 1) g2_product_role(A);
}

#endif   // gather_mfma2_kernel

// ---------------------------------------------------------------------------- host side
static int tiled_max_items(int N, int H8, int W8) { return cdiv(W8, TS) * cdiv(H8, TS) + N / GMAX + 1; }

size_t tiled_gather_scratch_bytes(int B, int N, int H8, int W8) {
    const int F = B * S;
    const int max_items = tiled_max_items(N, H8, W8);
    const size_t ntiles = (size_t)cdiv(W8, TS) * cdiv(H8, TS);
    // (records and bf16 feature rows: GMAX slots of slack behind the last frame -- an item's run is fetched whole)
    return align_up(((size_t)F * N + GMAX) * PIPS_LEVELS * sizeof(int4), 256) + align_up((size_t)F * max_items * sizeof(int4), 256) +
           align_up((size_t)F * sizeof(int), 256) + align_up(ntiles * NW * 32 * sizeof(int), 256) +
           align_up(ntiles * NW * MAXP * 64 * sizeof(unsigned), 256) + align_up((size_t)F * N * sizeof(int), 256) +
           align_up(((size_t)F * N + GMAX) * C * 2, 256);
}

// Selection: dense query sets (on average >= 16 particles per 16x16 level-0 tile) take the tiled kernel;
// PIPS_GATHER_TILED=0/1 forces it off/on.
// The kernel's hard limits (32-bit byte offsets into the maps and into X, the binning histogram in 64 KiB of LDS) are part
// of the selection: a problem beyond them falls back to the direct kernel instead of failing the forward.
bool tiled_gather_fits(int B, int N, int H8, int W8) {
    size_t px = 0;
    for (int l = 0, h = H8, w = W8; l < PIPS_LEVELS; ++l, h /= 2, w /= 2) px += ((size_t)B * S * h * w * C + 63) / 64 * 64;
    return px * 4 < (1ull << 32) && (size_t)B * N * S * PIPS_KIN_PAD * 4 < (1ull << 32) &&
           (size_t)B * S * N * PIPS_LEVELS * sizeof(int4) < (1ull << 32) && H8 < 16384 && W8 < 16384 &&
           ((size_t)33 * cdiv(W8, TS) * cdiv(H8, TS) + 1) * sizeof(int) <= 64 * 1024;
}
// bf16_maps: the bf16 mode's kernel (gather_mfma_kernel) pays from 256 particles per frame on -- BASELINE configs[2] (N = 256, 21 per tile):
// bin 10 + embed 12 + gather 44 us against 77 us for the direct bf16-map kernel, config 3 12.55 -> 12.45 ms (same box, interleaved; round 6).
// Both routes multiply bf16 features with bf16 maps since round 6, so the choice moves a result by fp32 summation order only.
bool tiled_gather_wanted(int B, int N, int H8, int W8, bool bf16_maps) {
    if (!tiled_gather_fits(B, N, H8, W8)) return false;
    const int force = PIPS_TUNE("PIPS_GATHER_TILED", -1);
    if (force >= 0) return force > 0;
    return (long)N >= 16L * cdiv(W8, TS) * cdiv(H8, TS) && N >= (bf16_maps ? 256 : 1024);
}

// mirror != nullptr: the bf16 mode -- the work items run on gather_mfma_kernel, which reads the pyramid's bf16 mirror (element
// offsets of the fp32 levels) and needs no tile tables
int launch_mixer_input_tiled(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B,
                             int S_, const float* ffeats, const float* coords, const float* times, int N,
                             float* X, void* scratch, size_t scratch_bytes, hipStream_t st, hipEvent_t* ev,
                             const unsigned short* mirror) {
    const int F = B * S, H8 = lvlH[0], W8 = lvlW[0];
    PIPS_CHECK_ARG(S_ == S, "tiled gather: the map buffer must hold %d frames per clip", S);
    if (scratch_bytes < tiled_gather_scratch_bytes(B, N, H8, W8)) {
        set_error("tiled gather: scratch %zu < %zu bytes", scratch_bytes, tiled_gather_scratch_bytes(B, N, H8, W8));
        return PIPS_E_WORKSPACE;
    }
    const int tiles_x = cdiv(W8, TS), tiles_y = cdiv(H8, TS), ntiles = tiles_x * tiles_y;
    const int max_items = tiled_max_items(N, H8, W8);
    char* p = (char*)scratch;
    int4* order = (int4*)p; p += align_up(((size_t)F * N + GMAX) * PIPS_LEVELS * sizeof(int4), 256);
    int4* items = (int4*)p; p += align_up((size_t)F * max_items * sizeof(int4), 256);
    int* nitems = (int*)p; p += align_up((size_t)F * sizeof(int), 256);
    int* gpk_tab = (int*)p; p += align_up((size_t)ntiles * NW * 32 * sizeof(int), 256);
    unsigned* doff_tab = (unsigned*)p; p += align_up((size_t)ntiles * NW * MAXP * 64 * sizeof(unsigned), 256);
    int* slot_of = (int*)p; p += align_up((size_t)F * N * sizeof(int), 256);
    uint4* featb = (uint4*)p;                                        // bf16 mode: the features as bf16, in sorted order
    const size_t bin_lds = ((size_t)2 * 16 * ntiles + ntiles + 1) * sizeof(int);
    PIPS_CHECK_ARG(bin_lds <= 64 * 1024, "tiled gather: map too large for the tile histogram");
    PIPS_CHECK_ARG((lvl_off[PIPS_LEVELS - 1] + (size_t)F * lvlH[PIPS_LEVELS - 1] * lvlW[PIPS_LEVELS - 1] * C) * 4 < (1ull << 32) && (size_t)B * N * S * PIPS_KIN_PAD * 4 < (1ull << 32) &&
                       (size_t)F * N * PIPS_LEVELS * sizeof(int4) < (1ull << 32),
                   "tiled gather: maps / features / records beyond 32-bit byte offsets");
    PIPS_CHECK_ARG(H8 < 16384 && W8 < 16384, "tiled gather: map too large for 16-bit window anchors");
    TiledLevels lv;
    FrameStrides fs;
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l];
        fs.b[l] = (unsigned)((size_t)lvlH[l] * lvlW[l] * C * sizeof(float));
    }
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(bin_particles_kernel, dim3(F), dim3(1024), bin_lds, st, coords, N, lv, tiles_x, tiles_y, max_items,
                       order, items, nitems, mirror ? slot_of : nullptr);
    PIPS_CHECK_LAUNCH("bin_particles_kernel");
    if (mirror == nullptr) {
        hipLaunchKernelGGL(tile_table_kernel, dim3(ntiles), dim3(NW * 64), 0, st, lv, tiles_x, gpk_tab, doff_tab);
        PIPS_CHECK_LAUNCH("tile_table_kernel");
    }
    const int M = B * N * S;
    if (ev) (void)hipEventRecord(ev[1], st);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, ffeats, coords, times, M, X,
                       mirror ? slot_of : nullptr, mirror ? featb : nullptr);
    PIPS_CHECK_LAUNCH("embed_rows_kernel");
    {
        static std::atomic<unsigned long long> raised{0};
        const int rc = ensure_dynamic_lds(raised, (const void*)gather_tiled_kernel, LDS_BYTES_V3);
        if (rc != PIPS_OK) return rc;
    }
    // one persistent block per CU, a multiple of 8 so that block id mod 8 stays the XCD
    const int cus = device_cus();
    if (cus <= 0) {
        set_error("tiled gather: cannot query the device");
        return PIPS_E_LAUNCH;
    }
    const int grid = max(cus / 8, 1) * 8;
    if (ev) (void)hipEventRecord(ev[2], st);
    if (mirror != nullptr) {                     // one persistent block of 12 product + 4 loader / aux waves per compute unit
#if PIPS_GM_V_DEFAULT == 2 || defined(PIPS_TUNING)
        if (PIPS_TUNE("PIPS_GATHER_MFMA_V", PIPS_GM_V_DEFAULT) == 2) {                   // (variant builds: round 6's re-cut)
            static std::atomic<unsigned long long> raised_g2{0};
            const int rc2 = ensure_dynamic_lds(raised_g2, (const void*)gather_mfma2_kernel, G2_LDS);
            if (rc2 != PIPS_OK) return rc2;
            hipLaunchKernelGGL(gather_mfma2_kernel, dim3(grid), dim3(GM_THREADS), G2_LDS, st, mirror, lv, featb, N, max_items, F,
                               order, items, nitems, tiles_x, X);
            if (ev) (void)hipEventRecord(ev[3], st);
            PIPS_CHECK_LAUNCH("gather_mfma2_kernel");
            return PIPS_OK;
        }
#endif
        static std::atomic<unsigned long long> raised_gm{0};
        const int rc = ensure_dynamic_lds(raised_gm, (const void*)gather_mfma_kernel, GM_LDS);
        if (rc != PIPS_OK) return rc;
        hipLaunchKernelGGL(gather_mfma_kernel, dim3(grid), dim3(GM_THREADS), GM_LDS, st, mirror, lv, featb, N, max_items, F,
                           order, items, nitems, tiles_x, X);
        if (ev) (void)hipEventRecord(ev[3], st);
        PIPS_CHECK_LAUNCH("gather_mfma_kernel");
        return PIPS_OK;
    }
    hipLaunchKernelGGL(gather_tiled_kernel, dim3(grid), dim3(NW * 64), LDS_BYTES_V3, st, pyramid, fs, S_, ffeats,
                       N, max_items, F, order, items, nitems, gpk_tab, doff_tab, X);
    if (ev) (void)hipEventRecord(ev[3], st);
    PIPS_CHECK_LAUNCH("gather_tiled_kernel");
    return PIPS_OK;
}

}  // namespace pips

#if defined(G2_TRACE) && (PIPS_GM_V_DEFAULT == 2 || defined(PIPS_TUNING))
extern "C" int pips_g2_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_g2_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif
#ifdef GM_TRACE
extern "C" int pips_gm_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_gm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif
#ifdef PIPS_TILED_TRACE
extern "C" int pips_tiled_trace(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pips::g_tiled_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif
