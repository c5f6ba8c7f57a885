// Fused channel-mixing FeedForward of one mixer layer for the bf16-operand mode at large M (BASELINE configs[2]):
//
//     x  <-  x + W2 . gelu(W1 . xn + b1) + b2          (nets/pips.py:102-109 inside PreNormResidual :93-100, 115-118)
//
// with xn = LayerNorm2(x) (bf16, written by the token-mix kernel), W1 (2048 x 512), W2 (512 x 2048), M = B*N*8 rows.
// As two GEMMs (gemm_bf16_asm.hip) the 2048-wide hidden activation goes to HBM and back -- 67 MB written + 67 MB read
// per layer at M = 16384 -- and the down-projection reads and re-writes the residual stream around a K loop that nothing
// else overlaps with: 49.9 + 47.5 us per layer of which the MFMAs need 27.  Here a block owns 64 rows for the whole
// FeedForward and the hidden activation never leaves the CU:
//   * 4 waves, ONE per SIMD (512 registers each).  The residual tile x[64][512] lives in the accumulators of the
//     down-projection (C^T: lane = row, registers = columns; wave w owns columns 128w..128w+127): 128 registers, loaded
//     once (+ b2), stored once.
//   * xn[64][512] (bf16) is staged in LDS once; the hidden dimension goes by chunks of 256: U(c) = xn . W1[chunk]^T + b1
//     (wave tile 64 x 64), GELU, bf16 -> a 64 x 256 LDS tile (double buffered), D(c) = tile . W2[:, chunk]^T accumulated
//     into x.  D runs one chunk behind U, with the GELU of chunk c spread between the MFMA groups of D(c-1).
//   * the WEIGHTS never touch LDS: every element of W1 / W2 is used by exactly one wave (its column range), so each wave
//     streams its own MFMA fragments straight from L2 into registers.  For that the bf16 copies are packed
//     FRAGMENT-MAJOR at weight-pack time (pack_frag_kernel): the 32 rows x 16 K values of one MFMA operand are 1 KiB
//     in lane order, K steps consecutive -- a fragment is one fully coalesced global_load_dwordx4 per wave.  A ring of
//     16 fragment registers per wave keeps ~16 KiB per wave in flight; no barrier is involved in the weight stream.
//   * LDS traffic is the A operand only (2 fragment reads per 4 / 8 MFMAs); rows are padded by 16 B (conflict-free
//     b128 reads); ONE barrier per chunk.
// Per CU and layer: 4 MiB of weights from L2 (64 B/clk = 65 k clocks, as many as the 8192 MFMAs take), 64 KiB of xn and
// 2 x 128 KiB of x from / to HBM.
#include "common.h"
#include "token_mix_mfma.h"

namespace pips {

typedef __bf16 bf16x8_ff __attribute__((ext_vector_type(8)));

#ifndef PIPS_FFN_ROT
#define PIPS_FFN_ROT 1       // tuning builds: 0 = every block walks the hidden chunks in the same order
#endif
#ifndef PIPS_FFN_RING
#define PIPS_FFN_RING 16     // weight fragments (1 KiB each) in flight per wave; divides 64
#endif
#ifndef PIPS_FFN_ROT_MASK
#define PIPS_FFN_ROT_MASK 7  // rotation groups - 1 (blocks of an XCD with equal (id >> 3) & mask walk together)
#define PIPS_FFN_ROT_STEP 1  // chunks between neighbouring groups
#endif
#ifndef PIPS_FFN_NT
#define PIPS_FFN_NT 0        // 1 = non-temporal weight loads
#endif
#ifndef PIPS_FFN_KROT
#define PIPS_FFN_KROT 0      // 1 = blocks start the K loop of every phase at a different K step (fine stagger, keeps L2 locality)
#endif

constexpr int FF_ROWS = 64, FF_D = PIPS_DMIX, FF_HID = 4 * PIPS_DMIX, FF_HC = 256, FF_NCH = FF_HID / FF_HC;
constexpr int FF_XN_PITCH = FF_D * 2 + 16, FF_H_PITCH = FF_HC * 2 + 16;
constexpr int FF_XN_BYTES = FF_ROWS * FF_XN_PITCH, FF_H_BYTES = FF_ROWS * FF_H_PITCH;
constexpr int FF_LDS = FF_XN_BYTES + 2 * FF_H_BYTES;
constexpr int FF_RING = PIPS_FFN_RING;                        // weight fragments in flight per wave
constexpr int FF_UF = 2 * (FF_D / 16), FF_DF = 4 * (FF_HC / 16);      // fragments of a U / D phase per wave: 64, 64

// [N][K] bf16 row-major -> the fragment STREAMS of ffn_fused_kernel.  A fragment = the 32 rows x 16 K values of one MFMA
// operand as 64 lanes x 16 B (1 KiB): lane L = 32 * half + n holds W[row0 + n][k0 + 8 half .. + 7].  Fragments are stored in
// the order a wave consumes them, so a wave's stream of a phase is one contiguous 64 KiB run:
//   which = 0 (W1, 2048 x 512):  [chunk c (8)][wave (4)][K step kt (32)][jn (2)]   rows 256 c + 64 wave + 32 jn, K 16 kt
//   which = 1 (W2, 512 x 2048):  [chunk c (8)][wave (4)][K step kt (16)][j (4)]    rows 128 wave + 32 j,      K 256 c + 16 kt
__global__ __launch_bounds__(256) void pack_ffn_frag_kernel(const unsigned short* __restrict__ src, uint4* __restrict__ dst, int which) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // one 16-byte piece; 2048 * 512 / 8 of them
    if (i >= (size_t)FF_HID * FF_D / 8) return;
    const int L = (int)(i & 63), n = L & 31, half = L >> 5;
    const int frag = (int)(i >> 6);                                        // 0 .. 2047
    const int f = frag & 63, wave = (frag >> 6) & 3, c = frag >> 8;
    int row, k, K;
    if (which == 0) { row = 256 * c + 64 * wave + 32 * (f & 1) + n; k = 16 * (f >> 1); K = FF_D; }
    else { row = 128 * wave + 32 * (f & 3) + n; k = 256 * c + 16 * (f >> 2); K = FF_HID; }
    dst[i] = *reinterpret_cast<const uint4*>(src + (size_t)row * K + k + 8 * half);
}

int launch_pack_frag(const void* src_bf16, void* dst, int N, int K, hipStream_t st) {
    PIPS_CHECK_ARG((N == FF_HID && K == FF_D) || (N == FF_D && K == FF_HID), "pack_frag: only the channel-mix weight shapes");
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(pack_ffn_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned short*>(src_bf16), reinterpret_cast<uint4*>(dst), N == FF_HID ? 0 : 1);
    PIPS_CHECK_LAUNCH("pack_ffn_frag_kernel");
    return PIPS_OK;
}

// TOKMIX = true: the WHOLE mixer layer (nets/pips.py:115-118) in this launch -- token mixing + LayerNorm-2 as the prologue
// (token_mix_mfma.h: each of the four waves takes two of the block's eight particles, side by side), the new residual
// rows go to x (this block re-reads them as the down-projection's accumulator seed: they are its own rows, L2-hot) and the
// LayerNorm-2 output straight into the LDS tile the up-projection reads -- xn never exists in memory.  The weight ring is
// requested before the prologue, so the first 64 KiB of W1 arrive under it.  `xn` is unused then.
template <bool TOKMIX>
__global__ __launch_bounds__(256) void ffn_fused_kernel(const float* __restrict__ arena, MixLayerW L, const unsigned short* __restrict__ xn,
                                                        float* __restrict__ x, const uint4* __restrict__ w1f, const float* __restrict__ b1,
                                                        const uint4* __restrict__ w2f, const float* __restrict__ b2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XN = smem;
    char* HB = smem + FF_XN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const size_t row0 = (size_t)blockIdx.x * FF_ROWS;

    // ---- the weight-fragment streams of this wave: 64 fragments of 1 KiB per phase, contiguous (pack_ffn_frag_kernel):
    // scalar base of the phase + fragment index * 1 KiB (an immediate) + one per-lane 16-byte offset
    const unsigned lane16 = (unsigned)lane * 16u;
    // The blocks of a launch run in lock-step through the SAME 4 MiB of weights: 32 CUs of an XCD asking for the same few
    // L2 lines at the same time serialise on their channels.  The hidden chunks are therefore taken in a rotated order,
    // by block: chunk index c of the loops below is chunk (c + rot) & 7 of the weights (the sum over chunks is order-free
    // up to fp32 rounding; the order is fixed per block, so results stay deterministic).
    const int rot = PIPS_FFN_ROT ? (((int)(blockIdx.x >> 3) & PIPS_FFN_ROT_MASK) * PIPS_FFN_ROT_STEP) & (FF_NCH - 1) : 0;
    auto base_u = [&](int c) __attribute__((always_inline)) -> const char* {
        return reinterpret_cast<const char*>(w1f) + (size_t)(((c + rot) & (FF_NCH - 1)) * 4 + wave) * (64 * 1024);
    };
    auto base_d = [&](int c) __attribute__((always_inline)) -> const char* {
        return reinterpret_cast<const char*>(w2f) + (size_t)(((c + rot) & (FF_NCH - 1)) * 4 + wave) * (64 * 1024);
    };
    // fine stagger (PIPS_FFN_KROT): consumption position p of a phase is K step ((p / per) + ks) mod steps of the stream
    const int ks_u = PIPS_FFN_KROT ? ((int)(blockIdx.x >> 3) & (FF_D / 16 - 1)) : 0;
    const int ks_d = PIPS_FFN_KROT ? ((int)(blockIdx.x >> 3) & (FF_HC / 16 - 1)) : 0;
    auto ld = [&](const char* base, int f, bool is_u) __attribute__((always_inline)) -> uint4 {
        typedef unsigned u32x4_ff __attribute__((ext_vector_type(4)));
        const char* a = base + f * 1024 + lane16;
        if (PIPS_FFN_KROT) {
            const int fe = is_u ? ((((f >> 1) + ks_u) & (FF_D / 16 - 1)) * 2 + (f & 1)) : ((((f >> 2) + ks_d) & (FF_HC / 16 - 1)) * 4 + (f & 3));
            a = base + (size_t)fe * 1024 + lane16;
        }
        if (PIPS_FFN_NT) {                              // streamed once per CU: keep it out of the vector L1
            const u32x4_ff v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_ff*>(a));
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        return *reinterpret_cast<const uint4*>(a);
    };
    uint4 wq[FF_RING];
#pragma unroll
    for (int f = 0; f < FF_RING; ++f) wq[f] = ld(base_u(0), f, true);

    if (TOKMIX) {
        // ---- token mixing of particles 2 wave, 2 wave + 1 of the block (tile rows 16 wave .. 16 wave + 15)
        float* const xp[2] = {x + (row0 + 16 * wave + 4 * half) * FF_D + 4 * l31, x + (row0 + 16 * wave + 8 + 4 * half) * FF_D + 4 * l31};
        char* const xr = XN + (16 * wave + 4 * half) * FF_XN_PITCH + 8 * l31;
        token_mix_mfma_particles<2>(arena, L, xp, [&](int i, int r, int g, uint2 v) __attribute__((always_inline)) {
            *reinterpret_cast<uint2*>(xr + (8 * i + r) * FF_XN_PITCH + g * 256) = v;
        }, l31, half);
        // the block's other waves read these rows of x below (the down-projection's seed)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    } else {
        // ---- xn tile -> LDS (rows padded by 16 B)
#pragma unroll
        for (int it = 0; it < FF_ROWS * (FF_D / 8) / 256; ++it) {
            const int i = tid + it * 256, r = i >> 6, cc = i & 63;
            *reinterpret_cast<uint4*>(XN + r * FF_XN_PITCH + cc * 16) = *reinterpret_cast<const uint4*>(xn + (row0 + r) * FF_D + cc * 8);
        }
    }
    f32x16 ax[2][4];
    float* xrow = x + (row0 + l31) * FF_D + 128 * wave + 4 * half;
    __syncthreads();
    if (TOKMIX) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    f32x16 au[2][2];
    const char* xa = XN + l31 * FF_XN_PITCH + half * 16;              // A fragment of row tile i, K step kt: + i*32*pitch + kt*32
    // U(c): au = b1 + xn . W1[chunk c]^T.  NEXT: what the ring prefetches behind this phase's own fragments
    auto phase_u = [&](int c, const char* own, const char* nxt, bool nxt_is_u) __attribute__((always_inline)) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + ((c + rot) & (FF_NCH - 1)) * FF_HC + 64 * wave + 32 * jn + 8 * g + 4 * half);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    au[i][jn][4 * g] = bb.x; au[i][jn][4 * g + 1] = bb.y; au[i][jn][4 * g + 2] = bb.z; au[i][jn][4 * g + 3] = bb.w;
                }
            }
        uint4 fa[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const uint4*>(xa + i * 32 * FF_XN_PITCH + ((0 + ks_u) & (FF_D / 16 - 1)) * 32);
#pragma unroll
        for (int kt = 0; kt < FF_D / 16; ++kt) {
            if (kt + 1 < FF_D / 16) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[(kt + 1) & 1][i] = *reinterpret_cast<const uint4*>(xa + i * 32 * FF_XN_PITCH + ((kt + 1 + ks_u) & (FF_D / 16 - 1)) * 32);
            }
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int f = 2 * kt + jn, slot = f % FF_RING;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    au[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_ff*>(&wq[slot]),
                                                                        *reinterpret_cast<const bf16x8_ff*>(&fa[kt & 1][i]), au[i][jn], 0, 0, 0);
                wq[slot] = f + FF_RING < FF_UF ? ld(own, f + FF_RING, true) : ld(nxt, f + FF_RING - FF_UF, nxt_is_u);
            }
        }
    };
    // GELU of piece p (0..15) of au -> bf16 -> hidden tile `hb`: 4 consecutive hidden columns of one row per lane
    auto gelu_piece = [&](int p, char* hb) __attribute__((always_inline)) {
        const int i = p >> 3, jn = (p >> 2) & 1, g = p & 3;
        const f2 lo = gelu_exact2((f2){au[i][jn][4 * g], au[i][jn][4 * g + 1]});
        const f2 hi = gelu_exact2((f2){au[i][jn][4 * g + 2], au[i][jn][4 * g + 3]});
        *reinterpret_cast<uint2*>(hb + (32 * i + l31) * FF_H_PITCH + (64 * wave + 32 * jn + 8 * g + 4 * half) * 2) =
            make_uint2(pack2_bf16(lo.x, lo.y), pack2_bf16(hi.x, hi.y));
    };
    // D(c): ax += hidden tile . W2[:, chunk c]^T, with the GELU pieces of the chunk U has just finished in between
    auto phase_d = [&](const char* hb_in, char* hb_out, bool with_gelu, const char* own, const char* nxt, bool nxt_is_u) __attribute__((always_inline)) {
        const char* ha = hb_in + l31 * FF_H_PITCH + half * 16;
        uint4 fh[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fh[0][i] = *reinterpret_cast<const uint4*>(ha + i * 32 * FF_H_PITCH + ((0 + ks_d) & (FF_HC / 16 - 1)) * 32);
#pragma unroll
        for (int kt = 0; kt < FF_HC / 16; ++kt) {
            if (kt + 1 < FF_HC / 16) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fh[(kt + 1) & 1][i] = *reinterpret_cast<const uint4*>(ha + i * 32 * FF_H_PITCH + ((kt + 1 + ks_d) & (FF_HC / 16 - 1)) * 32);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = 4 * kt + j, slot = g % FF_RING;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    ax[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_ff*>(&wq[slot]),
                                                                       *reinterpret_cast<const bf16x8_ff*>(&fh[kt & 1][i]), ax[i][j], 0, 0, 0);
                if (g + FF_RING < FF_DF) wq[slot] = ld(own, g + FF_RING, false);
                else if (nxt != nullptr) wq[slot] = ld(nxt, g + FF_RING - FF_DF, nxt_is_u);
            }
            if (with_gelu) gelu_piece(kt, hb_out);
        }
    };

    // chunk 0: U alone, its GELU alone; chunks 1..7: U(c), then D(c-1) with GELU(c) in between; then D(7)
    phase_u(0, base_u(0), base_u(1), true);
    // ---- residual tile + b2 into the down-projection's accumulators: ax[i][j][4g+e] = x[32i + l31][128 wave + 32j + 8g + 4 half + e]
    // (requested behind U(0): first needed by D(0), a whole U phase later)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4*>(b2 + 128 * wave + 32 * j + 8 * g + 4 * half);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(xrow + (size_t)i * 32 * FF_D + 32 * j + 8 * g);
                ax[i][j][4 * g] = v.x + bb.x; ax[i][j][4 * g + 1] = v.y + bb.y;
                ax[i][j][4 * g + 2] = v.z + bb.z; ax[i][j][4 * g + 3] = v.w + bb.w;
            }
        }
#pragma unroll
    for (int p = 0; p < 16; ++p) gelu_piece(p, HB);
    lds_barrier();
#pragma unroll 1
    for (int c = 1; c < FF_NCH; ++c) {
        char* hb_c = HB + (c & 1) * FF_H_BYTES;
        const char* hb_p = HB + ((c - 1) & 1) * FF_H_BYTES;
        phase_u(c, base_u(c), base_d(c - 1), false);
        phase_d(hb_p, hb_c, true, base_d(c - 1), c + 1 == FF_NCH ? base_d(c) : base_u(c + 1), c + 1 != FF_NCH);
        lds_barrier();
    }
    phase_d(HB + ((FF_NCH - 1) & 1) * FF_H_BYTES, nullptr, false, base_d(FF_NCH - 1), nullptr, false);      // (nothing left to prefetch)

    // ---- the new residual tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(xrow + (size_t)i * 32 * FF_D + 32 * j + 8 * g) =
                    make_float4(ax[i][j][4 * g], ax[i][j][4 * g + 1], ax[i][j][4 * g + 2], ax[i][j][4 * g + 3]);
}

// Whether the bf16 mixer runs each layer as ONE launch (token mixing + FeedForward, ffn_fused_kernel<true>) for M rows: from one
// 64-row block per CU (M = 16384 on 256 CUs: BASELINE configs[2]'s per-GPU share).  Tuning builds: PIPS_MIXER_LAYER=0/1 forces
// the answer, PIPS_FFN_FUSED=1 selects the FeedForward-only form behind the separate token-mix launch (round 3's measured
// alternative: 1-2 % behind the two assembly GEMMs end to end).
int mixer_layer_route(int M) {
    if (M % FF_ROWS != 0) return 0;
    const int cus = device_cus();
    const bool big = cus > 0 && M / FF_ROWS >= cus;
    const int force = PIPS_TUNE("PIPS_MIXER_LAYER", -1);
    (void)big;
    if (force == 1) return 2;                                    // whole layer (tuning builds only: measured SLOWER)
    if (PIPS_TUNE("PIPS_FFN_FUSED", 0)) {
        const int min_blocks = PIPS_TUNE("PIPS_FFN_MIN_BLOCKS", 0);
        if (cus > 0 && M / FF_ROWS >= (min_blocks > 0 ? min_blocks : cus)) return 1;   // FeedForward only
    }
    return 0;
}

int launch_ffn_fused(const float* arena, const MixLayerW& L, bool tokmix, const void* xn_bf16, float* x, const void* w1_frag,
                     const void* w2_frag, int M, hipStream_t st) {
    PIPS_CHECK_ARG(M > 0 && M % FF_ROWS == 0, "ffn_fused: M=%d must be a multiple of %d", M, FF_ROWS);
    static std::atomic<unsigned long long> raised0{0}, raised1{0};
    const void* fn = tokmix ? (const void*)ffn_fused_kernel<true> : (const void*)ffn_fused_kernel<false>;
    const int rc = ensure_dynamic_lds(tokmix ? raised1 : raised0, fn, FF_LDS);
    if (rc != PIPS_OK) return rc;
    if (tokmix)
        hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3(M / FF_ROWS), dim3(256), FF_LDS, st, arena, L, nullptr, x,
                           reinterpret_cast<const uint4*>(w1_frag), arena + L.b1, reinterpret_cast<const uint4*>(w2_frag), arena + L.b2);
    else
        hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3(M / FF_ROWS), dim3(256), FF_LDS, st, arena, L,
                           reinterpret_cast<const unsigned short*>(xn_bf16), x, reinterpret_cast<const uint4*>(w1_frag), arena + L.b1,
                           reinterpret_cast<const uint4*>(w2_frag), arena + L.b2);
    PIPS_CHECK_LAUNCH("ffn_fused_kernel");
    return PIPS_OK;
}

}  // namespace pips
