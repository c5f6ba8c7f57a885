// The three big layer-2 convolutions of the bf16 encoder (nets/pips.py:135-136,173-181: 96 -> 96 channels, 3x3, stride 1, pad 1, at
// 92 x 124 for a 368 x 496 frame) on bf16 channel-last maps, in the style of the four-wave bf16 GEMMs (gemm_bf16_t4.hip):
//
//   conv3x3_c96_t4_kernel   implicit GEMM, 256 pixels x 96 channels per tile on four waves (64 pixels x 96 channels each) of
//                           v_mfma_f32_16x16x32_bf16, operands global -> registers -> LDS two taps ahead, ONE generated assembly
//                           statement (conv_bf16_t4c_asm.inc <- tools/gen_conv_bf16_t4c.py; the schedule is described there).  With the
//                           pixels of a frame numbered row-major, tap (kh, kw) of pixel p is pixel p + (kh-1) W + (kw-1): a tile's A
//                           operand per tap is one contiguous run of 48 KiB; the frame's buffer descriptor returns zeros above /
//                           below the image and drops the stores of a ragged last tile, two flags per staged piece send the pieces
//                           of image column 0 / W-1 out of range for the kw = 0 / kw = 2 taps.  A block walks every `bpf`-th tile of
//                           ONE frame.  Output: raw bf16 map (+ bias); 16-byte stores (the weight rows sit in LDS permuted so that a
//                           lane owns 8 consecutive channels).
//   conv_stats_bf16_kernel  the InstanceNorm partials of that map {sum(x-p), sum((x-p)^2), p, n} per (frame, 256-pixel part,
//                           channel), from the STORED bf16 values -- what torch.autocast's instance_norm sees (it runs in fp32 on the
//                           convolution's bf16 output).  The implicit-GEMM kernels take theirs from the fp32 accumulators; here the
//                           accumulators sit pixel-major in the lanes (per-channel sums would be 16-lane reductions of 48 values per
//                           wave and tile), so a second pass over the 140 MB map does it at the HBM rate instead.
// The register-staged implicit GEMM of gemm_bf16.hip needs 263 us per layer at BASELINE configs[2] (MFMA pipe 0.18-0.24 busy,
// Cout = 96 on a 128-wide tile, 27 K blocks of 32 with a barrier each); [measured] profiles/r4_probe_conv_c96_t4.txt.
#include "common.h"

#ifndef PIPS_T4C_INC
#define PIPS_T4C_INC "conv_bf16_t4c_asm.inc"
#endif
#include PIPS_T4C_INC

namespace pips {

constexpr int T4C_C = 96, T4C_PIX = 256, T4C_ROWB = 256;                   // channels, pixels per tile, bytes of an LDS row (192 + pad)
constexpr int T4C_LDS_A = T4C_PIX * T4C_ROWB, T4C_LDS_W = T4C_C * T4C_ROWB;   // 65 536 + 24 576
constexpr int T4C_LDS = T4C_LDS_A + T4C_LDS_W;

__device__ __forceinline__ unsigned t4c_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#define T4C_LO(ptr) t4c_sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define T4C_HI(ptr) t4c_sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))

// grid = frames x bpf blocks; block (f, b) walks tiles b, b + bpf, ... of frame f
__global__ __launch_bounds__(256) void conv3x3_c96_t4_kernel(const unsigned short* __restrict__ in, const unsigned short* __restrict__ wgt,
                                                             const float* __restrict__ bias, unsigned short* __restrict__ out, int M,
                                                             int Wimg, unsigned invW, int bpf, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int f = blockIdx.x / bpf, b = blockIdx.x - f * bpf;
    const int ntile = (tiles - b + bpf - 1) / bpf;                         // tiles of this block (>= 1: bpf <= tiles)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- the per-thread table of the staging (46 dwords, entry k at 1024 k + 4 tid; read once by the statement, then overwritten):
    //   A piece s (12): q = tid + 256 s -> pixel r = q / 12 of the tile, 16-byte chunk c = q % 12: global offset 16 q (+ tile, tap),
    //                   LDS address of (row r, chunk c ^ (r & 15)), r
    //   W piece s (5):  q = tid + 256 s < 1152 -> LDS row n = q / 12 (MFMA order: column block j = n >> 4, row rho = n & 15 of it), chunk
    //                   c: global row = the channel that position stands for (8 consecutive channels per lane and block pair)
    unsigned* tab = reinterpret_cast<unsigned*>(smem);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        const int q = tid + 256 * s, r = q / 12, c = q - 12 * r;
        tab[(0 + s) * 256 + tid] = 16u * q;
        tab[(12 + s) * 256 + tid] = lds0 + r * T4C_ROWB + ((c ^ (r & 15)) * 16);
        tab[(24 + s) * 256 + tid] = (unsigned)r;
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int q = tid + 256 * s, n = q / 12, c = q - 12 * n;
        const int j = n >> 4, rho = n & 15;
        const int ch = 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
        tab[(36 + s) * 256 + tid] = q < T4C_C * 12 ? (unsigned)(ch * 9 * T4C_C * 2 + c * 16) : 0x80000000u;      // (outside W: zeros)
        tab[(41 + s) * 256 + tid] = lds0 + T4C_LDS_A + (q < T4C_C * 12 ? n * T4C_ROWB + ((c ^ (n & 15)) * 16) : (T4C_C - 1) * T4C_ROWB);   // (slot 0 of row 95: free, its chunks sit at c ^ 15 = 4 .. 15)
    }
    // ---- fragment addresses: lane = row r16 of a 16-row block, K group g; K step ks = chunk 4 ks + g, XORed with the row's key
    unsigned rA[3], rW[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        rA[ks] = lds0 + (64 * wave + r16) * T4C_ROWB + (((4 * ks + g) ^ r16) * 16);         // + i * 4096
        rW[ks] = lds0 + T4C_LDS_A + r16 * T4C_ROWB + (((4 * ks + g) ^ r16) * 16);           // + j * 4096
    }
    const unsigned short* Ab = in + (size_t)f * M * T4C_C;
    unsigned short* Cb = out + (size_t)f * M * T4C_C;
    const unsigned voC = (unsigned)(((64 * wave + r16) * T4C_C + 8 * g) * 2), voB = (unsigned)(32 * g);
    const unsigned tabaddr = lds0 + 4 * tid;
    const unsigned nrec = (unsigned)M * (T4C_C * 2);
    const unsigned p0 = (unsigned)b * T4C_PIX, pstep = (unsigned)bpf * T4C_PIX, plast = (unsigned)(tiles - 1) * T4C_PIX;
    asm volatile(PIPS_T4C_TEXT
                 :
                 : [rA0] "v"(rA[0]), [rA1] "v"(rA[1]), [rA2] "v"(rA[2]), [rW0] "v"(rW[0]), [rW1] "v"(rW[1]), [rW2] "v"(rW[2]),
                   [tab] "v"(tabaddr), [voC] "v"(voC), [voB] "v"(voB), [alo] "s"(T4C_LO(Ab)), [ahi] "s"(T4C_HI(Ab)), [wlo] "s"(T4C_LO(wgt)),
                   [whi] "s"(T4C_HI(wgt)), [clo] "s"(T4C_LO(Cb)), [chi] "s"(T4C_HI(Cb)), [blo] "s"(T4C_LO(bias)), [bhi] "s"(T4C_HI(bias)),
                   [nrec] "s"(t4c_sgpr(nrec)), [p0] "s"(t4c_sgpr(p0)), [pstep] "s"(t4c_sgpr(pstep)), [plast] "s"(t4c_sgpr(plast)),
                   [ntile] "s"(t4c_sgpr((unsigned)ntile)), [imgW] "s"(t4c_sgpr((unsigned)Wimg)), [wm1] "s"(t4c_sgpr((unsigned)(Wimg - 1))),
                   [invW] "s"(t4c_sgpr(invW))
                 : PIPS_T4C_CLOBBER);
}

// InstanceNorm partials of a bf16 channel-last map with C = 96: block = (frame, part of 256 pixels), thread = (8-channel chunk cg of 12,
// pixel lane pl of 21): pixels pl, pl + 21, ... of the part, sums about the part's first pixel; the 21 lanes of a chunk meet in LDS in a
// fixed order (bitwise deterministic).  stats: float4 [frame][parts][96].
__global__ __launch_bounds__(256) void conv_stats_bf16_c96_kernel(const unsigned short* __restrict__ map, int M, int parts, float4* __restrict__ stats) {
    __shared__ float red[2][21][T4C_C];
    const int tid = threadIdx.x, cg = tid % 12, pl = tid / 12;
    const int f = blockIdx.x / parts, part = blockIdx.x - f * parts;
    const int p0 = part * T4C_PIX, n = min(T4C_PIX, M - p0);
    const unsigned short* base = map + ((size_t)f * M + p0) * T4C_C + cg * 8;
    float piv[8], s1[8], s2[8];
    {
        const uint4 v = *reinterpret_cast<const uint4*>(base);                    // the part's first pixel: the pivot of every lane
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { piv[2 * k] = bf16_lo(w[k]); piv[2 * k + 1] = bf16_hi(w[k]); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    if (pl < 21) {
        for (int p = pl; p < n; p += 21) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)p * T4C_C);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = bf16_lo(w[k]) - piv[2 * k], d1 = bf16_hi(w[k]) - piv[2 * k + 1];
                s1[2 * k] += d0; s2[2 * k] = fmaf(d0, d0, s2[2 * k]);
                s1[2 * k + 1] += d1; s2[2 * k + 1] = fmaf(d1, d1, s2[2 * k + 1]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[0][pl][cg * 8 + k] = s1[k]; red[1][pl][cg * 8 + k] = s2[k]; }
    }
    __syncthreads();
    if (tid < T4C_C) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int l = 0; l < 21; ++l) { a += red[0][l][tid]; q += red[1][l][tid]; }
        const unsigned short pv = map[((size_t)f * M + p0) * T4C_C + tid];
        stats[((size_t)f * parts + part) * T4C_C + tid] = make_float4(a, q, __uint_as_float((unsigned)pv << 16), (float)n);
    }
}

// Whether launch_conv_bf16 hands a layer to the pair above: 96 -> 96, 3x3, stride 1, pad 1, bf16 maps in and out, no normalise-on-load,
// and enough tiles for the blocks' tile runs to balance (>= 4 tiles per compute unit).
bool conv_c96_t4_takes(const GemmArgs& a, int frames, int in_bf16, int out_bf16) {
    if (!PIPS_TUNE("PIPS_CONV_C96_T4", 1)) return false;
    // (the generated kernel always reads the bias through a buffer descriptor: a layer without one stays on the register-staged kernel)
    if (!in_bf16 || !out_bf16 || a.in_norm != nullptr || a.bias == nullptr || a.Cin != T4C_C || a.N != T4C_C || a.KH != 3 || a.KW != 3 ||
        a.cstride != 1 || a.pad != 1)
        return false;
    if (a.Ho != a.H || a.Wo != a.Win || a.Win < 3 || (unsigned long long)a.M * T4C_C * 2ull >= (1ull << 31)) return false;
    // pixel -> image row by multiplication with ceil(2^32 / W) is exact only while (M + W) * W < 2^32
    if ((unsigned long long)((unsigned long long)a.M + (unsigned)a.Win + 1ull) * (unsigned)a.Win >= (1ull << 32)) return false;
    const int cus = device_cus();
    const int ptiles = cdiv(a.M, T4C_PIX);
    if (a.stats != nullptr) {                                              // one statistics partial per tile: a route decision, not a late error
        const int cap = a.stats_parts_cap > 0 ? a.stats_parts_cap : 2 * cdiv(a.M, 64) + 4;
        if (ptiles > cap) return false;
    }
    const long tiles = (long)ptiles * frames;
    return cus > 0 && tiles >= 4L * cus;
}

int launch_conv_c96_t4(const GemmArgs& a, int frames, int* parts_out, hipStream_t st) {
    const int tiles = cdiv(a.M, T4C_PIX), cus = device_cus();
    if (a.stats != nullptr) {                                               // (conv_c96_t4_takes already routed such a layer away)
        const int cap = a.stats_parts_cap > 0 ? a.stats_parts_cap : 2 * cdiv(a.M, 64) + 4;
        PIPS_CHECK_ARG(tiles <= cap, "conv_c96_t4: %d statistics partials per frame, room for %d", tiles, cap);
    }
    int bpf = max(1, min(tiles, cdiv(cus, frames)));                        // blocks per frame: about one block per compute unit
    static std::atomic<unsigned long long> raised{0};
    const int rc = ensure_dynamic_lds(raised, (const void*)conv3x3_c96_t4_kernel, T4C_LDS);
    if (rc != PIPS_OK) return rc;
    const unsigned invW = (unsigned)((0x100000000ull + (unsigned)a.Win - 1) / (unsigned)a.Win);      // ceil(2^32 / W): exact rows for p < 2^32 / W
    hipLaunchKernelGGL(conv3x3_c96_t4_kernel, dim3(frames * bpf), dim3(256), T4C_LDS, st, reinterpret_cast<const unsigned short*>(a.A),
                       reinterpret_cast<const unsigned short*>(a.W), a.bias, reinterpret_cast<unsigned short*>(a.C), a.M, a.Win, invW, bpf, tiles);
    PIPS_CHECK_LAUNCH("conv3x3_c96_t4_kernel");
    if (a.stats != nullptr) {
        hipLaunchKernelGGL(conv_stats_bf16_c96_kernel, dim3(frames * tiles), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(a.C), a.M,
                           tiles, reinterpret_cast<float4*>(a.stats));
        PIPS_CHECK_LAUNCH("conv_stats_bf16_c96_kernel");
    }
    if (parts_out) *parts_out = tiles;
    return PIPS_OK;
}

}  // namespace pips
