// The big 3x3 / stride 1 / pad 1 convolutions of the fp32 encoder (nets/pips.py:135-136,173-181,221: 64 -> 64 at 1/2 resolution,
// 96 -> 96 at 1/4, 416 -> 256 at 1/8) as exact-fp32 implicit GEMMs on four waves with a generated static schedule -- the
// convolution counterpart of gemm_f32_t4.hip (round 4).  Same arithmetic and K order as igemm_f32_kernel<..., CONV> (gemm.hip:
// v_mfma_f32_32x32x2_f32, tap-major, 32 channels per stage, accumulators in C layout): the output map is BITWISE that kernel's; the
// InstanceNorm partials come in a different partition (one per wave: 32 NI pixels) and meet in fp64 in inorm_finalize as before.
//
// What changes is how the matrix pipe is fed (igemm_f32_kernel: 97-108 TFLOP/s on these layers): one block per CU,
// one wave per SIMD, wave tile 32 NI pixels x 32 NJ channels, operands global -> registers -> LDS with three stages in flight,
// three LDS buffers and one barrier per stage, a block walking every bpf-th tile of ONE frame with the pipeline running on across
// tiles.  The A operand of a tile and tap is a run of consecutive pixel rows (pixels numbered row-major inside the frame); the
// frame's buffer descriptor supplies the zero padding above / below the image and drops the ragged last tile's stores, two flags
// per staged piece send the pieces of image column 0 / W - 1 out of range for the kw = 0 / kw = 2 taps.  Bodies:
// conv_f32_t4_asm.inc <- tools/gen_conv_f32_t4.py (the schedule and the register map are described there).
#include "common.h"
#ifndef PIPS_CF32T4_INC
#define PIPS_CF32T4_INC "conv_f32_t4_asm.inc"
#endif
#include PIPS_CF32T4_INC

namespace pips {

constexpr int CF4_ROW = 144;                                  // LDS row: 32 channels + 16 bytes

__device__ __forceinline__ unsigned cf4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#define CF4_LO(ptr) cf4_sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define CF4_HI(ptr) cf4_sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))

struct ConvT4Args {
    const float* in; const float* wgt; const float* bias; float* out; float* stats;
    int M, Wimg, bpf, tiles, parts; unsigned invW;
    int by_xcd;      // frames % 8 == 0: blocks are dealt so that all blocks of a frame run on ONE XCD (block & 7 = frame & 7)
};

// CFG 0: 64 -> 64 (tile 256 pixels x 64 channels), 1: 96 -> 96 (128 x 96), 2: 416 -> 256 (128 x 64, four column tiles)
template <int CFG> struct ConvT4Cfg;
template <> struct ConvT4Cfg<0> { static constexpr int CIN = 64, COUT = 64, NI = 2, NJ = 2; };
template <> struct ConvT4Cfg<1> { static constexpr int CIN = 96, COUT = 96, NI = 1, NJ = 3; };
template <> struct ConvT4Cfg<2> { static constexpr int CIN = 416, COUT = 256, NI = 1, NJ = 2; };

// grid = frames x column tiles x bpf blocks; block (f, tn, b) walks tiles b, b + bpf, ... of frame f
template <int CFG>
__global__ __launch_bounds__(256) void conv3x3_f32_t4_kernel(ConvT4Args p) {
    using C = ConvT4Cfg<CFG>;
    constexpr int PX = 128 * C::NI, TN = 32 * C::NJ, NCOL = C::COUT / TN, STAGE = (PX + TN) * CF4_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // Neighbouring tiles share their halo rows (tap row kh = 0 / 2 of a tile is the pixel range of the tile before / behind it, an image
    // row being about a tile long at 1/2 resolution): with the blocks of a frame on one XCD the halo comes out of that XCD's L2
    int f, rem;
    if (p.by_xcd) {
        const int per = 8 * NCOL * p.bpf, grp = blockIdx.x / per, j = blockIdx.x - grp * per;
        f = 8 * grp + (j & 7); rem = j >> 3;
    } else {
        f = blockIdx.x / (NCOL * p.bpf); rem = blockIdx.x - f * (NCOL * p.bpf);
    }
    const int tn = rem / p.bpf, b = rem - tn * p.bpf;
    const int ntile = (p.tiles - b + p.bpf - 1) / p.bpf;                  // tiles of this block (>= 1: bpf <= tiles)
    const int col0 = tn * TN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // staging: thread = (row lr of a 32-row pass, 16-byte chunk lc of the row's 128 bytes); LDS rows 0 .. PX - 1: pixels, PX ..: channels
    const int lr = tid >> 3, lc = tid & 7;
    const unsigned wb0 = lds0 + lr * CF4_ROW + lc * 16, wb1 = wb0 + STAGE, wb2 = wb1 + STAGE;
    const unsigned voA = (unsigned)(lr * C::CIN * 4 + lc * 16), voW = (unsigned)(lr * 9 * C::CIN * 4 + lc * 16);
    // fragments: lane = row l31 of a 32-row block, K values 8 kk + 4 half .. + 3
    const unsigned rA0 = lds0 + (32 * C::NI * wave + l31) * CF4_ROW + half * 16, rA1 = rA0 + STAGE, rA2 = rA1 + STAGE;
    const unsigned rW0 = lds0 + (PX + l31) * CF4_ROW + half * 16, rW1 = rW0 + STAGE, rW2 = rW1 + STAGE;
    const float* Ab = p.in + (size_t)f * p.M * C::CIN;
    const float* Wb = p.wgt + (size_t)col0 * 9 * C::CIN;
    float* Cb = p.out + (size_t)f * p.M * C::COUT;
    float* Sb = p.stats + (size_t)f * p.parts * C::COUT * 4;
    // output: lane = channel col0 + 32 j + l31, register r = pixel 32 NI wave + 32 i + (r & 3) + 8 (r >> 2) + 4 half of the tile
    const unsigned vrow = (unsigned)(32 * C::NI * wave + 4 * half);
    const unsigned voC = (vrow * C::COUT + col0 + l31) * 4u, voB = (unsigned)((col0 + l31) * 4);
    // partials: one float4 per (wave, channel), written by the lower lane half; part = 4 (tile index) + wave
    const unsigned voS = half ? 0x80000000u : (unsigned)((wave * C::COUT + col0 + l31) * 16);
    const unsigned vl31x4 = (unsigned)(l31 * 4), vswap = (unsigned)((lane ^ 32) * 4), vlr = (unsigned)lr;
    const unsigned nrecA = (unsigned)p.M * (C::CIN * 4), nrecC = (unsigned)p.M * (C::COUT * 4), nrecS = (unsigned)p.parts * (C::COUT * 16);
    const unsigned p0 = (unsigned)b * PX, pstep = (unsigned)p.bpf * PX, plast = (unsigned)(p.tiles - 1) * PX;
    const unsigned sbytes = 64u * C::COUT / PX;
#define CF4_OPERANDS                                                                                                               \
    : [rA0] "v"(rA0), [rA1] "v"(rA1), [rA2] "v"(rA2), [rW0] "v"(rW0), [rW1] "v"(rW1), [rW2] "v"(rW2), [wb0] "v"(wb0),            \
      [wb1] "v"(wb1), [wb2] "v"(wb2), [voA] "v"(voA), [voW] "v"(voW), [voB] "v"(voB), [voC] "v"(voC), [voS] "v"(voS),            \
      [vlr] "v"(vlr), [vrow] "v"(vrow), [vl31x4] "v"(vl31x4), [vswap] "v"(vswap), [alo] "s"(CF4_LO(Ab)), [ahi] "s"(CF4_HI(Ab)),   \
      [wlo] "s"(CF4_LO(Wb)), [whi] "s"(CF4_HI(Wb)), [clo] "s"(CF4_LO(Cb)), [chi] "s"(CF4_HI(Cb)), [blo] "s"(CF4_LO(p.bias)),      \
      [bhi] "s"(CF4_HI(p.bias)), [slo] "s"(CF4_LO(Sb)), [shi] "s"(CF4_HI(Sb)), [nrecA] "s"(cf4_sgpr(nrecA)),                      \
      [nrecC] "s"(cf4_sgpr(nrecC)), [nrecS] "s"(cf4_sgpr(nrecS)), [p0] "s"(cf4_sgpr(p0)), [pstep] "s"(cf4_sgpr(pstep)),            \
      [plast] "s"(cf4_sgpr(plast)), [ntile] "s"(cf4_sgpr((unsigned)ntile)), [imgW] "s"(cf4_sgpr((unsigned)p.Wimg)),               \
      [wm1] "s"(cf4_sgpr((unsigned)(p.Wimg - 1))), [invW] "s"(cf4_sgpr(p.invW)), [npix] "s"(cf4_sgpr((unsigned)p.M)),             \
      [sbytes] "s"(cf4_sgpr(sbytes))
    if (CFG == 0)      asm volatile(PIPS_CF32T4_C64_TEXT : CF4_OPERANDS : PIPS_CF32T4_CLOBBER);
    else if (CFG == 1) asm volatile(PIPS_CF32T4_C96_TEXT : CF4_OPERANDS : PIPS_CF32T4_CLOBBER);
    else               asm volatile(PIPS_CF32T4_C416_TEXT : CF4_OPERANDS : PIPS_CF32T4_CLOBBER);
#undef CF4_OPERANDS
}

// Which configuration takes a layer (-1: none -- igemm_f32_kernel): 3x3, stride 1, pad 1, one of the three channel pairs, statistics
// wanted, and at least `min` tiles per compute unit (hook PIPS_CONV_F32_T4_MINPCT, in percent) for the blocks' tile runs to balance.
int conv_f32_t4_config(const GemmArgs& a, int frames) {
    if (!PIPS_TUNE("PIPS_CONV_F32_T4", 1)) return -1;
    if (a.KH != 3 || a.KW != 3 || a.cstride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.Win || a.Win < 3 || a.bias == nullptr ||
        a.stats == nullptr)
        return -1;
    int cfg = -1, px = 0, ncol = 1;
    if (a.Cin == 64 && a.N == 64) { cfg = 0; px = 256; }
    else if (a.Cin == 96 && a.N == 96) { cfg = 1; px = 128; }
    else if (a.Cin == 416 && a.N == 256) { cfg = 2; px = 128; ncol = 4; }
    else return -1;
    if (!((PIPS_TUNE("PIPS_CONV_F32_T4_MASK", 7) >> cfg) & 1)) return -1;
    if ((unsigned long long)a.M * (unsigned)max(a.Cin, a.N) * 4ull >= (1ull << 31)) return -1;
    // pixel -> image row by multiplication with ceil(2^32 / W) is exact only while (M + W) * W < 2^32
    if ((unsigned long long)((unsigned long long)a.M + (unsigned)a.Win + 1ull) * (unsigned)a.Win >= (1ull << 32)) return -1;
    const int cus = device_cus();
    const long tiles = (long)cdiv(a.M, px) * ncol * frames;
    if (cus <= 0 || tiles * 100 < (long)cus * PIPS_TUNE("PIPS_CONV_F32_T4_MINPCT", 250)) return -1;
    const int parts = cdiv(a.M, px / 4);
    const int cap = a.stats_parts_cap > 0 ? a.stats_parts_cap : 2 * cdiv(a.M, 64) + 4;
    return parts <= cap ? cfg : -1;
}

template <int CFG>
static int launch_conv_f32_t4_cfg(const GemmArgs& a, int frames, int* parts_out, hipStream_t st) {
    using C = ConvT4Cfg<CFG>;
    constexpr int PX = 128 * C::NI, TN = 32 * C::NJ, NCOL = C::COUT / TN, LDS = 3 * (PX + TN) * CF4_ROW;
    const int tiles = cdiv(a.M, PX), cus = device_cus();
    const int bpf = max(1, min(tiles, cdiv(cus, frames * NCOL)));          // blocks per frame and column tile: about one block per compute unit
    static std::atomic<unsigned long long> raised{0};
    const int rc = ensure_dynamic_lds(raised, (const void*)conv3x3_f32_t4_kernel<CFG>, LDS);
    if (rc != PIPS_OK) return rc;
    ConvT4Args p;
    p.in = a.A; p.wgt = a.W; p.bias = a.bias; p.out = a.C; p.stats = a.stats;
    p.M = a.M; p.Wimg = a.Win; p.bpf = bpf; p.tiles = tiles; p.parts = cdiv(a.M, PX / 4);
    p.by_xcd = (frames % 8 == 0) && PIPS_TUNE("PIPS_CONV_F32_T4_XCD", 1);
    p.invW = (unsigned)((0x100000000ull + (unsigned)a.Win - 1) / (unsigned)a.Win);      // ceil(2^32 / W): exact rows for p < 2^32 / W
    hipLaunchKernelGGL(conv3x3_f32_t4_kernel<CFG>, dim3(frames * NCOL * bpf), dim3(256), LDS, st, p);
    PIPS_CHECK_LAUNCH("conv3x3_f32_t4_kernel");
    if (parts_out) *parts_out = p.parts;
    return PIPS_OK;
}

int launch_conv_f32_t4(const GemmArgs& a, int cfg, int frames, int* parts_out, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_conv_f32_t4_cfg<0>(a, frames, parts_out, st);
        case 1: return launch_conv_f32_t4_cfg<1>(a, frames, parts_out, st);
        default: return launch_conv_f32_t4_cfg<2>(a, frames, parts_out, st);
    }
}

}  // namespace pips
