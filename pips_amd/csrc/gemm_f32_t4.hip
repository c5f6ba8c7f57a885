// The mixer's exact-fp32 Linears on four waves with a static schedule (round 4):
//
//     C = gelu(A W^T + b)         up-projection   (nets/pips.py:102-109: Linear, GELU)
//     C = R + A W^T + b           down-projection (the second Linear of the FeedForward inside PreNormResidual :93-100)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulation, the K order of igemm_f32_kernel (gemm.hip); what
// changes is how the matrix pipe is kept fed.  igemm_f32_kernel at M = 2048 runs four 64 x 64 blocks per CU (four waves per
// SIMD) whose K loop reaches 70 % of the MFMA rate, plus a prologue and an epilogue per block: 39-41 us per Linear, 0.68 of the
// fp32 MFMA peak.  Here ONE block per CU, one wave per SIMD, every wave a 64 x 64 output block (64 AccVGPR
// accumulators, 0.25 fragment reads per MFMA), operands global -> registers -> LDS two stages ahead, two LDS buffers and one
// barrier per 64 MFMAs, fragments of the next 8 K values read under the MFMAs of the current 8: no wait in the loop is ever
// reached with its data still in flight.  The whole body is generated assembly (gemm_f32_t4_asm.inc <- tools/gen_gemm_f32_t4.py).
//
// Block shapes (the generator's header has the details):
//     U   128 x 128 tile, waves 2 x 2; a block walks `tpb` consecutive row tiles of one column tile
//     D   64 x 64 tile, every wave the whole tile on one of the four groups of 8 K values of each 32-wide stage, the four partial
//         tiles summed through LDS in a fixed order -- when there are too few 128 x 128 tiles for the chip (the down-projection
//         at M = 2048: 64 of them)
//     E   shape D staged by LDS-DMA (no staging registers, no ds_write): what the product runs for shape D's problems; bitwise D
#include "common.h"
#ifndef PIPS_F32T4_INC
#define PIPS_F32T4_INC "gemm_f32_t4_asm.inc"      // tuning builds point this at another schedule of the generator
#endif
#include PIPS_F32T4_INC

namespace pips {

constexpr int F4_ROW = 144;                                  // LDS row: 32 K values + 16 bytes
constexpr int F4_LDS_U = 2 * 256 * F4_ROW;                   // 73 728 bytes: two stages of [A rows 0..127 | W rows 0..127]
constexpr int F4_STAGE_D = 128 * F4_ROW;                     // a stage of shape D: [A rows 0..63 | W rows 0..63]
constexpr int F4_LDS_D = 4 * F4_STAGE_D;                     // 73 728 bytes: four stages; the four partial tiles at the end use 64 KiB of it

__device__ __forceinline__ unsigned f4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#define F4_LO(ptr) f4_sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define F4_HI(ptr) f4_sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))
#define F4_OPERANDS                                                                                                              \
    : [rA0] "v"(rA0), [rW0] "v"(rW0), [rA1] "v"(rA1), [rW1] "v"(rW1), [wA0] "v"(wA0), [wW0] "v"(wW0), [wA1] "v"(wA1),          \
      [wW1] "v"(wW1), [voA] "v"(voA), [voW] "v"(voW), [voR] "v"(voR), [voC] "v"(voC), [voB] "v"(voB), [redW] "v"(redW),        \
      [redR] "v"(redR), [alo] "s"(F4_LO(Ab)), [ahi] "s"(F4_HI(Ab)), [wlo] "s"(F4_LO(Wb)), [whi] "s"(F4_HI(Wb)),                 \
      [rlo] "s"(F4_LO(Rb)), [rhi] "s"(F4_HI(Rb)), [clo] "s"(F4_LO(Cb)), [chi] "s"(F4_HI(Cb)), [blo] "s"(F4_LO(Bb)),             \
      [bhi] "s"(F4_HI(Bb)), [passA] "s"(f4_sgpr(passA)), [passW] "s"(f4_sgpr(passW)), [rstep] "s"(f4_sgpr(rstep)),              \
      [cstep] "s"(f4_sgpr(cstep)), [kt] "s"(f4_sgpr(kt)), [tstepA] "s"(f4_sgpr(tstepA)), [tstepC] "s"(f4_sgpr(tstepC)),         \
      [tstepR] "s"(f4_sgpr(tstepR)), [ntile] "s"(f4_sgpr(ntile))

// Block -> (row unit, column tile).  Blocks go to the XCDs round robin (block & 7), every XCD has its own L2, and the operands
// come out of the Infinity Cache: in the linear order (column tile fastest) the eight column tiles of the M = 2048 down-projection
// land on eight XCDs and EVERY XCD pulls all of A through its L2 -- 132 MB per launch for 20 MB of operands.  Instead the XCDs
// split the tile grid gm x gn (host: the split with the smallest per-XCD footprint) and each works through its own sub-grid.
struct F4Grid { int units_m, tiles_n, gm, gn; };
__device__ __forceinline__ void f4_tile(const F4Grid& g, int* um, int* tn) {
    const int b = blockIdx.x;
    if (g.gm == 0) { *um = b / g.tiles_n; *tn = b - *um * g.tiles_n; return; }
    const int xcd = b & 7, local = b >> 3, xm = xcd / g.gn, xn = xcd - xm * g.gn;
    const int pm = g.units_m / g.gm, pn = g.tiles_n / g.gn, lm = local / pn, ln = local - lm * pn;
    *um = xm * pm + lm; *tn = xn * pn + ln;
    (void)pm;
}
static F4Grid f4_grid(int units_m, int tiles_n, long bytes_unit_m, long bytes_tile_n) {
    F4Grid g = {units_m, tiles_n, 0, 1};
    if (!PIPS_TUNE("PIPS_F32_T4_XCD", 1) || ((long)units_m * tiles_n) % 8 != 0) return g;
    long best = -1;
    for (int gm = 8; gm >= 1; gm >>= 1) {
        const int gn = 8 / gm;
        if (units_m % gm != 0 || tiles_n % gn != 0) continue;
        const long foot = (units_m / gm) * bytes_unit_m + (tiles_n / gn) * bytes_tile_n;
        if (best < 0 || foot < best) { best = foot; g.gm = gm; g.gn = gn; }
    }
    return g;
}

// EPI: 0 = + bias + GELU, 1 = + bias + residual
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_t4u_kernel(GemmArgs p, F4Grid grid, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    int um, tn;                                                // row unit = `tpb` consecutive row tiles
    f4_tile(grid, &um, &tn);
    const int m0 = um * tpb * 128, n0 = tn * 128;

    // staging: thread = (row lr of a 32-row pass, 16-byte chunk lc of the row's 128 bytes)
    const int lr = tid >> 3, lc = tid & 7;
    const float* Ab = p.A + (size_t)m0 * p.lda;
    const float* Wb = p.W + (size_t)n0 * p.K;
    const unsigned voA = (unsigned)(lr * p.lda * 4 + lc * 16), voW = (unsigned)(lr * p.K * 4 + lc * 16);
    const unsigned passA = (unsigned)(32 * p.lda * 4), passW = (unsigned)(32 * p.K * 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wA0 = lds0 + lr * F4_ROW + lc * 16, wW0 = wA0 + 128 * F4_ROW;               // + piece * 32 rows
    const unsigned wA1 = wA0 + F4_LDS_U / 2, wW1 = wW0 + F4_LDS_U / 2;
    // fragments: lane = row l31 of a 32-row block, K values 8 kk + 4 half .. + 3
    const unsigned rA0 = lds0 + (64 * wm + l31) * F4_ROW + half * 16, rW0 = lds0 + (128 + 64 * wn + l31) * F4_ROW + half * 16;
    const unsigned rA1 = rA0 + F4_LDS_U / 2, rW1 = rW0 + F4_LDS_U / 2;
    // output / residual / bias: MFMA block (i, j), quad q = C[m0 + 64 wm + 32 i + l31][n0 + 64 wn + 32 j + 8 q + 4 half .. + 3]
    const float* Cb = p.C + (size_t)(m0 + 64 * wm) * p.ldc + n0 + 64 * wn;
    const float* Rb = EPI == 1 ? p.R + (size_t)(m0 + 64 * wm) * p.ldr + n0 + 64 * wn : p.C;
    const float* Bb = p.bias + n0 + 64 * wn;
    const unsigned voC = (unsigned)((l31 * p.ldc + 4 * half) * 4), voR = (unsigned)((l31 * p.ldr + 4 * half) * 4), voB = (unsigned)(16 * half);
    const unsigned cstep = (unsigned)(32 * p.ldc * 4), rstep = (unsigned)(32 * p.ldr * 4);
    const unsigned tstepC = (unsigned)(128 * p.ldc * 4), tstepR = (unsigned)(128 * p.ldr * 4);
    const unsigned kt = (unsigned)(p.K / 32), tstepA = (unsigned)(128 * p.lda * 4) - 128u * kt, ntile = (unsigned)tpb;
    const unsigned redW = 0, redR = 0;
    if (EPI == 0) asm volatile(PIPS_F32T4_U_GELU_TEXT : F4_OPERANDS : PIPS_F32T4_CLOBBER);
    else          asm volatile(PIPS_F32T4_U_RES_TEXT : F4_OPERANDS : PIPS_F32T4_CLOBBER);
}

__global__ __launch_bounds__(256) void gemm_f32_t4d_kernel(GemmArgs p, F4Grid grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the K group of a stage in the loop, = the MFMA block (i + 2 j) at the end
    const int l31 = lane & 31, half = lane >> 5;
    int tm, tn;
    f4_tile(grid, &tm, &tn);
    const int m0 = tm * 64, n0 = tn * 64;

    // staging: piece s of a thread = row 32 s + lr of the tile, 16-byte chunk lc of the stage's 128 bytes
    const int lr = tid >> 3, lc = tid & 7;
    const float* Ab = p.A + (size_t)m0 * p.lda;
    const float* Wb = p.W + (size_t)n0 * p.K;
    const unsigned voA = (unsigned)(lr * p.lda * 4 + lc * 16), voW = (unsigned)(lr * p.K * 4 + lc * 16);
    const unsigned passA = (unsigned)(32 * p.lda * 4), passW = (unsigned)(32 * p.K * 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wA0 = lds0 + lr * F4_ROW + lc * 16, wW0 = wA0 + 64 * F4_ROW;
    const unsigned wA1 = wA0 + F4_STAGE_D, wW1 = wW0 + F4_STAGE_D;
    // fragments: this wave's 8 K values of the stage = bytes 32 wave .. of a row
    const unsigned rA0 = lds0 + l31 * F4_ROW + wave * 32 + half * 16, rW0 = lds0 + (64 + l31) * F4_ROW + wave * 32 + half * 16;
    const unsigned rA1 = rA0 + F4_STAGE_D, rW1 = rW0 + F4_STAGE_D;
    // the sum over the K quarters: red[quarter][4 block + quad][lane] (16 bytes each, 64 KiB over the stage buffers)
    const unsigned redW = lds0 + wave * 16384 + lane * 16, redR = lds0 + wave * 4096 + lane * 16;
    const int bi = wave & 1, bj = wave >> 1;
    const float* Cb = p.C + (size_t)(m0 + 32 * bi) * p.ldc + n0 + 32 * bj;
    const float* Rb = p.R + (size_t)(m0 + 32 * bi) * p.ldr + n0 + 32 * bj;
    const float* Bb = p.bias + n0 + 32 * bj;
    const unsigned voC = (unsigned)((l31 * p.ldc + 4 * half) * 4), voR = (unsigned)((l31 * p.ldr + 4 * half) * 4), voB = (unsigned)(16 * half);
    const unsigned cstep = 0, rstep = 0, tstepC = 0, tstepR = 0;
    const unsigned kt = (unsigned)(p.K / 32), tstepA = (unsigned)(64 * p.lda * 4) - 128u * kt, ntile = 1;
    asm volatile(PIPS_F32T4_D_RES_TEXT : F4_OPERANDS : PIPS_F32T4_CLOBBER);
}

// Shape E = shape D with the operands staged by LDS-DMA (no staging registers, no ds_write; tools/gen_gemm_f32_t4.py: body_e).  LDS: four
// buffers of 128 dense 128-byte rows, chunk slot j of row r = global chunk j ^ ((r >> 1) & 7).
constexpr int F4_LDS_E = 4 * 128 * 128;                      // 65 536 bytes (the four partial tiles at the end use the same 64 KiB)
__global__ __launch_bounds__(256) void gemm_f32_t4e_kernel(GemmArgs p, F4Grid grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    int tm, tn;
    f4_tile(grid, &tm, &tn);
    const int m0 = tm * 64, n0 = tn * 64;
    // staging: wave w fills LDS rows 32 w .. 32 w + 31 (waves 0, 1: the tile's A rows, waves 2, 3: its W rows); DMA instruction k:
    // rows 8 k .. 8 k + 7 of them, lane = (row q = lane >> 3, chunk slot j = lane & 7) fetching chunk j ^ ((row >> 1) & 7)
    const bool isA = wave < 2;
    const int ld = isA ? p.lda : p.K, row0 = 32 * (wave & 1);
    const float* Xb = isA ? p.A + (size_t)m0 * p.lda : p.W + (size_t)n0 * p.K;
    const int q = lane >> 3, j = lane & 7;
    const unsigned vo0 = (unsigned)((row0 + q) * ld * 4 + ((j ^ (q >> 1)) * 16));
    const unsigned vo1 = (unsigned)((row0 + 8 + q) * ld * 4 + ((j ^ (4 + (q >> 1))) * 16));
    const unsigned pass2 = (unsigned)(16 * ld * 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = lds0 + 32 * wave * 128;
    // fragments: lane = row l31 of a 32-row block, this wave's 8 K values of the stage = chunks 2 wave, 2 wave + 1 (lane half)
    const unsigned rA0 = lds0 + l31 * 128 + (((2 * wave + half) ^ ((l31 >> 1) & 7)) * 16), rW0 = rA0 + 64 * 128;
    const unsigned redW = lds0 + wave * 16384 + lane * 16, redR = lds0 + wave * 4096 + lane * 16;
    const int bi = wave & 1, bj = wave >> 1;
    const float* Cb = p.C + (size_t)(m0 + 32 * bi) * p.ldc + n0 + 32 * bj;
    const float* Rb = p.R + (size_t)(m0 + 32 * bi) * p.ldr + n0 + 32 * bj;
    const float* Bb = p.bias + n0 + 32 * bj;
    const unsigned voC = (unsigned)((l31 * p.ldc + 4 * half) * 4), voR = (unsigned)((l31 * p.ldr + 4 * half) * 4), voB = (unsigned)(16 * half);
    const unsigned kt = (unsigned)(p.K / 32);
    asm volatile(PIPS_F32T4_E_RES_TEXT
                 :
                 : [rA0] "v"(rA0), [rW0] "v"(rW0), [vo0] "v"(vo0), [vo1] "v"(vo1), [voR] "v"(voR), [voC] "v"(voC), [voB] "v"(voB),
                   [redW] "v"(redW), [redR] "v"(redR), [xlo] "s"(F4_LO(Xb)), [xhi] "s"(F4_HI(Xb)), [rlo] "s"(F4_LO(Rb)), [rhi] "s"(F4_HI(Rb)),
                   [clo] "s"(F4_LO(Cb)), [chi] "s"(F4_HI(Cb)), [blo] "s"(F4_LO(Bb)), [bhi] "s"(F4_HI(Bb)), [pass2] "s"(f4_sgpr(pass2)),
                   [ldsw] "s"(f4_sgpr(ldsw)), [kt] "s"(f4_sgpr(kt))
                 : PIPS_F32T4_CLOBBER);
}

// Which of the kernels above takes a plain fp32 GEMM: 0 none (igemm_f32_kernel), 1 shape U, 2 shape D; *tpb = row tiles per block (U).
int gemm_f32_t4_route(const GemmArgs& a, int* tpb) {
    if (!PIPS_TUNE("PIPS_F32_T4", 1)) return 0;               // tuning hook: 0 = igemm_f32_kernel everywhere
    const int epi = a.epi & 0xff;
    if (a.bias == nullptr || (epi != EPI_GELU && epi != EPI_RESIDUAL) || (epi == EPI_RESIDUAL && a.R == nullptr)) return 0;
    if (a.lda % 4 != 0 || a.ldc % 4 != 0 || (epi == EPI_RESIDUAL && a.ldr % 4 != 0)) return 0;
    if ((unsigned long long)a.M * a.lda * 4ull >= (1ull << 31) || (unsigned long long)a.N * a.K * 4ull >= (1ull << 31) ||
        (unsigned long long)160 * a.ldc * 4ull >= (1ull << 31) || (unsigned long long)160 * a.ldr * 4ull >= (1ull << 31)) return 0;
    const int cus = device_cus();
    if (cus <= 0) return 0;
    const long t128 = (long)(a.M / 128) * (a.N / 128);
    if (a.M % 128 == 0 && a.N % 128 == 0 && a.K % 64 == 0 && a.K >= 128 &&
        t128 * 100 >= (long)cus * PIPS_TUNE("PIPS_F32_T4U_MINPCT", 75)) {
        int t = PIPS_TUNE("PIPS_F32_T4U_TPB", 2);
        while (t > 1 && ((a.M / 128) % t != 0 || t128 / t < cus)) --t;
        if (tpb) *tpb = t;
        return 1;
    }
    const long t64 = (long)(a.M / 64) * (a.N / 64);
    if (epi == EPI_RESIDUAL && a.M % 64 == 0 && a.N % 64 == 0 && a.K % 128 == 0 && a.K >= 256 &&
        t64 * 100 >= (long)cus * PIPS_TUNE("PIPS_F32_T4D_MINPCT", 75) && t64 <= 2l * cus)
        return 2;
    return 0;
}

int launch_gemm_f32_t4(const GemmArgs& a, int route, int tpb, hipStream_t st) {
    const int epi = a.epi & 0xff;
    if (route == 1) {
        const int units_m = a.M / 128 / tpb, blocks = units_m * (a.N / 128);
        const F4Grid grid = f4_grid(units_m, a.N / 128, (long)tpb * 128 * a.K * 4, (long)128 * a.K * 4);
        static std::atomic<unsigned long long> raised0{0}, raised1{0};
        if (epi == EPI_GELU) {
            const int rc = ensure_dynamic_lds(raised0, (const void*)gemm_f32_t4u_kernel<0>, F4_LDS_U);
            if (rc != PIPS_OK) return rc;
            hipLaunchKernelGGL(gemm_f32_t4u_kernel<0>, dim3(blocks), dim3(256), F4_LDS_U, st, a, grid, tpb);
        } else {
            const int rc = ensure_dynamic_lds(raised1, (const void*)gemm_f32_t4u_kernel<1>, F4_LDS_U);
            if (rc != PIPS_OK) return rc;
            hipLaunchKernelGGL(gemm_f32_t4u_kernel<1>, dim3(blocks), dim3(256), F4_LDS_U, st, a, grid, tpb);
        }
        PIPS_CHECK_LAUNCH("gemm_f32_t4u_kernel");
        return PIPS_OK;
    }
    if (PIPS_TUNE("PIPS_F32_T4_E", 1)) {                      // tuning hook: shape E (LDS-DMA staging) in place of shape D
        static std::atomic<unsigned long long> raised3{0};
        const int rc3 = ensure_dynamic_lds(raised3, (const void*)gemm_f32_t4e_kernel, F4_LDS_E);
        if (rc3 != PIPS_OK) return rc3;
        const F4Grid grid = f4_grid(a.M / 64, a.N / 64, (long)64 * a.K * 4, (long)64 * a.K * 4);
        hipLaunchKernelGGL(gemm_f32_t4e_kernel, dim3((a.M / 64) * (a.N / 64)), dim3(256), F4_LDS_E, st, a, grid);
        PIPS_CHECK_LAUNCH("gemm_f32_t4e_kernel");
        return PIPS_OK;
    }
    static std::atomic<unsigned long long> raised2{0};
    const int rc = ensure_dynamic_lds(raised2, (const void*)gemm_f32_t4d_kernel, F4_LDS_D);
    if (rc != PIPS_OK) return rc;
    const F4Grid grid = f4_grid(a.M / 64, a.N / 64, (long)64 * a.K * 4, (long)64 * a.K * 4);
    hipLaunchKernelGGL(gemm_f32_t4d_kernel, dim3((a.M / 64) * (a.N / 64)), dim3(256), F4_LDS_D, st, a, grid);
    PIPS_CHECK_LAUNCH("gemm_f32_t4d_kernel");
    return PIPS_OK;
}

}  // namespace pips
