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
//     D   64 x 64 tile, waves = four K quarters summed through LDS in a fixed order -- when there are too few 128 x 128 tiles
//         for the chip (the down-projection at M = 2048: 64 of them)
#include "common.h"
#include "gemm_f32_t4_asm.inc"

namespace pips {

constexpr int F4_ROW = 144;                                  // LDS row: 32 K values + 16 bytes
constexpr int F4_LDS_U = 2 * 256 * F4_ROW;                   // 73 728 bytes: two stages of [A rows 0..127 | W rows 0..127]
constexpr int F4_LDS_D = 2 * 512 * F4_ROW;                   // 147 456 bytes: two stages of [A 4 x 64 rows | W 4 x 64 rows]

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

// EPI: 0 = + bias + GELU, 1 = + bias + residual
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_t4u_kernel(GemmArgs p, int tiles_m, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int runs = tiles_m / tpb;                            // blocks per column tile
    const int tn = blockIdx.x / runs, tm0 = (blockIdx.x - tn * runs) * tpb;
    const int m0 = tm0 * 128, n0 = tn * 128;

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

__global__ __launch_bounds__(256) void gemm_f32_t4d_kernel(GemmArgs p, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the K quarter in the loop, = the MFMA block (i + 2 j) at the end
    const int l31 = lane & 31, half = lane >> 5;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;

    // staging: piece s of a thread = rows 32 (s & 1) + lr of the tile, K quarter s >> 1 (128 bytes each) -> LDS row 32 s + lr
    const int lr = tid >> 3, lc = tid & 7;
    const float* Ab = p.A + (size_t)m0 * p.lda;
    const float* Wb = p.W + (size_t)n0 * p.K;
    const unsigned voA = (unsigned)(lr * p.lda * 4 + lc * 16), voW = (unsigned)(lr * p.K * 4 + lc * 16);
    const unsigned passA = (unsigned)(32 * p.lda * 4), passW = (unsigned)(32 * p.K * 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wA0 = lds0 + lr * F4_ROW + lc * 16, wW0 = wA0 + 256 * F4_ROW;
    const unsigned wA1 = wA0 + F4_LDS_D / 2, wW1 = wW0 + F4_LDS_D / 2;
    const unsigned rA0 = lds0 + (64 * wave + l31) * F4_ROW + half * 16, rW0 = lds0 + (256 + 64 * wave + l31) * F4_ROW + half * 16;
    const unsigned rA1 = rA0 + F4_LDS_D / 2, rW1 = rW0 + F4_LDS_D / 2;
    // the sum over the K quarters: red[quarter][4 block + quad][lane] (16 bytes each, 64 KiB over the stage buffers)
    const unsigned redW = lds0 + wave * 16384 + lane * 16, redR = lds0 + wave * 4096 + lane * 16;
    const int bi = wave & 1, bj = wave >> 1;
    const float* Cb = p.C + (size_t)(m0 + 32 * bi) * p.ldc + n0 + 32 * bj;
    const float* Rb = p.R + (size_t)(m0 + 32 * bi) * p.ldr + n0 + 32 * bj;
    const float* Bb = p.bias + n0 + 32 * bj;
    const unsigned voC = (unsigned)((l31 * p.ldc + 4 * half) * 4), voR = (unsigned)((l31 * p.ldr + 4 * half) * 4), voB = (unsigned)(16 * half);
    const unsigned cstep = 0, rstep = 0, tstepC = 0, tstepR = 0;
    const unsigned kt = (unsigned)(p.K / 128), tstepA = (unsigned)(64 * p.lda * 4) - 512u * kt, ntile = 1;
    asm volatile(PIPS_F32T4_D_RES_TEXT : F4_OPERANDS : PIPS_F32T4_CLOBBER);
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
    if (epi == EPI_RESIDUAL && a.M % 64 == 0 && a.N % 64 == 0 && a.K % 256 == 0 && a.K >= 512 &&
        t64 * 100 >= (long)cus * PIPS_TUNE("PIPS_F32_T4D_MINPCT", 75) && t64 <= 2l * cus)
        return 2;
    return 0;
}

int launch_gemm_f32_t4(const GemmArgs& a, int route, int tpb, hipStream_t st) {
    const int epi = a.epi & 0xff;
    if (route == 1) {
        const int tiles_m = a.M / 128, blocks = tiles_m / tpb * (a.N / 128);
        static std::atomic<unsigned long long> raised0{0}, raised1{0};
        if (epi == EPI_GELU) {
            const int rc = ensure_dynamic_lds(raised0, (const void*)gemm_f32_t4u_kernel<0>, F4_LDS_U);
            if (rc != PIPS_OK) return rc;
            hipLaunchKernelGGL(gemm_f32_t4u_kernel<0>, dim3(blocks), dim3(256), F4_LDS_U, st, a, tiles_m, tpb);
        } else {
            const int rc = ensure_dynamic_lds(raised1, (const void*)gemm_f32_t4u_kernel<1>, F4_LDS_U);
            if (rc != PIPS_OK) return rc;
            hipLaunchKernelGGL(gemm_f32_t4u_kernel<1>, dim3(blocks), dim3(256), F4_LDS_U, st, a, tiles_m, tpb);
        }
        PIPS_CHECK_LAUNCH("gemm_f32_t4u_kernel");
        return PIPS_OK;
    }
    static std::atomic<unsigned long long> raised2{0};
    const int rc = ensure_dynamic_lds(raised2, (const void*)gemm_f32_t4d_kernel, F4_LDS_D);
    if (rc != PIPS_OK) return rc;
    const int tiles_n = a.N / 64;
    hipLaunchKernelGGL(gemm_f32_t4d_kernel, dim3((a.M / 64) * tiles_n), dim3(256), F4_LDS_D, st, a, tiles_n);
    PIPS_CHECK_LAUNCH("gemm_f32_t4d_kernel");
    return PIPS_OK;
}

}  // namespace pips
