// The mixer's channel-mix Linears in the split-bf16 matrix mode (gemm_x3.hip: every fp32 operand = three exact bf16 terms, six bf16
// MFMA products per fp32 product, fp32 accumulation) on four waves with a generated static schedule -- gemm_f32_t4.hip's kernel
// (shape U: 128 x 128 tile, 2 x 2 waves of 64 x 64, one block per CU, a block walking `tpb` row tiles of one column tile) with
// v_mfma_f32_32x32x16_bf16 in place of the fp32 MFMA: same accumulator layout, same epilogues (bias + exact GELU, bias + residual).
// gemm_x3_kernel keeps its matrix pipe 0.30 busy (one wave per SIMD cannot hide its hipcc-scheduled staging); here the
// six planes of a 32-wide stage are written to LDS one stage ahead and requested two ahead, the fragments of a 16-wide K step are
// refilled behind their last use inside the step, and the split of A (9 vector instructions per pair of values) is dealt out over
// the MFMA slots.  Body: gemm_x3_t4_asm.inc <- tools/gen_gemm_x3_t4.py (schedule and register map there).  Same products in the
// same order as gemm_x3_kernel; the K order of the accumulation is that kernel's unsplit form.
#include "common.h"
#ifndef PIPS_X3T4_INC
#define PIPS_X3T4_INC "gemm_x3_t4_asm.inc"
#endif
#include PIPS_X3T4_INC

namespace pips {

constexpr int X4_ROW = 80;                                   // LDS row of a plane: 32 K values (64 bytes) + 16
constexpr int X4_PLANE = 128 * X4_ROW, X4_STAGE = 6 * X4_PLANE, X4_LDS = 2 * X4_STAGE;      // 122 880 bytes

__device__ __forceinline__ unsigned x4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#define X4_LO(ptr) x4_sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define X4_HI(ptr) x4_sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))

// EPI: 0 = + bias + GELU, 1 = + bias + residual.  W: three bf16 planes [3][N][K].
template <int EPI>
__global__ __launch_bounds__(256) void gemm_x3_t4_kernel(GemmArgs p, int units_m, int tiles_n, int gm, int gn, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    int um, tn;                                                // XCD-aware 2-D partition of the tile grid (gemm_f32_t4.hip: f4_grid)
    {
        const int b = blockIdx.x;
        if (gm == 0) { um = b / tiles_n; tn = b - um * tiles_n; }
        else {
            const int xcd = b & 7, local = b >> 3, xm = xcd / gn, xn = xcd - xm * gn;
            const int pn = tiles_n / gn, lm = local / pn, ln = local - lm * pn;
            um = xm * (units_m / gm) + lm; tn = xn * pn + ln;
        }
    }
    const int m0 = um * tpb * 128, n0 = tn * 128;

    // staging: thread = (row lr of a 64-row pass, 8 K values lc of the stage's 32): 32 bytes of A, 16 bytes of each W plane
    const int lr = tid >> 2, lc = tid & 3;
    const float* Ab = p.A + (size_t)m0 * p.lda;
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W) + (size_t)n0 * p.K;
    const unsigned voA = (unsigned)(lr * p.lda * 4 + lc * 32), voW = (unsigned)(lr * p.K * 2 + lc * 16);
    const unsigned passA = (unsigned)(64 * p.lda * 4), passW = (unsigned)(64 * p.K * 2), wplane = (unsigned)p.N * (unsigned)p.K * 2u;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wb0 = lds0 + lr * X4_ROW + lc * 16, wb1 = wb0 + X4_STAGE;
    // fragments: lane = row l31 of a 32-row block, K values 16 ks + 8 half .. + 7
    const unsigned rA0 = lds0 + (64 * wm + l31) * X4_ROW + half * 16, rA1 = rA0 + X4_STAGE;
    const unsigned rW0 = lds0 + 3 * X4_PLANE + (64 * wn + l31) * X4_ROW + half * 16, rW1 = rW0 + X4_STAGE;
    float* Cb = p.C + (size_t)(m0 + 64 * wm) * p.ldc + n0 + 64 * wn;
    const float* Rb = EPI == 1 ? p.R + (size_t)(m0 + 64 * wm) * p.ldr + n0 + 64 * wn : p.C;
    const float* Bb = p.bias + n0 + 64 * wn;
    const unsigned voC = (unsigned)((l31 * p.ldc + 4 * half) * 4), voR = (unsigned)((l31 * p.ldr + 4 * half) * 4), voB = (unsigned)(16 * half);
    const unsigned cstep = (unsigned)(32 * p.ldc * 4), rstep = (unsigned)(32 * p.ldr * 4);
    const unsigned tstepC = (unsigned)(128 * p.ldc * 4), tstepR = (unsigned)(128 * p.ldr * 4);
    const unsigned kt = (unsigned)(p.K / 32), tstepA = (unsigned)(128 * p.lda * 4) - 128u * kt, ntile = (unsigned)tpb;
#define X4_OPERANDS                                                                                                              \
    : [rA0] "v"(rA0), [rW0] "v"(rW0), [rA1] "v"(rA1), [rW1] "v"(rW1), [wb0] "v"(wb0), [wb1] "v"(wb1), [voA] "v"(voA),          \
      [voW] "v"(voW), [voR] "v"(voR), [voC] "v"(voC), [voB] "v"(voB), [alo] "s"(X4_LO(Ab)), [ahi] "s"(X4_HI(Ab)),               \
      [wlo] "s"(X4_LO(Wb)), [whi] "s"(X4_HI(Wb)), [rlo] "s"(X4_LO(Rb)), [rhi] "s"(X4_HI(Rb)), [clo] "s"(X4_LO(Cb)),             \
      [chi] "s"(X4_HI(Cb)), [blo] "s"(X4_LO(Bb)), [bhi] "s"(X4_HI(Bb)), [passA] "s"(x4_sgpr(passA)), [passW] "s"(x4_sgpr(passW)), \
      [wplane] "s"(x4_sgpr(wplane)), [rstep] "s"(x4_sgpr(rstep)), [cstep] "s"(x4_sgpr(cstep)), [kt] "s"(x4_sgpr(kt)),            \
      [tstepA] "s"(x4_sgpr(tstepA)), [tstepC] "s"(x4_sgpr(tstepC)), [tstepR] "s"(x4_sgpr(tstepR)), [ntile] "s"(x4_sgpr(ntile))
    if (EPI == 0) asm volatile(PIPS_X3T4_U_GELU_TEXT : X4_OPERANDS : PIPS_X3T4_CLOBBER);
    else          asm volatile(PIPS_X3T4_U_RES_TEXT : X4_OPERANDS : PIPS_X3T4_CLOBBER);
#undef X4_OPERANDS
}

// Whether a split-bf16 GEMM goes to the kernel above; *tpb = row tiles per block.
bool gemm_x3_t4_takes(const GemmArgs& a, int* tpb) {
    if (!PIPS_TUNE("PIPS_X3_T4", 1)) return false;            // tuning hook: 0 = gemm_x3_kernel everywhere
    const int epi = a.epi & 0xff;
    if (a.bias == nullptr || (epi != EPI_GELU && epi != EPI_RESIDUAL) || (epi == EPI_RESIDUAL && a.R == nullptr)) return false;
    if (a.M % 128 != 0 || a.N % 128 != 0 || a.K % 64 != 0 || a.K < 128) return false;
    if (a.lda % 8 != 0 || a.ldc % 4 != 0 || (epi == EPI_RESIDUAL && a.ldr % 4 != 0)) return false;
    if ((unsigned long long)a.M * a.lda * 4ull >= (1ull << 31) || (unsigned long long)a.N * a.K * 6ull + 128ull * a.K * 2ull >= (1ull << 31) ||
        (unsigned long long)160 * a.ldc * 4ull >= (1ull << 31) || (unsigned long long)160 * a.ldr * 4ull >= (1ull << 31)) return false;
    const int cus = device_cus();
    const long t128 = (long)(a.M / 128) * (a.N / 128);
    if (cus <= 0 || t128 * 100 < (long)cus * PIPS_TUNE("PIPS_X3_T4_MINPCT", 75)) return false;
    int t = PIPS_TUNE("PIPS_X3_T4_TPB", 2);
    while (t > 1 && ((a.M / 128) % t != 0 || t128 / t < cus)) --t;
    if (tpb) *tpb = t;
    return true;
}

int launch_gemm_x3_t4(const GemmArgs& a, int tpb, hipStream_t st) {
    const int units_m = a.M / 128 / tpb, tiles_n = a.N / 128, blocks = units_m * tiles_n;
    int gm = 0, gn = 1;                                        // the XCD split with the smallest per-XCD operand footprint
    if (blocks % 8 == 0) {
        long best = -1;
        for (int m = 8; m >= 1; m >>= 1) {
            const int n = 8 / m;
            if (units_m % m != 0 || tiles_n % n != 0) continue;
            const long foot = (long)(units_m / m) * tpb * 128 * a.K * 4 + (long)(tiles_n / n) * 128 * a.K * 6;
            if (best < 0 || foot < best) { best = foot; gm = m; gn = n; }
        }
    }
    static std::atomic<unsigned long long> raised0{0}, raised1{0};
    if ((a.epi & 0xff) == EPI_GELU) {
        const int rc = ensure_dynamic_lds(raised0, (const void*)gemm_x3_t4_kernel<0>, X4_LDS);
        if (rc != PIPS_OK) return rc;
        hipLaunchKernelGGL(gemm_x3_t4_kernel<0>, dim3(blocks), dim3(256), X4_LDS, st, a, units_m, tiles_n, gm, gn, tpb);
    } else {
        const int rc = ensure_dynamic_lds(raised1, (const void*)gemm_x3_t4_kernel<1>, X4_LDS);
        if (rc != PIPS_OK) return rc;
        hipLaunchKernelGGL(gemm_x3_t4_kernel<1>, dim3(blocks), dim3(256), X4_LDS, st, a, units_m, tiles_n, gm, gn, tpb);
    }
    PIPS_CHECK_LAUNCH("gemm_x3_t4_kernel");
    return PIPS_OK;
}

}  // namespace pips
