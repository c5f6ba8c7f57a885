// bf16 down-projection of the mixer at large M (BASELINE configs[2]: M = 16384, N = 512, K = 2048, + bias + fp32 residual):
//
//     C[m][n] = R[m][n] + sum_k A[m][k] * W[n][k] + bias[n]           (nets/pips.py:102-109 inside PreNormResidual :93-100)
//
// One 128 x 256 tile per block, FOUR waves (one per SIMD, 512 registers each), wave tile 64 x 128 on v_mfma_f32_16x16x32_bf16
// (C^T: the W fragment is the MFMA's row operand, so a lane owns an output ROW and 4 consecutive columns per 16 x 16 tile).
// What is different from gemm_bf16_res_asm_kernel (256 x 128 tile, eight waves of 64 x 64, LDS-DMA ring) and why
// [measured, profiles/r4_probe_bf16_down_t4.txt]:
//   * the 16-pass-shorter MFMA (16 clk instead of 32) lets ONE memory instruction sit between two MFMAs everywhere in the
//     loop: per 64 K values a wave issues 64 MFMAs, 24 ds_read_b128 (12 fragments per 32-K step: 4 of A, 8 of W), 12
//     ds_write_b128 and 12 buffer_load_dwordx4 -- 48 memory instructions in 64 slots;
//   * operands go global -> registers -> LDS (two tiles ahead in registers), not by LDS-DMA: the DMA's LDS writes collided with
//     the fragment reads (MFMA + reads 11.3, MFMA + DMA 10.3, all three 18.1 us per 1 024 K, DESIGN.md 4b iii), a ds_write
//     is placed where the schedule wants it;
//   * ONE LDS buffer (48 KiB) and two barriers per 64 K: [MFMAs of K step 0 | reads of K step 1] barrier [MFMAs | writes of
//     the next tile + loads of the one after] barrier [MFMAs of K step 1 | reads of the next tile's K step 0];
//   * the residual tile seeds the accumulators (its 32 loads per lane fly during the prologue), the bias is added at the end.
// LDS image: rows of 64 K values (128 B), the 16-byte chunk index XORed with (row >> 1) & 7: conflict-free for the fragment
// ds_read_b128 (lane = row & 15, K group = lane >> 4) and for the staging ds_write_b128 (8 lanes = one row).
// The whole body is ONE generated assembly statement (gemm_bf16_t4_asm.inc <- tools/gen_gemm_bf16_t4.py: static schedule, counted
// waits): as C++ with builtins + sched_barrier pins hipcc kept a third of the accumulators in ArchVGPRs and moved ~200 registers
// per iteration between the two register files (88 v_accvgpr_read + 88 _write + 32 _mov per 64 MFMAs).
#include "common.h"
#ifndef PIPS_T4_INC
#define PIPS_T4_INC "gemm_bf16_t4_asm.inc"      // tuning builds point this at another schedule of the generator
#endif
#include PIPS_T4_INC
#ifndef PIPS_T4UP_INC
#define PIPS_T4UP_INC "gemm_bf16_t4up_asm.inc"
#endif
#include PIPS_T4UP_INC

namespace pips {

constexpr int T4_BM = 128, T4_BN = 256, T4_BK = 64;
constexpr int T4_LDS = (T4_BM + T4_BN) * T4_BK * 2;          // 49 152 bytes: one buffer, [A rows 0..127 | W rows 0..255] x 128 B

__device__ __forceinline__ unsigned t4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#define T4_LO(ptr) t4_sgpr((unsigned)(unsigned long long)reinterpret_cast<uintptr_t>(ptr))
#define T4_HI(ptr) t4_sgpr((unsigned)((unsigned long long)reinterpret_cast<uintptr_t>(ptr) >> 32))

#ifdef PIPS_T4_CLOCK
// tools/t4_clock.py (variant build): per wave of the LAST launch of either kernel, shader clocks (s_memtime) and 100 MHz ticks (s_memrealtime)
// around the generated statement -- the clock the kernel actually ran at, and a wave's share of the launch
__device__ unsigned long long g_t4_clock[2][1024][2];
extern "C" int pips_debug_t4_clock(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_t4_clock), sizeof(unsigned long long) * 2 * 1024 * 2);
}
#define PIPS_T4_CLOCK_BEGIN const unsigned long long clk0__ = clock64(), rt0__ = wall_clock64();
#define PIPS_T4_CLOCK_END(which)                                                                                       \
    {                                                                                                                  \
        const unsigned long long clk1__ = clock64(), rt1__ = wall_clock64();                                           \
        const int w__ = blockIdx.x * 4 + (threadIdx.x >> 6);                                                           \
        if ((threadIdx.x & 63) == 0 && w__ < 1024) { g_t4_clock[which][w__][0] = clk1__ - clk0__; g_t4_clock[which][w__][1] = rt1__ - rt0__; } \
    }
#else
#define PIPS_T4_CLOCK_BEGIN
#define PIPS_T4_CLOCK_END(which)
#endif

// SB: the residual stream is bf16 -- R and C are bf16 (the reference's PreNormResidual under autocast adds two bf16 tensors,
// nets/pips.py:93-100): the residual tile is widened into the accumulators, the result rounded once (RNE) on the way out.
template <bool SB>
__global__ __launch_bounds__(256) void gemm_bf16_t4_res_kernel(GemmArgs p, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;

    // XCD-aware tile order: XCD (block & 7) owns one contiguous run of the (row tile, column tile) sequence, column tile
    // fastest -- the column tiles of a row block read the same 128 rows of A out of one L2
    int tile;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, q = ntiles >> 3, r = ntiles & 7;
        tile = xcd * q + (xcd < r ? xcd : r) + local;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * T4_BM, n0 = tn * T4_BN;

    // ---- staging: thread = (row lr of a 32-row pass, 16-byte chunk lc of the row's 128 bytes); LDS chunk = lc ^ ((row >> 1) & 7)
    const int lr = tid >> 3, lc = tid & 7;
    const unsigned short* Ab = reinterpret_cast<const unsigned short*>(p.A) + (size_t)m0 * p.lda;
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W) + (size_t)n0 * p.K;
    const unsigned voA = (unsigned)(lr * p.lda * 2 + lc * 16), voW = (unsigned)(lr * p.K * 2 + lc * 16);
    const unsigned passA = (unsigned)(32 * p.lda * 2), passW = (unsigned)(32 * p.K * 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wA = lds0 + lr * 128 + ((lc ^ ((lr >> 1) & 7)) * 16);                         // + piece * 4096
    const unsigned wW = wA + T4_BM * 128;
    // ---- fragments: lane = row r16 of a 16-row block, K group g (8 K values = one 16-byte chunk); K step 1 = chunk ^ 4
    const int sw = (r16 >> 1) & 7;
    const unsigned rA0 = lds0 + (64 * wm + r16) * 128 + ((g ^ sw) * 16);                          // + i * 2048
    const unsigned rW0 = lds0 + T4_BM * 128 + (128 * wn + r16) * 128 + ((g ^ sw) * 16);          // + j * 2048
    const unsigned rA1 = lds0 + ((rA0 - lds0) ^ 64), rW1 = lds0 + ((rW0 - lds0) ^ 64);
    // ---- residual / output / bias: acc tile (i, j) = R[m0 + 64 wm + 16 i + r16][n0 + 128 wn + 16 j + 4 g .. + 3]
    constexpr int ES = SB ? 2 : 4;                                                          // bytes per element of R and C
    const char* Rb = reinterpret_cast<const char*>(p.R) + ((size_t)(m0 + 64 * wm) * p.ldr + n0 + 128 * wn) * ES;
    const char* Cb = reinterpret_cast<const char*>(p.C) + ((size_t)(m0 + 64 * wm) * p.ldc + n0 + 128 * wn) * ES;
    const float* Bb = p.bias + n0 + 128 * wn;
    const unsigned voR = (unsigned)((r16 * p.ldr + 4 * g) * ES), voC = (unsigned)((r16 * p.ldc + 4 * g) * ES), voB = (unsigned)(16 * g);
    const unsigned rstep = (unsigned)(16 * p.ldr * ES), cstep = (unsigned)(16 * p.ldc * ES);
    const unsigned kt = (unsigned)(p.K / T4_BK);
#define T4_OPERANDS \
                 : [rA0] "v"(rA0), [rW0] "v"(rW0), [rA1] "v"(rA1), [rW1] "v"(rW1), [wA] "v"(wA), [wW] "v"(wW), [voA] "v"(voA), \
                   [voW] "v"(voW), [voR] "v"(voR), [voC] "v"(voC), [voB] "v"(voB), [alo] "s"(T4_LO(Ab)), [ahi] "s"(T4_HI(Ab)), \
                   [wlo] "s"(T4_LO(Wb)), [whi] "s"(T4_HI(Wb)), [rlo] "s"(T4_LO(Rb)), [rhi] "s"(T4_HI(Rb)), [clo] "s"(T4_LO(Cb)), \
                   [chi] "s"(T4_HI(Cb)), [blo] "s"(T4_LO(Bb)), [bhi] "s"(T4_HI(Bb)), [passA] "s"(t4_sgpr(passA)), \
                   [passW] "s"(t4_sgpr(passW)), [rstep] "s"(t4_sgpr(rstep)), [cstep] "s"(t4_sgpr(cstep)), [kt] "s"(t4_sgpr(kt))
    PIPS_T4_CLOCK_BEGIN
    if (SB) asm volatile(PIPS_T4B_TEXT : T4_OPERANDS : PIPS_T4_CLOBBER);
    else asm volatile(PIPS_T4_TEXT : T4_OPERANDS : PIPS_T4_CLOBBER);
    PIPS_T4_CLOCK_END(1)
#undef T4_OPERANDS
}

// ---------------------------------------------------------------------------------------------------------------------
// The up-projection C = bf16(gelu(bf16(A W^T + b))), K = 512, in the same style (round 4): 256 x 256 tile, four waves with 128 x 128
// wave tiles and all 256 AccVGPRs as accumulators (0.25 fragment reads per MFMA against 0.375 above), a block walking `tpb`
// consecutive row tiles of one column tile with the K pipeline running on across the tile boundary, exact polynomial GELU
// (gelu_exact2's arithmetic: 4.8e-7, where the LDS table of round 3's gemm_bf16_gelu256_asm_kernel carried 2.4e-5 -- profiles/r4_probe_bf16_up_t4.txt has the A/B against that kernel) on the Linear's
// bf16-rounded output.  The W rows of a tile are staged in a permuted order (w_perm) so that the 16 x 16 tiles 2 j', 2 j' + 1
// together give a lane 8 consecutive output columns: one 16-byte store.  Body: gemm_bf16_t4up_asm.inc <- tools/gen_gemm_bf16_t4up.py.
constexpr int T4U_B = 256, T4U_K = 512;
constexpr int T4U_LDS = 2 * T4U_B * T4_BK * 2;              // 65 536 bytes

__global__ __launch_bounds__(256) void gemm_bf16_t4_gelu_kernel(GemmArgs p, int tiles_m, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int runs = tiles_m / tpb;                            // blocks per column tile
    const int tn = blockIdx.x / runs, tm0 = (blockIdx.x - tn * runs) * tpb;
    const int m0 = tm0 * T4U_B, n0 = tn * T4U_B;

    const int lr = tid >> 3, lc = tid & 7;
    const unsigned short* Ab = reinterpret_cast<const unsigned short*>(p.A) + (size_t)m0 * p.lda;
    const unsigned short* Wb = reinterpret_cast<const unsigned short*>(p.W) + (size_t)n0 * p.K;
    // LDS row lr + 32 s of the W tile holds global row 32 s + w_perm(lr): rows are in MFMA order (column block j, row rho of it)
    const int w_perm = 8 * ((lr & 15) >> 2) + 4 * (lr >> 4) + (lr & 3);
    const unsigned voA = (unsigned)(lr * p.lda * 2 + lc * 16), voW = (unsigned)(w_perm * p.K * 2 + lc * 16);
    const unsigned passA = (unsigned)(32 * p.lda * 2), passW = (unsigned)(32 * p.K * 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wA = lds0 + lr * 128 + ((lc ^ ((lr >> 1) & 7)) * 16), wW = wA + T4U_B * 128;
    const int sw = (r16 >> 1) & 7;
    const unsigned oA = (128 * wm + r16) * 128 + ((g ^ sw) * 16), oW = T4U_B * 128 + (128 * wn + r16) * 128 + ((g ^ sw) * 16);
    const unsigned rA0 = lds0 + oA, rW0 = lds0 + oW, rA1 = lds0 + (oA ^ 64), rW1 = lds0 + (oW ^ 64);
    // output: lane = row r16 of a 16-row block, 8 consecutive bf16 columns 32 j' + 8 g .. + 7 of the wave's 128
    unsigned short* Cb = reinterpret_cast<unsigned short*>(p.C) + (size_t)(m0 + 128 * wm) * p.ldc + n0 + 128 * wn;
    const float* Bb = p.bias + n0 + 128 * wn;
    const unsigned voC = (unsigned)((r16 * p.ldc + 8 * g) * 2), voB = (unsigned)(32 * g);
    const unsigned cstep = (unsigned)(16 * p.ldc * 2), tstepC = (unsigned)(T4U_B * p.ldc * 2);
    const unsigned tstepA = (unsigned)(T4U_B * p.lda * 2 - 128 * (T4U_K / T4_BK));
    PIPS_T4_CLOCK_BEGIN
    asm volatile(PIPS_T4UP_TEXT
                 :
                 : [rA0] "v"(rA0), [rW0] "v"(rW0), [rA1] "v"(rA1), [rW1] "v"(rW1), [wA] "v"(wA), [wW] "v"(wW), [voA] "v"(voA),
                   [voW] "v"(voW), [voC] "v"(voC), [voB] "v"(voB), [alo] "s"(T4_LO(Ab)), [ahi] "s"(T4_HI(Ab)), [wlo] "s"(T4_LO(Wb)),
                   [whi] "s"(T4_HI(Wb)), [clo] "s"(T4_LO(Cb)), [chi] "s"(T4_HI(Cb)), [blo] "s"(T4_LO(Bb)), [bhi] "s"(T4_HI(Bb)),
                   [passA] "s"(t4_sgpr(passA)), [passW] "s"(t4_sgpr(passW)), [cstep] "s"(t4_sgpr(cstep)), [tstepC] "s"(t4_sgpr(tstepC)),
                   [tstepA] "s"(t4_sgpr(tstepA)), [ntile] "s"(t4_sgpr((unsigned)tpb))
                 : PIPS_T4UP_CLOBBER);
    PIPS_T4_CLOCK_END(0)
}

// Whether the up-projection form (bf16 A and C, GELU, K = 512) goes to gemm_bf16_t4_gelu_kernel; *tpb = row tiles per block.
bool gemm_bf16_t4up_takes(const GemmArgs& a, int a_bf16, int out_bf16, int* tpb) {
    if (!PIPS_TUNE("PIPS_BF16_T4UP", 1)) return false;       // tuning hook: 0 = the register-staged kernel
    if (!a_bf16 || !out_bf16 || (a.epi & 0xff) != EPI_GELU || a.bias == nullptr || a.K != T4U_K) return false;
    if (a.M % T4U_B != 0 || a.N % T4U_B != 0 || a.lda % 8 != 0 || a.ldc % 8 != 0) return false;
    if ((unsigned long long)a.M * a.lda * 2ull >= (1ull << 31) || (unsigned long long)288 * a.ldc * 2ull >= (1ull << 31)) return false;
    const int cus = device_cus();
    const long tiles = (long)(a.M / T4U_B) * (a.N / T4U_B);
    if (cus <= 0 || tiles * 100 < (long)cus * PIPS_TUNE("PIPS_BF16_T4UP_MINPCT", 75)       /* M = 6144: -7.5 %; at 50 %, M = 4096: +3.4 % */) return false;
    int t = PIPS_TUNE("PIPS_BF16_T4UP_TPB", 2);
    while (t > 1 && ((a.M / T4U_B) % t != 0 || tiles / t < cus)) --t;
    if (tpb) *tpb = t;
    return true;
}

int launch_gemm_bf16_t4up(const GemmArgs& a, int tpb, hipStream_t st) {
    const int tiles_m = a.M / T4U_B, blocks = tiles_m / tpb * (a.N / T4U_B);
    static std::atomic<unsigned long long> raised{0};
    const int rc = ensure_dynamic_lds(raised, (const void*)gemm_bf16_t4_gelu_kernel, T4U_LDS);
    if (rc != PIPS_OK) return rc;
    hipLaunchKernelGGL(gemm_bf16_t4_gelu_kernel, dim3(blocks), dim3(256), T4U_LDS, st, a, tiles_m, tpb);
    PIPS_CHECK_LAUNCH("gemm_bf16_t4_gelu_kernel");
    return PIPS_OK;
}

// Whether the down-projection form (bf16 A, fp32 C, + bias + fp32 residual) of a bf16-operand GEMM goes to this kernel.
bool gemm_bf16_t4_takes(const GemmArgs& a, int a_bf16, int out_bf16) {
    const int mode = PIPS_TUNE("PIPS_BF16_T4", 1);          // tuning hook: 0 = off, 2 = any tile count
    // fp32 residual stream (fp32 R, fp32 C) or bf16 residual stream (EPI_RES_BF16: bf16 R, bf16 C)
    if (!mode || !a_bf16 || (out_bf16 != 0) != ((a.epi & EPI_RES_BF16) != 0) || (a.epi & 0xff) != EPI_RESIDUAL || a.R == nullptr ||
        a.bias == nullptr)
        return false;
    if (a.M % T4_BM != 0 || a.N % T4_BN != 0 || a.K % T4_BK != 0 || a.K < 2 * T4_BK) return false;
    if (a.lda % 8 != 0 || a.ldc % 4 != 0 || a.ldr % 4 != 0) return false;
    if ((unsigned long long)80 * a.ldr * 4ull >= (1ull << 31) || (unsigned long long)80 * a.ldc * 4ull >= (1ull << 31)) return false;   // (buffer offsets inside a wave tile)
    if ((unsigned long long)(T4_BM + 32) * a.lda * 2ull >= (1ull << 31) || (unsigned long long)(T4_BN + 32) * a.K * 2ull >= (1ull << 31)) return false;
    const int cus = device_cus();
    // tiles needed, in percent of the CU count: from HALF a tile per CU on this kernel is ahead of the register-staged one (bf16 mixer pass
    // at M = 8192 / 12288: -9.4 % / -13.7 %; at a quarter, M = 4096: +14 %; profiles/r4_probe_bf16_t4_min_tiles.txt)
    return mode == 2 || (cus > 0 && (long)(a.M / T4_BM) * (a.N / T4_BN) * 100 >= (long)cus * PIPS_TUNE("PIPS_BF16_T4_MINPCT", 50));
}

int launch_gemm_bf16_t4(const GemmArgs& a, hipStream_t st) {
    const int tiles_n = a.N / T4_BN, ntiles = (a.M / T4_BM) * tiles_n;
    if (a.epi & EPI_RES_BF16) hipLaunchKernelGGL(gemm_bf16_t4_res_kernel<true>, dim3(ntiles), dim3(256), T4_LDS, st, a, tiles_n, ntiles);
    else hipLaunchKernelGGL(gemm_bf16_t4_res_kernel<false>, dim3(ntiles), dim3(256), T4_LDS, st, a, tiles_n, ntiles);
    PIPS_CHECK_LAUNCH("gemm_bf16_t4_res_kernel");
    return PIPS_OK;
}

// Which kernel a bf16-operand GEMM goes to: 0 = the register-staged gemm_bf16_kernel (gemm_bf16.hip), 3 = gemm_bf16_t4_res_kernel
// (down-projection + residual), 4 = gemm_bf16_t4_gelu_kernel (up-projection + GELU).  Pure function of the problem and the
// device's CU count -- behind pips_gemm_bf16_route(), which lets a test assert that a forward's geometry reaches these kernels.
// (1 and 2 were the LDS-DMA assembly kernels of rounds 2-3, gemm_bf16_asm.hip: removed in round 4.)
int gemm_bf16_asm_route(const GemmArgs& a, int a_bf16, int out_bf16) {
    if (gemm_bf16_t4_takes(a, a_bf16, out_bf16)) return 3;
    if (gemm_bf16_t4up_takes(a, a_bf16, out_bf16, nullptr)) return 4;
    return 0;
}

}  // namespace pips
