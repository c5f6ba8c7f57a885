// fp32-MFMA GEMM with direct global->LDS loads (global_load_lds_dwordx4) and a 3-stage ring.
//
// Same arithmetic and C^T-accumulator epilogue as gemm.hip's plain-GEMM path; what changes is
// how operands reach LDS.  gemm.hip stages each K block through VGPRs (global_load -> registers
// -> ds_write), one block ahead.  Here every wave issues LDS-DMA loads (1 KiB = 8 rows x 128 B per
// wave-instruction, no VGPR round trip, no ds_write pass) TWO K blocks ahead and waits with a
// counted s_waitcnt vmcnt(LPW), so a load has two MFMA phases to land instead of one.
//
// LDS image: per stage and per K-split half, A[BM][32] and B[BN][32] floats, rows UNPADDED (the
// DMA writes lane-linear: lane l lands at base + 16*l) with an XOR swizzle of the 16-byte slots,
// phys = slot ^ ((row >> 1) & 7), applied to the per-lane GLOBAL source address and to the
// fragment reads; for every 16-lane ds_read_b128 service group the 16 (row parity, slot) pairs
// are distinct -> conflict-free.
// Ordering: ONE raw s_barrier per K block.  Iteration kb: wait vmcnt (stage kb of THIS wave has
// landed; the loads of stage kb+1 may still fly) -> s_barrier (everybody's part of stage kb has
// landed, and everybody is done reading stage kb-1) -> issue stage kb+2 into the buffer stage kb-1
// used -> MFMAs on stage kb.
#include "common.h"

namespace pips {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int WGM, int WGN, int KS>
__global__ __launch_bounds__(WGM * WGN * KS * 64) void igemm_f32_glds_kernel(GemmArgs p) {
    constexpr int NWB = WGM * WGN;                   // waves per K-split group
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int HALF = (BM + BN) * 32;             // floats per (stage, ks) image: A rows then B rows
    constexpr int STAGE = HALF * KS;                 // floats per stage
    constexpr int NSTAGE = 3;
    constexpr int LPW = (BM + BN) / 8 / NWB;         // wave-instructions per wave per stage
    static_assert((BM + BN) % (8 * NWB) == 0 && BM % 8 == 0, "loader/tile mismatch");
    static_assert(KS == 1 || (KS - 1) * BM * BN <= NSTAGE * STAGE, "K-split reduction does not fit");

    // Fragment reads are issued from inline asm (ds_read_b128 + hand-counted lgkmcnt): hipcc then
    // sees no LDS load that could alias an in-flight LDS-DMA and inserts no vmcnt(0) of its own.
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / NWB, wmn = wave - ks * NWB;
    const int wm = wmn / WGN, wn = wmn % WGN;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // ---- loader: wave (ks, wmn) brings rows [wmn*LPW*8, +LPW*8) of the combined A|B row list of
    // its K half; lane -> row (lane>>3) of the 8-row group, physical slot lane&7
    static_assert(LPW <= 8, "extend the loader macros");
#define PIPS_Q(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define PIPS_LSRC(q)                                                                             \
    const float* lsrc##q = p.A;                                                                  \
    if constexpr (q < LPW) {                                                                     \
        const int cr = (wmn * LPW + q) * 8 + (lane >> 3);            /* combined row */          \
        const bool isA = (wmn * LPW + q) * 8 < BM;                   /* wave-uniform */          \
        const int row = isA ? cr : cr - BM;                          /* row inside its array */  \
        const int slot = (lane & 7) ^ ((row >> 1) & 7);              /* logical slot fetched */  \
        if (isA) {                                                                               \
            int m = m0 + row; m = m < p.M ? m : p.M - 1;             /* clamp: never stored */   \
            lsrc##q = p.A + (size_t)m * p.lda + ks * 32 + slot * 4;                              \
        } else {                                                                                 \
            int n = n0 + row; n = n < p.N ? n : p.N - 1;                                         \
            lsrc##q = p.W + (size_t)n * p.K + ks * 32 + slot * 4;                                \
        }                                                                                        \
    }
    PIPS_Q(PIPS_LSRC)
    const int ldst = ks * HALF + wmn * LPW * 8 * 32;                 // floats, wave-uniform

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define PIPS_ISSUE1(q)                                                                           \
    if constexpr (q < LPW)                                                                       \
        __builtin_amdgcn_global_load_lds((gptr_t)(lsrc##q + koff_), (lptr_t)(ldsb_ + q * 256), 16, 0, 0);
#define PIPS_ISSUE(kb_, buf_)                                                                    \
    {                                                                                            \
        const size_t koff_ = (size_t)(kb_) * (32 * KS);                                          \
        float* ldsb_ = smem + (buf_) * STAGE + ldst;                                             \
        PIPS_Q(PIPS_ISSUE1)                                                                      \
    }

    // fragment byte addresses inside a stage: row = tile row + l31, logical slot 2*kk + half, swizzled
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int sw = (l31 >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const unsigned a_byte = lds0 + (ks * HALF + (wm * WTM + l31) * 32) * 4;
    const unsigned b_byte = lds0 + (ks * HALF + BM * 32 + (wn * WTN + l31) * 32) * 4;
    unsigned koffb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffb[kk] = ((kk * 2 + half) ^ sw) * 16;

    const int nk = p.K / (32 * KS);
    PIPS_ISSUE(0, 0)
    if (nk > 1) PIPS_ISSUE(1, 1)
    int buf = 0;
    for (int kb = 0; kb < nk; ++kb) {
        if (kb + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kb + 2 < nk) {
            const int nb = buf >= 1 ? buf - 1 : NSTAGE - 1;          // buffer of stage kb-1 == (kb+2) % 3
            PIPS_ISSUE(kb + 2, nb)
        }
        const unsigned ab = a_byte + buf * (STAGE * 4), bb = b_byte + buf * (STAGE * 4);
        f32x4 fa[4][TM], fb[4][TN];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk][i]) : "v"(ab + koffb[kk]), "n"(i * 32 * 32 * 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[kk][j]) : "v"(bb + koffb[kk]), "n"(j * 32 * 32 * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk][j].x, fa[kk][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk][j].y, fa[kk][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk][j].z, fa[kk][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[kk][j].w, fa[kk][i].w, acc[i][j], 0, 0, 0);
                }
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    }
#undef PIPS_ISSUE
#undef PIPS_ISSUE1
#undef PIPS_LSRC
#undef PIPS_Q

    if (KS > 1) {
        __syncthreads();
        float* red = smem;
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((ks - 1) * NWB + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int g = 0; g < KS - 1; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += red[(((g * NWB + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
    }

    // epilogue on C^T accumulators (as gemm.hip): MFMA row = output column n, MFMA column = output row m
    const int epi = p.epi & 0xff;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WTM + i * 32 + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * WTN + j * 32 + 8 * g + 4 * half;
                if (col + 3 >= p.N) continue;                      // N % 4 == 0 (checked by the launcher)
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.bias != nullptr) {
                    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
                if (epi == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
                } else if (epi == EPI_RESIDUAL) {
                    const float4 r4 = *reinterpret_cast<const float4*>(p.R + (size_t)row * p.ldr + col);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int KS>
static int launch_glds_tile(const GemmArgs& a, hipStream_t st) {
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), 1);
    dim3 block(WGM * WGN * KS * 64);
    const size_t lds = (size_t)3 * (BM + BN) * 32 * KS * sizeof(float);
    auto kern = igemm_f32_glds_kernel<BM, BN, WGM, WGN, KS>;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    PIPS_CHECK_LAUNCH("igemm_f32_glds_kernel");
    return PIPS_OK;
}

// returns PIPS_OK if the problem was taken, 1 if the caller should use the register-staged kernel
int launch_gemm_glds(const GemmArgs& a, hipStream_t st) {
    if (a.N % 4 != 0 || a.ldc % 4 != 0 || a.lda % 4 != 0 || a.K % 32 != 0) return 1;
    if ((a.epi & 0xff) == EPI_RESIDUAL && a.ldr % 4 != 0) return 1;
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const long b64 = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    if (b128 >= 400) return 1;                                   // large problems: keep the 128x128 register-staged tile
    if (b64 >= 800 || a.K % 64 != 0) return launch_glds_tile<64, 64, 2, 2, 1>(a, st);
    return launch_glds_tile<64, 64, 2, 2, 2>(a, st);
}

}  // namespace pips
