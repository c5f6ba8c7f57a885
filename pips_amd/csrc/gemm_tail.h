// Tail of a GEMM / implicit-GEMM tile shared by the matrix-core kernels: K-split reduction,
// plain-GEMM epilogue (C^T accumulators) and convolution epilogue (C accumulators + the
// pivoted instance-norm partials).  Accumulator map of the 32x32 MFMA C/D operand:
// element r of lane l sits at MFMA row (r&3) + 8*(r>>2) + 4*(l>>5), MFMA column l&31.
#pragma once
#include "common.h"

namespace pips {

// Wave groups ks > 0 hand their accumulators to group 0 through LDS (red: >= (KS-1)*BM*BN floats,
// the pipeline stages are dead by now).  Returns false for the waves that are done.
template <int KS, int WG, int TM, int TN>
__device__ __forceinline__ bool ksplit_reduce(f32x16 (&acc)[TM][TN], float* red, int ks, int wmn, int lane) {
    if (KS == 1) return true;
    __syncthreads();                                  // all waves are done with the stages
    if (ks > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[((((ks - 1) * WG + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (ks > 0) return false;
#pragma unroll
    for (int g = 0; g < KS - 1; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += red[(((g * WG + wmn) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
    return true;
}

// Plain GEMM, C^T accumulators: acc[i][j][4g..4g+3] = C[row0 + 32 i][col0 + 32 j + 8 g + 0..3]
// with row0 = m0 + wave row offset + (lane&31), col0 = n0 + wave column offset + 4*(lane>>5).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM][TN], const GemmArgs& p, bool tile_inside,
                                              int row0, int col0) {
    const int epi = p.epi & 0xff;
    if (tile_inside && (p.ldc & 3) == 0 && (epi != EPI_RESIDUAL || (p.ldr & 3) == 0)) {
        if (epi == EPI_GELU) epilogue_full_tile<EPI_GELU, false, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        else if (epi == EPI_RESIDUAL) epilogue_full_tile<EPI_RESIDUAL, false, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        else epilogue_full_tile<EPI_BIAS, false, TM, TN>(acc, p.bias, p.R, p.ldr, p.C, p.ldc, row0, col0);
        return;
    }
    // edge tiles and odd strides: element-wise.  Loads are unconditional on clamped indices and
    // finish before the predicated stores (a load first used inside a predicated block makes
    // hipcc serialise every store behind a full s_waitcnt).
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row0 + i * 32;
        const int rowc = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = col0 + j * 32 + 8 * (r >> 2) + (r & 3);
                const int colc = col < p.N ? col : p.N - 1;
                t[r] = acc[i][j][r] + (p.bias != nullptr ? p.bias[colc] : 0.f);
                if (epi == EPI_GELU) t[r] = gelu_exact(t[r]);
                else if (epi == EPI_RESIDUAL) t[r] += p.R[(size_t)rowc * p.ldr + colc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = col0 + j * 32 + 8 * (r >> 2) + (r & 3);
                if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = t[r];
            }
        }
    }
}

// Convolution, C accumulators: raw output + bias, per-(frame, m-tile x wave row, channel) pivoted partials
// (store_conv_partial, common.h) from the stored values (no atomics, no LDS: bitwise deterministic).
// Bias is added to every accumulator BEFORE the (predicated) stores: a load first used inside a
// predicated block makes hipcc wait for the previous store's acknowledgement in front of each store.
template <int BM, int BN, int WGM, int WTM, int WTN, int NT, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[TM][TN], const GemmArgs& p, float* Cframe,
                                              float* red, int frame, int m_tile, int m0, int n0, int wm, int wn,
                                              int l31, int half, int tid) {
    float csum[TN], csq[TN], piv[TN];
    const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        csum[j] = csq[j] = 0.f;
        const int col = n0 + wn * WTN + j * 32 + l31;
        const bool col_ok = col < p.N;
        const float bv = p.bias != nullptr ? p.bias[col_ok ? col : p.N - 1] : 0.f;
        piv[j] = __shfl(acc[0][j][0] + bv, l31);      // the wave's first row in this column (lanes of half 0, r = 0)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            const int rbase = m0 + wm * WTM + i * 32 + 4 * half;
            float* cp = Cframe + (size_t)rbase * p.ldc + col;
            if (full_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
                    const float d = v[r] - piv[j];
                    csum[j] += d;
                    csq[j] += d * d;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M && col_ok) {
                        cp[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v[r];
                        const float d = v[r] - piv[j];
                        csum[j] += d;
                        csq[j] += d * d;
                    }
                }
            }
        }
    }
    if (p.stats == nullptr) return;
    const int left = p.M - (m0 + wm * WTM), nvalid = left < 0 ? 0 : (left > WTM ? WTM : left);
#pragma unroll
    for (int j = 0; j < TN; ++j)
        store_conv_partial(p.stats, frame, (int)gridDim.x * WGM, m_tile * WGM + wm, p.N, n0 + wn * WTN + j * 32 + l31, half, csum[j],
                           csq[j], piv[j], nvalid);
}

}  // namespace pips
