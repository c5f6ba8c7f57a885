// The token-mixing half of a mixer block for the bf16-operand mixer, one WAVE per particle (nets/pips.py:93-109,117-118):
//   y  = x + Conv1d(32->8)(GELU(Conv1d(8->32)(LN1(x))))     tokens = the S = 8 axis
//   xn = LN2(y)                                             (bf16: the A operand of the 512 -> 2048 up-projection)
// Shared by token_mix_mfma_kernel (track.hip: one launch per layer below M = 16384) and mixer_layer_kernel (ffn_fused.hip:
// the whole mixer layer in one launch, where it is the prologue that fills the block's xn tile in LDS).
//
// Under torch.autocast the token-mixing Conv1d layers take bf16 operands like every other Linear, so the 8 -> 32 -> 8 MLP
// per channel runs as three v_mfma_f32_32x32x16_bf16 per 32 channels:
//   H[32 hidden][32 ch] = W0[32][8 tok -> K = 16, zero padded] * Xn[tok][ch]        (1 MFMA; N = channels)
//   Y[8 tok -> M = 32][32 ch] = W3[tok][32 hidden] * gelu(H + b0)                    (2 MFMAs of K = 16)
// A lane owns 4 tokens (lanes 0-31: tokens 0-3, lanes 32-63: tokens 4-7) x 16 channels (c = 128 g + 4 (lane & 31) + q): the
// fp32 loads are float4s, the MFMA for channel slot (g, q) takes the lane's four tokens of that channel as its K values, H
// comes back with this lane's channel in all 16 registers (hidden units m(r) = (r&3) + 8(r>>2) + 4*half) -- exactly the K
// values the second product wants from this lane once W3's columns are permuted the same way -- and Y's rows 0-7 are the
// lane's own four tokens again: no cross-lane traffic between the three products.  LayerNorm statistics in one pass (sum,
// sum of squares; the operands are rounded to bf16 anyway) inside the wave (one DPP transpose-reduce, no LDS, no barrier);
// fp32 residual stream.
#pragma once
#include "common.h"

namespace pips {

typedef __bf16 bf16x8_tm __attribute__((ext_vector_type(8)));

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK_MASK, false));
}
// Eight wave sums at once by transpose-reduce: three exchange steps (lane^1, ^2, ^4) in which a lane
// keeps the half of its values that matches its lane bit and adds the partner's copy of it -- 8 -> 4
// -> 2 -> 1 value per lane -- then three plain steps (^8, ^16, ^32).  Lane l returns the sum over the
// wave of v[l & 7]: 26 instructions against ~200 for eight separate wave reductions.  DPP quad_perm /
// row_shl / row_shr / row_ror within rows, ds_swizzle across rows, one bpermute across the halves.
// (Neutral at B=1, where one block per CU leaves the kernel latency-bound; -20 % at 2048+ particles,
// where it is VALU-issue bound.)
__device__ __forceinline__ float wave_sum8(const float (&v)[PIPS_S]) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w[i] = keep + dpp_mov<0xB1, 0xf>(0.f, send);                   // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b1 ? w[2 * j + 1] : w[2 * j], send = b1 ? w[2 * j] : w[2 * j + 1];
        x[j] = keep + dpp_mov<0x4E, 0xf>(0.f, send);                   // quad_perm [2,3,0,1]
    }
    const float keep = b2 ? x[1] : x[0], send = b2 ? x[0] : x[1];
    float recv = dpp_mov<0x104, 0x5>(0.f, send);                       // row_shl:4 into lanes 0-3, 8-11
    recv = dpp_mov<0x114, 0xa>(recv, send);                            // row_shr:4 into lanes 4-7, 12-15
    float y = keep + recv;
    y += dpp_mov<0x128, 0xf>(0.f, y);                                  // row_ror:8  (lane ^ 8)
    y += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(y), 0x401F));   // lane ^ 16
    y += __shfl_xor(y, 32);
    return y;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// 1/sqrt(v) for v >= eps: v_rsq_f32 (1 ulp) + one Newton step -- 5 instructions where the IEEE 1.0f / sqrtf(v) of the
// compiler is ~30 (denormal scaling, div_scale / div_fmas / div_fixup); every thread needs it 16 times per launch
__device__ __forceinline__ float rsqrt_nr(float v) {
    const float r = __builtin_amdgcn_rsqf(v);
    return r * fmaf(-0.5f * v * r, r, 1.5f);
}

// The wave-resident weights of the token MLP as MFMA A fragments + the biases in the accumulator layouts.
struct TokenMixFrags {
    uint4 a1, a2[2];
    float b0r[16], b3r[4];
};
__device__ __forceinline__ void token_mix_load_frags(const float* __restrict__ arena, const MixLayerW& L, int l31, int half, TokenMixFrags& F) {
    const float* w0 = arena + L.tw0 + l31 * 8 + 4 * half;                 // w0[hidden = l31][token 4*half + i]
    F.a1 = make_uint4(pack2_bf16(w0[0], w0[1]), pack2_bf16(w0[2], w0[3]), 0u, 0u);
    // w3[token = l31 (< 8)][hidden]: register r of the lane = hidden unit (r & 3) + 8 (r >> 2) + 4 half, i.e. four runs of four
    // consecutive floats.  Four UNCONDITIONAL 16-byte loads, masked afterwards: written as `l31 < 8 ? pack(w3[..]) : 0` hipcc
    // sank every load into its own predicated block -- eight load -> s_waitcnt vmcnt(0) round trips in a row at the head of
    // every wave, before the particle's own tile was even requested
    const float4* w3 = reinterpret_cast<const float4*>(arena + L.tw3 + (l31 & 7) * 32 + 4 * half);
    const float4 wq[4] = {w3[0], w3[2], w3[4], w3[6]};
    const unsigned keep = l31 < 8 ? 0xffffffffu : 0u;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        const float4 p0 = wq[2 * kc], p1 = wq[2 * kc + 1];
        F.a2[kc] = make_uint4(pack2_bf16(p0.x, p0.y) & keep, pack2_bf16(p0.z, p0.w) & keep, pack2_bf16(p1.x, p1.y) & keep,
                              pack2_bf16(p1.z, p1.w) & keep);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) F.b0r[r] = arena[L.tb0 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
    for (int r = 0; r < 4; ++r) F.b3r[r] = arena[L.tb3 + 4 * half + r];
}

// per-token mean / rstd of the lane's four tokens: one pass of sums, reduced over the wave
__device__ __forceinline__ void token_mix_ln_stats(const float (&v)[4][16], int half, float (&mean)[4], float (&rstd)[4]) {
    float s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            a += v[r][k]; b += v[r][k + 1];
            c = fmaf(v[r][k], v[r][k], c); d = fmaf(v[r][k + 1], v[r][k + 1], d);
        }
        s1[r] = a + b; s2[r] = c + d;
    }
    float sa[PIPS_S], sb[PIPS_S];
#pragma unroll
    for (int t = 0; t < PIPS_S; ++t) {
        const bool mine = (t >> 2) == half;
        sa[t] = mine ? s1[t & 3] : 0.f;
        sb[t] = mine ? s2[t & 3] : 0.f;
    }
    const float ta = wave_sum8(sa), tb = wave_sum8(sb);                   // lane l: totals of token l & 7
    const float m = ta * (1.0f / PIPS_DMIX);
    const float var = fmaxf(tb * (1.0f / PIPS_DMIX) - m * m, 0.f);
    const float rs = rsqrt_nr(var + 1e-5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float m_lo = lane_bcast(m, r), m_hi = lane_bcast(m, 4 + r);
        const float r_lo = lane_bcast(rs, r), r_hi = lane_bcast(rs, 4 + r);
        mean[r] = half ? m_hi : m_lo;
        rstd[r] = half ? r_hi : r_lo;
    }
}

// The token MLP of channel slot c (of 16) of one particle, in place on the residual stream xv[token r][c].
__device__ __forceinline__ void token_mix_slot(float (&xv)[4][16], int c, const float (&mean)[4], const float (&rstd)[4], float g1, float be1,
                                               const TokenMixFrags& F) {
    float n[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) n[r] = (xv[r][c] - mean[r]) * (rstd[r] * g1) + be1;
    const uint4 bx = make_uint4(pack2_bf16(n[0], n[1]), pack2_bf16(n[2], n[3]), 0u, 0u);
    f32x16 h;
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = F.b0r[r];                             // the MFMA accumulates onto the bias
    h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&F.a1), *reinterpret_cast<const bf16x8_tm*>(&bx), h, 0, 0, 0);
    unsigned hb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f2 gq = gelu_exact2((f2){h[2 * i], h[2 * i + 1]});
        hb[i] = pack2_bf16(gq.x, gq.y);
    }
    const uint4 k0 = make_uint4(hb[0], hb[1], hb[2], hb[3]), k1 = make_uint4(hb[4], hb[5], hb[6], hb[7]);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = r < 4 ? F.b3r[r] : 0.f;
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&F.a2[0]), *reinterpret_cast<const bf16x8_tm*>(&k0), o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_tm*>(&F.a2[1]), *reinterpret_cast<const bf16x8_tm*>(&k1), o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) xv[r][c] += o[r];                             // residual stream, in place
}

// NP particles of one wave, side by side (their instruction streams are independent: the MFMA -> GELU -> MFMA chain of one
// fills the waits of the other).  xp[i]: the lane's corner of particle i's rows in the fp32 residual stream (global;
// + r * 512 + g * 128 floats), rewritten in place.  sink(i, r, g, v): takes the lane's 4 bf16 of LayerNorm-2 of particle i,
// token r (of the lane's four), channel group g (channels 128 g + 4 (lane & 31) ..+3) -- a global row or an LDS tile.
template <int NP, class Sink>
__device__ __forceinline__ void token_mix_mfma_particles(const float* __restrict__ arena, const MixLayerW& L, float* const (&xp)[NP],
                                                         Sink sink, int l31, int half) {
    // ---- the particles' tiles first: their loads are the long ones (HBM / Infinity Cache), the weights below hit L2
    float xv[NP][4][16];                                                      // [particle][token r][g * 4 + q]
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(xp[i] + r * PIPS_DMIX + g * 128);
                xv[i][r][4 * g] = v.x; xv[i][r][4 * g + 1] = v.y; xv[i][r][4 * g + 2] = v.z; xv[i][r][4 * g + 3] = v.w;
            }
    TokenMixFrags F;
    token_mix_load_frags(arena, L, l31, half, F);
    float mean[NP][4], rstd[NP][4];
#pragma unroll
    for (int i = 0; i < NP; ++i) token_mix_ln_stats(xv[i], half, mean[i], rstd[i]);

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 g1 = *reinterpret_cast<const float4*>(arena + L.ln1g + g * 128 + 4 * l31);
        const float4 be1 = *reinterpret_cast<const float4*>(arena + L.ln1b + g * 128 + 4 * l31);
        const float g1a[4] = {g1.x, g1.y, g1.z, g1.w}, be1a[4] = {be1.x, be1.y, be1.z, be1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NP; ++i) token_mix_slot(xv[i], 4 * g + q, mean[i], rstd[i], g1a[q], be1a[q], F);
        // the new residual stream of these 128 channels goes out while the next group is computed (all waves of the
        // launch run in one round, in lock-step: stores held back to the end would queue behind one another)
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(xp[i] + r * PIPS_DMIX + g * 128) =
                    make_float4(xv[i][r][4 * g], xv[i][r][4 * g + 1], xv[i][r][4 * g + 2], xv[i][r][4 * g + 3]);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) token_mix_ln_stats(xv[i], half, mean[i], rstd[i]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 g2 = *reinterpret_cast<const float4*>(arena + L.ln2g + g * 128 + 4 * l31);
        const float4 be2 = *reinterpret_cast<const float4*>(arena + L.ln2b + g * 128 + 4 * l31);
        const float g2a[4] = {g2.x, g2.y, g2.z, g2.w}, be2a[4] = {be2.x, be2.y, be2.z, be2.w};
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float n[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) n[q] = (xv[i][r][4 * g + q] - mean[i][r]) * (rstd[i][r] * g2a[q]) + be2a[q];
                sink(i, r, g, make_uint2(pack2_bf16(n[0], n[1]), pack2_bf16(n[2], n[3])));
            }
    }
}

}  // namespace pips
