// PARKED EXPERIMENT (round 4) -- not built, not part of the product library.  Was a kernel of pips_amd/csrc/track.hip behind
// launch_token_mix (commit 29368b5 + 1); needs that file's helpers (ln_stats2, gelu_exact2, S, MixLayerW).
// Measured (profiles/r4_probe_token_mix_f32_mfma_ab.txt): 9.34 us per launch at 256 particles (the VALU token_mix_kernel: 8.99),
// 233 us at 16384 particles (203); the fp32 mixer pass is unchanged within noise at M = 2048 and M = 131072.  The matrix pipe has
// the same fp32 rate as the packed vector FMAs and the two do not overlap on a SIMD (DESIGN.md 4b iii), so moving the 512 FMAs
// per channel to v_mfma_f32_4x4x1_16B_f32 buys nothing; the A operands cost 46 LDS reads + hazard nops on top.
// ---------------------------------------------------------------------------------------------------------------------
// token_mix_kernel with the token MLP's 512 fp32 FMAs per channel on the matrix cores, EXACT fp32: v_mfma_f32_4x4x1_16B_f32
// is sixteen independent 4 x 4 outer products per instruction -- D_b[i][j] += A_b[i] * B_b[j], one fused multiply-add per
// element, lane 4b + j holding column j of block b in four registers (rows i).  With B = the lane's own channel value and A =
// four weights that depend on (lane & 3) only,
//     H[4 hg + i][ch(lane)] += W0[4 hg + i][t] * xn[t][ch(lane)]            (8 tokens x 8 hidden groups  = 64 instructions)
//     Y[4 tg + i][ch(lane)] += W3[4 tg + i][j] * gelu(H)[j][ch(lane)]       (32 hidden x 2 token groups = 64 instructions)
// every lane ends up with the 32 hidden units / 8 tokens of ITS channel in registers: lane = channel exactly as in the VALU
// form, no padding (a 32x32x2 tile would spend 3/4 of the second product on zero rows), the same summation order (tokens /
// hidden units ascending onto the bias), so the result is what the fmaf loop of token_mix_kernel gives.  The matrix pipe has
// the same fp32 peak as the packed vector FMAs (157 TF); what this buys is the VECTOR pipe: 512 of the kernel's ~2 000
// vector instructions per wave (the half-rate packed FMAs) leave it, and the GELU / LayerNorm work of one channel set runs
// beside the MFMAs of the other.  Thread = channels 2 tid, 2 tid + 1 = channel sets 0 / 1 of its lane.
typedef float f32x4_tm __attribute__((ext_vector_type(4)));

template <bool XN_BF16>
__global__ __launch_bounds__(256, 2) void token_mix_f32mfma_kernel(const float* __restrict__ arena, MixLayerW L,
                                                                   float* __restrict__ x, float* __restrict__ xn) {
    __shared__ __attribute__((aligned(16))) float red[S][4];
    __shared__ __attribute__((aligned(16))) float wsm[32 * 8 + 32 + 8 * 32 + 8];
    const int tid = threadIdx.x;
    float* xp = x + (size_t)blockIdx.x * S * PIPS_DMIX + 2 * tid;
    float* xnp = xn + (size_t)blockIdx.x * S * PIPS_DMIX + 2 * tid;
    f2 xv[S];
    float mean[S], rstd[S];
#pragma unroll
    for (int t = 0; t < S; ++t) xv[t] = *reinterpret_cast<const f2*>(xp + t * PIPS_DMIX);
    const f2 g1 = *reinterpret_cast<const f2*>(arena + L.ln1g + 2 * tid), be1 = *reinterpret_cast<const f2*>(arena + L.ln1b + 2 * tid);
    const f2 g2 = *reinterpret_cast<const f2*>(arena + L.ln2g + 2 * tid), be2 = *reinterpret_cast<const f2*>(arena + L.ln2b + 2 * tid);
    {   // w0[32][8], b0[32], w3[8][32], b3[8] -> LDS (all four values requested before the first is stored)
        const float w0v = arena[L.tw0 + tid], w3v = arena[L.tw3 + tid];
        const float b0v = arena[L.tb0 + (tid & 31)], b3v = arena[L.tb3 + (tid & 7)];
        wsm[tid] = w0v;
        wsm[288 + tid] = w3v;
        if (tid < 32) wsm[256 + tid] = b0v;
        if (tid >= 64 && tid < 72) wsm[544 + (tid - 64)] = b3v;
    }
    ln_stats2(xv, mean, rstd, red);          // (its barriers also publish wsm)

    // A operands: the lane's row (lane & 3) of every 4-row weight block
    const int l3 = tid & 3;
    float wa0[8][S], wa3[2][32];
#pragma unroll
    for (int hg = 0; hg < 8; ++hg) {
        const float4 p = *reinterpret_cast<const float4*>(&wsm[(4 * hg + l3) * 8]), q = *reinterpret_cast<const float4*>(&wsm[(4 * hg + l3) * 8 + 4]);
        wa0[hg][0] = p.x; wa0[hg][1] = p.y; wa0[hg][2] = p.z; wa0[hg][3] = p.w;
        wa0[hg][4] = q.x; wa0[hg][5] = q.y; wa0[hg][6] = q.z; wa0[hg][7] = q.w;
    }
#pragma unroll
    for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            const float4 p = *reinterpret_cast<const float4*>(&wsm[288 + (4 * tg + l3) * 32 + j]);
            wa3[tg][j] = p.x; wa3[tg][j + 1] = p.y; wa3[tg][j + 2] = p.z; wa3[tg][j + 3] = p.w;
        }
    f2 h[S], y[S];
#pragma unroll
    for (int t = 0; t < S; ++t) h[t] = (xv[t] - (f2){mean[t], mean[t]}) * (g1 * (f2){rstd[t], rstd[t]}) + be1;

    f32x4_tm H[2][8];
#pragma unroll
    for (int hg = 0; hg < 8; ++hg) {
        const float4 b = *reinterpret_cast<const float4*>(&wsm[256 + 4 * hg]);
        H[0][hg] = H[1][hg] = (f32x4_tm){b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
        for (int t = 0; t < S; ++t)
#pragma unroll
            for (int hg = 0; hg < 8; ++hg)
                H[cs][hg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa0[hg][t], cs ? h[t].y : h[t].x, H[cs][hg], 0, 0, 0);
    f32x4_tm Y[2][2];
    {
        const float4 b3a = *reinterpret_cast<const float4*>(&wsm[544]), b3b = *reinterpret_cast<const float4*>(&wsm[548]);
        Y[0][0] = Y[1][0] = (f32x4_tm){b3a.x, b3a.y, b3a.z, b3a.w};
        Y[0][1] = Y[1][1] = (f32x4_tm){b3b.x, b3b.y, b3b.z, b3b.w};
    }
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
        for (int hg = 0; hg < 8; ++hg) {
            const f2 ga = gelu_exact2((f2){H[cs][hg][0], H[cs][hg][1]}), gb = gelu_exact2((f2){H[cs][hg][2], H[cs][hg][3]});
            const float gl[4] = {ga.x, ga.y, gb.x, gb.y};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tg = 0; tg < 2; ++tg)
                    Y[cs][tg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa3[tg][4 * hg + i], gl[i], Y[cs][tg], 0, 0, 0);
        }
#pragma unroll
    for (int t = 0; t < S; ++t) y[t] = (f2){Y[0][t >> 2][t & 3], Y[1][t >> 2][t & 3]} + xv[t];

    ln_stats2(y, mean, rstd, red);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        *reinterpret_cast<f2*>(xp + t * PIPS_DMIX) = y[t];
        const f2 o = (y[t] - (f2){mean[t], mean[t]}) * (g2 * (f2){rstd[t], rstd[t]}) + be2;
        if (XN_BF16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t ob = __builtin_convertvector(o, bf16x2_t);
            reinterpret_cast<unsigned*>(xn)[((size_t)blockIdx.x * S + t) * (PIPS_DMIX / 2) + tid] = *reinterpret_cast<const unsigned*>(&ob);
        } else {
            *reinterpret_cast<f2*>(xnp + t * PIPS_DMIX) = o;
        }
    }
}

