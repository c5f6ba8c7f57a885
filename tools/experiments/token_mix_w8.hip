// Parked experiment (round 4): cut out of pips_amd/csrc/track.hip -- needs its helpers (wave_sum8, lane_bcast, rsqrt_nr, gelu_exact2, f2, S).
// launch: hipLaunchKernelGGL(token_mix_w8_kernel<false>, dim3(particles), dim3(512), 0, st, arena, L, x, xn);

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same layer step on EIGHT waves -- thread = ONE channel, the token axis in the packed lanes (pairs of tokens).
// token_mix_kernel above is latency-bound at the headline's 256 particles (one block per CU = one wave per SIMD: 2.0 k instructions
// per wave at 10.7 clocks each, 9 us); here a particle's work is 8 x ~0.8 k instructions on two waves per SIMD.  Per pair of hidden
// units: 2 x (4 packed FMAs over token pairs + 1 add) for the first Linear, ONE packed GELU for the pair, 8 packed FMAs for the
// second Linear (the hidden value broadcast over a token pair).  Weights in LDS as w0[j][t] and w3^T[j][t] (16-byte broadcast reads).
// Same arithmetic per element as above except the order of the eight-term token sums (even + odd tokens).
__device__ __forceinline__ float block_sum8_w8(const float (&v)[S], float (*red)[8]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float w = wave_sum8(v);
    __syncthreads();                                   // previous readers of red are done
    if (lane < S) red[lane][wave] = w;
    __syncthreads();
    const float4 r0 = *reinterpret_cast<const float4*>(&red[lane & 7][0]), r1 = *reinterpret_cast<const float4*>(&red[lane & 7][4]);
    return ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
}
__device__ __forceinline__ void ln_stats_w8(const f2 (&x)[S / 2], float (&mean)[S], float (&rstd)[S], float (*red)[8]) {
    float a[S];
#pragma unroll
    for (int i = 0; i < S / 2; ++i) { a[2 * i] = x[i].x; a[2 * i + 1] = x[i].y; }
    const float m = block_sum8_w8(a, red) * (1.0f / PIPS_DMIX);
#pragma unroll
    for (int t = 0; t < S; ++t) mean[t] = lane_bcast(m, t);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const float d = a[t] - mean[t];
        a[t] = d * d;
    }
    const float r = rsqrt_nr(block_sum8_w8(a, red) * (1.0f / PIPS_DMIX) + 1e-5f);
#pragma unroll
    for (int t = 0; t < S; ++t) rstd[t] = lane_bcast(r, t);
}

template <bool XN_BF16>
__global__ __launch_bounds__(512) void token_mix_w8_kernel(const float* __restrict__ arena, MixLayerW L,
                                                           float* __restrict__ x, float* __restrict__ xn) {
    __shared__ __attribute__((aligned(16))) float red[S][8];
    __shared__ __attribute__((aligned(16))) float wsm[256 + 256 + 32 + 8];      // w0[j][t] | w3^T[j][t] | b0[j] | b3[t]
    const int c = threadIdx.x;
    float* xp = x + (size_t)blockIdx.x * S * PIPS_DMIX + c;
    f2 xv[S / 2];
#pragma unroll
    for (int i = 0; i < S / 2; ++i) { xv[i].x = xp[(2 * i) * PIPS_DMIX]; xv[i].y = xp[(2 * i + 1) * PIPS_DMIX]; }
    const float g1 = arena[L.ln1g + c], be1 = arena[L.ln1b + c], g2 = arena[L.ln2g + c], be2 = arena[L.ln2b + c];
    if (c < 256) {
        const float w0v = arena[L.tw0 + c], w3v = arena[L.tw3 + c];             // w0 [32][8]; w3 [8][32] -> transposed
        wsm[c] = w0v;
        wsm[256 + (c & 31) * 8 + (c >> 5)] = w3v;
    } else if (c < 288) {
        wsm[512 + (c - 256)] = arena[L.tb0 + (c - 256)];
    } else if (c < 296) {
        wsm[544 + (c - 288)] = arena[L.tb3 + (c - 288)];
    }
    float mean[S], rstd[S];
    ln_stats_w8(xv, mean, rstd, red);          // (its barriers also publish wsm)

    f2 h[S / 2], y[S / 2];
#pragma unroll
    for (int i = 0; i < S / 2; ++i) {
        const f2 m2 = {mean[2 * i], mean[2 * i + 1]}, r2 = {rstd[2 * i] * g1, rstd[2 * i + 1] * g1};
        h[i] = (xv[i] - m2) * r2 + (f2){be1, be1};
        y[i] = *reinterpret_cast<const f2*>(&wsm[544 + 2 * i]);
    }
    const float4* w0q = reinterpret_cast<const float4*>(wsm);
    const float4* w3q = reinterpret_cast<const float4*>(wsm + 256);
#pragma unroll 4
    for (int jp = 0; jp < 16; ++jp) {
        const float4 a0 = w0q[4 * jp], a1 = w0q[4 * jp + 1], b0 = w0q[4 * jp + 2], b1 = w0q[4 * jp + 3];
        const f2 bias = *reinterpret_cast<const f2*>(&wsm[512 + 2 * jp]);
        f2 s0 = {bias.x, 0.f}, s1 = {bias.y, 0.f};
        s0 = h[0] * (f2){a0.x, a0.y} + s0; s0 = h[1] * (f2){a0.z, a0.w} + s0; s0 = h[2] * (f2){a1.x, a1.y} + s0; s0 = h[3] * (f2){a1.z, a1.w} + s0;
        s1 = h[0] * (f2){b0.x, b0.y} + s1; s1 = h[1] * (f2){b0.z, b0.w} + s1; s1 = h[2] * (f2){b1.x, b1.y} + s1; s1 = h[3] * (f2){b1.z, b1.w} + s1;
        const f2 u = gelu_exact2((f2){s0.x + s0.y, s1.x + s1.y});
        const float4 c0 = w3q[4 * jp], c1 = w3q[4 * jp + 1], d0 = w3q[4 * jp + 2], d1 = w3q[4 * jp + 3];
        const f2 u0 = {u.x, u.x}, u1 = {u.y, u.y};
        y[0] = u0 * (f2){c0.x, c0.y} + y[0]; y[1] = u0 * (f2){c0.z, c0.w} + y[1]; y[2] = u0 * (f2){c1.x, c1.y} + y[2]; y[3] = u0 * (f2){c1.z, c1.w} + y[3];
        y[0] = u1 * (f2){d0.x, d0.y} + y[0]; y[1] = u1 * (f2){d0.z, d0.w} + y[1]; y[2] = u1 * (f2){d1.x, d1.y} + y[2]; y[3] = u1 * (f2){d1.z, d1.w} + y[3];
    }
#pragma unroll
    for (int i = 0; i < S / 2; ++i) y[i] += xv[i];

    ln_stats_w8(y, mean, rstd, red);
#pragma unroll
    for (int i = 0; i < S / 2; ++i) {
        xp[(2 * i) * PIPS_DMIX] = y[i].x;
        xp[(2 * i + 1) * PIPS_DMIX] = y[i].y;
        const f2 m2 = {mean[2 * i], mean[2 * i + 1]}, r2 = {rstd[2 * i] * g2, rstd[2 * i + 1] * g2};
        const f2 o = (y[i] - m2) * r2 + (f2){be2, be2};
        if (XN_BF16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t ob = __builtin_convertvector(o, bf16x2_t);
            const unsigned pk = *reinterpret_cast<const unsigned*>(&ob);
            unsigned short* xb = reinterpret_cast<unsigned short*>(xn) + (size_t)blockIdx.x * S * PIPS_DMIX + c;
            xb[(2 * i) * PIPS_DMIX] = (unsigned short)(pk & 0xffffu);
            xb[(2 * i + 1) * PIPS_DMIX] = (unsigned short)(pk >> 16);
        } else {
            float* xnp = xn + (size_t)blockIdx.x * S * PIPS_DMIX + c;
            xnp[(2 * i) * PIPS_DMIX] = o.x;
            xnp[(2 * i + 1) * PIPS_DMIX] = o.y;
        }
    }
}

