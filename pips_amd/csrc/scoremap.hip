// Score-map loss terms (evaluation only): nets/pips.py:501-511 builds, per update iteration, the dense score map of
// every particle -- the correlation of its feature with every map pixel at the four pyramid levels, each level
// upsampled (bilinear, align_corners=True) to the level-0 size and summed -- and score_map_loss (:58-92) puts a
// balanced cross-entropy on it with a one-hot target at the rounded ground-truth position.  Both steps are linear in
// the maps up to the loss, so the four levels are upsampled and summed ONCE per forward into U (frames x H8 x W8 x 128),
// and a score map is one dense correlation of a particle feature with U: the (B,S,N,H8,W8) volume is never formed,
// only the two sums a heat map contributes to the loss.  (Summation order differs from the reference's
// sum-of-upsampled-correlations; evaluation metric, fp32.)
#include "common.h"

namespace pips {

struct ScoreLevels {
    size_t off[PIPS_LEVELS];
    int H[PIPS_LEVELS], W[PIPS_LEVELS];
    float sh[PIPS_LEVELS], sw[PIPS_LEVELS];        // area_pixel_compute_scale(align_corners=True): (in-1)/(out-1)
};

// U[f][y][x][c] = sum over levels of F.interpolate(level, (H8,W8), 'bilinear', align_corners=True)  (:505-510)
__global__ __launch_bounds__(256) void score_upsum_kernel(const float* __restrict__ pyramid, ScoreLevels lv, int C4,
                                                          float4* __restrict__ U, size_t total4) {
    const int Hd = lv.H[0], Wd = lv.W[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int x = (int)(p % Wd); p /= Wd;
        const int y = (int)(p % Hd);
        const int f = (int)(p / Hd);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                        // fcp = zeros (:503)
#pragma unroll
        for (int l = 0; l < PIPS_LEVELS; ++l) {
            const int Hs = lv.H[l], Ws = lv.W[l];
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            if (Hd == Hs) { y0 = y1 = y; ly0 = 1.f; ly1 = 0.f; }
            else {
                const float r = lv.sh[l] * (float)y;
                y0 = min((int)floorf(r), Hs - 1);
                ly1 = fminf(fmaxf(r - (float)y0, 0.f), 1.f);
                y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
                ly0 = 1.f - ly1;
            }
            if (Wd == Ws) { x0 = x1 = x; lx0 = 1.f; lx1 = 0.f; }
            else {
                const float r = lv.sw[l] * (float)x;
                x0 = min((int)floorf(r), Ws - 1);
                lx1 = fminf(fmaxf(r - (float)x0, 0.f), 1.f);
                x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
                lx0 = 1.f - lx1;
            }
            const float4* s4 = reinterpret_cast<const float4*>(pyramid + lv.off[l]) + (size_t)f * Hs * Ws * C4 + c4;
            const float4 v00 = s4[((size_t)y0 * Ws + x0) * C4], v01 = s4[((size_t)y0 * Ws + x1) * C4];
            const float4 v10 = s4[((size_t)y1 * Ws + x0) * C4], v11 = s4[((size_t)y1 * Ws + x1) * C4];
            acc.x += ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
            acc.y += ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
            acc.z += ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
            acc.w += ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        }
        U[i] = acc;
    }
}

// One block per mixer row m = (b*N+n)*S+s.  tgt[m] = {x, y, use}: rounded target pixel (map coordinates) and whether
// the heat map enters the loss (score_map_loss: target inside the map, valid > 0, vis > 0).  out[m] = {loss at the
// target pixel (label +1), sum of the losses of all other pixels (label -1)}, loss = balanced_ce_loss's stable
// softplus b + log(exp(-b) + exp(a-b)), a = -label * score, b = relu(a)  (:26-29).
__global__ __launch_bounds__(256) void score_terms_kernel(const float* __restrict__ U, int S, int N, int HW, int W8,
                                                          const float* __restrict__ ffeats,
                                                          const float* __restrict__ tgt, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float ff[PIPS_C];
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float use = tgt[(size_t)m * 3 + 2];
    if (!(use > 0.f)) {                                   // block-uniform
        if (tid == 0) { out[(size_t)m * 2 + 0] = 0.f; out[(size_t)m * 2 + 1] = 0.f; }
        return;
    }
    const int s = m % S, b = m / (S * N);
    const int f = b * S + s;
    if (tid < PIPS_C) ff[tid] = ffeats[(size_t)m * PIPS_C + tid];
    __syncthreads();
    const int pt = (int)tgt[(size_t)m * 3 + 1] * W8 + (int)tgt[(size_t)m * 3 + 0];
    const float scale = sqrtf((float)PIPS_C);
    const float4* Uf = reinterpret_cast<const float4*>(U) + (size_t)f * HW * (PIPS_C / 4);
    const float4* f4 = reinterpret_cast<const float4*>(ff);
    float neg = 0.f, pos = 0.f;
    for (int p = tid; p < HW; p += 256) {
        const float4* u = Uf + (size_t)p * (PIPS_C / 4);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int c = 0; c < PIPS_C / 4; ++c) {
            const float4 v = u[c], w = f4[c];
            a0 += v.x * w.x; a1 += v.y * w.y; a2 += v.z * w.z; a3 += v.w * w.w;
        }
        const float score = ((a0 + a1) + (a2 + a3)) / scale;               // corr / sqrt(C) (:397)
        const float a = p == pt ? -score : score;
        const float bb = fmaxf(a, 0.f);
        const float loss = bb + logf(expf(-bb) + expf(a - bb));
        if (p == pt) pos = loss; else neg += loss;
    }
    // block sums (the one thread that met the target pixel holds pos)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { neg += __shfl_xor(neg, o); pos += __shfl_xor(pos, o); }
    __shared__ float redp[4];
    if ((tid & 63) == 0) { red[tid >> 6] = neg; redp[tid >> 6] = pos; }
    __syncthreads();
    if (tid == 0) {
        out[(size_t)m * 2 + 0] = (redp[0] + redp[1]) + (redp[2] + redp[3]);
        out[(size_t)m * 2 + 1] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

int launch_score_upsum(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int F, float* U,
                       hipStream_t st) {
    ScoreLevels lv;
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        lv.off[l] = lvl_off[l]; lv.H[l] = lvlH[l]; lv.W[l] = lvlW[l];
        lv.sh[l] = lvlH[0] > 1 ? (float)(lvlH[l] - 1) / (float)(lvlH[0] - 1) : 0.f;
        lv.sw[l] = lvlW[0] > 1 ? (float)(lvlW[l] - 1) / (float)(lvlW[0] - 1) : 0.f;
    }
    const size_t total4 = (size_t)F * lvlH[0] * lvlW[0] * (PIPS_C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(score_upsum_kernel, dim3(blocks), dim3(256), 0, st, pyramid, lv, PIPS_C / 4,
                       reinterpret_cast<float4*>(U), total4);
    PIPS_CHECK_LAUNCH("score_upsum_kernel");
    return PIPS_OK;
}

int launch_score_terms(const float* U, int B, int S, int H8, int W8, const float* ffeats, int N, const float* tgt,
                       float* out, hipStream_t st) {
    hipLaunchKernelGGL(score_terms_kernel, dim3(B * N * S), dim3(256), 0, st, U, S, N, H8 * W8, W8, ffeats, tgt, out);
    PIPS_CHECK_LAUNCH("score_terms_kernel");
    return PIPS_OK;
}

}  // namespace pips
