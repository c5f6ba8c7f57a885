// C ABI of libpips_hip.so (include/pips_hip.h): weight arena, stage entry points and the
// whole-forward driver that replaces Pips.forward (nets/pips.py:428-611).  Everything here
// is host-side launch logic; no allocation, no synchronisation, no global mutable state.
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace pips {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef PIPS_TUNING
int tune_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

int device_cus() {
    static std::atomic<int> cache[64];                 // zero-initialised; one slot per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::atomic<int>& slot = cache[dev & 63];
    int v = slot.load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 0;
    slot.store(v, std::memory_order_relaxed);
    return v;
}

// ------------------------------------------------------------------ arena layout
// conv geometry in execution (= state-dict) order: nets/pips.py:206-223, 135-136, 169-170
static ArenaLayout build_layout(int S) {
    ArenaLayout A;
    memset(&A, 0, sizeof(A));
    A.S = S; A.nout = S * (PIPS_C + 2); A.nout_pad = (A.nout + 3) & ~3;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };   // 256-B aligned
    int ci = 0, idx = 0;
    auto add_conv = [&](int cout, int cin, int k, int s, int p) {
        ConvW& c = A.conv[idx++];
        c.cout = cout; c.cin = cin; c.k = k; c.stride = s; c.pad = p;
        c.w = take((size_t)cout * cin * k * k);
        c.b = take(cout);
    };
    add_conv(64, 3, 7, 2, 3);
    ci = 64;
    const int dims[4] = {64, 96, 128, 128}, strides[4] = {1, 2, 2, 2};
    for (int l = 0; l < 4; ++l)
        for (int b = 0; b < 2; ++b) {
            const int s = b == 0 ? strides[l] : 1;
            add_conv(dims[l], ci, 3, s, 1);
            add_conv(dims[l], dims[l], 3, 1, 1);
            if (s != 1) add_conv(dims[l], ci, 1, s, 0);
            ci = dims[l];
        }
    add_conv(256, 416, 3, 1, 1);
    add_conv(128, 256, 1, 1, 0);
    A.w_in = take((size_t)PIPS_DMIX * PIPS_KIN_PAD);
    A.b_in = take(PIPS_DMIX);
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        MixLayerW& L = A.mix[d];
        L.ln1g = take(PIPS_DMIX); L.ln1b = take(PIPS_DMIX);
        L.w1 = take((size_t)4 * PIPS_DMIX * PIPS_DMIX); L.b1 = take(4 * PIPS_DMIX);
        L.w2 = take((size_t)4 * PIPS_DMIX * PIPS_DMIX); L.b2 = take(PIPS_DMIX);
        L.ln2g = take(PIPS_DMIX); L.ln2b = take(PIPS_DMIX);
    }
    A.lnf_g = take(PIPS_DMIX); A.lnf_b = take(PIPS_DMIX);
    A.norm_g = take(PIPS_C); A.norm_b = take(PIPS_C);
    A.w_upd_t = take(PIPS_C * PIPS_C); A.b_upd = take(PIPS_C);
    A.w_vis = take(PIPS_C); A.b_vis = take(1);
    A.total = off;
    size_t hoff = 0;
    auto take_h = [&](size_t n) { size_t o = hoff; hoff += (n + 127) / 128 * 128; return o; };   // 256-B aligned
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        A.h_w1[d] = take_h((size_t)4 * PIPS_DMIX * PIPS_DMIX);
        A.h_w2[d] = take_h((size_t)4 * PIPS_DMIX * PIPS_DMIX);
    }
    A.h_conv[0] = 0;                                   // the 7x7 stem stays fp32 (VALU kernel)
    for (int i = 1; i < 22; ++i) A.h_conv[i] = take_h((size_t)A.conv[i].cout * A.conv[i].cin * A.conv[i].k * A.conv[i].k);
    A.h_in = take_h((size_t)PIPS_DMIX * PIPS_KIN_PAD);
    A.total_h = hoff;
    size_t toff = 0;
    auto take_t = [&](size_t n) { size_t o = toff; toff += (3 * n + 127) / 128 * 128; return o; };
    A.t_in = take_t((size_t)PIPS_DMIX * PIPS_KIN_PAD);
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        A.t_w1[d] = take_t((size_t)4 * PIPS_DMIX * PIPS_DMIX);
        A.t_w2[d] = take_t((size_t)4 * PIPS_DMIX * PIPS_DMIX);
    }
    A.t_conv[0] = 0;
    for (int i = 1; i < 22; ++i) A.t_conv[i] = take_t((size_t)A.conv[i].cout * A.conv[i].cin * A.conv[i].k * A.conv[i].k);
    A.total_t = toff;
    // ---- the S-dependent block, behind the three sections (offsets in floats from the arena base)
    off = A.total + (A.total_h + A.total_t + 1) / 2;
    off = (off + 63) / 64 * 64;
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        MixLayerW& L = A.mix[d];
        L.tw0 = take((size_t)4 * S * S); L.tb0 = take(4 * S); L.tw3 = take((size_t)S * 4 * S); L.tb3 = take(S);
    }
    A.w_head = take((size_t)A.nout_pad * PIPS_DMIX); A.b_head = take(A.nout_pad);
    // bf16 copy / split planes of the head: addressed like the members of their sections, i.e. in ushorts from
    // (arena + total) and from (arena + total) + total_h
    const size_t hh = take(((size_t)A.nout_pad * PIPS_DMIX + 1) / 2);
    const size_t th = take((3 * (size_t)A.nout_pad * PIPS_DMIX + 1) / 2);
    A.h_head = 2 * (hh - A.total);
    A.t_head = 2 * (th - A.total) - A.total_h;
    A.total_all = off;
    return A;
}

const ArenaLayout& arena_layout(int S) {
    static ArenaLayout table[PIPS_S_MAX + 1];
    static std::once_flag once[PIPS_S_MAX + 1];
    if (S < 1 || S > PIPS_S_MAX) S = PIPS_S;          // (callers validate S; the S-independent members are the same anyway)
    std::call_once(once[S], [S]() { table[S] = build_layout(S); });
    return table[S];
}

// ------------------------------------------------------------------ repack kernels
// OIHW -> O(HW)I   (implicit-GEMM K order = kh, kw, ci)
__global__ void repack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int T) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)O * I * T) return;
    const int ci = (int)(i % I);
    const int t = (int)((i / I) % T);
    const int o = (int)(i / ((size_t)I * T));
    dst[i] = src[((size_t)o * I + ci) * T + t];
}
// [R][Cc] -> [Cc][R]
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * Cc) return;
    const int r = (int)(i % R), c = (int)(i / R);
    dst[i] = src[(size_t)r * Cc + c];
}
// [R][Kin] -> [R][Kpad] zero padded
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Kin, int Kpad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * Kpad) return;
    const int k = (int)(i % Kpad), r = (int)(i / Kpad);
    dst[i] = k < Kin ? src[(size_t)r * Kin + k] : 0.f;
}

// fp32 -> bf16 (round to nearest even, hardware v_cvt_pk_bf16_f32)
__global__ void cvt_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (__bf16)src[i];
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace pips

using namespace pips;

extern "C" {

const char* pips_last_error(void) { return g_err; }
int pips_abi_version(void) { return 3; }

size_t pips_weight_arena_bytes(void) { return pips_weight_arena_bytes_s(PIPS_S); }
int pips_delta_stride(int S) { return (S < 1 || S > PIPS_S_MAX) ? 0 : arena_layout(S).nout_pad; }
size_t pips_weight_arena_bytes_s(int S) {
    if (S < 1 || S > PIPS_S_MAX) return 0;
    return arena_layout(S).total_all * sizeof(float);
}

int pips_repack_weights(const void* const* params, int nparams, void* arena_v, void* stream) {
    return pips_repack_weights_s(params, nparams, arena_v, PIPS_S, PIPS_PACK_FP32 | PIPS_PACK_BF16 | PIPS_PACK_SPLIT, stream);
}
int pips_repack_weights_ex(const void* const* params, int nparams, void* arena_v, int sections, void* stream) {
    return pips_repack_weights_s(params, nparams, arena_v, PIPS_S, sections, stream);
}

int pips_repack_weights_s(const void* const* params, int nparams, void* arena_v, int S, int sections, void* stream) {
    PIPS_CHECK_ARG(arena_v != nullptr, "repack: null pointer");
    PIPS_CHECK_ARG(S >= 1 && S <= PIPS_S_MAX, "repack: S=%d outside 1..%d", S, PIPS_S_MAX);
    PIPS_CHECK_ARG(sections != 0 && (sections & ~(PIPS_PACK_FP32 | PIPS_PACK_BF16 | PIPS_PACK_SPLIT)) == 0, "repack: bad section mask %d", sections);
    hipStream_t st = (hipStream_t)stream;
    const ArenaLayout& A = arena_layout(S);
    float* arena = (float*)arena_v;
    int pi = 0;
  if (sections & PIPS_PACK_FP32) {
    PIPS_CHECK_ARG(params != nullptr, "repack: null pointer");
    PIPS_CHECK_ARG(nparams == PIPS_NPARAMS, "repack: expected %d tensors, got %d", PIPS_NPARAMS, nparams);
    for (int i = 0; i < nparams; ++i) PIPS_CHECK_ARG(params[i] != nullptr, "repack: tensor %d is null", i);
    auto src = [&]() { return (const float*)params[pi++]; };
    auto copy = [&](size_t dst_off, size_t n) {
        (void)hipMemcpyAsync(arena + dst_off, src(), n * sizeof(float), hipMemcpyDeviceToDevice, st);
    };
    // stem: [64][3*49] -> [147][64]
    {
        const ConvW& c = A.conv[0];
        hipLaunchKernelGGL(transpose_kernel, dim3(nblk(64 * 147)), dim3(256), 0, st, src(), arena + c.w, 64, 147);
        copy(c.b, 64);
    }
    for (int i = 1; i < 22; ++i) {
        const ConvW& c = A.conv[i];
        const size_t n = (size_t)c.cout * c.cin * c.k * c.k;
        hipLaunchKernelGGL(repack_conv_kernel, dim3(nblk(n)), dim3(256), 0, st, src(), arena + c.w, c.cout,
                           c.cin, c.k * c.k);
        copy(c.b, c.cout);
    }
    hipLaunchKernelGGL(pad_rows_kernel, dim3(nblk((size_t)PIPS_DMIX * PIPS_KIN_PAD)), dim3(256), 0, st, src(),
                       arena + A.w_in, PIPS_DMIX, PIPS_KIN, PIPS_KIN_PAD);
    copy(A.b_in, PIPS_DMIX);
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        const MixLayerW& L = A.mix[d];
        copy(L.tw0, (size_t)4 * S * S); copy(L.tb0, 4 * S); copy(L.tw3, (size_t)S * 4 * S); copy(L.tb3, S);     // nets/pips.py:102-109 over S tokens
        copy(L.ln1g, PIPS_DMIX); copy(L.ln1b, PIPS_DMIX);
        copy(L.w1, (size_t)4 * PIPS_DMIX * PIPS_DMIX); copy(L.b1, 4 * PIPS_DMIX);
        copy(L.w2, (size_t)4 * PIPS_DMIX * PIPS_DMIX); copy(L.b2, PIPS_DMIX);
        copy(L.ln2g, PIPS_DMIX); copy(L.ln2b, PIPS_DMIX);
    }
    copy(A.lnf_g, PIPS_DMIX); copy(A.lnf_b, PIPS_DMIX);
    if (A.nout_pad != A.nout) {                          // odd S: zero rows up to a multiple of 4
        (void)hipMemsetAsync(arena + A.w_head, 0, (size_t)A.nout_pad * PIPS_DMIX * sizeof(float), st);
        (void)hipMemsetAsync(arena + A.b_head, 0, (size_t)A.nout_pad * sizeof(float), st);
    }
    copy(A.w_head, (size_t)A.nout * PIPS_DMIX); copy(A.b_head, A.nout);
    copy(A.norm_g, PIPS_C); copy(A.norm_b, PIPS_C);
    hipLaunchKernelGGL(transpose_kernel, dim3(nblk(PIPS_C * PIPS_C)), dim3(256), 0, st, src(), arena + A.w_upd_t,
                       PIPS_C, PIPS_C);
    copy(A.b_upd, PIPS_C);
    copy(A.w_vis, PIPS_C); copy(A.b_vis, 1);
    if (pi != PIPS_NPARAMS) return PIPS_E_ARG;
  }
  if (sections & PIPS_PACK_BF16) {
    // bf16 copies of the channel-mix / head / conv weights (bf16-operand modes), from the fp32 section
    __bf16* hb = reinterpret_cast<__bf16*>(arena + A.total);
    auto to_h = [&](size_t src_off, size_t dst_off, size_t n) {
        hipLaunchKernelGGL(cvt_bf16_kernel, dim3(nblk(n)), dim3(256), 0, st, arena + src_off, hb + dst_off, n);
    };
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        to_h(A.mix[d].w1, A.h_w1[d], (size_t)4 * PIPS_DMIX * PIPS_DMIX);
        to_h(A.mix[d].w2, A.h_w2[d], (size_t)4 * PIPS_DMIX * PIPS_DMIX);
    }
    to_h(A.w_head, A.h_head, (size_t)A.nout_pad * PIPS_DMIX);
    to_h(A.w_in, A.h_in, (size_t)PIPS_DMIX * PIPS_KIN_PAD);

    for (int i = 1; i < 22; ++i)
        to_h(A.conv[i].w, A.h_conv[i], (size_t)A.conv[i].cout * A.conv[i].cin * A.conv[i].k * A.conv[i].k);
  }
  if (sections & PIPS_PACK_SPLIT) {
    // split-bf16 planes of the same weights (fp32-grade matrix path on the bf16 cores), from the fp32 section
    unsigned short* tb = reinterpret_cast<unsigned short*>(arena + A.total) + A.total_h;
    auto to_t = [&](size_t src_off, size_t dst_off, size_t n) {
        (void)launch_split_bf16x3(arena + src_off, n, tb + dst_off, st);
    };
    to_t(A.w_in, A.t_in, (size_t)PIPS_DMIX * PIPS_KIN_PAD);
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        to_t(A.mix[d].w1, A.t_w1[d], (size_t)4 * PIPS_DMIX * PIPS_DMIX);
        to_t(A.mix[d].w2, A.t_w2[d], (size_t)4 * PIPS_DMIX * PIPS_DMIX);
    }
    to_t(A.w_head, A.t_head, (size_t)A.nout_pad * PIPS_DMIX);
    for (int i = 1; i < 22; ++i)
        to_t(A.conv[i].w, A.t_conv[i], (size_t)A.conv[i].cout * A.conv[i].cin * A.conv[i].k * A.conv[i].k);
  }
    PIPS_CHECK_LAUNCH("pips_repack_weights");
    return PIPS_OK;
}

// ------------------------------------------------------------------ building blocks
int pips_gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N,
                  int K, int epi, const float* R, int ldr, void* stream) {
    PIPS_CHECK_ARG(A && W && C, "gemm: null pointer");
    PIPS_CHECK_ARG((epi & 0xff) <= 2 && ((epi & 0xff) != EPI_RESIDUAL || R != nullptr), "gemm: bad epilogue");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.W = W; g.bias = bias; g.C = C; g.R = R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.epi = epi;
    return launch_gemm(g, (hipStream_t)stream);
}

int pips_split_bf16x3(const float* src, size_t n, void* dst3, void* stream) {
    PIPS_CHECK_ARG(src && dst3, "split_bf16x3: null pointer");
    return launch_split_bf16x3(src, n, dst3, (hipStream_t)stream);
}

int pips_gemm_f32x3(const float* A, int lda, const void* W3, const float* bias, float* C, int ldc, int M, int N,
                    int K, int epi, const float* R, int ldr, void* stream) {
    PIPS_CHECK_ARG(A && W3 && C, "gemm_x3: null pointer");
    PIPS_CHECK_ARG((epi & 0xff) <= 2 && ((epi & 0xff) != EPI_RESIDUAL || R != nullptr), "gemm_x3: bad epilogue");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.W = (const float*)W3; g.bias = bias; g.C = C; g.R = R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.epi = epi;
    return launch_gemm_x3(g, (hipStream_t)stream);
}

// mm: 0 exact-fp32 MFMA, 1 bf16 operands (RNE), 2 split-bf16 (wgt points at the matching weight form)
// in_bf16 / out_bf16 (mm == 1 only): the NHWC maps themselves are bf16; in_norm: see GemmArgs
static int conv_nhwc(const float* in, int F, int H, int W, int Cin, const float* wgt, const float* bias, int Cout,
                     int k, int s, int p, float* out, float* stats, int* tiles, hipStream_t st, int bf16 = 0,
                     int in_bf16 = 0, int out_bf16 = 0, const float* in_norm = nullptr, int parts_cap = 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = in; g.W = wgt; g.bias = bias; g.C = out; g.stats = stats; g.in_norm = in_norm; g.stats_parts_cap = parts_cap;
    g.H = H; g.Win = W; g.Cin = Cin; g.KH = g.KW = k; g.cstride = s; g.pad = p;
    g.Ho = conv_out(H, k, s, p); g.Wo = conv_out(W, k, s, p);
    g.M = g.Ho * g.Wo; g.N = Cout; g.K = k * k * Cin; g.ldc = Cout; g.epi = EPI_BIAS;
    PIPS_CHECK_ARG(g.Ho > 0 && g.Wo > 0, "conv: empty output");
    if (bf16 == 2) return launch_conv_x3(g, F, tiles, st);
    PIPS_CHECK_ARG(bf16 == 1 || (!in_bf16 && !out_bf16 && !in_norm), "conv: bf16 maps need the bf16-operand kernels");
    return bf16 ? launch_conv_bf16(g, F, tiles, st, in_bf16, out_bf16) : launch_conv(g, F, tiles, st);   // bf16: wgt points at bf16 data
}

int pips_conv_nhwc_bf16_maps(const void* in_bf16, const float* in_norm, int F, int H, int W, int Cin, const void* wgt_bf16,
                             const float* bias, int Cout, int ksize, int cstride, int pad, void* out, int out_is_bf16,
                             float* stats, int stats_parts_cap, int* tiles_m_host, void* stream) {
    PIPS_CHECK_ARG(in_bf16 && wgt_bf16 && out, "conv_bf16_maps: null pointer");
    return conv_nhwc((const float*)in_bf16, F, H, W, Cin, (const float*)wgt_bf16, bias, Cout, ksize, cstride, pad, (float*)out,
                     stats, tiles_m_host, (hipStream_t)stream, 1, 1, out_is_bf16 ? 1 : 0, in_norm, stats_parts_cap);
}

int pips_conv_nhwc_f32x3(const float* in, int F, int H, int W, int Cin, const void* wgt3, const float* bias,
                         int Cout, int ksize, int cstride, int pad, float* out, float* stats, int* tiles_m_host,
                         void* stream) {
    PIPS_CHECK_ARG(in && wgt3 && out, "conv_x3: null pointer");
    return conv_nhwc(in, F, H, W, Cin, (const float*)wgt3, bias, Cout, ksize, cstride, pad, out, stats, tiles_m_host,
                     (hipStream_t)stream, 2);
}

int pips_conv_nhwc_bf16(const float* in, int F, int H, int W, int Cin, const void* wgt_bf16, const float* bias,
                        int Cout, int ksize, int cstride, int pad, float* out, float* stats, int* tiles_m_host,
                        void* stream) {
    PIPS_CHECK_ARG(in && wgt_bf16 && out, "conv_bf16: null pointer");
    return conv_nhwc(in, F, H, W, Cin, (const float*)wgt_bf16, bias, Cout, ksize, cstride, pad, out, stats, tiles_m_host,
                     (hipStream_t)stream, 1);
}

int pips_conv_nhwc_f32(const float* in, int F, int H, int W, int Cin, const float* wgt, const float* bias,
                       int Cout, int ksize, int cstride, int pad, float* out, float* stats, int* tiles_m_host,
                       void* stream) {
    PIPS_CHECK_ARG(in && wgt && out, "conv: null pointer");
    return conv_nhwc(in, F, H, W, Cin, wgt, bias, Cout, ksize, cstride, pad, out, stats, tiles_m_host,
                     (hipStream_t)stream);
}

// ------------------------------------------------------------------ encoder
namespace {

struct Bump {
    size_t off = 0;
    size_t take(size_t floats) { size_t o = off; off += (floats + 63) / 64 * 64; return o; }
};

struct EncPlan {
    int F, H, W, stride;
    int Hs[5], Ws[5];         // stem/layer1, layer2, layer3, layer4 resolutions; [4] = target H8,W8
    size_t raw, mid, xa, xb, ds, outs[4], cat, partial, partial2, st_a, st_b;
    size_t total;             // floats
};

// room for InstanceNorm partials per frame of a layer with H x W output pixels: the documented bound of the conv entry
// points, or one partial per wave of the 4 x 32-pixel tiles of the ping-pong 64 -> 64 kernel (conv_bf16_c64.hip)
int enc_parts_cap(int H, int W) {
    const int a = 2 * cdiv(H * W, 64) + 4, b = cdiv(W, 32) * cdiv(H, 4) * 4;
    return a > b ? a : b;
}

EncPlan plan_encoder(int F, int H, int W, int stride) {
    EncPlan P;
    P.F = F; P.H = H; P.W = W; P.stride = stride;
    P.Hs[0] = conv_out(H, 7, 2, 3); P.Ws[0] = conv_out(W, 7, 2, 3);
    for (int l = 1; l < 4; ++l) { P.Hs[l] = conv_out(P.Hs[l - 1], 3, 2, 1); P.Ws[l] = conv_out(P.Ws[l - 1], 3, 2, 1); }
    P.Hs[4] = H / stride; P.Ws[4] = W / stride;
    const int ch[4] = {64, 96, 128, 128};
    size_t big = 0;
    for (int l = 0; l < 4; ++l) big = big > (size_t)F * P.Hs[l] * P.Ws[l] * ch[l] ? big : (size_t)F * P.Hs[l] * P.Ws[l] * ch[l];
    const size_t tgt = (size_t)F * P.Hs[4] * P.Ws[4];
    if (big < tgt * 256) big = tgt * 256;
    Bump b;
    P.raw = b.take(big); P.mid = b.take(big); P.xa = b.take(big); P.xb = b.take(big); P.ds = b.take(big);
    for (int l = 0; l < 4; ++l) P.outs[l] = b.take((size_t)F * P.Hs[l] * P.Ws[l] * ch[l]);
    P.cat = b.take(tgt * 416);
    // partial statistics: float4 [F][parts][C]; parts <= 2 wave rows x (rows/64 + 1) m tiles
    size_t pmax = 0;
    for (int l = 0; l < 4; ++l) {
        size_t t = (size_t)F * enc_parts_cap(P.Hs[l], P.Ws[l]) * ch[l] * 4;
        pmax = pmax > t ? pmax : t;
    }
    {
        const size_t ts = (size_t)F * stem_tiles_m(P.Hs[0], P.Ws[0]) * 64 * 4;        // the stem's partials are float4
        pmax = pmax > ts ? pmax : ts;
    }
    size_t t2 = (size_t)F * (2 * cdiv(P.Hs[4] * P.Ws[4], 64) + 4) * 256 * 4;
    pmax = pmax > t2 ? pmax : t2;
    P.partial = b.take(pmax); P.partial2 = b.take(pmax);
    P.st_a = b.take((size_t)F * 256 * 2); P.st_b = b.take((size_t)F * 256 * 2);
    P.total = b.off;
    return P;
}

void pyramid_dims(int H, int W, int stride, int* lh, int* lw) {
    lh[0] = H / stride; lw[0] = W / stride;
    for (int l = 1; l < PIPS_LEVELS; ++l) { lh[l] = lh[l - 1] / 2; lw[l] = lw[l - 1] / 2; }
}

int check_geometry(int F, int H, int W, int stride) {
    PIPS_CHECK_ARG(F > 0 && H > 0 && W > 0 && stride >= 1, "bad geometry F=%d H=%d W=%d stride=%d", F, H, W, stride);
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    PIPS_CHECK_ARG(lh[PIPS_LEVELS - 1] >= 1 && lw[PIPS_LEVELS - 1] >= 1,
                   "input %dx%d too small for a 4-level pyramid at stride %d", H, W, stride);
    return PIPS_OK;
}

#define RUN(x) do { int rc__ = (x); if (rc__ != PIPS_OK) return rc__; } while (0)

// Matrix mode of one convolution.  mm: 0 exact-fp32 MFMA, 1 bf16 operands, 2 split-bf16.  The
// split path is used where it measured faster than the exact kernel (tools/x3_check.py): every
// layer with >= 8000 output pixels over the batch (at config 2: all but the 23x31 maps).
int layer_mm(const ConvW& c, int F, int H, int W, int mm) {
    if (mm != 2) return mm;
    const long rows = (long)F * conv_out(H, c.k, c.stride, c.pad) * conv_out(W, c.k, c.stride, c.pad);
    return rows >= 8000 ? 2 : 0;
}
// weight pointer for that mode, handed over as float* (bf16 copy / split planes live behind the fp32 arena)
const float* conv_w(const float* arena, const ArenaLayout& A, int ci, int mm) {
    if (mm == 0) return arena + A.conv[ci].w;
    const unsigned short* hb = reinterpret_cast<const unsigned short*>(arena + A.total);
    return reinterpret_cast<const float*>(mm == 1 ? hb + A.h_conv[ci] : hb + A.total_h + A.t_conv[ci]);
}

// conv -> partial stats -> mean/rstd
int conv_stats(const float* arena, const ArenaLayout& A, int ci, const float* in, int F, int H, int W, float* out,
               float* partial, float* mean_rstd, hipStream_t st, int mm) {
    const ConvW& c = A.conv[ci];
    const int lm = layer_mm(c, F, H, W, mm);
    int tiles = 0;
    RUN(conv_nhwc(in, F, H, W, c.cin, conv_w(arena, A, ci, lm), arena + c.b, c.cout, c.k, c.stride, c.pad, out, partial,
                  &tiles, st, lm));
    return launch_inorm_finalize_pivot(partial, F, tiles, c.cout, mean_rstd, st);
}

// ResidualBlock.forward, nets/pips.py:173-181
int res_block(const float* arena, const ArenaLayout& A, int& ci, bool down, const float* x, int F, int H, int W,
              float* ws, const EncPlan& P, float* out, hipStream_t st, int bf16) {
    const int i1 = ci++, i2 = ci++;
    const ConvW& c1 = A.conv[i1];
    const ConvW& c2 = A.conv[i2];
    const int Ho = conv_out(H, 3, c1.stride, 1), Wo = conv_out(W, 3, c1.stride, 1);
    float* raw = ws + P.raw; float* mid = ws + P.mid;
    RUN(conv_stats(arena, A, i1, x, F, H, W, raw, ws + P.partial, ws + P.st_a, st, bf16));
    RUN(launch_inorm_apply(raw, ws + P.st_a, nullptr, nullptr, mid, F, Ho * Wo, c1.cout, st));
    RUN(conv_stats(arena, A, i2, mid, F, Ho, Wo, raw, ws + P.partial, ws + P.st_a, st, bf16));
    if (down) {
        const int id = ci++;
        RUN(conv_stats(arena, A, id, x, F, H, W, ws + P.ds, ws + P.partial2, ws + P.st_b, st, bf16));
        RUN(launch_inorm_apply(raw, ws + P.st_a, ws + P.ds, ws + P.st_b, out, F, Ho * Wo, c2.cout, st));
    } else {
        RUN(launch_inorm_apply(raw, ws + P.st_a, x, nullptr, out, F, Ho * Wo, c2.cout, st));
    }
    return PIPS_OK;
}

// ---- the encoder with every activation in bf16 (PIPS_FLAG_BF16_ENCODER; kernels: encoder_bf16.hip, conv_bf16_c64.hip,
// gemm_bf16.hip).  Buffers are the fp32 plan's, used as bf16.
// conv (bf16 map in, optional normalise-on-load) -> bf16 raw map + partials -> {mean, rstd}
int conv_stats_h(const float* arena, const ArenaLayout& A, int ci, const void* in, const float* in_norm, int F, int H, int W,
                 void* out, float* partial, float* mean_rstd, hipStream_t st) {
    const ConvW& c = A.conv[ci];
    int tiles = 0;
    RUN(conv_nhwc((const float*)in, F, H, W, c.cin, conv_w(arena, A, ci, 1), arena + c.b, c.cout, c.k, c.stride, c.pad,
                  (float*)out, partial, &tiles, st, 1, 1, 1, in_norm,
                  enc_parts_cap(conv_out(H, c.k, c.stride, c.pad), conv_out(W, c.k, c.stride, c.pad))));
    return launch_inorm_finalize_pivot(partial, F, tiles, c.cout, mean_rstd, st);
}

// ResidualBlock.forward (nets/pips.py:173-181), materialised bf16 activations
int res_block_h(const float* arena, const ArenaLayout& A, int& ci, bool down, const void* x, int F, int H, int W, float* ws,
                const EncPlan& P, void* out, hipStream_t st) {
    const int i1 = ci++, i2 = ci++;
    const ConvW& c1 = A.conv[i1];
    const ConvW& c2 = A.conv[i2];
    const int Ho = conv_out(H, 3, c1.stride, 1), Wo = conv_out(W, 3, c1.stride, 1);
    void* raw = ws + P.raw; void* mid = ws + P.mid;
    RUN(conv_stats_h(arena, A, i1, x, nullptr, F, H, W, raw, ws + P.partial, ws + P.st_a, st));
    RUN(launch_inorm_apply_bf16(raw, ws + P.st_a, nullptr, nullptr, 0, mid, F, Ho * Wo, c1.cout, st));
    RUN(conv_stats_h(arena, A, i2, mid, nullptr, F, Ho, Wo, raw, ws + P.partial, ws + P.st_a, st));
    if (down) {
        const int id = ci++;
        RUN(conv_stats_h(arena, A, id, x, nullptr, F, H, W, ws + P.ds, ws + P.partial2, ws + P.st_b, st));
        return launch_inorm_apply_bf16(raw, ws + P.st_a, ws + P.ds, ws + P.st_b, 2, out, F, Ho * Wo, c2.cout, st);
    }
    return launch_inorm_apply_bf16(raw, ws + P.st_a, x, nullptr, 1, out, F, Ho * Wo, c2.cout, st);
}

int encoder_bf16_acts(const float* arena, const ArenaLayout& A, const void* rgbs, int rgb_u8, int F, int H, int W, int stride,
                      float* pyramid, float* ws, const EncPlan& P, hipStream_t st) {
    const int H0 = P.Hs[0], W0 = P.Ws[0];
    int tiles = 0, ci = 1;
    void* xa = ws + P.xa; void* xb = ws + P.xb; void* raw = ws + P.raw; void* mid = ws + P.mid;
    float* st_a = ws + P.st_a; float* st_b = ws + P.st_b;
    // stem: conv1 (:251); its norm1 + relu (:252-253) is applied by the consumers
    RUN(launch_stem_bf16(rgbs, rgb_u8, arena + A.conv[0].w, arena + A.conv[0].b, xa, ws + P.partial, F, H, W, H0, W0, &tiles, st));
    RUN(launch_inorm_finalize_pivot(ws + P.partial, F, tiles, 64, st_a, st));
    const void* x;
    if (conv3x3_c64_takes(H0, W0, F)) {
        // layer1 on the LDS-resident 64 -> 64 kernel: a convolution normalises its input while staging it, so relu(norm(.))
        // of the stem and of each block's first convolution never goes to HBM.  xa = raw stem map (statistics st_a).
        RUN(conv_stats_h(arena, A, ci++, xa, st_a, F, H0, W0, raw, ws + P.partial, st_b, st));          // block 1 conv1
        RUN(conv_stats_h(arena, A, ci++, raw, st_b, F, H0, W0, mid, ws + P.partial, st_b, st));         // block 1 conv2
        // relu(x + y), x = relu(norm1(stem)) recomputed from the raw stem map, y = relu(norm2(conv2)) (:176-181)
        RUN(launch_inorm_apply_bf16(mid, st_b, xa, st_a, 3, xb, F, H0 * W0, 64, st));
        RUN(conv_stats_h(arena, A, ci++, xb, nullptr, F, H0, W0, raw, ws + P.partial, st_a, st));       // block 2 conv1
        RUN(conv_stats_h(arena, A, ci++, raw, st_a, F, H0, W0, mid, ws + P.partial, st_b, st));         // block 2 conv2
        RUN(launch_inorm_apply_bf16(mid, st_b, xb, nullptr, 1, ws + P.outs[0], F, H0 * W0, 64, st));
    } else {
        RUN(launch_inorm_apply_bf16(xa, st_a, nullptr, nullptr, 0, xb, F, H0 * W0, 64, st));
        RUN(res_block_h(arena, A, ci, false, xb, F, H0, W0, ws, P, xa, st));
        RUN(res_block_h(arena, A, ci, false, xa, F, H0, W0, ws, P, ws + P.outs[0], st));
    }
    x = ws + P.outs[0];
    int Hc = H0, Wc = W0;
    for (int l = 1; l < 4; ++l) {
        RUN(res_block_h(arena, A, ci, true, x, F, Hc, Wc, ws, P, xb, st));
        Hc = P.Hs[l]; Wc = P.Ws[l];
        RUN(res_block_h(arena, A, ci, false, xb, F, Hc, Wc, ws, P, ws + P.outs[l], st));
        x = ws + P.outs[l];
    }
    const int ch[4] = {64, 96, 128, 128};
    const int H8 = P.Hs[4], W8 = P.Ws[4];
    for (int l = 0, coff = 0; l < 4; coff += ch[l], ++l)
        RUN(launch_resize_into_bf16(ws + P.outs[l], F, P.Hs[l], P.Ws[l], ch[l], ws + P.cat, H8, W8, 416, coff, st));
    const int i2 = ci++, i3 = ci++;
    const ConvW& c3 = A.conv[i3];
    RUN(conv_stats_h(arena, A, i2, ws + P.cat, nullptr, F, H8, W8, raw, ws + P.partial, st_a, st));
    RUN(launch_inorm_apply_bf16(raw, st_a, nullptr, nullptr, 0, mid, F, H8 * W8, 256, st));
    // conv3 writes the fp32 level-0 map of the correlation pyramid
    RUN(conv_nhwc((const float*)mid, F, H8, W8, 256, conv_w(arena, A, i3, 1), arena + c3.b, 128, 1, 1, 0, pyramid, nullptr, nullptr,
                  st, 1, 1, 0));
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    for (int l = 1; l < PIPS_LEVELS; ++l)
        RUN(launch_avgpool2(pyramid + pips_pyramid_offset(F, H, W, stride, l - 1), F, lh[l - 1], lw[l - 1], PIPS_C,
                            pyramid + pips_pyramid_offset(F, H, W, stride, l), st));
    // the bf16 mirror the gather of the bf16 mode reads (PIPS_FLAG_BF16_MAPS)
    RUN(pips_pyramid_mirror(pyramid, F, H, W, stride, st));
    return PIPS_OK;
}

}  // namespace

size_t pips_encoder_workspace_bytes(int F, int H, int W, int stride) {
    if (F <= 0 || H <= 0 || W <= 0 || stride < 1) return 0;
    return plan_encoder(F, H, W, stride).total * sizeof(float);
}

// fp32 levels (pips_pyramid_offset) + the bf16 mirror of all of them behind (pips_pyramid_mirror_offset; written by the
// bf16 encoder or pips_pyramid_mirror, read by the gather under PIPS_FLAG_BF16_MAPS)
size_t pips_pyramid_mirror_offset(int F, int H, int W, int stride) {
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    size_t n = 0;
    for (int l = 0; l < PIPS_LEVELS; ++l) n += ((size_t)F * lh[l] * lw[l] * PIPS_C + 63) / 64 * 64;
    return n;
}
size_t pips_pyramid_floats(int F, int H, int W, int stride) {
    const size_t n = pips_pyramid_mirror_offset(F, H, W, stride);
    // behind the mirror: a slack of a few map rows of the coarsest level (+ a pixel block).  gather_mfma_kernel fetches whole
    // 8 x 4 pixel blocks; the slots of a border block that hang over the last level's last frame must still lie inside the buffer
    // (their values are never used)
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    const size_t slack = ((size_t)4 * lw[PIPS_LEVELS - 1] + 16) * PIPS_C / 2;          // floats: (4 rows + 16 pixels) of bf16 channels
    return n + (n / 2 + 63) / 64 * 64 + (slack + 63) / 64 * 64;
}
int pips_pyramid_mirror(float* pyramid, int F, int H, int W, int stride, void* stream) {
    PIPS_CHECK_ARG(pyramid != nullptr && F > 0, "pyramid_mirror: bad argument");
    const size_t n = pips_pyramid_mirror_offset(F, H, W, stride);
    return launch_pyramid_mirror(pyramid, n, pyramid + n, (hipStream_t)stream);
}

size_t pips_pyramid_offset(int F, int H, int W, int stride, int level) {
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    size_t n = 0;
    for (int l = 0; l < level && l < PIPS_LEVELS; ++l) n += ((size_t)F * lh[l] * lw[l] * PIPS_C + 63) / 64 * 64;
    return n;
}

static int encoder_impl(const void* arena_v, const void* rgbs, int F, int H, int W, int stride, float* pyramid,
                        void* workspace, size_t workspace_bytes, void* stream, int mode);

int pips_encoder_fwd(const void* arena_v, const float* rgbs, int F, int H, int W, int stride, float* pyramid,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_impl(arena_v, rgbs, F, H, W, stride, pyramid, workspace, workspace_bytes, stream, 0);
}

int pips_encoder_fwd_bf16(const void* arena_v, const float* rgbs, int F, int H, int W, int stride, float* pyramid,
                          void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_impl(arena_v, rgbs, F, H, W, stride, pyramid, workspace, workspace_bytes, stream, 1);
}

int pips_encoder_fwd_ex(const void* arena_v, const void* rgbs, int F, int H, int W, int stride, int flags,
                        float* pyramid, void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_impl(arena_v, rgbs, F, H, W, stride, pyramid, workspace, workspace_bytes, stream,
                        ((flags & PIPS_FLAG_BF16_ENCODER) ? 1 : 0) | ((flags & PIPS_FLAG_RGB_U8) ? 2 : 0) |
                            ((flags & PIPS_FLAG_SPLIT_BF16) ? 4 : 0));
}

// mode bit0: bf16 MFMA operands in all 22 convolutions AND bf16 activation maps (the rounding points of the reference
// under torch.autocast(bfloat16)); statistics, normalisation, adds and resizes are fp32 arithmetic, the pyramid is fp32;
// mode bit1: rgbs is uint8 (B,S,3,H,W) instead of float;
// mode bit2: split-bf16 (fp32-grade) convolutions where layer_mm() picks them; wins over bit0
static int encoder_impl(const void* arena_v, const void* rgbs, int F, int H, int W, int stride, float* pyramid,
                        void* workspace, size_t workspace_bytes, void* stream, int mode) {
    const int bf16 = (mode & 4) ? 2 : (mode & 1);      // matrix mode handed to the convolutions
    PIPS_CHECK_ARG(arena_v && rgbs && pyramid && workspace, "encoder: null pointer");
    RUN(check_geometry(F, H, W, stride));
    const EncPlan P = plan_encoder(F, H, W, stride);
    if (workspace_bytes < P.total * sizeof(float)) {
        set_error("encoder: workspace %zu < %zu bytes", workspace_bytes, P.total * sizeof(float));
        return PIPS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const ArenaLayout& A = arena_layout();
    const float* arena = (const float*)arena_v;
    float* ws = (float*)workspace;
    if (bf16 == 1)
        return encoder_bf16_acts(arena, A, rgbs, (mode & 2) ? 1 : 0, F, H, W, stride, pyramid, ws, P, st);

    // stem: conv1 + norm1 + relu (nets/pips.py:251-253)
    int tiles = 0;
    RUN(launch_stem(rgbs, (mode & 2) ? 1 : 0, arena + A.conv[0].w, arena + A.conv[0].b, ws + P.raw, ws + P.partial, F, H, W, P.Hs[0],
                    P.Ws[0], &tiles, st));
    RUN(launch_inorm_finalize_pivot(ws + P.partial, F, tiles, 64, ws + P.st_a, st));
    RUN(launch_inorm_apply(ws + P.raw, ws + P.st_a, nullptr, nullptr, ws + P.xa, F, P.Hs[0] * P.Ws[0], 64, st));

    // layer1..4 (:265-268)
    int ci = 1;
    const float* x = ws + P.xa;
    int Hc = P.Hs[0], Wc = P.Ws[0];
    for (int l = 0; l < 4; ++l) {
        const bool down = l > 0;
        RUN(res_block(arena, A, ci, down, x, F, Hc, Wc, ws, P, ws + P.xb, st, bf16));
        Hc = P.Hs[l]; Wc = P.Ws[l];
        RUN(res_block(arena, A, ci, false, ws + P.xb, F, Hc, Wc, ws, P, ws + P.outs[l], st, bf16));
        x = ws + P.outs[l];
    }
    // resize a,b,c,d to (H//stride, W//stride) and concatenate (:269-273)
    const int ch[4] = {64, 96, 128, 128};
    const int H8 = P.Hs[4], W8 = P.Ws[4];
    for (int l = 0, coff = 0; l < 4; coff += ch[l], ++l)
        RUN(launch_resize_into(ws + P.outs[l], F, P.Hs[l], P.Ws[l], ch[l], ws + P.cat, H8, W8, 416, coff, st));
    // conv2 + norm2 + relu + conv3 (:273-276)
    const int i2 = ci++, i3 = ci++;
    const ConvW& c3 = A.conv[i3];
    RUN(conv_stats(arena, A, i2, ws + P.cat, F, H8, W8, ws + P.raw, ws + P.partial, ws + P.st_a, st, bf16));
    RUN(launch_inorm_apply(ws + P.raw, ws + P.st_a, nullptr, nullptr, ws + P.mid, F, H8 * W8, 256, st));
    const int m3 = layer_mm(c3, F, H8, W8, bf16);
    RUN(conv_nhwc(ws + P.mid, F, H8, W8, 256, conv_w(arena, A, i3, m3), arena + c3.b, 128, 1, 1, 0, pyramid, nullptr,
                  nullptr, st, m3));
    // CorrBlock.__init__ pyramid (:346-352)
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_dims(H, W, stride, lh, lw);
    for (int l = 1; l < PIPS_LEVELS; ++l)
        RUN(launch_avgpool2(pyramid + pips_pyramid_offset(F, H, W, stride, l - 1), F, lh[l - 1], lw[l - 1], PIPS_C,
                            pyramid + pips_pyramid_offset(F, H, W, stride, l), st));
    return PIPS_OK;
}

int pips_resize_frames(const void* src, int src_is_u8, int planes, int h, int w, float* dst, int H, int W, void* stream) {
    PIPS_CHECK_ARG(src && dst, "resize_frames: null pointer");
    PIPS_CHECK_ARG(planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "resize_frames: empty image");
    return launch_resize_frames(src, src_is_u8, planes, h, w, dst, H, W, (hipStream_t)stream);
}

// ------------------------------------------------------------------ tracker stages
int pips_point_sample(const float* level0, int B, int S, int H8, int W8, const float* xy, int N, float* out,
                      void* stream) {
    PIPS_CHECK_ARG(level0 && xy && out && B > 0 && N > 0 && S > 0, "point_sample: bad argument");
    return launch_point_sample(level0, B, S, H8, W8, xy, N, out, (hipStream_t)stream);
}

// scratch != null and a dense, un-windowed query set: LDS-tiled kernel; otherwise the direct one
static int mixer_input(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats, const float* coords,
                       const float* times, int N, const int* win_start, float* X, hipStream_t st,
                       void* scratch = nullptr, size_t scratch_bytes = 0, int force_tiled = -1, hipEvent_t* ev = nullptr,
                       int Sw = PIPS_S,         // S: frames per clip in the pyramid; Sw: window length = mixer rows per particle
                       bool bf16_maps = false) {   // the direct gather reads the bf16 mirror behind the fp32 levels
    size_t off[PIPS_LEVELS];
    int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    lh[0] = H8; lw[0] = W8;
    for (int l = 1; l < PIPS_LEVELS; ++l) { lh[l] = lh[l - 1] / 2; lw[l] = lw[l - 1] / 2; }
    size_t o = 0;
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        off[l] = o;
        o += ((size_t)B * S * lh[l] * lw[l] * PIPS_C + 63) / 64 * 64;
    }
    PIPS_CHECK_ARG(lh[PIPS_LEVELS - 1] >= 1 && lw[PIPS_LEVELS - 1] >= 1, "mixer_input: map too small");
    const bool can_tile = scratch != nullptr && win_start == nullptr && S == PIPS_S && Sw == PIPS_S &&
                          scratch_bytes >= tiled_gather_scratch_bytes(B, N, H8, W8);
    const bool tiled = force_tiled >= 0 ? (force_tiled != 0) : tiled_gather_wanted(B, N, H8, W8, bf16_maps);
    if (tiled && can_tile)      // (bf16 mode: the same work items on the matrix cores, reading the bf16 mirror behind the fp32 levels)
        return launch_mixer_input_tiled(pyramid, off, lh, lw, B, S, ffeats, coords, times, N, X, scratch, scratch_bytes, st, ev,
                                        bf16_maps ? reinterpret_cast<const unsigned short*>(pyramid + o) : nullptr);
    PIPS_CHECK_ARG(force_tiled != 1, "tiled gather needs scratch of %zu bytes, no win_start and 8 frames per clip",
                   tiled_gather_scratch_bytes(B, N, H8, W8));
    if (bf16_maps) return launch_mixer_input_bf16maps(pyramid + o, off, lh, lw, B, S, ffeats, coords, times, N, win_start, X, st, Sw);
    return launch_mixer_input(pyramid, off, lh, lw, B, S, ffeats, coords, times, N, win_start, X, st, Sw);
}

int pips_mixer_input_build(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats,
                           const float* coords, const float* times, int N, float* X, void* stream) {
    PIPS_CHECK_ARG(pyramid && ffeats && coords && times && X, "mixer_input: null pointer");
    PIPS_CHECK_ARG(S >= 1 && B > 0 && N > 0, "mixer_input: empty problem");
    return mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, nullptr, X, (hipStream_t)stream);
}

int pips_mixer_input_build_ex(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats, const float* coords,
                              const float* times, int N, const int* win_start, int flags, float* X, void* stream) {
    PIPS_CHECK_ARG(pyramid && ffeats && coords && times && X, "mixer_input: null pointer");
    PIPS_CHECK_ARG(S >= 1 && B > 0 && N > 0, "mixer_input: empty problem");
    return mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, win_start, X, (hipStream_t)stream, nullptr, 0, 0, nullptr,
                       PIPS_S, (flags & PIPS_FLAG_BF16_MAPS) != 0);
}

size_t pips_gather_scratch_bytes(int B, int N, int H8, int W8) {
    if (B <= 0 || N <= 0 || H8 <= 0 || W8 <= 0) return 0;
    return tiled_gather_scratch_bytes(B, N, H8, W8);
}

int pips_mixer_input_build_tiled(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats,
                                 const float* coords, const float* times, int N, float* X, void* scratch,
                                 size_t scratch_bytes, void* stream) {
    PIPS_CHECK_ARG(pyramid && ffeats && coords && times && X && scratch, "mixer_input_tiled: null pointer");
    PIPS_CHECK_ARG(S == PIPS_S && B > 0 && N > 0, "mixer_input_tiled: S must be %d", PIPS_S);
    return mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, nullptr, X, (hipStream_t)stream, scratch,
                       scratch_bytes, 1);
}

int pips_mixer_input_build_tiled_timed(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats,
                                       const float* coords, const float* times, int N, float* X, void* scratch,
                                       size_t scratch_bytes, void* stream, float* ms3_host) {
    PIPS_CHECK_ARG(pyramid && ffeats && coords && times && X && scratch && ms3_host, "mixer_input_tiled_timed: null pointer");
    PIPS_CHECK_ARG(S == PIPS_S && B > 0 && N > 0, "mixer_input_tiled: S must be %d", PIPS_S);
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("hipEventCreate failed"); return PIPS_E_LAUNCH; }
    int rc = mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, nullptr, X, (hipStream_t)stream, scratch,
                         scratch_bytes, 1, ev);
    if (rc == PIPS_OK && hipEventSynchronize(ev[3]) != hipSuccess) rc = PIPS_E_LAUNCH;
    if (rc == PIPS_OK)
        for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&ms3_host[i], ev[i], ev[i + 1]);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

int pips_gather_route(int B, int N, int H8, int W8, int flags) {
    if (B <= 0 || N <= 0 || H8 <= 0 || W8 <= 0) return 0;
    if (!tiled_gather_wanted(B, N, H8, W8, (flags & PIPS_FLAG_BF16_MAPS) != 0)) return 0;
    return (flags & PIPS_FLAG_BF16_MAPS) ? 2 : 1;
}

int pips_mixer_input_build_tiled_ex(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats, const float* coords,
                                    const float* times, int N, int flags, float* X, void* scratch, size_t scratch_bytes,
                                    void* stream, float* ms3_host) {
    PIPS_CHECK_ARG(pyramid && ffeats && coords && times && X && scratch, "mixer_input_tiled: null pointer");
    PIPS_CHECK_ARG(S == PIPS_S && B > 0 && N > 0, "mixer_input_tiled: S must be %d", PIPS_S);
    const bool bf16_maps = (flags & PIPS_FLAG_BF16_MAPS) != 0;
    if (ms3_host == nullptr)
        return mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, nullptr, X, (hipStream_t)stream, scratch, scratch_bytes, 1,
                           nullptr, PIPS_S, bf16_maps);
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("hipEventCreate failed"); return PIPS_E_LAUNCH; }
    int rc = mixer_input(pyramid, B, S, H8, W8, ffeats, coords, times, N, nullptr, X, (hipStream_t)stream, scratch, scratch_bytes, 1,
                         ev, PIPS_S, bf16_maps);
    if (rc == PIPS_OK && hipEventSynchronize(ev[3]) != hipSuccess) rc = PIPS_E_LAUNCH;
    if (rc == PIPS_OK)
        for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&ms3_host[i], ev[i], ev[i + 1]);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

size_t pips_mixer_workspace_bytes(int M) { return pips_mixer_workspace_bytes_s(M, PIPS_S); }
size_t pips_mixer_workspace_bytes_s(int M, int S) {
    if (M <= 0 || S < 1 || S > PIPS_S_MAX) return 0;
    Bump b;
    b.take((size_t)M * PIPS_DMIX); b.take((size_t)M * PIPS_DMIX); b.take((size_t)M * 4 * PIPS_DMIX);
    b.take((size_t)(M / S) * PIPS_DMIX);
    return b.off * sizeof(float);
}

// ev != nullptr: record ev[2g], ev[2g+1] around GEMM g (g = 0 in-proj, 1+2d up, 2+2d down, 25 head)
static int gemm_h(const float* A, int a_bf16, int lda, const unsigned short* W, const float* bias, float* C,
                  int out_bf16, int ldc, int M, int N, int K, int epi, const float* R, int ldr, hipStream_t st) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.W = reinterpret_cast<const float*>(W); g.bias = bias; g.C = C; g.R = R;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.ldr = ldr; g.epi = epi;
    return launch_gemm_bf16(g, a_bf16, out_bf16, st);
}

int pips_gemm_bf16(const void* A, int a_bf16, int lda, const void* W, const float* bias, void* C, int out_bf16, int ldc,
                   int M, int N, int K, int epi, const float* R, int ldr, void* stream) {
    PIPS_CHECK_ARG(A && W && C, "gemm_bf16: null pointer");
    return gemm_h(reinterpret_cast<const float*>(A), a_bf16, lda, reinterpret_cast<const unsigned short*>(W), bias,
                  reinterpret_cast<float*>(C), out_bf16, ldc, M, N, K, epi, R, ldr, (hipStream_t)stream);
}

int pips_gemm_bf16_route(int M, int N, int K, int epi, int a_bf16, int out_bf16) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    static const float dummy = 0.f;                     // only null-ness of bias / R is inspected
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.bias = &dummy; g.R = &dummy;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N; g.epi = epi;
    return gemm_bf16_asm_route(g, a_bf16, out_bf16);
}

int pips_device_cus(void) { return device_cus(); }

int pips_gemm_f32_route(int M, int N, int K, int epi) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    static const float dummy = 0.f;                     // only null-ness of bias / R is inspected
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.bias = &dummy; g.R = &dummy;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N; g.epi = epi;
    return gemm_f32_t4_route(g, nullptr);
}

// bf16 == 1: bf16 MFMA operands for every Linear of the mixer (weights pre-converted; the LayerNorm-2 output and the
// 2048-wide hidden activation stored as bf16, everything else fp32); the 544-wide input projection rides
// 32-element K blocks (544 = 17 x 32).
// S: the window length the arena was packed for (tokens per particle); delta rows are nout_pad(S) wide.
// matrix mode of the mixer from the PIPS_FLAG_* word: 0 exact fp32, 1 bf16 operands, 2 split-bf16, 3 bf16 operands + bf16 residual stream
static int mixer_mode(int flags) {
    if (flags & PIPS_FLAG_SPLIT_BF16) return 2;
    if (flags & PIPS_FLAG_BF16_MIXER) return (flags & PIPS_FLAG_BF16_STREAM) ? 3 : 1;
    return 0;
}

// bf16 == 3: bf16 operands AND a bf16 residual stream (PIPS_FLAG_BF16_STREAM, S = 8): x is stored as bf16 -- written by the input
// projection, read and rewritten by token mixing and the down-projection (whose fp32 sums take the bf16 residual and are rounded
// once), read by the final LayerNorm; what PreNormResidual holds under autocast (nets/pips.py:93-100).
static int mixer_impl(const void* arena_v, const float* X, int M, float* delta, void* workspace,
                      size_t workspace_bytes, void* stream, hipEvent_t* ev, int bf16 = 0, int S = PIPS_S) {
    PIPS_CHECK_ARG(arena_v && X && delta && workspace, "mixer: null pointer");
    const int xb = (bf16 == 3 && S == PIPS_S) ? 1 : 0;
    if (bf16 == 3) bf16 = 1;
    PIPS_CHECK_ARG(S >= 1 && S <= PIPS_S_MAX, "mixer: S=%d outside 1..%d", S, PIPS_S_MAX);
    PIPS_CHECK_ARG(M > 0 && M % S == 0, "mixer: M=%d must be a positive multiple of S=%d", M, S);
    if (workspace_bytes < pips_mixer_workspace_bytes_s(M, S)) {
        set_error("mixer: workspace %zu < %zu bytes", workspace_bytes, pips_mixer_workspace_bytes_s(M, S));
        return PIPS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const ArenaLayout& A = arena_layout(S);
    const int NOUT = A.nout_pad;
    const float* arena = (const float*)arena_v;
    Bump b;
    float* ws = (float*)workspace;
    float* x = ws + b.take((size_t)M * PIPS_DMIX);
    float* xn = ws + b.take((size_t)M * PIPS_DMIX);
    float* h = ws + b.take((size_t)M * 4 * PIPS_DMIX);
    float* pooled = ws + b.take((size_t)(M / S) * PIPS_DMIX);
    const int P = M / S;
    int g = 0;
#define TIMED(call)                                                        \
    do {                                                                   \
        if (ev) (void)hipEventRecord(ev[2 * g], st);                       \
        RUN(call);                                                         \
        if (ev) (void)hipEventRecord(ev[2 * g + 1], st);                   \
        ++g;                                                               \
    } while (0)

    const unsigned short* tw = reinterpret_cast<const unsigned short*>(arena + A.total) + A.total_h;
    if (bf16 == 2) {                                   // split-bf16: every GEMM of the mixer
        TIMED(pips_gemm_f32x3(X, PIPS_KIN_PAD, tw + A.t_in, arena + A.b_in, x, PIPS_DMIX, M, PIPS_DMIX, PIPS_KIN_PAD,
                              EPI_BIAS, nullptr, 0, stream));
        for (int d = 0; d < PIPS_DEPTH; ++d) {
            const MixLayerW& L = A.mix[d];
            RUN(launch_token_mix(arena, L, x, xn, P, st, 0, S));
            TIMED(pips_gemm_f32x3(xn, PIPS_DMIX, tw + A.t_w1[d], arena + L.b1, h, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX,
                                  PIPS_DMIX, EPI_GELU, nullptr, 0, stream));
            TIMED(pips_gemm_f32x3(h, 4 * PIPS_DMIX, tw + A.t_w2[d], arena + L.b2, x, PIPS_DMIX, M, PIPS_DMIX,
                                  4 * PIPS_DMIX, EPI_RESIDUAL, x, PIPS_DMIX, stream));
        }
        RUN(launch_ln_mean(x, arena + A.lnf_g, arena + A.lnf_b, pooled, P, st, S));
        TIMED(pips_gemm_f32x3(pooled, PIPS_DMIX, tw + A.t_head, arena + A.b_head, delta, NOUT, P, NOUT,
                              PIPS_DMIX, EPI_BIAS, nullptr, 0, stream));
        return PIPS_OK;
    }
    if (bf16) {
        const unsigned short* hw = reinterpret_cast<const unsigned short*>(arena + A.total);
        TIMED(gemm_h(X, 0, PIPS_KIN_PAD, hw + A.h_in, arena + A.b_in, x, xb, PIPS_DMIX, M, PIPS_DMIX, PIPS_KIN_PAD,
                     EPI_BIAS, nullptr, 0, st));
    } else {
        TIMED(pips_gemm_f32(X, PIPS_KIN_PAD, arena + A.w_in, arena + A.b_in, x, PIPS_DMIX, M, PIPS_DMIX, PIPS_KIN_PAD,
                            EPI_BIAS, nullptr, 0, stream));
    }
    for (int d = 0; d < PIPS_DEPTH; ++d) {
        const MixLayerW& L = A.mix[d];
        RUN(launch_token_mix(arena, L, x, xn, P, st, bf16 == 1, S, xb));
        if (bf16) {
            const unsigned short* hw = reinterpret_cast<const unsigned short*>(arena + A.total);
            TIMED(gemm_h(xn, 1, PIPS_DMIX, hw + A.h_w1[d], arena + L.b1, h, 1, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX,
                         PIPS_DMIX, EPI_GELU, nullptr, 0, st));
            TIMED(gemm_h(h, 1, 4 * PIPS_DMIX, hw + A.h_w2[d], arena + L.b2, x, xb, PIPS_DMIX, M, PIPS_DMIX,
                         4 * PIPS_DMIX, EPI_RESIDUAL | (xb ? EPI_RES_BF16 : 0), x, PIPS_DMIX, st));
            continue;
        }
        TIMED(pips_gemm_f32(xn, PIPS_DMIX, arena + L.w1, arena + L.b1, h, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX, PIPS_DMIX,
                            EPI_GELU, nullptr, 0, stream));
        TIMED(pips_gemm_f32(h, 4 * PIPS_DMIX, arena + L.w2, arena + L.b2, x, PIPS_DMIX, M, PIPS_DMIX, 4 * PIPS_DMIX,
                            EPI_RESIDUAL, x, PIPS_DMIX, stream));
    }
    RUN(launch_ln_mean(x, arena + A.lnf_g, arena + A.lnf_b, pooled, P, st, S, xb));
    if (bf16) {
        const unsigned short* hw = reinterpret_cast<const unsigned short*>(arena + A.total);
        TIMED(gemm_h(pooled, 0, PIPS_DMIX, hw + A.h_head, arena + A.b_head, delta, 0, NOUT, P, NOUT,
                     PIPS_DMIX, EPI_BIAS, nullptr, 0, st));
    } else {
        TIMED(pips_gemm_f32(pooled, PIPS_DMIX, arena + A.w_head, arena + A.b_head, delta, NOUT, P, NOUT,
                            PIPS_DMIX, EPI_BIAS, nullptr, 0, stream));
    }
#undef TIMED
    return PIPS_OK;
}

int pips_mixer_fwd(const void* arena_v, const float* X, int M, float* delta, void* workspace, size_t workspace_bytes,
                   void* stream) {
    return mixer_impl(arena_v, X, M, delta, workspace, workspace_bytes, stream, nullptr);
}

int pips_mixer_fwd_bf16(const void* arena_v, const float* X, int M, float* delta, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return mixer_impl(arena_v, X, M, delta, workspace, workspace_bytes, stream, nullptr, 1);
}

int pips_mixer_fwd_x3(const void* arena_v, const float* X, int M, float* delta, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return mixer_impl(arena_v, X, M, delta, workspace, workspace_bytes, stream, nullptr, 2);
}

int pips_mixer_fwd_s(const void* arena_v, const float* X, int M, int S, int flags, float* delta, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return mixer_impl(arena_v, X, M, delta, workspace, workspace_bytes, stream, nullptr, mixer_mode(flags), S);
}

int pips_mixer_fwd_timed(const void* arena_v, const float* X, int M, float* delta, void* workspace,
                         size_t workspace_bytes, void* stream, float* ms_host) {
    return pips_mixer_fwd_timed_ex(arena_v, X, M, 0, delta, workspace, workspace_bytes, stream, ms_host);
}

int pips_mixer_fwd_timed_ex(const void* arena_v, const float* X, int M, int flags, float* delta, void* workspace,
                            size_t workspace_bytes, void* stream, float* ms_host) {
    PIPS_CHECK_ARG(ms_host != nullptr, "mixer_timed: null output");
    const int mm = mixer_mode(flags);
    constexpr int NG = 2 * PIPS_DEPTH + 2;
    constexpr int NCAL = 8;                      // empty event pairs: the marker-to-marker overhead
    hipEvent_t ev[2 * NG], cal[2 * NCAL];
    for (int i = 0; i < 2 * NG; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("hipEventCreate failed"); return PIPS_E_LAUNCH; }
    for (int i = 0; i < 2 * NCAL; ++i)
        if (hipEventCreate(&cal[i]) != hipSuccess) { set_error("hipEventCreate failed"); return PIPS_E_LAUNCH; }
    hipStream_t st = (hipStream_t)stream;
    int rc = mixer_impl(arena_v, X, M, delta, workspace, workspace_bytes, stream, ev, mm);
    for (int i = 0; i < 2 * NCAL; ++i) (void)hipEventRecord(cal[i], st);
    if (rc == PIPS_OK && hipEventSynchronize(cal[2 * NCAL - 1]) != hipSuccess) rc = PIPS_E_LAUNCH;
    if (rc == PIPS_OK) {
        float up = 0.f, down = 0.f, t = 0.f, ovh = 0.f;
        for (int i = 0; i < NCAL; ++i) { (void)hipEventElapsedTime(&t, cal[2 * i], cal[2 * i + 1]); ovh += t; }
        ovh /= NCAL;
        (void)hipEventElapsedTime(&ms_host[0], ev[0], ev[1]);
        for (int d = 0; d < PIPS_DEPTH; ++d) {
            (void)hipEventElapsedTime(&t, ev[2 * (1 + 2 * d)], ev[2 * (1 + 2 * d) + 1]); up += t;
            (void)hipEventElapsedTime(&t, ev[2 * (2 + 2 * d)], ev[2 * (2 + 2 * d) + 1]); down += t;
        }
        ms_host[1] = up / PIPS_DEPTH;
        ms_host[2] = down / PIPS_DEPTH;
        (void)hipEventElapsedTime(&ms_host[3], ev[2 * (NG - 1)], ev[2 * (NG - 1) + 1]);
        ms_host[4] = ovh;
    }
    for (int i = 0; i < 2 * NG; ++i) (void)hipEventDestroy(ev[i]);
    for (int i = 0; i < 2 * NCAL; ++i) (void)hipEventDestroy(cal[i]);
    return rc;
}

// The two channel-mix GEMM shapes as launch TRAINS: the 12 layers' up-projections (then their down-projections) back to back on the
// layers' own weights between ONE event pair, reps times -> ms2_host = {up, down} milliseconds per launch.  No per-launch event and
// no overhead to subtract: start-to-start durations as the forward pays them (what a rocprofv3 kernel trace of the forward shows).
// The workspace must hold the activations of a mixer pass at this M (pips_mixer_fwd* on the same workspace first).
int pips_mixer_gemm_train(const void* arena_v, int M, int flags, void* workspace, size_t workspace_bytes, void* stream, int reps,
                          float* ms2_host) {
    PIPS_CHECK_ARG(arena_v && workspace && ms2_host && reps > 0, "mixer_gemm_train: bad argument");
    PIPS_CHECK_ARG(M > 0 && M % PIPS_S == 0, "mixer_gemm_train: M=%d must be a positive multiple of %d", M, PIPS_S);
    if (workspace_bytes < pips_mixer_workspace_bytes_s(M, PIPS_S)) {
        set_error("mixer_gemm_train: workspace %zu < %zu bytes", workspace_bytes, pips_mixer_workspace_bytes_s(M, PIPS_S));
        return PIPS_E_WORKSPACE;
    }
    // the mode pips_mixer_fwd_s derives from the same flags (incl. PIPS_FLAG_BF16_STREAM: the workspace's x is then a bf16 stream and
    // the down-projection the bf16-residual kernel the forward launches)
    const int mm0 = mixer_mode(flags);
    const int xb = mm0 == 3 ? 1 : 0;
    const int mm = mm0 == 3 ? 1 : mm0;
    hipStream_t st = (hipStream_t)stream;
    const ArenaLayout& A = arena_layout(PIPS_S);
    const float* arena = (const float*)arena_v;
    Bump b;
    float* ws = (float*)workspace;
    float* x = ws + b.take((size_t)M * PIPS_DMIX);
    float* xn = ws + b.take((size_t)M * PIPS_DMIX);
    float* h = ws + b.take((size_t)M * 4 * PIPS_DMIX);
    const unsigned short* hw = reinterpret_cast<const unsigned short*>(arena + A.total);
    const unsigned short* tw = hw + A.total_h;
    hipEvent_t ev[3];
    for (int i = 0; i < 3; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { set_error("hipEventCreate failed"); return PIPS_E_LAUNCH; }
    int rc = PIPS_OK;
    auto up = [&](int d) {
        const MixLayerW& L = A.mix[d];
        if (mm == 2) return pips_gemm_f32x3(xn, PIPS_DMIX, tw + A.t_w1[d], arena + L.b1, h, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX, PIPS_DMIX,
                                            EPI_GELU, nullptr, 0, stream);
        if (mm == 1) return gemm_h(xn, 1, PIPS_DMIX, hw + A.h_w1[d], arena + L.b1, h, 1, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX, PIPS_DMIX,
                                   EPI_GELU, nullptr, 0, st);
        return pips_gemm_f32(xn, PIPS_DMIX, arena + L.w1, arena + L.b1, h, 4 * PIPS_DMIX, M, 4 * PIPS_DMIX, PIPS_DMIX, EPI_GELU,
                             nullptr, 0, stream);
    };
    auto down = [&](int d) {                            // C = x (in place, like the mixer): R = x
        const MixLayerW& L = A.mix[d];
        if (mm == 2) return pips_gemm_f32x3(h, 4 * PIPS_DMIX, tw + A.t_w2[d], arena + L.b2, x, PIPS_DMIX, M, PIPS_DMIX, 4 * PIPS_DMIX,
                                            EPI_RESIDUAL, x, PIPS_DMIX, stream);
        if (mm == 1) return gemm_h(h, 1, 4 * PIPS_DMIX, hw + A.h_w2[d], arena + L.b2, x, xb, PIPS_DMIX, M, PIPS_DMIX, 4 * PIPS_DMIX,
                                   EPI_RESIDUAL | (xb ? EPI_RES_BF16 : 0), x, PIPS_DMIX, st);
        return pips_gemm_f32(h, 4 * PIPS_DMIX, arena + L.w2, arena + L.b2, x, PIPS_DMIX, M, PIPS_DMIX, 4 * PIPS_DMIX, EPI_RESIDUAL,
                             x, PIPS_DMIX, stream);
    };
    for (int d = 0; d < PIPS_DEPTH && rc == PIPS_OK; ++d) rc = up(d);                   // warm: clocks, caches
    (void)hipEventRecord(ev[0], st);
    for (int r = 0; r < reps && rc == PIPS_OK; ++r)
        for (int d = 0; d < PIPS_DEPTH && rc == PIPS_OK; ++d) rc = up(d);
    (void)hipEventRecord(ev[1], st);
    for (int r = 0; r < reps && rc == PIPS_OK; ++r)
        for (int d = 0; d < PIPS_DEPTH && rc == PIPS_OK; ++d) rc = down(d);
    (void)hipEventRecord(ev[2], st);
    if (rc == PIPS_OK && hipEventSynchronize(ev[2]) != hipSuccess) rc = PIPS_E_LAUNCH;
    if (rc == PIPS_OK) {
        (void)hipEventElapsedTime(&ms2_host[0], ev[0], ev[1]);
        (void)hipEventElapsedTime(&ms2_host[1], ev[1], ev[2]);
        ms2_host[0] /= (float)(reps * PIPS_DEPTH);
        ms2_host[1] /= (float)(reps * PIPS_DEPTH);
    }
    for (int i = 0; i < 3; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

int pips_state_update(const void* arena, const float* delta, float* ffeats, float* coords, const float* coords0,
                      int B, int N, float stride, float* out_traj, float* out_vis, void* stream) {
    PIPS_CHECK_ARG(arena && delta && ffeats && coords && coords0 && out_traj, "state_update: null pointer");
    PIPS_CHECK_ARG(B > 0 && N > 0, "state_update: empty");
    return launch_state_update((const float*)arena, delta, ffeats, coords, coords0, B, N, stride, out_traj, out_vis,
                               (hipStream_t)stream);
}

// ------------------------------------------------------------------ tracker driver / whole forward
namespace {
struct TrackPlan { size_t coords, coords0, ffeats, ffeat0, X, delta, mixer, total; };   // floats
TrackPlan plan_track(int B, int N, int S = PIPS_S) {
    TrackPlan P;
    Bump b;
    const int M = B * N * S;
    P.coords = b.take((size_t)M * 2);
    P.coords0 = b.take((size_t)M * 2);
    P.ffeats = b.take((size_t)M * PIPS_C);
    P.ffeat0 = b.take((size_t)B * N * PIPS_C);
    P.X = b.take((size_t)M * PIPS_KIN_PAD);
    P.delta = b.take((size_t)B * N * arena_layout(S).nout_pad);
    P.mixer = b.take(pips_mixer_workspace_bytes_s(M, S) / sizeof(float));
    P.total = b.off;
    return P;
}
struct FwdPlan { size_t pyramid, enc, track, total; };   // floats
FwdPlan plan_forward(int B, int S, int H, int W, int N, int stride) {
    FwdPlan P;
    Bump b;
    const int F = B * S;
    P.pyramid = b.take(pips_pyramid_floats(F, H, W, stride));
    P.enc = b.take(pips_encoder_workspace_bytes(F, H, W, stride) / sizeof(float));
    P.track = b.take(plan_track(B, N, S).total);
    P.total = b.off;
    return P;
}
}  // namespace

size_t pips_track_workspace_bytes(int B, int N) { return pips_track_workspace_bytes_s(B, N, PIPS_S); }
size_t pips_track_workspace_bytes_s(int B, int N, int S) {
    if (B <= 0 || N <= 0 || S < 1 || S > PIPS_S_MAX) return 0;
    return plan_track(B, N, S).total * sizeof(float);
}

// level table of a pyramid with frames*H8*W8 level-0 pixels (pips_pyramid_offset's packing)
static void pyramid_table(int frames, int H8, int W8, size_t* off, int* lh, int* lw) {
    lh[0] = H8; lw[0] = W8;
    for (int l = 1; l < PIPS_LEVELS; ++l) { lh[l] = lh[l - 1] / 2; lw[l] = lw[l - 1] / 2; }
    size_t o = 0;
    for (int l = 0; l < PIPS_LEVELS; ++l) {
        off[l] = o;
        o += ((size_t)frames * lh[l] * lw[l] * PIPS_C + 63) / 64 * 64;
    }
}

size_t pips_score_map_workspace_bytes(int B, int S, int H8, int W8) {
    if (B <= 0 || S <= 0 || H8 <= 0 || W8 <= 0) return 0;
    return (size_t)B * S * H8 * W8 * PIPS_C * sizeof(float);
}

int pips_score_map_prepare(const float* pyramid, int B, int S, int H8, int W8, float* U, void* stream) {
    PIPS_CHECK_ARG(pyramid && U && B > 0 && S > 0 && H8 >= 8 && W8 >= 8, "score_map_prepare: bad argument");
    size_t off[PIPS_LEVELS]; int lh[PIPS_LEVELS], lw[PIPS_LEVELS];
    pyramid_table(B * S, H8, W8, off, lh, lw);
    return launch_score_upsum(pyramid, off, lh, lw, B * S, U, (hipStream_t)stream);
}

int pips_score_map_terms(const float* U, int B, int S, int H8, int W8, const float* ffeats, int N, const float* tgt,
                         float* out, void* stream) {
    PIPS_CHECK_ARG(U && ffeats && tgt && out && B > 0 && S > 0 && N > 0, "score_map_terms: bad argument");
    return launch_score_terms(U, B, S, H8, W8, ffeats, N, tgt, out, (hipStream_t)stream);
}

}  // extern "C"

// S = window length (tokens per particle) the arena was packed for; S == PIPS_S runs the specialised kernels
static int track_impl(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
                      const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
                      int stride, int iters, int flags, int S, void* workspace, size_t workspace_bytes, float* out_trajs,
                      float* out_vis, float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws,
                      size_t ce_ws_bytes, void* stream);

extern "C" {

int pips_track(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
               const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
               int stride, int iters, int flags, void* workspace, size_t workspace_bytes, float* out_trajs,
               float* out_vis, float* out_ffeat0, void* stream) {
    return track_impl(arena, pyramid, B, T, H8, W8, xys, coords_init, feat_init, win_start, times, N, stride, iters,
                      flags, PIPS_S, workspace, workspace_bytes, out_trajs, out_vis, out_ffeat0, nullptr, nullptr, nullptr, 0,
                      stream);
}

int pips_track_s(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
                 const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
                 int stride, int iters, int flags, int S, void* workspace, size_t workspace_bytes, float* out_trajs,
                 float* out_vis, float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws,
                 size_t ce_ws_bytes, void* stream) {
    return track_impl(arena, pyramid, B, T, H8, W8, xys, coords_init, feat_init, win_start, times, N, stride, iters,
                      flags, S, workspace, workspace_bytes, out_trajs, out_vis, out_ffeat0, ce_tgt, ce_terms, ce_ws, ce_ws_bytes,
                      stream);
}

int pips_track_ce(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
                  const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
                  int stride, int iters, int flags, void* workspace, size_t workspace_bytes, float* out_trajs,
                  float* out_vis, float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws,
                  size_t ce_ws_bytes, void* stream) {
    return track_impl(arena, pyramid, B, T, H8, W8, xys, coords_init, feat_init, win_start, times, N, stride, iters,
                      flags, PIPS_S, workspace, workspace_bytes, out_trajs, out_vis, out_ffeat0, ce_tgt, ce_terms, ce_ws,
                      ce_ws_bytes, stream);
}

}  // extern "C"

static int track_impl(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
                      const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
                      int stride, int iters, int flags, int S, void* workspace, size_t workspace_bytes, float* out_trajs,
                      float* out_vis, float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws,
                      size_t ce_ws_bytes, void* stream) {
    PIPS_CHECK_ARG(arena && pyramid && xys && times && workspace && out_trajs && out_vis, "track: null pointer");
    PIPS_CHECK_ARG(B > 0 && N > 0 && T >= 1 && iters >= 0 && stride >= 1, "track: need B,N,T,stride >= 1 and iters >= 0");
    PIPS_CHECK_ARG(S >= 1 && S <= PIPS_S_MAX, "track: window length S=%d outside 1..%d", S, PIPS_S_MAX);
    PIPS_CHECK_ARG(H8 >= 8 && W8 >= 8, "track: map %dx%d too small for a 4-level pyramid", H8, W8);
    const TrackPlan P = plan_track(B, N, S);
    if (workspace_bytes < P.total * sizeof(float)) {
        set_error("track: workspace %zu < %zu bytes", workspace_bytes, P.total * sizeof(float));
        return PIPS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int M = B * N * S;
    float* coords = ws + P.coords; float* coords0 = ws + P.coords0; float* ffeats = ws + P.ffeats;
    float* ffeat0 = out_ffeat0 != nullptr ? out_ffeat0 : ws + P.ffeat0;
    const size_t traj_sz = (size_t)B * S * N * 2;
    RUN(launch_init_coords(xys, coords_init, B, N, (float)stride, coords, coords0, out_trajs, st, S));
    if (feat_init != nullptr) {
        if (feat_init != ffeat0)
            (void)hipMemcpyAsync(ffeat0, feat_init, (size_t)B * N * PIPS_C * sizeof(float), hipMemcpyDeviceToDevice, st);
    } else {
        RUN(launch_point_sample_strided(pyramid, B, T, H8, W8, coords, S * 2, N, win_start, ffeat0, st));   // :463
    }
    RUN(launch_init_ffeats(ffeat0, B * N, ffeats, st, S));                                                    // :466
    if (ce_tgt != nullptr) {            // score-map loss terms of every iteration (:501-511, 58-92): evaluation only
        PIPS_CHECK_ARG(ce_terms && ce_ws && win_start == nullptr && T == S,
                       "track: score-map terms need their output and workspace, S frames per clip and no windows");
        if (ce_ws_bytes < pips_score_map_workspace_bytes(B, T, H8, W8)) {
            set_error("track: score-map workspace %zu < %zu bytes", ce_ws_bytes, pips_score_map_workspace_bytes(B, T, H8, W8));
            return PIPS_E_WORKSPACE;
        }
        RUN(pips_score_map_prepare(pyramid, B, T, H8, W8, (float*)ce_ws, stream));
    }
    if (iters == 0)       // the loop body never runs: vis_e comes from the initial features (:559)
        RUN(launch_vis_head((const float*)arena, ffeats, B, N, out_vis, st, S));
    for (int it = 0; it < iters; ++it) {                                                                     // :499
        if (ce_tgt != nullptr)           // fcorr_fn.corr(ffeats) of this iteration (:501), before the update
            RUN(launch_score_terms((const float*)ce_ws, B, S, H8, W8, ffeats, N, ce_tgt, ce_terms + (size_t)it * M * 2, st));
        // the mixer workspace is idle while the gather runs: it doubles as the binning scratch
        RUN(mixer_input(pyramid, B, T, H8, W8, ffeats, coords, times, N, win_start, ws + P.X, st, ws + P.mixer,
                        pips_mixer_workspace_bytes_s(M, S), -1, nullptr, S, (flags & PIPS_FLAG_BF16_MAPS) != 0));
        RUN(mixer_impl(arena, ws + P.X, M, ws + P.delta, ws + P.mixer, pips_mixer_workspace_bytes_s(M, S), stream, nullptr,
                       mixer_mode(flags), S));
        RUN(launch_state_update((const float*)arena, ws + P.delta, ffeats, coords, coords0, B, N, (float)stride,
                                out_trajs + (size_t)(it + 1) * traj_sz, it + 1 == iters ? out_vis : nullptr, st, S));
    }
    PIPS_CHECK_LAUNCH("pips_track");
    return PIPS_OK;
}

extern "C" {

size_t pips_workspace_bytes(int B, int S, int H, int W, int N, int stride) {
    if (B <= 0 || S < 1 || S > PIPS_S_MAX || H <= 0 || W <= 0 || N <= 0 || stride < 1) return 0;
    return plan_forward(B, S, H, W, N, stride).total * sizeof(float);
}

int pips_forward(const void* arena, const float* rgbs, const float* xys, const float* coords_init,
                 const float* feat_init, const float* times, int B, int S, int H, int W, int N, int stride,
                 int iters, int flags, void* workspace, size_t workspace_bytes, float* out_trajs, float* out_vis,
                 float* out_ffeat0, void* stream) {
    return pips_forward_ce(arena, rgbs, xys, coords_init, feat_init, times, B, S, H, W, N, stride, iters, flags, workspace,
                           workspace_bytes, out_trajs, out_vis, out_ffeat0, nullptr, nullptr, nullptr, 0, stream);
}

int pips_forward_ce(const void* arena, const float* rgbs, const float* xys, const float* coords_init,
                    const float* feat_init, const float* times, int B, int S, int H, int W, int N, int stride,
                    int iters, int flags, void* workspace, size_t workspace_bytes, float* out_trajs, float* out_vis,
                    float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws, size_t ce_ws_bytes,
                    void* stream) {
    PIPS_CHECK_ARG(arena && xys && times && workspace && out_trajs && out_vis, "forward: null pointer");
    PIPS_CHECK_ARG((flags & PIPS_FLAG_REUSE_MAPS) || rgbs != nullptr, "forward: rgbs is null");
    PIPS_CHECK_ARG(S >= 1 && S <= PIPS_S_MAX, "forward: S=%d outside 1..%d (the arena must be packed for the same S, nets/pips.py:295-301)", S, PIPS_S_MAX);
    PIPS_CHECK_ARG(B > 0 && N > 0 && iters >= 0, "forward: need B,N >= 1 and iters >= 0");
    RUN(check_geometry(B * S, H, W, stride));
    const FwdPlan P = plan_forward(B, S, H, W, N, stride);
    if (workspace_bytes < P.total * sizeof(float)) {
        set_error("forward: workspace %zu < %zu bytes", workspace_bytes, P.total * sizeof(float));
        return PIPS_E_WORKSPACE;
    }
    float* ws = (float*)workspace;
    float* pyramid = ws + P.pyramid;
    if (!(flags & PIPS_FLAG_REUSE_MAPS))
        RUN(encoder_impl(arena, rgbs, B * S, H, W, stride, pyramid, ws + P.enc,
                         pips_encoder_workspace_bytes(B * S, H, W, stride), stream,
                         ((flags & PIPS_FLAG_BF16_ENCODER) ? 1 : 0) | ((flags & PIPS_FLAG_RGB_U8) ? 2 : 0) |
                             ((flags & PIPS_FLAG_SPLIT_BF16) ? 4 : 0)));
    // both bf16 modes on: the bf16 encoder wrote the mirror with the maps and the gather reads it (with REUSE_MAPS the flags
    // describe the call that produced the maps, so the same forward gives the same result with and without the encoder pass)
    if (!(flags & PIPS_FLAG_SPLIT_BF16) &&
        (flags & (PIPS_FLAG_BF16_ENCODER | PIPS_FLAG_BF16_MIXER)) == (PIPS_FLAG_BF16_ENCODER | PIPS_FLAG_BF16_MIXER) &&
        PIPS_TUNE("PIPS_BF16_MAPS", 1))
        flags |= PIPS_FLAG_BF16_MAPS;
    return track_impl(arena, pyramid, B, S, H / stride, W / stride, xys, coords_init, feat_init, nullptr, times, N,
                      stride, iters, flags, S, ws + P.track, plan_track(B, N, S).total * sizeof(float), out_trajs, out_vis,
                      out_ffeat0, ce_tgt, ce_terms, ce_ws, ce_ws_bytes, stream);
}

}  // extern "C"
