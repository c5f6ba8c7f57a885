// Internal declarations shared by the HIP translation units of libpips_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pips_hip.h"

namespace pips {

void set_error(const char* fmt, ...);

#define PIPS_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) { ::pips::set_error(__VA_ARGS__); return PIPS_E_ARG; } \
    } while (0)

#define PIPS_CHECK_LAUNCH(what)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ::pips::set_error("%s: %s", what, hipGetErrorString(e__));              \
            return PIPS_E_LAUNCH;                                                   \
        }                                                                           \
    } while (0)

// ---------------------------------------------------------------- exact-form GELU
// nn.GELU() default = 0.5*x*(1+erf(x/sqrt(2))) (nets/pips.py:105,419), evaluated branch-free as
//     gelu(x) = max(x, 0) - 0.5 * t * erfc(t / sqrt(2)),   t = min(|x|, 4*sqrt(2)),
//     erfc(t / sqrt(2)) = exp2(t * A8(t))
// (erf(x) = sign(x)(1 - erfc(|x|)), so 0.5 x (1 + erf) = x - 0.5 x erfc for x > 0 and 0.5 x erfc(|.|) for x < 0; beyond
// the clamp erfc < 1.6e-8).  A8 = -log2(e)/sqrt(2) * Q8(t/sqrt(2)) with Q8 a weighted minimax fit of -ln(erfc(u))/u
// on [0, 4].  Max abs error of the GELU 4.8e-7 over [-8, 8] (numpy fp32 emulation against double, 6 M points),
// as much as the fp32 rounding of the textbook formula itself carries.  Per PAIR of values: 2 min, 2 max, 9 packed
// FMA/mul for the polynomial, 2 v_exp, 2 packed ops for the result -- it runs on 64 values per lane in the GEMM
// epilogues and 64 per thread in the token-mix kernel.
#define PIPS_GELU_TMAX 5.65685425f
#define PIPS_GELU_A8(q, t, C)                                             \
    q = C(3.208326405e-07f);                                              \
    q = q * t + C(-6.917509381e-06f);                                     \
    q = q * t + C(6.041429151e-05f);                                      \
    q = q * t + C(-2.428356966e-04f);                                     \
    q = q * t + C(-5.105399032e-05f);                                     \
    q = q * t + C(6.989960559e-03f);                                      \
    q = q * t + C(-5.246259645e-02f);                                     \
    q = q * t + C(-4.592153430e-01f);                                     \
    q = q * t + C(-1.151104689e+00f);
__device__ __forceinline__ float gelu_exact(float x) {
    const float t = fminf(fabsf(x), PIPS_GELU_TMAX);
    float q;
#define PIPS_C1(v) v
    PIPS_GELU_A8(q, t, PIPS_C1)
#undef PIPS_C1
    return fmaf(t * __builtin_amdgcn_exp2f(q * t), -0.5f, fmaxf(x, 0.0f));
}

typedef float f2 __attribute__((ext_vector_type(2)));
// the same on two values at once: the polynomial runs as packed FMAs (v_pk_fma_f32)
__device__ __forceinline__ f2 gelu_exact2(f2 x) {
    const f2 t = __builtin_elementwise_min(__builtin_elementwise_abs(x), (f2){PIPS_GELU_TMAX, PIPS_GELU_TMAX});
    f2 q;
#define PIPS_C2(v) ((f2){v, v})
    PIPS_GELU_A8(q, t, PIPS_C2)
#undef PIPS_C2
    const f2 a = q * t;
    const f2 w = t * (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    return w * -0.5f + __builtin_elementwise_max(x, (f2){0.0f, 0.0f});
}

// GELU whose result is rounded to bf16 right away (hidden activations of the bf16 mixer): the same form with a degree-5 exponent
// polynomial -- max |error| 5.1e-6, 3.3e-5 relative where |gelu| > 1e-3, two orders below the bf16 rounding (2^-9); three
// packed FMAs per pair fewer.  The coefficients of tools/gen_gemm_bf16_t4up.py (the up-projection's epilogue).
// min(|x|, c) and max(x, 0) as ONE instruction each: v_med3_f32 (the abs is a source modifier; med3(|x|, c, -1) = min(|x|, c) for c > 0,
// med3(x, 0, +inf) = max(x, 0)).  Written with the min / max / abs builtins hipcc emits v_max |x|,|x| for the abs and a v_max x,x in
// front of every IEEE min / max (it quiets a signalling NaN): 768 of the 3 600 vector instructions of a token_mix_mfma_kernel wave,
// which is bound by exactly those (tools/token_trace_bf16.py: the 16 channel slots are 9.3 of its 12.4 us)
__device__ __forceinline__ float min_abs(float x, float c) { return __builtin_amdgcn_fmed3f(__builtin_fabsf(x), c, -1.0f); }
__device__ __forceinline__ float relu_med3(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_huge_valf()); }
__device__ __forceinline__ f2 gelu_bf16out2(f2 x) {
    const f2 t = {min_abs(x.x, PIPS_GELU_TMAX), min_abs(x.y, PIPS_GELU_TMAX)};
#define PIPS_C2(v) ((f2){v, v})
    f2 q = PIPS_C2(2.554670494e-05f);
    q = q * t + PIPS_C2(-6.529359078e-04f);
    q = q * t + PIPS_C2(7.452824686e-03f);
    q = q * t + PIPS_C2(-5.192063601e-02f);
    q = q * t + PIPS_C2(-4.602978599e-01f);
    q = q * t + PIPS_C2(-1.150685204e+00f);
#undef PIPS_C2
    const f2 a = q * t;
    const f2 w = t * (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    return w * -0.5f + (f2){relu_med3(x.x), relu_med3(x.y)};
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic ONLY.  hipcc's __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it
// also drains the wave's global loads and stores, so a prefetch issued before a barrier is waited for AT the barrier and the
// full memory latency is exposed once per barrier (measured in conv3x3_c64_pp_kernel: 8k clocks per half step whatever the
// half step did).  Register dependences on outstanding loads are still tracked by the compiler's own s_waitcnt insertion.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- bf16 activations (bf16 encoder mode)
// two floats -> one dword of two bf16 (hardware round-to-nearest-even, v_cvt_pk_bf16_f32); lo in bits 0-15
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t_ __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    const bf16x2_t_ b = __builtin_convertvector(v, bf16x2_t_);
    return *reinterpret_cast<const unsigned*>(&b);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// the value of lane ^ 1 (DPP quad_perm [1,0,3,2]: one VALU instruction, no LDS)
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}
// One 32-pixel x 32-channel MFMA accumulator tile in C orientation -- lane: channel c0 + l31, register r: pixel
// (r&3) + 8*(r>>2) + 4*half -- stored as bf16 into a channel-last map.  Neighbouring lanes (channels n, n+1) trade every
// other pixel, so each lane writes packed channel PAIRS: one dword per store, 64-byte runs per pixel and half-wave.
// px_ptr(px) returns the pixel's row (bf16 units, at the tile's first channel) or nullptr when the pixel is outside.
template <typename PxPtr>
__device__ __forceinline__ void store_c_tile_bf16(const float (&v)[16], int l31, int half, PxPtr px_ptr) {
    const bool odd = l31 & 1;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r0 = 2 * rr, r1 = r0 + 1;                                  // consecutive pixels
        float a = v[r0], b = v[r1];
        // opaque copies: hipcc otherwise reads "odd ? v[r0] : v[r1]" as a dynamically indexed element of the accumulator
        // vector and lowers it to a 16-step compare/select chain per value (measured: +1100 VALU instructions per tile)
        asm("" : "+v"(a), "+v"(b));
        const float recv = lane_xor1(odd ? a : b);
        const unsigned d = pack2_bf16(odd ? recv : a, odd ? b : recv);
        unsigned short* row = px_ptr((r0 & 3) + 8 * (r0 >> 2) + 4 * half + (odd ? 1 : 0));
        if (row != nullptr) *reinterpret_cast<unsigned*>(row + (l31 & ~1)) = d;
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// InstanceNorm partials of a convolution tile, one per WAVE ROW (m tile, wm) and channel, about a PIVOT -- the value of
// the wave's first output row in that channel: {sum(x-p), sum((x-p)^2), p, n}.  E[x^2] - mean^2 of raw fp32 sums loses
// digits where |mean| >> std (flat or letterboxed frames, large biases); sums about a value of the data do not, and
// the partials are combined in fp64 (inorm_finalize_pivot_kernel).  csum / csq: the lane's sums over its rows of column
// col (lanes l and l+32 hold the same column); no LDS, no barrier, bitwise deterministic.
__device__ __forceinline__ void store_conv_partial(float* stats, int frame, int parts, int part, int N, int col, int half,
                                                   float csum, float csq, float pivot, int nvalid) {
    const float s = csum + __shfl_xor(csum, 32);
    const float q = csq + __shfl_xor(csq, 32);
    if (half == 0 && col < N)
        reinterpret_cast<float4*>(stats)[((size_t)frame * parts + part) * N + col] = make_float4(s, q, pivot, (float)nvalid);
}

// Dynamic-LDS limit of one kernel instantiation, raised once per (instantiation, device).  The only state
// the launchers keep: an idempotent attribute, tracked per device, safe from several host threads.
inline int ensure_dynamic_lds(std::atomic<unsigned long long>& done, const void* kern, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return PIPS_E_LAUNCH; }
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return PIPS_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("cannot raise the dynamic LDS limit of a kernel to %zu bytes on device %d", bytes, dev);
        return PIPS_E_LAUNCH;
    }
    done.fetch_or(bit, std::memory_order_release);
    return PIPS_OK;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Tuning hooks.  The product library is built WITHOUT -DPIPS_TUNING: PIPS_TUNE() is then its compile-time default -- no
// getenv, no mutable state reachable from the drop-in.  `python -m pips_amd._build --tuning` builds libpips_hip_tune.so
// (load it through PIPS_LIB_PATH) in which every hook is read from the environment ONCE per call site, in a C++11
// function-local static initialiser (thread-safe).  The A/B numbers quoted in DESIGN.md come from that build.
#ifdef PIPS_TUNING
int tune_env(const char* name, int dflt);
#define PIPS_TUNE(name, dflt) ([]() -> int { static const int v__ = ::pips::tune_env(name, dflt); return v__; }())
#else
#define PIPS_TUNE(name, dflt) (dflt)
#endif

// XCD-aware tile order over the WHOLE grid (x = m tiles fastest, then y = n tiles, then z = frames).  Workgroups are
// dispatched round-robin over the 8 XCDs in linear id order and every XCD has its own L2: with the identity order the tiles
// that run on one XCD at a time are eight apart -- every XCD ends up pulling (nearly) the whole operand / the whole map of
// every frame through its L2.  Re-dealt, XCD k owns ONE contiguous run of the linear tile sequence: tiles that are neighbours
// in a frame (shared conv halo rows, shared W rows) run side by side on one L2, and a frame is touched by one or two XCDs.
// Bijective for any grid; returns (bx, by, bz).
struct Tile3 { int x, y, z; };
__device__ __forceinline__ Tile3 xcd_tile_order(bool on) {
    Tile3 t = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    if (on) {
        const int gx = gridDim.x, gxy = gx * gridDim.y, T = gxy * gridDim.z;
        const int id = t.x + gx * t.y + gxy * t.z;
        const int xcd = id & 7, local = id >> 3, q = T >> 3, r = T & 7;
        const int nid = xcd * q + (xcd < r ? xcd : r) + local;
        t.z = nid / gxy;
        const int rem = nid - t.z * gxy;
        t.y = rem / gx;
        t.x = rem - t.y * gx;
    }
    return t;
}

// Compute units of the current device, queried once per device (persistent kernels size their grid by it).
int device_cus();

// ---------------------------------------------------------------- GEMM / conv core
enum Epilogue { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESIDUAL = 2 };
// flag in GemmArgs::epi beside EPI_RESIDUAL (bf16-operand GEMMs with a bf16 output): the residual R is bf16 as well -- the mixer's
// residual stream under autocast (nets/pips.py:93-100 adds two bf16 tensors)
constexpr int EPI_RES_BF16 = PIPS_EPI_RES_BF16;      // public: include/pips_hip.h

struct GemmArgs {
    const float* A;      // plain: [M][lda]; conv: NHWC input of frame 0
    const float* W;      // [N][K], K contiguous
    const float* bias;   // [N] or null
    float* C;            // [M][ldc] (conv: output of frame 0, ldc = N)
    const float* R;      // residual [M][ldr] (EPI_RESIDUAL)
    float* stats;        // optional pivoted partials {sum(x-p), sum((x-p)^2), p, n} (float4): [frame][parts][N], parts = m tiles x wave rows
    int M, N, K;
    int lda, ldc, ldr;
    int epi;
    int swz;             // XCD-aware tile order (set by the launcher)
    // implicit-GEMM geometry (conv only); M = Ho*Wo rows per frame, gridDim.z = frames
    int H, Win, Cin, Ho, Wo, KH, KW, cstride, pad;
    // bf16-activation convolutions: {mean, rstd} [frame][Cin] of the PRODUCING layer -- relu((x - mean) * rstd) is applied
    // to the bf16 input map while it is staged (conv_bf16_c64.hip only); null = the map is read as it is
    const float* in_norm;
    int stats_parts_cap;     // room in stats in partials per frame; 0 = the documented 2*ceil(Ho*Wo/64)+4
#ifdef PIPS_GEMM_TRACE
    unsigned long long* trace;   // tools/gemm_trace.py: per-block phase timestamps
#endif
};

// Epilogue of a plain-GEMM tile that lies wholly inside C (the common case): straight-line code,
// every bias / residual load issued before the first use, GELU on packed pairs (shared by the
// fp32 and bf16-operand kernels; OUT_BF16 stores the tile as bf16).  The lane holds
// C^T: acc[i][j][4g..4g+3] = C[row = i*32 + l31][col = j*32 + 8g + 4*half + 0..3].
template <int E, bool OUT_BF16, int TM, int TN, bool R_BF16 = false>
__device__ __forceinline__ void epilogue_full_tile(const f32x16 (&acc)[TM][TN], const float* __restrict__ bias,
                                                   const float* __restrict__ R, int ldr, void* __restrict__ Cv,
                                                   int ldc, int row0, int col0) {
    float4 b4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b4[j][g] = bias != nullptr ? *reinterpret_cast<const float4*>(bias + col0 + j * 32 + 8 * g)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const size_t row = (size_t)(row0 + i * 32);
        float4 r4[TN][4];
        if (E == EPI_RESIDUAL) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (R_BF16) {           // four bf16 = 8 bytes, widened to fp32
                        const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(R) + row * ldr + col0 + j * 32 + 8 * g);
                        r4[j][g] = make_float4(__uint_as_float(rb.x << 16), __uint_as_float(rb.x & 0xffff0000u),
                                               __uint_as_float(rb.y << 16), __uint_as_float(rb.y & 0xffff0000u));
                    } else {
                        r4[j][g] = *reinterpret_cast<const float4*>(R + row * ldr + col0 + j * 32 + 8 * g);
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f2 lo = (f2){acc[i][j][4 * g] + b4[j][g].x, acc[i][j][4 * g + 1] + b4[j][g].y};
                f2 hi = (f2){acc[i][j][4 * g + 2] + b4[j][g].z, acc[i][j][4 * g + 3] + b4[j][g].w};
                if (E == EPI_GELU) {
                    lo = gelu_exact2(lo);
                    hi = gelu_exact2(hi);
                } else if (E == EPI_RESIDUAL) {
                    lo += (f2){r4[j][g].x, r4[j][g].y};
                    hi += (f2){r4[j][g].z, r4[j][g].w};
                }
                const size_t o = row * ldc + col0 + j * 32 + 8 * g;
                if (OUT_BF16) {
                    typedef float f32x4_ __attribute__((ext_vector_type(4)));
                    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
                    const f32x4_ t = {lo.x, lo.y, hi.x, hi.y};
                    bf16x4_ ob = __builtin_convertvector(t, bf16x4_);              // hardware RNE
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(Cv) + o) = *reinterpret_cast<uint2*>(&ob);
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + o) = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
    }
}

int launch_gemm(const GemmArgs& a, hipStream_t st);          // plain GEMM, picks a tile
// the mixer's exact-fp32 Linears (bias + GELU / bias + residual) on four waves with a generated static schedule (gemm_f32_t4.hip):
// 0 = not taken (igemm_f32_kernel), 1 = 128 x 128 tiles, 2 = 64 x 64 tiles with the K range split over the waves
int gemm_f32_t4_route(const GemmArgs& a, int* tpb);
int launch_gemm_f32_t4(const GemmArgs& a, int route, int tpb, hipStream_t st);
int launch_conv(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st);
// the big 3x3 / stride 1 layers of the fp32 encoder on four waves with a generated static schedule (conv_f32_t4.hip): the configuration
// that takes a layer (-1: none); *parts_out = InstanceNorm partials per frame
int conv_f32_t4_config(const GemmArgs& a, int frames);
int launch_conv_f32_t4(const GemmArgs& a, int cfg, int frames, int* parts_out, hipStream_t st);
// bf16-operand GEMM (gemm_bf16.hip): A fp32 or bf16, W bf16, C fp32 or bf16; pointers passed as float*
int launch_gemm_bf16(const GemmArgs& a, int a_bf16, int out_bf16, hipStream_t st);
// in_bf16 / out_bf16: the NHWC maps are bf16 instead of fp32 (the bf16 encoder keeps every activation in bf16)
int launch_conv_bf16(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st, int in_bf16 = 0, int out_bf16 = 0);
// 3x3 / 64 -> 64 channels with the weights of all taps and the tile's halo patch resident in LDS (conv_bf16_c64.hip); 1 = not taken
int launch_conv3x3_c64_bf16(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st, int in_bf16 = 0, int out_bf16 = 0);
bool conv3x3_c64_takes(int H, int W, int frames);      // whether that kernel takes a 64 -> 64 3x3 layer of this size
// 96 -> 96, 3x3, stride 1 on bf16 maps at >= 4 tiles of 256 pixels per compute unit: four-wave implicit GEMM + a statistics pass (conv_bf16_t4c.hip)
bool conv_c96_t4_takes(const GemmArgs& a, int frames, int in_bf16, int out_bf16);
int launch_conv_c96_t4(const GemmArgs& a, int frames, int* parts_out, hipStream_t st);
int gemm_bf16_asm_route(const GemmArgs& a, int a_bf16, int out_bf16);   // 0 register-staged, 3 / 4 the kernels of gemm_bf16_t4.hip
// the large-M bf16 down-projection (+ bias + fp32 residual) on 128x256 tiles, four waves, 16x16x32 MFMAs (gemm_bf16_t4.hip)
bool gemm_bf16_t4_takes(const GemmArgs& a, int a_bf16, int out_bf16);
int launch_gemm_bf16_t4(const GemmArgs& a, hipStream_t st);
bool gemm_bf16_t4up_takes(const GemmArgs& a, int a_bf16, int out_bf16, int* tpb);      // the up-projection form (GELU, K = 512)
int launch_gemm_bf16_t4up(const GemmArgs& a, int tpb, hipStream_t st);
// split-bf16 (bf16x3) fp32-grade GEMM / conv (gemm_x3.hip): A fp32, W = three bf16 planes [3][N][K]
int launch_split_bf16x3(const float* src, size_t n, void* dst, hipStream_t st);
int launch_gemm_x3(const GemmArgs& a, hipStream_t st);
int launch_conv_x3(const GemmArgs& a, int frames, int* tiles_m, hipStream_t st);

// ---------------------------------------------------------------- weight arena
// Offsets in floats into the packed arena (see api.hip: build_layout()).
struct ConvW { size_t w, b; int cout, cin, k, stride, pad; };
struct MixLayerW { size_t tw0, tb0, tw3, tb3, ln1g, ln1b, ln2g, ln2b, w1, b1, w2, b2; };
struct ArenaLayout {
    ConvW conv[22];          // execution order, conv[0] = stem ([147][64] layout)
    size_t w_in, b_in;       // first Linear, [512][544]
    MixLayerW mix[PIPS_DEPTH];
    size_t lnf_g, lnf_b, w_head, b_head;
    size_t norm_g, norm_b, w_upd_t, b_upd, w_vis, b_vis;
    size_t total;            // floats (fp32 section)
    // bf16 copies of the big Linear weights for the bf16-operand mixer, in ushort units from
    // the end of the fp32 section (arena + total)
    size_t h_w1[PIPS_DEPTH], h_w2[PIPS_DEPTH], h_head, h_conv[22], h_in, total_h;
    // split-bf16 planes [3][N][K] of every matrix-core weight (gemm_x3.hip), in ushort units from
    // the end of the bf16 section
    size_t t_in, t_w1[PIPS_DEPTH], t_w2[PIPS_DEPTH], t_head, t_conv[22], total_t;
    // Everything whose SHAPE depends on the window length S (nets/pips.py:295-301: the token-mixing weights 4S x S / S x 4S
    // and the head S*(C+2) x 512, with its bf16 copy and split planes) sits in a fourth block behind the three sections, so that
    // every other offset is the same for every S.  S-dependent members: mix[].tw0/tb0/tw3/tb3, w_head, b_head, h_head, t_head.
    int S, nout, nout_pad;   // nout = S*(C+2); rows nout .. nout_pad-1 of the head are zero (N % 4 == 0 for the bf16 GEMM)
    size_t total_all;        // floats, the whole arena
};
// window lengths 1 .. PIPS_S_MAX (pips_hip.h): the generic kernels keep the S tokens of two channels in registers
const ArenaLayout& arena_layout(int S = PIPS_S);    // S-independent members are valid whatever S was asked for

// ---------------------------------------------------------------- encoder pieces (encoder.hip)
inline int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

int launch_stem(const void* rgbs, int rgb_u8, const float* w, const float* bias, float* out, float* stats,
                int F, int H, int W, int Ho, int Wo, int* tiles_m, hipStream_t st);
int stem_tiles_m(int Ho, int Wo);
int launch_inorm_finalize_pivot(const float* partial, int F, int parts, int C, float* mean_rstd, hipStream_t st);
// y = relu((x-m)*r)                               (res == null)
// y = relu(res + relu((x-m)*r))                   (res != null, res_stats == null)
// y = relu((res-m2)*r2 + relu((x-m)*r))           (res_stats != null)
int launch_inorm_apply(const float* x, const float* stats, const float* res, const float* res_stats,
                       float* y, int F, int HW, int C, hipStream_t st);
int launch_resize_into(const float* src, int F, int Hs, int Ws, int C, float* dst, int Hd, int Wd,
                       int Cdst, int coff, hipStream_t st);
int launch_avgpool2(const float* src, int F, int H, int W, int C, float* dst, hipStream_t st);
int launch_resize_frames(const void* src, int src_u8, int planes, int h, int w, float* dst, int H, int W, hipStream_t st);
// bf16-activation forms (encoder_bf16.hip): bf16 NHWC maps in and out, fp32 arithmetic
int launch_stem_bf16(const void* rgbs, int rgb_u8, const float* w, const float* bias, void* out_bf16, float* stats,
                     int F, int H, int W, int Ho, int Wo, int* tiles_m, hipStream_t st);
// mode 0: y = relu(n(x));  1: relu(res + relu(n(x)));  2: relu(n2(res) + relu(n(x)));  3: relu(relu(n2(res)) + relu(n(x)))
int launch_inorm_apply_bf16(const void* x, const float* stats, const void* res, const float* res_stats, int mode,
                            void* y, int F, int HW, int C, hipStream_t st);
int launch_resize_into_bf16(const void* src, int F, int Hs, int Ws, int C, void* dst, int Hd, int Wd, int Cdst, int coff,
                            hipStream_t st);

// ---------------------------------------------------------------- tracker pieces (track.hip)
int launch_point_sample(const float* level0, int B, int S, int H8, int W8, const float* xy, int N,
                        float* out, hipStream_t st);
int launch_point_sample_strided(const float* level0, int B, int S, int H8, int W8, const float* xy,
                                int xy_stride, int N, const int* win_start, float* out, hipStream_t st);
// Sw (last argument of the tracker launchers): the window length = mixer rows per particle; PIPS_S runs the specialised kernels
int launch_init_coords(const float* xys, const float* coords_init, int B, int N, float stride,
                       float* coords, float* coords0, float* out_traj0, hipStream_t st, int Sw = PIPS_S);
int launch_init_ffeats(const float* ffeat0, int BN, float* ffeats, hipStream_t st, int Sw = PIPS_S);
int launch_mixer_input(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW,
                       int B, int S, const float* ffeats, const float* coords, const float* times,
                       int N, const int* win_start, float* X, hipStream_t st, int Sw = PIPS_S);
// the direct gather on the bf16 mirror of the pyramid (PIPS_FLAG_BF16_MAPS), and the pass that writes the mirror
int launch_mixer_input_bf16maps(const void* mirror, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B, int S,
                                const float* ffeats, const float* coords, const float* times, int N, const int* win_start,
                                float* X, hipStream_t st, int Sw = PIPS_S);
int launch_pyramid_mirror(const float* pyramid, size_t floats, void* mirror, hipStream_t st);
// LDS-tiled gather for dense query sets (gather_tiled.hip)
size_t tiled_gather_scratch_bytes(int B, int N, int H8, int W8);
bool tiled_gather_wanted(int B, int N, int H8, int W8, bool bf16_maps = false);
// ev != null: 4 events recorded around the three launches (bin, embed, gather)
int launch_mixer_input_tiled(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int B,
                             int S, const float* ffeats, const float* coords, const float* times, int N,
                             float* X, void* scratch, size_t scratch_bytes, hipStream_t st, hipEvent_t* ev = nullptr,
                             const unsigned short* mirror = nullptr);   // mirror: the bf16 mode's matrix-core kernel on the pyramid's bf16 mirror
// x_bf16: the residual stream x is bf16 in memory (bf16 mixer, S = 8 only)
int launch_token_mix(const float* arena, const MixLayerW& L, float* x, float* xn, int particles,
                     hipStream_t st, int xn_bf16 = 0, int Sw = PIPS_S, int x_bf16 = 0);
int launch_ln_mean(const float* x, const float* g, const float* b, float* out, int particles,
                   hipStream_t st, int Sw = PIPS_S, int x_bf16 = 0);
int launch_score_upsum(const float* pyramid, const size_t* lvl_off, const int* lvlH, const int* lvlW, int F, float* U,
                       hipStream_t st);
int launch_score_terms(const float* U, int B, int S, int H8, int W8, const float* ffeats, int N, const float* tgt,
                       float* out, hipStream_t st);
int launch_vis_head(const float* arena, const float* ffeats, int B, int N, float* out_vis, hipStream_t st, int Sw = PIPS_S);
int launch_state_update(const float* arena, const float* delta, float* ffeats, float* coords,
                        const float* coords0, int B, int N, float stride, float* out_traj,
                        float* out_vis, hipStream_t st, int Sw = PIPS_S);

}  // namespace pips
