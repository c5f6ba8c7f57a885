/*
 * pips_hip.h -- C ABI of libpips_hip.so: the PIPs particle-tracker inference hot path
 * (reference: aharley/pips  nets/pips.py:428-611, Pips.forward) as hand-written HIP for
 * gfx950 (MI355X / CDNA4).
 *
 * The reference has no FFI of its own: the operator API of the path is the Python class
 * nets.pips.Pips (nets/pips.py:400-611).  The replacement keeps that class
 * (pips_amd.Pips) and puts ALL arithmetic behind the entry points below; each one names
 * the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PIPS_E_* code; the text of the
 *     last error of the calling thread is available from pips_last_error();
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - the caller owns every buffer (inputs, outputs, weight arena, workspace); the
 *     library never allocates, frees or synchronises (the *_timed entry points excepted).  The
 *     only state kept between calls is idempotent: the dynamic-LDS attribute of each kernel
 *     instantiation (raised on its first launch on a device, tracked per device with atomics)
 *     and the device's compute-unit count (one atomic slot per device).  The library reads no
 *     environment variables (the PIPS_* tuning hooks exist only in a -DPIPS_TUNING build,
 *     pips_amd/_build.py --tuning, where each is read once in a thread-safe static initialiser).
 *     Calls are re-entrant and stream-ordered
 *     on `stream` (a hipStream_t passed as void*).  hipGraph capture: run the same call once
 *     eagerly first (so no attribute is set while capturing), then capture and replay -- replays
 *     are bit-identical to the eager call (tests/test_forward_gpu.py::test_forward_in_hip_graph);
 *   - all tensors are dense fp32 unless stated; "frames" F = B*S; mixer rows are ordered
 *     m = (b*N + n)*S + s ("particle-major"), map levels are channel-last
 *     [F][H_l][W_l][128].
 */
#ifndef PIPS_HIP_H
#define PIPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIPS_OK            0
#define PIPS_E_ARG        -1   /* bad shape / null pointer / unsupported size        */
#define PIPS_E_WORKSPACE  -2   /* workspace too small                                */
#define PIPS_E_LAUNCH     -3   /* hipLaunch error (see pips_last_error)              */

#define PIPS_S        8        /* frames per window of the shipped checkpoints; other S: the _s entry points */
#define PIPS_C        128      /* latent channels            nets/pips.py:408        */
#define PIPS_LEVELS   4        /* correlation pyramid levels nets/pips.py:409        */
#define PIPS_RADIUS   3        /* correlation radius         nets/pips.py:410        */
#define PIPS_NCORR    196      /* LEVELS * (2R+1)^2                                  */
#define PIPS_KIN      519      /* mixer input width          nets/pips.py:289        */
#define PIPS_KIN_PAD  544      /* same, zero-padded to a multiple of 32              */
#define PIPS_DMIX     512      /* mixer width                nets/pips.py:298        */
#define PIPS_DEPTH    12       /* mixer depth                nets/pips.py:300        */
#define PIPS_NOUT     1040     /* S*(C+2)                    nets/pips.py:299        */
#define PIPS_NPARAMS  200      /* tensors in the reference state dict                */

/* flags of pips_forward / pips_track */
#define PIPS_FLAG_REUSE_MAPS  1   /* pips_forward: skip the encoder, the workspace already holds the maps (of a call with the
                                     same B,S,H,W,stride AND the same encoder flags: they describe the maps) */
#define PIPS_FLAG_BF16_ENCODER 4  /* the encoder the way torch.autocast(bfloat16) runs it: bf16 MFMA operands in all 22
                                     convolutions (7x7 stem included) and bf16 activation maps between the layers;
                                     statistics from the fp32 accumulators, normalisation / ReLU / adds / resizes in
                                     fp32 arithmetic rounded once on store; the pyramid handed to the tracker is fp32 */
#define PIPS_FLAG_RGB_U8      8   /* rgbs points at uint8 (B,S,3,H,W) frames instead of float: same values, a
                                     quarter of the input bytes (the decoded frames of demo.py:136-144) */
#define PIPS_FLAG_BF16_MIXER  2   /* bf16 MFMA operands in the channel-mix and head Linear layers (BASELINE
                                     config 3); accumulation, norms, GELU, residual stream, gather stay fp32 */

#define PIPS_FLAG_BF16_STREAM 64  /* with PIPS_FLAG_BF16_MIXER (window length 8): the mixer's residual stream is a bf16 tensor, as
                                     PreNormResidual's `fn(norm(x)) + x` is under autocast (nets/pips.py:93-100: both terms bf16) --
                                     token mixing and the down-projections read it, add in fp32 and round once on the way out;
                                     LayerNorm statistics, GELU and accumulation stay fp32.  IGNORED for window lengths other than
                                     8 (the generic-S token mixing keeps an fp32 stream).  pips_amd.Pips sets it by default with a
                                     bf16 mixer since round 5 (outputs differ from rounds 1-4 by ~1e-2 px at BASELINE configs[2];
                                     Pips.mixer_stream_dtype = torch.float32 restores the fp32 stream) */
#define PIPS_FLAG_BF16_MAPS   32  /* pips_track / pips_mixer_input_build_ex: the correlation gather reads the bf16 MIRROR of the
                                     pyramid (behind the fp32 levels, pips_pyramid_mirror_offset; written by the bf16 encoder or
                                     pips_pyramid_mirror) -- the reference's rounding point under autocast, where the encoder's
                                     output is a bf16 tensor; pips_forward sets it itself when both BF16 flags are given */
#define PIPS_FLAG_SPLIT_BF16  16  /* fp32-grade "split-bf16" matrix path: every fp32 operand is split exactly into
                                     three bf16 terms (round-to-nearest at each step), six exact bf16 products per fp32 product, fp32 accumulation
                                     (pips_gemm_f32x3) -- all mixer Linear layers and the convolutions where it is
                                     faster; same accuracy class as the exact-fp32 MFMA path, not bitwise equal to
                                     it; takes precedence over the two BF16 flags */

const char* pips_last_error(void);
/* 3 (round 6).  History: 2 -> 3 (buffer contracts that grew in round 5; a caller holding buffers sized by the v2 rules gets
 * out-of-range reads on a dense query set with PIPS_FLAG_BF16_MAPS): pips_pyramid_floats() includes a SLACK behind the bf16 mirror
 * (4 rows + 16 pixels of the coarsest level: gather_mfma_kernel fetches whole 8 x 4 pixel blocks unmasked past the last level's
 * last frame), and the tiled gather's scratch (inside pips_track_workspace_bytes* / pips_workspace_bytes) carries the slot map,
 * the bf16 feature rows of the sorted order and a batch of slots of slack -- size every buffer with the query functions of THIS
 * version; PIPS_EPI_RES_BF16 is a public constant; pips_mixer_gemm_train honours PIPS_FLAG_BF16_STREAM.
 * 1 -> 2: the pyramid buffer of EVERY encoder mode is pips_pyramid_floats() long (the bf16 encoder writes
 * a bf16 mirror of the four levels behind them, at pips_pyramid_mirror_offset -- a buffer sized from the level offsets alone
 * is too short); the PIPS_PACK_FFN arena section and pips_mixer_fwd_bf16_fused (round 3's fused FeedForward, measured slower than
 * the two GEMMs again in round 4 -- tools/experiments/) are gone. */
int         pips_abi_version(void);

/* ---- weights ------------------------------------------------------------------------
 * Replaces: nn.Module parameter storage + load_state_dict (saverloader.py:58-59).
 * `params[i]` is the device pointer of the i-th tensor of the reference state dict in
 * its canonical order (nets/pips.py:400-426; pips_amd/weights.py:param_table), in the
 * reference's own layout.  The arena receives the kernel-side layouts
 * (conv [Cout][kh][kw][Cin], stem [ci*kh*kw][64], first Linear zero-padded to 544
 * columns, updater Linear transposed, the rest verbatim). */
size_t pips_weight_arena_bytes(void);
int    pips_repack_weights(const void* const* params_host, int nparams, void* arena, void* stream);
/* The arena's three sections can be (re)built separately: PIPS_PACK_FP32 (the fp32 layouts, from params), PIPS_PACK_BF16
 * (bf16 copies of every matrix-core weight: the bf16-operand modes) and PIPS_PACK_SPLIT (three bf16 planes per weight: the
 * split-bf16 mode) -- the last two are derived from the arena's fp32 section (params may be null without PIPS_PACK_FP32),
 * so a process packs only what its matrix mode reads.  pips_repack_weights = all sections. */
#define PIPS_PACK_FP32  1
#define PIPS_PACK_BF16  2
#define PIPS_PACK_SPLIT 4
int    pips_repack_weights_ex(const void* const* params_host, int nparams, void* arena, int sections, void* stream);

/* ---- any window length: Pips(S != 8) ----------------------------------------------------
 * Replaces: the S argument of nets.pips.Pips.__init__ (nets/pips.py:401-402), which sizes the token-mixing weights
 * (4S x S, S x 4S: :102-109,117) and the head (S*(C+2) x 512: :295-301).  Every entry point above and below without an S
 * argument means S = PIPS_S = 8, the window of every shipped checkpoint, and runs kernels specialised for it; the _s
 * variants take 1 <= S <= PIPS_S_MAX (the arena must have been packed for the same S) and run the token mixing, the final
 * LayerNorm + mean and the state update on generic kernels (same arithmetic, not tuned), the GEMMs and the gather on
 * the same kernels as S = 8.  pips_forward / pips_workspace_bytes take S from their own argument.  Rows of the mixer output
 * (delta) are pips_delta_stride(S) = S*(C+2) rounded up to a multiple of 4 floats apart.  (Round 6: 32, was 16 -- the generic
 * kernels are instantiated for 16 and for 32 token registers and picked by S.) */
#define PIPS_S_MAX 32
size_t pips_weight_arena_bytes_s(int S);
int    pips_repack_weights_s(const void* const* params_host, int nparams, void* arena, int S, int sections, void* stream);
int    pips_delta_stride(int S);
size_t pips_mixer_workspace_bytes_s(int M, int S);
int    pips_mixer_fwd_s(const void* arena, const float* X, int M, int S, int flags, float* delta,
                        void* workspace, size_t workspace_bytes, void* stream);     /* flags: BF16_MIXER (+ BF16_STREAM at S = 8) / SPLIT_BF16 */
size_t pips_track_workspace_bytes_s(int B, int N, int S);
/* pips_track / pips_track_ce (below) with the window length as an argument; ce_* may be NULL */
int    pips_track_s(const void* arena, const float* pyramid, int B, int T, int H8, int W8, const float* xys,
                    const float* coords_init, const float* feat_init, const int* win_start, const float* times, int N,
                    int stride, int iters, int flags, int S, void* workspace, size_t workspace_bytes, float* out_trajs,
                    float* out_vis, float* out_ffeat0, const float* ce_tgt, float* ce_terms, void* ce_ws,
                    size_t ce_ws_bytes, void* stream);

/* ---- whole forward -------------------------------------------------------------------
 * Replaces: Pips.forward, inference branch (nets/pips.py:428-611 minus the dead fcp
 * upsample :504-511, the sw visualisation branches and the losses :600-606).
 *   rgbs        (B,S,3,H,W) fp32 0..255, NCHW exactly as callers pass it (demo.py:40)
 *   xys         (B,N,2) pixels
 *   coords_init (B,S,N,2) pixels or NULL   (nets/pips.py:452-455)
 *   feat_init   (B,N,128) or NULL          (nets/pips.py:461-465)
 *   times       (S) = torch.linspace(0,S,S) (nets/pips.py:519)
 *   out_trajs   (iters+1,B,S,N,2) pixels: entry 0 = initial coords*stride
 *               (coord_predictions2[0]), entries 1.. = coord_predictions
 *   out_vis     (B,S,N) logits             (nets/pips.py:559)
 *   out_ffeat0  (B,N,128) initial feature  (nets/pips.py:463, returned when return_feat)
 * stride is 4 or 8 in the reference's callers; any value >=1 with non-empty level-3 map.
 * flags: PIPS_FLAG_REUSE_MAPS = skip the encoder and reuse the pyramid already in the workspace
 *        (same B,S,H,W,stride as the call that produced it); PIPS_FLAG_BF16_MIXER. */
size_t pips_workspace_bytes(int B, int S, int H, int W, int N, int stride);
int    pips_forward(const void* arena, const float* rgbs, const float* xys,
                    const float* coords_init, const float* feat_init, const float* times,
                    int B, int S, int H, int W, int N, int stride, int iters, int flags,
                    void* workspace, size_t workspace_bytes,
                    float* out_trajs, float* out_vis, float* out_ffeat0, void* stream);

/* ---- tracker on cached maps ------------------------------------------------------------
 * Replaces: Pips.forward minus the encoder, for callers that re-run the tracker on maps they
 * already have: the dense-grid loop of test_on_davis.py:103-130 (one encoder pass, many query
 * chunks) and the visibility-aware chaining of chain_demo.py:40-83 / test_on_badja.py:64-112
 * (one encoder pass per video frame instead of per particle and hop).
 *   pyramid    packed 4-level channel-last maps of B clips x T frames (pips_encoder_fwd with
 *              F = B*T; per-frame InstanceNorm makes a frame's maps independent of its clip)
 *   win_start  (B*N) int32 first frame of each particle's 8-frame window, or NULL (= 0);
 *              frames past T-1 repeat frame T-1 (chain_demo.py:50-52)
 * All other arguments as pips_forward.  T = 8 and win_start = NULL is exactly the forward. */
size_t pips_track_workspace_bytes(int B, int N);
int    pips_track(const void* arena, const float* pyramid, int B, int T, int H8, int W8,
                  const float* xys, const float* coords_init, const float* feat_init,
                  const int* win_start, const float* times, int N, int stride, int iters, int flags,
                  void* workspace, size_t workspace_bytes,
                  float* out_trajs, float* out_vis, float* out_ffeat0, void* stream);

/* ---- stages (same kernels, exposed for parity tests and for callers that cache maps) --*/

/* BasicEncoder.forward (nets/pips.py:247-281) incl. the 2*(x/255)-1 of :436 and
 * CorrBlock.__init__ (:346-352).  Writes the 4-level channel-last pyramid:
 * level l at pyramid + pips_pyramid_offset(l), shape [F][H_l][W_l][128]. */
size_t pips_encoder_workspace_bytes(int F, int H, int W, int stride);
size_t pips_pyramid_floats(int F, int H, int W, int stride);
size_t pips_pyramid_offset(int F, int H, int W, int stride, int level);   /* in floats */
/* pips_pyramid_floats = the four fp32 levels + their bf16 mirror (same element offsets, half the bytes) behind them */
size_t pips_pyramid_mirror_offset(int F, int H, int W, int stride);       /* in floats: where the mirror starts = size of the fp32 levels */
int    pips_pyramid_mirror(float* pyramid, int F, int H, int W, int stride, void* stream);   /* (re)write the mirror from the fp32 levels */
int    pips_encoder_fwd(const void* arena, const float* rgbs, int F, int H, int W, int stride,
                        float* pyramid, void* workspace, size_t workspace_bytes, void* stream);

int    pips_encoder_fwd_bf16(const void* arena, const float* rgbs, int F, int H, int W, int stride,
                             float* pyramid, void* workspace, size_t workspace_bytes, void* stream);
/* general form: flags = PIPS_FLAG_BF16_ENCODER | PIPS_FLAG_RGB_U8 */
int    pips_encoder_fwd_ex(const void* arena, const void* rgbs, int F, int H, int W, int stride, int flags,
                           float* pyramid, void* workspace, size_t workspace_bytes, void* stream);

/* The callers' frame pre-processing (demo.py:22-28, chain_demo.py:26-28, test_on_davis.py:93-95):
 * F.interpolate(rgbs, (H,W), mode='bilinear') (align_corners=False) of decoded frames, on the device.
 * src: planes = F*3 images of h x w, uint8 (src_is_u8 != 0) or float; dst: float (planes,H,W), values 0..255,
 * i.e. exactly what pips_forward / pips_encoder_fwd take as rgbs. */
int    pips_resize_frames(const void* src, int src_is_u8, int planes, int h, int w,
                          float* dst, int H, int W, void* stream);

/* utils.samp.bilinear_sample2d (utils/samp.py:5-78): clamped-index point sample of frame
 * 0 of every clip.  xy (B,N,2) in map pixels -> out (B,N,128). */
int    pips_point_sample(const float* level0, int B, int S, int H8, int W8,
                         const float* xy, int N, float* out, void* stream);

/* CorrBlock.corr + CorrBlock.sample + get_3d_embedding + the concat of
 * DeltaBlock.forward (nets/pips.py:384-398, 355-382, 517-522, 304-308; utils/misc.py:44-69):
 * builds the mixer input X (B*N*S, 544) = [ffeat 128 | corr 196 | sincos 192 | dx dy t | 0..].
 * ffeats (B*N*S,128), coords (B*N*S,2) in map pixels, both particle-major. */
int    pips_mixer_input_build(const float* pyramid, int B, int S, int H8, int W8,
                              const float* ffeats, const float* coords, const float* times,
                              int N, float* X, void* stream);
/* the same with per-particle window starts (pips_track's win_start, may be NULL) and flags = 0 | PIPS_FLAG_BF16_MAPS */
int    pips_mixer_input_build_ex(const float* pyramid, int B, int S, int H8, int W8, const float* ffeats, const float* coords,
                                 const float* times, int N, const int* win_start, int flags, float* X, void* stream);

/* Same result as pips_mixer_input_build through the LDS-tiled kernels meant for dense query sets
 * (BASELINE configs[3], test_on_davis.py:103-130): particles binned by 16x16 map tile, the tile's
 * halo region at each level staged in LDS once per tile (csrc/gather_tiled.hip).  pips_track /
 * pips_forward pick it by themselves when the query set is dense (>= 16 particles per tile on average and >= 1024 per
 * frame -- >= 256 with PIPS_FLAG_BF16_MAPS, whose matrix-core kernel pays earlier: BASELINE configs[2] takes it since
 * round 6 -- within the kernel's 32-bit offset limits; otherwise the direct kernel).  scratch holds the per-frame sort,
 * the per-tile tables and (read by the bf16 mode's kernel) the features as bf16 in the sorted order; values agree
 * with the direct kernel to fp32 summation order.  The _timed form also returns the HIP-event
 * durations (ms) of its three launches {bin_particles, embed_rows, gather_tiled} in ms3_host and
 * synchronises the stream (measurement only). */
size_t pips_gather_scratch_bytes(int B, int N, int H8, int W8);
int    pips_mixer_input_build_tiled(const float* pyramid, int B, int S, int H8, int W8,
                                    const float* ffeats, const float* coords, const float* times,
                                    int N, float* X, void* scratch, size_t scratch_bytes, void* stream);
int    pips_mixer_input_build_tiled_timed(const float* pyramid, int B, int S, int H8, int W8,
                                          const float* ffeats, const float* coords, const float* times,
                                          int N, float* X, void* scratch, size_t scratch_bytes, void* stream,
                                          float* ms3_host);
/* The same with flags.  PIPS_FLAG_BF16_MAPS: the bf16 mode of a dense query set -- under torch.autocast the reference correlates
 * bf16 features with bf16 maps (nets/pips.py:394-397), so the work items of the tiled path run on the matrix cores
 * (gather_mfma_kernel: the tile's region of the pyramid's bf16 MIRROR times all of the item's particles' features, rounded to
 * bf16, as v_mfma_f32_32x32x16_bf16 products with fp32 sums; each particle keeps its 8 x 8 window; same blend, same taps).
 * pips_forward / pips_track take this route by themselves when PIPS_FLAG_BF16_MAPS is set and the query set is dense.
 * ms3_host != NULL: as the _timed form ({bin_particles, embed_rows, gather} ms; synchronises the stream). */
/* Which correlation-gather kernel pips_forward / pips_track (8 frames per clip, no per-particle windows) run for this query set:
 * 0 = the direct kernel (mixer_input_kernel, or mixer_input_bf16maps_kernel with PIPS_FLAG_BF16_MAPS), 1 = gather_tiled_kernel
 * (fp32, LDS-tiled), 2 = gather_mfma_kernel (PIPS_FLAG_BF16_MAPS on a dense query set).  Host function; needs a current device. */
int    pips_gather_route(int B, int N, int H8, int W8, int flags);
int    pips_mixer_input_build_tiled_ex(const float* pyramid, int B, int S, int H8, int W8,
                                       const float* ffeats, const float* coords, const float* times,
                                       int N, int flags, float* X, void* scratch, size_t scratch_bytes,
                                       void* stream, float* ms3_host);

/* MLPMixer (nets/pips.py:111-123): X (M,544) -> delta (M/8, 1040).  M = B*N*8. */
size_t pips_mixer_workspace_bytes(int M);
int    pips_mixer_fwd(const void* arena, const float* X, int M, float* delta,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Same with bf16 MFMA operands (PIPS_FLAG_BF16_MIXER): weights converted once at pack time,
 * activations rounded to bf16 (RNE) as they are staged, fp32 accumulation and epilogues. */
int    pips_mixer_fwd_bf16(const void* arena, const float* X, int M, float* delta,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Same with every GEMM on the split-bf16 (bf16x3) path: fp32-grade results, see pips_gemm_f32x3. */
int    pips_mixer_fwd_x3(const void* arena, const float* X, int M, float* delta,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Profiling variant (the ONLY entry point that creates events and synchronises): same work as
 * pips_mixer_fwd with a hipEvent pair around every GEMM launch on `stream`; ms_host[5] receives
 * {in-proj, mean of the 12 up-projections (512->2048 +GELU), mean of the 12 down-projections
 * (2048->512 +residual), head, marker-pair overhead (empty event pair)} in milliseconds, the
 * first four RAW (not overhead-corrected).  Used by bench.py for the roofline object. */
int    pips_mixer_fwd_timed(const void* arena, const float* X, int M, float* delta,
                            void* workspace, size_t workspace_bytes, void* stream, float* ms_host);
/* Same for the matrix mode selected by flags (0, PIPS_FLAG_BF16_MIXER or PIPS_FLAG_SPLIT_BF16). */
int    pips_mixer_fwd_timed_ex(const void* arena, const float* X, int M, int flags, float* delta,
                               void* workspace, size_t workspace_bytes, void* stream, float* ms_host);

/* Profiling: the two channel-mix Linear shapes (nets/pips.py:102-109) as launch trains -- the 12 layers' up-projections, then
 * their down-projections, back to back on the layers' own weights between ONE HIP event pair, `reps` times.
 * ms2_host = {up-projection, down-projection} milliseconds PER LAUNCH, start to start as the forward pays them (no per-launch
 * marker, nothing subtracted).  The workspace must hold a mixer pass at this M (pips_mixer_fwd* on it first).  bench.py's
 * roofline.frac comes from this. */
int    pips_mixer_gemm_train(const void* arena, int M, int flags, void* workspace, size_t workspace_bytes,
                             void* stream, int reps, float* ms2_host);

/* State update nets/pips.py:525-539 (+ vis head :559 when out_vis != NULL).
 * delta (B*N,1040); ffeats/coords updated in place; coords0 = locked frame-0 coords;
 * out_traj (B,S,N,2) receives coords*stride. */
int    pips_state_update(const void* arena, const float* delta, float* ffeats, float* coords,
                         const float* coords0, int B, int N, float stride,
                         float* out_traj, float* out_vis, void* stream);

/* Generic fp32-MFMA building blocks (exposed for unit tests).
 * C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N]); K % 32 == 0; epi: 0 none, 1 GELU(erf),
 * 2 add residual R (ldr). */
int    pips_gemm_f32(const float* A, int lda, const float* W, const float* bias,
                     float* C, int ldc, int M, int N, int K, int epi,
                     const float* R, int ldr, void* stream);
/* NHWC convolution as implicit GEMM, weights [Cout][kh][kw][Cin], Cin % 32 == 0.
 * stats (optional) receives InstanceNorm partials of the output about a pivot, float4 {sum(x-p), sum((x-p)^2), p, n}
 * per (frame, part, channel), part = m-tile x wave row: room for F * (2*ceil(Ho*Wo/64) + 4) * Cout * 4 floats;
 * returns the number of parts per frame through *tiles_m_host. */
int    pips_conv_nhwc_f32(const float* in, int F, int H, int W, int Cin,
                          const float* wgt, const float* bias, int Cout, int ksize, int cstride, int pad,
                          float* out, float* stats, int* tiles_m_host, void* stream);

/* epi values of the GEMM building blocks; PIPS_EPI_RES_BF16 is OR-ed to PIPS_EPI_RESIDUAL for pips_gemm_bf16 with out_bf16 = 1:
 * the residual R is a bf16 tensor too (the mixer's bf16 residual stream, PIPS_FLAG_BF16_STREAM).  Any other combination with it is
 * rejected (PIPS_E_ARG) -- without out_bf16 a bf16 R would be read as fp32. */
#define PIPS_EPI_BIAS      0
#define PIPS_EPI_GELU      1
#define PIPS_EPI_RESIDUAL  2
#define PIPS_EPI_RES_BF16  0x1000
/* bf16-operand building block (BASELINE config 3): A fp32 (rounded to bf16 while staged) or bf16 [M][lda], W bf16 [N][K]
 * (round-to-nearest-even of the fp32 weights), fp32 accumulation, C fp32 or bf16 [M][ldc]; epi as pips_gemm_f32
 * (+ PIPS_EPI_RES_BF16).  K % 32 == 0 (K % 64 unless A and C are fp32). */
int    pips_gemm_bf16(const void* A, int a_bf16, int lda, const void* W, const float* bias, void* C, int out_bf16, int ldc,
                      int M, int N, int K, int epi, const float* R, int ldr, void* stream);
/* Which kernel pips_gemm_bf16 (and the bf16 mixer of pips_forward) takes for a problem with bias and, for epi = residual, an
 * fp32 residual of ldr = N: 0 = register-staged gemm_bf16_kernel; 3 = gemm_bf16_t4_res_kernel (down-projection: bf16 A, fp32 C,
 * bias + residual, M % 128 == 0, N % 256 == 0, K % 64 == 0, at least half a 128 x 256 tile per compute unit); 4 =
 * gemm_bf16_t4_gelu_kernel (up-projection: bf16 A and C, GELU, K = 512, M and N multiples of 256, at least three quarters of a 256 x 256
 * tile per compute unit; rounds the Linear output to bf16 ahead of the GELU as autocast does).  Both are four-wave kernels on
 * v_mfma_f32_16x16x32_bf16 with a generated static schedule (1 and 2 were kernels of rounds 2-3).  Host function;
 * needs a current device.  The mixer's bf16 numerics depend on M = B*N*8 through this choice. */
int    pips_gemm_bf16_route(int M, int N, int K, int epi, int a_bf16, int out_bf16);

/* Which kernel pips_gemm_f32 (and the fp32 mixer of pips_forward) takes for a problem with bias and, for epi = residual, a residual of
 * ldr = N: 0 = igemm_f32_kernel (gemm.hip); 1 = gemm_f32_t4u_kernel (128 x 128 tiles: epi GELU or residual, M and N multiples of
 * 128, K % 64 == 0, at least three quarters of a tile per compute unit); 2 = gemm_f32_t4e_kernel (64 x 64 tiles, LDS-DMA staging; the K range split
 * over the four waves and summed in a fixed order: epi residual, K % 128 == 0, between 0.75 and 2 tiles per compute unit -- the
 * down-projection at M = 2048).  Same exact-fp32 MFMA arithmetic everywhere; 1 is bitwise igemm_f32_kernel's unsplit form, 2
 * differs from it by the order of the four partial sums.  Host function; needs a current device. */
int    pips_gemm_f32_route(int M, int N, int K, int epi);
/* Compute units of the current device (0: no device).  Every "tiles per compute unit" threshold of the route functions is
 * relative to this number; tests derive their route expectations from it instead of assuming 256. */
int    pips_device_cus(void);

/* pips_conv_nhwc_f32 with bf16 MFMA operands: the fp32 map is rounded to bf16 while it is staged, wgt_bf16 is the
 * round-to-nearest-even bf16 copy of the [Cout][kh][kw][Cin] weights; fp32 accumulation and output, same stats. */
int    pips_conv_nhwc_bf16(const float* in, int F, int H, int W, int Cin,
                           const void* wgt_bf16, const float* bias, int Cout, int ksize, int cstride, int pad,
                           float* out, float* stats, int* tiles_m_host, void* stream);

/* The same convolution on bf16 MAPS (what the bf16 encoder mode uses between its layers): in_bf16 is a bf16 NHWC map;
 * in_norm (optional, 64 -> 64 3x3 stride-1 layers on maps the LDS-resident kernel takes: >= 512 tiles of 4x64 pixels)
 * holds {mean, rstd} per (frame, input channel) of the layer that produced the map, and relu((x - mean) * rstd) is
 * applied to it while it is staged (taps outside the image stay zero); out is bf16 (out_is_bf16) or fp32; the
 * statistics are taken from the fp32 accumulators.  stats_parts_cap = room in stats in partials per frame (0 = the
 * 2*ceil(Ho*Wo/64)+4 of pips_conv_nhwc_f32); the ping-pong 64 -> 64 kernel wants ceil(W/32)*ceil(H/4)*4 and is
 * only taken when that fits. */
int    pips_conv_nhwc_bf16_maps(const void* in_bf16, const float* in_norm, int F, int H, int W, int Cin,
                                const void* wgt_bf16, const float* bias, int Cout, int ksize, int cstride, int pad,
                                void* out, int out_is_bf16, float* stats, int stats_parts_cap, int* tiles_m_host,
                                void* stream);

/* Split-bf16 ("bf16x3") building blocks: fp32-grade results from the bf16 matrix cores.  Every
 * fp32 operand is split exactly into three bf16 terms and each product is formed from six exact
 * bf16 products accumulated in fp32 (same F.linear / F.conv2d contracts as the two calls above).
 * pips_split_bf16x3: src fp32 [n] (n even) -> dst bf16 planes [3][n] (6n bytes).
 * pips_gemm_f32x3 / pips_conv_nhwc_f32x3: W3 / wgt3 are the split planes of the fp32 weights. */
int    pips_split_bf16x3(const float* src, size_t n, void* dst3, void* stream);
int    pips_gemm_f32x3(const float* A, int lda, const void* W3, const float* bias,
                       float* C, int ldc, int M, int N, int K, int epi,
                       const float* R, int ldr, void* stream);
int    pips_conv_nhwc_f32x3(const float* in, int F, int H, int W, int Cin,
                            const void* wgt3, const float* bias, int Cout, int ksize, int cstride, int pad,
                            float* out, float* stats, int* tiles_m_host, void* stream);

/* Score-map loss terms of nets/pips.py:501-511 + score_map_loss :58-92 (evaluation: test_on_flt.py:87 and
 * test_on_crohd.py:133 pass trajs_g / vis_g / valids).  pips_forward_ce / pips_track_ce are pips_forward / pips_track
 * plus: ce_tgt (B*N*S,3) = per mixer row m=(b*N+n)*S+s {x, y, use}: the rounded target pixel in map coordinates
 * (trajs_g/stride rounded half-to-even) and use = 1 when that heat map enters the loss (target inside the map,
 * valids > 0, vis_g > 0), else 0;  ce_terms (iters, B*N*S, 2) receives per iteration and row {loss at the target
 * pixel, sum of the losses of all other pixels} (balanced_ce_loss's stable softplus, :26-29) -- the caller divides
 * the two totals by (#used rows * iters) and (#used rows * iters * (H8*W8 - 1)) (+1e-6, utils.basic.reduce_masked_mean)
 * and adds them;  ce_ws: pips_score_map_workspace_bytes(B,S,H8,W8) of scratch (the four pyramid levels upsampled and
 * summed once: the (B,S,N,H8,W8) volume itself is never formed).  ce_tgt == NULL: exactly pips_forward / pips_track.
 * pips_score_map_prepare / _terms are the two stages on their own. */
size_t pips_score_map_workspace_bytes(int B, int S, int H8, int W8);
int    pips_score_map_prepare(const float* pyramid, int B, int S, int H8, int W8, float* U, void* stream);
int    pips_score_map_terms(const float* U, int B, int S, int H8, int W8, const float* ffeats, int N,
                            const float* tgt, float* out, void* stream);
int    pips_forward_ce(const void* arena, const float* rgbs, const float* xys,
                       const float* coords_init, const float* feat_init, const float* times,
                       int B, int S, int H, int W, int N, int stride, int iters, int flags,
                       void* workspace, size_t workspace_bytes,
                       float* out_trajs, float* out_vis, float* out_ffeat0,
                       const float* ce_tgt, float* ce_terms, void* ce_ws, size_t ce_ws_bytes, void* stream);
int    pips_track_ce(const void* arena, const float* pyramid, int B, int T, int H8, int W8,
                     const float* xys, const float* coords_init, const float* feat_init,
                     const int* win_start, const float* times, int N, int stride, int iters, int flags,
                     void* workspace, size_t workspace_bytes,
                     float* out_trajs, float* out_vis, float* out_ffeat0,
                     const float* ce_tgt, float* ce_terms, void* ce_ws, size_t ce_ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIPS_HIP_H */
