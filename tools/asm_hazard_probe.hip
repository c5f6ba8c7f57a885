// Probes that pin the table of tools/asm_hazard_lint.py against hipcc's own hazard recognizer (gfx950): one kernel per producer /
// consumer pair, written with builtins and sched_barriers so that the two end up adjacent; the wait states hipcc puts between them
// (s_nop N = N + 1, any other instruction = 1) are what the pair needs.  Compiled with -S by asm_hazard_lint.measure_probes();
// never linked into the product.
#include <hip/hip_runtime.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define SB __builtin_amdgcn_sched_barrier(0)

// OP: 0 v_mfma_f32_32x32x16_bf16, 1 v_mfma_f32_16x16x32_bf16, 2 v_mfma_f32_32x32x2_f32
// KIND: 0 VALU reads the result, 1 ds_write reads it, 2 global_store reads it, 3 an MFMA takes it as A, 4 v_readlane reads it
template <int OP, int KIND>
__global__ void probe_mfma(const bf8* a, const bf8* b, float* out, const float* fa) {
    __shared__ float lds[4096];
    bf8 x = a[threadIdx.x], y = b[threadIdx.x];
    float fx = fa[threadIdx.x], fy = fa[threadIdx.x + 64];
    f16v c = {0};
    f4v d = {0};
    SB;
    if (OP == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    if (OP == 1) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d, 0, 0, 0);
    if (OP == 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c, 0, 0, 0);
    SB;
    float r = OP == 1 ? d[0] : c[0];
    if (KIND == 0) { out[threadIdx.x] = r + 1.0f; }
    if (KIND == 1) { lds[threadIdx.x] = r; __syncthreads(); out[threadIdx.x] = lds[threadIdx.x ^ 1]; }
    if (KIND == 2) { out[threadIdx.x] = r; }
    if (KIND == 3) { f16v e = {0}; e = __builtin_amdgcn_mfma_f32_32x32x2f32(r, fy, e, 0, 0, 0); out[threadIdx.x] = e[3]; }
    if (KIND == 4) { out[threadIdx.x] = __builtin_amdgcn_readlane(__float_as_int(r), 3); }
}
#define INST(O, K) template __global__ void probe_mfma<O, K>(const bf8*, const bf8*, float*, const float*);
INST(0, 0) INST(0, 1) INST(0, 2) INST(0, 3) INST(0, 4)
INST(1, 0) INST(1, 1) INST(1, 2) INST(1, 3) INST(1, 4)
INST(2, 0) INST(2, 1) INST(2, 2) INST(2, 3) INST(2, 4)

// the accumulators in AGPRs, read back by v_accvgpr_read
__global__ void probe_agpr_read(const bf8* a, const bf8* b, float* out) {
    bf8 x = a[threadIdx.x], y = b[threadIdx.x];
    f16v c0 = {0};
    asm volatile("" : "+a"(c0));
    SB; c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); SB;
    asm volatile("" : "+a"(c0));
    SB;
    float r = c0[0] + 1.0f;
    SB;
    out[threadIdx.x] = r;
}
// an MFMA takes part of another MFMA's result as C (overlapping, not identical)
__global__ void probe_mfma_c_overlap(const bf8* a, const bf8* b, float* out) {
    bf8 x = a[threadIdx.x], y = b[threadIdx.x];
    f16v c = {0}; f4v d;
    SB; c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); SB;
    d[0] = c[4]; d[1] = c[5]; d[2] = c[6]; d[3] = c[7];
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d, 0, 0, 0);
    out[threadIdx.x] = d[0] + c[0];
}
// the accumulate chain: same registers as C and D, back to back
__global__ void probe_mfma_c_same(const bf8* a, const bf8* b, float* out) {
    bf8 x = a[threadIdx.x], y = b[threadIdx.x];
    f16v c = {0};
    SB; c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); SB;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
// NB independent MFMAs between producer and consumer: LLVM counts each as ONE wait state
template <int NB>
__global__ void probe_between(const bf8* a, const bf8* b, float* out) {
    bf8 x = a[threadIdx.x], y = b[threadIdx.x];
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    SB; c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); SB;
    if (NB >= 1) { c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c1, 0, 0, 0); SB; }
    if (NB >= 2) { c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, c2, 0, 0, 0); SB; }
    if (NB >= 3) { c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, c3, 0, 0, 0); SB; }
    float r = c0[0] + 1.0f;
    SB;
    out[threadIdx.x] = r + c1[1] + c2[2] + c3[3];
}
template __global__ void probe_between<2>(const bf8*, const bf8*, float*);

__global__ void probe_valu_to_mfma(const float* fa, float* out) {
    float fx = fa[threadIdx.x], fy = fa[threadIdx.x + 64];
    f16v c = {0};
    SB; float r = fx * 2.0f; SB;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(r, fy, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
// v_readlane writes an SGPR, a VALU instruction reads it
__global__ void probe_sgpr_to_valu(const float* fa, float* out) {
    float fx = fa[threadIdx.x];
    SB; float r = fx * 2.0f; SB;
    out[threadIdx.x] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 5));
}
// v_cmp writes VCC, v_cndmask reads it;  v_cndmask writes a VGPR, v_readfirstlane reads it
__global__ void probe_vcc_and_readfirstlane(const float* fa, const float* fb, float* out) {
    int i = fa[threadIdx.x] > 0.f ? 3 : 7;
    SB; int s = __builtin_amdgcn_readfirstlane(i); SB;
    out[threadIdx.x] = fb[s * 1024];
}
// v_readfirstlane writes an SGPR, v_readlane takes it as lane select
__global__ void probe_sgpr_to_lanesel(const float* fa, float* out) {
    int i = (int)fa[threadIdx.x];
    float v = fa[threadIdx.x + 64];
    SB; int s = __builtin_amdgcn_readfirstlane(i); SB;
    out[threadIdx.x] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s));
}
__global__ void probe_trans_to_valu(const float* fa, float* out) {
    float fx = fa[threadIdx.x];
    SB; float r = __builtin_amdgcn_exp2f(fx); SB;
    out[threadIdx.x] = r + 1.0f;
}
__global__ void probe_valu_to_dpp(const float* fa, float* out) {
    float fx = fa[threadIdx.x];
    SB; float r = fx * 2.0f; SB;
    out[threadIdx.x] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0xB1, 0xf, 0xf, false));
}
// a 16-byte store, then a VALU instruction overwrites its data registers
__global__ void probe_store_data(const float4* fa, float4* out, float* o2) {
    float4 v = fa[threadIdx.x];
    SB; out[threadIdx.x] = v; SB;
    v.x = v.y * 2.0f; v.w = v.z * 3.0f;
    o2[threadIdx.x] = v.x + v.w;
}
// v_readfirstlane writes an SGPR, a buffer load takes it as scalar offset
__global__ void probe_sgpr_to_vmem(const float* fa, float* out, __amdgpu_buffer_rsrc_t rsrc) {
    int i = (int)fa[threadIdx.x];
    SB; int s = __builtin_amdgcn_readfirstlane(i); SB;
    out[threadIdx.x] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, threadIdx.x * 4, s, 0));
}
