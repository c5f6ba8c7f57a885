// Micro-benchmark: one wave per SIMD issuing MFMAs with LDS fragment reads between them at the bf16 up-projection's ratio (per 32 K values of a
// 128 x 128 wave tile: 16 ds_read_b128 + 8 more memory instructions standing in for the staging), for the two bf16 shapes:
//   v_mfma_f32_16x16x32_bf16: 64 MFMAs of 16 clocks per 32 K -- a memory instruction after every 2nd / 3rd MFMA
//   v_mfma_f32_32x32x16_bf16: 32 MFMAs of 32 clocks per 32 K -- a memory instruction after (nearly) every MFMA
// Same flops, same LDS bytes.  Does the longer MFMA hide the issue of the memory instructions of its own wave?  Tuning aid, not product code.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_lds_mix tools/mfma_lds_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CLOB "memory", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define M16(c) "v_mfma_f32_16x16x32_bf16 a[" #c "], v[2:5], v[6:9], a[" #c "]\n\t"
#define M32(c) "v_mfma_f32_32x32x16_bf16 a[" #c "], v[2:5], v[6:9], a[" #c "]\n\t"
#define RD(r, off) "ds_read_b128 v[" #r "], v1 offset:" #off "\n\t"

template <int SHAPE, int MEM>   // MEM: 0 none, 1 the 24 reads per 32 K
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(sm)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    // v1 = LDS address, v[2:5] / v[6:9] = the operands, v[10:105] = 24 read targets (never consumed: the reads only have to issue and land)
    asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, 0x3f803f80\n\tv_mov_b32 v3, 0x3f803f80\n\tv_mov_b32 v4, 0x3f803f80\n\tv_mov_b32 v5, 0x3f803f80\n\t"
                 "v_mov_b32 v6, 0x3f003f00\n\tv_mov_b32 v7, 0x3f003f00\n\tv_mov_b32 v8, 0x3f003f00\n\tv_mov_b32 v9, 0x3f003f00" ::"v"(lds)
                 : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9");
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
            if (MEM)   // 64 MFMAs, 24 reads: after MFMAs 0, 2, 5, 8, 10, 13, ... (pattern 3-3-2 over 8 MFMAs)
                asm volatile(
                    M16(0:3) RD(10:13, 0) M16(4:7) M16(8:11) RD(14:17, 4096) M16(12:15) M16(16:19) M16(20:23) RD(18:21, 8192) M16(24:27) M16(28:31)
                    M16(32:35) RD(22:25, 12288) M16(36:39) M16(40:43) RD(26:29, 16384) M16(44:47) M16(48:51) M16(52:55) RD(30:33, 20480) M16(56:59) M16(60:63)
                    M16(64:67) RD(34:37, 24576) M16(68:71) M16(72:75) RD(38:41, 28672) M16(76:79) M16(80:83) M16(84:87) RD(42:45, 32768) M16(88:91) M16(92:95)
                    M16(96:99) RD(46:49, 36864) M16(100:103) M16(104:107) RD(50:53, 40960) M16(108:111) M16(112:115) M16(116:119) RD(54:57, 45056) M16(120:123) M16(124:127)
                    M16(128:131) RD(58:61, 49152) M16(132:135) M16(136:139) RD(62:65, 53248) M16(140:143) M16(144:147) M16(148:151) RD(66:69, 57344) M16(152:155) M16(156:159)
                    M16(160:163) RD(70:73, 61440) M16(164:167) M16(168:171) RD(74:77, 512) M16(172:175) M16(176:179) M16(180:183) RD(78:81, 4608) M16(184:187) M16(188:191)
                    M16(192:195) RD(82:85, 8704) M16(196:199) M16(200:203) RD(86:89, 12800) M16(204:207) M16(208:211) M16(212:215) RD(90:93, 16896) M16(216:219) M16(220:223)
                    M16(224:227) RD(94:97, 20992) M16(228:231) M16(232:235) RD(98:101, 25088) M16(236:239) M16(240:243) M16(244:247) RD(102:105, 29184) M16(248:251) M16(252:255)
                    "s_waitcnt lgkmcnt(0)" ::: CLOB);
            else
                asm volatile(
                    M16(0:3) M16(4:7) M16(8:11) M16(12:15) M16(16:19) M16(20:23) M16(24:27) M16(28:31) M16(32:35) M16(36:39) M16(40:43) M16(44:47) M16(48:51) M16(52:55) M16(56:59) M16(60:63)
                    M16(64:67) M16(68:71) M16(72:75) M16(76:79) M16(80:83) M16(84:87) M16(88:91) M16(92:95) M16(96:99) M16(100:103) M16(104:107) M16(108:111) M16(112:115) M16(116:119) M16(120:123) M16(124:127)
                    M16(128:131) M16(132:135) M16(136:139) M16(140:143) M16(144:147) M16(148:151) M16(152:155) M16(156:159) M16(160:163) M16(164:167) M16(168:171) M16(172:175) M16(176:179) M16(180:183) M16(184:187) M16(188:191)
                    M16(192:195) M16(196:199) M16(200:203) M16(204:207) M16(208:211) M16(212:215) M16(216:219) M16(220:223) M16(224:227) M16(228:231) M16(232:235) M16(236:239) M16(240:243) M16(244:247) M16(248:251) M16(252:255)
                    "s_nop 0" ::: CLOB);
        } else {
            if (MEM)   // 32 MFMAs (two passes over the 16 accumulator blocks), 24 reads: after 3 of every 4 MFMAs
                asm volatile(
                    M32(0:15) RD(10:13, 0) M32(16:31) RD(14:17, 4096) M32(32:47) RD(18:21, 8192) M32(48:63)
                    M32(64:79) RD(22:25, 12288) M32(80:95) RD(26:29, 16384) M32(96:111) RD(30:33, 20480) M32(112:127)
                    M32(128:143) RD(34:37, 24576) M32(144:159) RD(38:41, 28672) M32(160:175) RD(42:45, 32768) M32(176:191)
                    M32(192:207) RD(46:49, 36864) M32(208:223) RD(50:53, 40960) M32(224:239) RD(54:57, 45056) M32(240:255)
                    M32(0:15) RD(58:61, 49152) M32(16:31) RD(62:65, 53248) M32(32:47) RD(66:69, 57344) M32(48:63)
                    M32(64:79) RD(70:73, 61440) M32(80:95) RD(74:77, 512) M32(96:111) RD(78:81, 4608) M32(112:127)
                    M32(128:143) RD(82:85, 8704) M32(144:159) RD(86:89, 12800) M32(160:175) RD(90:93, 16896) M32(176:191)
                    M32(192:207) RD(94:97, 20992) M32(208:223) RD(98:101, 25088) M32(224:239) RD(102:105, 29184) M32(240:255)
                    "s_waitcnt lgkmcnt(0)" ::: CLOB);
            else
                asm volatile(
                    M32(0:15) M32(16:31) M32(32:47) M32(48:63) M32(64:79) M32(80:95) M32(96:111) M32(112:127) M32(128:143) M32(144:159) M32(160:175) M32(176:191) M32(192:207) M32(208:223) M32(224:239) M32(240:255)
                    M32(0:15) M32(16:31) M32(32:47) M32(48:63) M32(64:79) M32(80:95) M32(96:111) M32(112:127) M32(128:143) M32(144:159) M32(160:175) M32(176:191) M32(192:207) M32(208:223) M32(224:239) M32(240:255)
                    "s_nop 0" ::: CLOB);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 v2, a0" ::: "v2", "memory");
    float s;
    asm volatile("v_mov_b32 %0, v2" : "=v"(s));
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int MEM>
void run(const char* name, float* d) {
    const int iters = 20000, blocks = 256;
    auto kern = k<SHAPE, MEM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double flops = (double)blocks * 4 * iters * 64 * 16384.0;          // per iteration and wave: 32 K of a 128 x 128 wave tile
    const double clk_per_iter = ms * 1e-3 * 2.39e9 / iters;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  %6.0f clocks (at 2.39 GHz) per 32 K values, the MFMAs alone need 1024\n", name, ms, flops / ms / 1e9, clk_per_iter);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>("16x16x32: 64 MFMAs", d);
    run<0, 1>("16x16x32: 64 MFMAs + 24 ds_read_b128 between them", d);
    run<1, 0>("32x32x16: 32 MFMAs", d);
    run<1, 1>("32x32x16: 32 MFMAs + 24 ds_read_b128 between them", d);
    return 0;
}
