// Micro-benchmark: 64 v_mfma_f32_16x16x32_bf16 per iteration on 64 independent accumulators, one wave per SIMD -- (a) every MFMA on the SAME two
// operand fragments, (b) on the up-projection's pattern (8 A fragments x 8 W fragments, v[0:31] x v[32:63]); operand data constant or random bf16.
// Does the matrix pipe's rate or the clock depend on which registers / what data the MFMAs read?  Tuning aid, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <int PATTERN>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ data, float* out, unsigned long long* stamps, int iters) {
    const unsigned* dp = data + (threadIdx.x & 63) * 64;
    unsigned long long c0, r0, c1, r1;
    asm volatile("global_load_dwordx4 v[0:3], %0, off offset:0\n\t"
                 "global_load_dwordx4 v[4:7], %0, off offset:16\n\t"
                 "global_load_dwordx4 v[8:11], %0, off offset:32\n\t"
                 "global_load_dwordx4 v[12:15], %0, off offset:48\n\t"
                 "global_load_dwordx4 v[16:19], %0, off offset:64\n\t"
                 "global_load_dwordx4 v[20:23], %0, off offset:80\n\t"
                 "global_load_dwordx4 v[24:27], %0, off offset:96\n\t"
                 "global_load_dwordx4 v[28:31], %0, off offset:112\n\t"
                 "global_load_dwordx4 v[32:35], %0, off offset:128\n\t"
                 "global_load_dwordx4 v[36:39], %0, off offset:144\n\t"
                 "global_load_dwordx4 v[40:43], %0, off offset:160\n\t"
                 "global_load_dwordx4 v[44:47], %0, off offset:176\n\t"
                 "global_load_dwordx4 v[48:51], %0, off offset:192\n\t"
                 "global_load_dwordx4 v[52:55], %0, off offset:208\n\t"
                 "global_load_dwordx4 v[56:59], %0, off offset:224\n\t"
                 "global_load_dwordx4 v[60:63], %0, off offset:240\n\t"
                  "s_waitcnt vmcnt(0)" :: "v"(dp) : CLOB);
    c0 = clock64(); r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0)
            asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], v[32:35], v[0:3], a[0:3]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[4:7], v[32:35], v[0:3], a[4:7]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[8:11], v[32:35], v[0:3], a[8:11]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[12:15], v[32:35], v[0:3], a[12:15]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[16:19], v[32:35], v[0:3], a[16:19]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[20:23], v[32:35], v[0:3], a[20:23]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[24:27], v[32:35], v[0:3], a[24:27]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[28:31], v[32:35], v[0:3], a[28:31]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[32:35], v[32:35], v[0:3], a[32:35]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[36:39], v[32:35], v[0:3], a[36:39]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[40:43], v[32:35], v[0:3], a[40:43]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[44:47], v[32:35], v[0:3], a[44:47]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[48:51], v[32:35], v[0:3], a[48:51]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[52:55], v[32:35], v[0:3], a[52:55]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[56:59], v[32:35], v[0:3], a[56:59]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[60:63], v[32:35], v[0:3], a[60:63]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[64:67], v[32:35], v[0:3], a[64:67]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[68:71], v[32:35], v[0:3], a[68:71]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[72:75], v[32:35], v[0:3], a[72:75]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[76:79], v[32:35], v[0:3], a[76:79]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[80:83], v[32:35], v[0:3], a[80:83]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[84:87], v[32:35], v[0:3], a[84:87]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[88:91], v[32:35], v[0:3], a[88:91]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[92:95], v[32:35], v[0:3], a[92:95]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[96:99], v[32:35], v[0:3], a[96:99]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[100:103], v[32:35], v[0:3], a[100:103]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[104:107], v[32:35], v[0:3], a[104:107]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[108:111], v[32:35], v[0:3], a[108:111]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[112:115], v[32:35], v[0:3], a[112:115]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[116:119], v[32:35], v[0:3], a[116:119]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[120:123], v[32:35], v[0:3], a[120:123]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[124:127], v[32:35], v[0:3], a[124:127]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[128:131], v[32:35], v[0:3], a[128:131]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[132:135], v[32:35], v[0:3], a[132:135]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[136:139], v[32:35], v[0:3], a[136:139]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[140:143], v[32:35], v[0:3], a[140:143]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[144:147], v[32:35], v[0:3], a[144:147]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[148:151], v[32:35], v[0:3], a[148:151]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[152:155], v[32:35], v[0:3], a[152:155]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[156:159], v[32:35], v[0:3], a[156:159]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[160:163], v[32:35], v[0:3], a[160:163]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[164:167], v[32:35], v[0:3], a[164:167]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[168:171], v[32:35], v[0:3], a[168:171]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[172:175], v[32:35], v[0:3], a[172:175]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[176:179], v[32:35], v[0:3], a[176:179]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[180:183], v[32:35], v[0:3], a[180:183]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[184:187], v[32:35], v[0:3], a[184:187]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[188:191], v[32:35], v[0:3], a[188:191]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[192:195], v[32:35], v[0:3], a[192:195]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[196:199], v[32:35], v[0:3], a[196:199]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[200:203], v[32:35], v[0:3], a[200:203]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[204:207], v[32:35], v[0:3], a[204:207]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[208:211], v[32:35], v[0:3], a[208:211]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[212:215], v[32:35], v[0:3], a[212:215]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[216:219], v[32:35], v[0:3], a[216:219]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[220:223], v[32:35], v[0:3], a[220:223]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[224:227], v[32:35], v[0:3], a[224:227]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[228:231], v[32:35], v[0:3], a[228:231]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[232:235], v[32:35], v[0:3], a[232:235]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[236:239], v[32:35], v[0:3], a[236:239]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[240:243], v[32:35], v[0:3], a[240:243]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[244:247], v[32:35], v[0:3], a[244:247]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[248:251], v[32:35], v[0:3], a[248:251]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[252:255], v[32:35], v[0:3], a[252:255]\n\t"
                "s_nop 0" ::: CLOB);
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], v[32:35], v[0:3], a[0:3]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[4:7], v[32:35], v[4:7], a[4:7]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[8:11], v[32:35], v[8:11], a[8:11]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[12:15], v[32:35], v[12:15], a[12:15]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[16:19], v[32:35], v[16:19], a[16:19]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[20:23], v[32:35], v[20:23], a[20:23]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[24:27], v[32:35], v[24:27], a[24:27]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[28:31], v[32:35], v[28:31], a[28:31]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[32:35], v[36:39], v[0:3], a[32:35]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[36:39], v[36:39], v[4:7], a[36:39]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[40:43], v[36:39], v[8:11], a[40:43]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[44:47], v[36:39], v[12:15], a[44:47]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[48:51], v[36:39], v[16:19], a[48:51]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[52:55], v[36:39], v[20:23], a[52:55]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[56:59], v[36:39], v[24:27], a[56:59]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[60:63], v[36:39], v[28:31], a[60:63]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[64:67], v[40:43], v[0:3], a[64:67]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[68:71], v[40:43], v[4:7], a[68:71]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[72:75], v[40:43], v[8:11], a[72:75]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[76:79], v[40:43], v[12:15], a[76:79]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[80:83], v[40:43], v[16:19], a[80:83]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[84:87], v[40:43], v[20:23], a[84:87]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[88:91], v[40:43], v[24:27], a[88:91]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[92:95], v[40:43], v[28:31], a[92:95]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[96:99], v[44:47], v[0:3], a[96:99]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[100:103], v[44:47], v[4:7], a[100:103]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[104:107], v[44:47], v[8:11], a[104:107]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[108:111], v[44:47], v[12:15], a[108:111]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[112:115], v[44:47], v[16:19], a[112:115]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[116:119], v[44:47], v[20:23], a[116:119]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[120:123], v[44:47], v[24:27], a[120:123]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[124:127], v[44:47], v[28:31], a[124:127]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[128:131], v[48:51], v[0:3], a[128:131]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[132:135], v[48:51], v[4:7], a[132:135]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[136:139], v[48:51], v[8:11], a[136:139]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[140:143], v[48:51], v[12:15], a[140:143]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[144:147], v[48:51], v[16:19], a[144:147]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[148:151], v[48:51], v[20:23], a[148:151]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[152:155], v[48:51], v[24:27], a[152:155]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[156:159], v[48:51], v[28:31], a[156:159]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[160:163], v[52:55], v[0:3], a[160:163]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[164:167], v[52:55], v[4:7], a[164:167]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[168:171], v[52:55], v[8:11], a[168:171]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[172:175], v[52:55], v[12:15], a[172:175]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[176:179], v[52:55], v[16:19], a[176:179]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[180:183], v[52:55], v[20:23], a[180:183]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[184:187], v[52:55], v[24:27], a[184:187]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[188:191], v[52:55], v[28:31], a[188:191]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[192:195], v[56:59], v[0:3], a[192:195]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[196:199], v[56:59], v[4:7], a[196:199]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[200:203], v[56:59], v[8:11], a[200:203]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[204:207], v[56:59], v[12:15], a[204:207]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[208:211], v[56:59], v[16:19], a[208:211]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[212:215], v[56:59], v[20:23], a[212:215]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[216:219], v[56:59], v[24:27], a[216:219]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[220:223], v[56:59], v[28:31], a[220:223]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[224:227], v[60:63], v[0:3], a[224:227]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[228:231], v[60:63], v[4:7], a[228:231]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[232:235], v[60:63], v[8:11], a[232:235]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[236:239], v[60:63], v[12:15], a[236:239]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[240:243], v[60:63], v[16:19], a[240:243]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[244:247], v[60:63], v[20:23], a[244:247]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[248:251], v[60:63], v[24:27], a[248:251]\n\t"
                "v_mfma_f32_16x16x32_bf16 a[252:255], v[60:63], v[28:31], a[252:255]\n\t"
                "s_nop 0" ::: CLOB);
    }
    c1 = clock64(); r1 = wall_clock64();
    float s;
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s) :: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0; }
}
template <int PATTERN>
void run(const char* name, const unsigned* data, float* d, unsigned long long* st) {
    const int iters = 40000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PATTERN>, dim3(blocks), dim3(256), 0, 0, data, d, st, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k<PATTERN>, dim3(blocks), dim3(256), 0, 0, data, d, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 40;
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] / 10.0;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s  %5.1f shader clocks per MFMA at the measured %.2f GHz\n", name, ms, (double)blocks * 4 * iters * 64 * 16384.0 / ms / 1e9,
           (double)h[0] / iters / 64.0, ghz);
}
int main() {
    unsigned *dc, *dr; float* d; unsigned long long* st;
    hipMalloc(&dc, 64 * 64 * 4); hipMalloc(&dr, 64 * 64 * 4); hipMalloc(&d, 256 * 256 * 4); hipMalloc(&st, 1024 * 16);
    unsigned hc[4096], hr[4096]; unsigned x = 12345u;
    for (int i = 0; i < 4096; ++i) {
        hc[i] = 0x3f803f80u;
        x = x * 1664525u + 1013904223u; const unsigned lo = 0x3c00u + ((x >> 8) & 0x07ffu) + ((x >> 30 & 1) << 15);     // bf16 in +-[2^-7, 2^-3)
        x = x * 1664525u + 1013904223u; const unsigned hi = 0x3c00u + ((x >> 8) & 0x07ffu) + ((x >> 30 & 1) << 15);
        hr[i] = lo | (hi << 16);
    }
    hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice); hipMemcpy(dr, hr, sizeof hr, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("same two fragments, constant data", dc, d, st);
        run<0>("same two fragments, random data", dr, d, st);
        run<1>("8 x 8 fragments (the up-projection's pattern), constant", dc, d, st);
        run<1>("8 x 8 fragments (the up-projection's pattern), random", dr, d, st);
    }
    return 0;
}
