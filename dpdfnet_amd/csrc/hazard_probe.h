// hazard_probe.h -- PROBE BUILDS ONLY (-DDPDF_HAZARD_PROBE; tools/hazard_probe.py, tools/hazard_battery*.sh).
// df_apply_probe_kernel<TAPS, WAIT, LATE>: df_apply_kernel with its loads, waits and products in selectable forms, storing what it consumed
// and produced to a side buffer.  This is the instrument that showed the round-5 corruption to be ARITHMETIC: taps and history registers
// right (copies taken in front of the products equal the registers at the end of the kernel, every load long complete), the packed
// products v_pk_fma_f32 wrong in one half of a 16-lane group, their scalar twins on the same registers at the same time right.
#pragma once
// Probe build only (tools/hazard_probe.py): TAPS = 0 agent-scope dword loads (what ships), 1 plain loads (the
// compiler merges them into two under-aligned dwordx4 + one dwordx2 -- the failing form), 2 five plain, naturally aligned float2 loads,
// 3 ten plain dword loads; 2-5 are inline asm with one explicit wait (same timing structure): 4 = the merged form by hand, 5 = 4 with sc1.
template <int TAPS, int WAIT, int LATE = 0>
__global__ void df_apply_probe_kernel(DfApplyArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.B * a.Tc * a.F;
    if (idx >= total) return;
    const int f = (int)(idx % a.F);
    const size_t bt = idx / a.F;
    const int b = (int)(bt / a.Tc), t = (int)(bt - (size_t)b * a.Tc);
    const float* xb = a.xm + (((size_t)b * (a.Tc + 4) + t) * a.F + f) * 2;
    const size_t fs = (size_t)a.F * 2;
    float re, im;
    float cv[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float2 sv[5], snap[5] = {};
    if (f < a.D) {
        const float* c = a.coefs + (((size_t)b * (a.Tc + 2) + t) * a.D + f) * 10;
        if (WAIT >= 6) {             // (the pinned stream below loads the taps itself)
        } else if (TAPS == 0) {
#pragma unroll
            for (int j = 0; j < 10; ++j) cv[j] = ld_agent(c + j);
        } else if (TAPS == 1) {
#pragma unroll
            for (int j = 0; j < 10; ++j) cv[j] = c[j];
        } else if (TAPS == 2) {      // inline asm: the compiler can neither merge these nor count them -- one explicit wait behind all
            float2 v0, v1, v2, v3, v4;
            asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:8\n\tglobal_load_dwordx2 %2, %5, off offset:16\n\t"
                         "global_load_dwordx2 %3, %5, off offset:24\n\tglobal_load_dwordx2 %4, %5, off offset:32\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4) : "v"(c) : "memory");
            cv[0] = v0.x; cv[1] = v0.y; cv[2] = v1.x; cv[3] = v1.y; cv[4] = v2.x; cv[5] = v2.y; cv[6] = v3.x; cv[7] = v3.y; cv[8] = v4.x; cv[9] = v4.y;
        } else if (TAPS == 4 || TAPS == 5) {      // the merged form by hand (two under-aligned dwordx4 + one dwordx2), 5: the same with sc1 (agent scope)
            f32x4 q0, q1; float2 v4;
            if (TAPS == 4)
                asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx2 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(v4) : "v"(c) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx2 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(v4) : "v"(c) : "memory");
            cv[0] = q0[0]; cv[1] = q0[1]; cv[2] = q0[2]; cv[3] = q0[3]; cv[4] = q1[0]; cv[5] = q1[1]; cv[6] = q1[2]; cv[7] = q1[3]; cv[8] = v4.x; cv[9] = v4.y;
        } else if (TAPS == 6) {      // no tap loads at all in front of the history loads: fixed taps (timing probe; the output is compared with a run of the same form)
#pragma unroll
            for (int j = 0; j < 10; ++j) cv[j] = 0.1f * (float)(j + 1);
        } else {
            asm volatile("global_load_dword %0, %10, off\n\tglobal_load_dword %1, %10, off offset:4\n\tglobal_load_dword %2, %10, off offset:8\n\t"
                         "global_load_dword %3, %10, off offset:12\n\tglobal_load_dword %4, %10, off offset:16\n\tglobal_load_dword %5, %10, off offset:20\n\t"
                         "global_load_dword %6, %10, off offset:24\n\tglobal_load_dword %7, %10, off offset:28\n\tglobal_load_dword %8, %10, off offset:32\n\t"
                         "global_load_dword %9, %10, off offset:36\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3]), "=&v"(cv[4]), "=&v"(cv[5]), "=&v"(cv[6]), "=&v"(cv[7]), "=&v"(cv[8]), "=&v"(cv[9])
                         : "v"(c) : "memory");
        }
        float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
        if (WAIT == 0) {     // the loop as the shipped kernel has it: the compiler stages vmcnt(4) .. vmcnt(0) between the products
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float2 s = *(const float2*)(xb + n * fs);
                sv[n] = s;
                const float cr = cv[2 * n], ci = cv[2 * n + 1];
                rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
            }
        } else if (WAIT == 3) {
            // the staged waits by hand, and behind each a COPY of the register pair that wait is supposed to have made valid: the products
            // use the copies, the dump stores copies and originals -- a pair that differs was written by its load AFTER the wait let the wave go
            const float* p0 = xb; const float* p1 = xb + fs; const float* p2 = xb + 2 * fs; const float* p3 = xb + 3 * fs; const float* p4 = xb + 4 * fs;
            unsigned long long l0, l1, l2, l3, l4, c0, c1, c2, c3, c4;
            asm volatile("global_load_dwordx2 %0, %10, off\n\tglobal_load_dwordx2 %1, %11, off\n\tglobal_load_dwordx2 %2, %12, off\n\t"
                         "global_load_dwordx2 %3, %13, off\n\tglobal_load_dwordx2 %4, %14, off\n\t"
                         "s_waitcnt vmcnt(4)\n\tv_lshl_add_u64 %5, %0, 0, 0\n\ts_waitcnt vmcnt(3)\n\tv_lshl_add_u64 %6, %1, 0, 0\n\t"
                         "s_waitcnt vmcnt(2)\n\tv_lshl_add_u64 %7, %2, 0, 0\n\ts_waitcnt vmcnt(1)\n\tv_lshl_add_u64 %8, %3, 0, 0\n\t"
                         "s_waitcnt vmcnt(0)\n\tv_lshl_add_u64 %9, %4, 0, 0"
                         : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
                         : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
            sv[0] = __builtin_bit_cast(float2, l0); sv[1] = __builtin_bit_cast(float2, l1); sv[2] = __builtin_bit_cast(float2, l2);
            sv[3] = __builtin_bit_cast(float2, l3); sv[4] = __builtin_bit_cast(float2, l4);
            snap[0] = __builtin_bit_cast(float2, c0); snap[1] = __builtin_bit_cast(float2, c1); snap[2] = __builtin_bit_cast(float2, c2);
            snap[3] = __builtin_bit_cast(float2, c3); snap[4] = __builtin_bit_cast(float2, c4);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float2 s = snap[n];
                const float cr = cv[2 * n], ci = cv[2 * n + 1];
                rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
            }
        } else if (WAIT >= 6) {
            // The failing build's own instruction stream, register for register (physical registers pinned), from the tap loads to the sums:
            // WAIT 6 as it is; 7 with s_nop 7 behind every staged wait; 8 with a copy of the pair a wait has just released in front of the two
            // products that consume it; 9 with the first staged wait turned into vmcnt(0).
            unsigned long long r23 = (unsigned long long)c, r2829 = (unsigned long long)xb;
            float2 T0, T1, T2, T3, T4, L1, L2, L3, L4, A, A2, S0 = {}, S1 = {}, S2 = {}, S3 = {}, S4 = {};
            const unsigned long long stride = (unsigned long long)fs * sizeof(float);
#define DPDF_EXACT(W0, NOP, C0, C1, C2, C3, C4) \
                    "global_load_dwordx2 v[4:5], v[2:3], off\n\tglobal_load_dwordx2 v[6:7], v[2:3], off offset:8\n\t" \
                    "global_load_dwordx2 v[16:17], v[2:3], off offset:16\n\tglobal_load_dwordx2 v[18:19], v[2:3], off offset:24\n\t" \
                    "global_load_dwordx2 v[0:1], v[2:3], off offset:32\n\ts_waitcnt vmcnt(0)\n\t" \
                    "global_load_dwordx2 v[2:3], v[28:29], off\n\tv_lshl_add_u64 v[10:11], v[28:29], 0, %[st]\n\t" \
                    "global_load_dwordx2 v[8:9], v[10:11], off\n\tv_lshl_add_u64 v[12:13], v[10:11], 0, %[st]\n\t" \
                    "global_load_dwordx2 v[10:11], v[12:13], off\n\tv_lshl_add_u64 v[14:15], v[12:13], 0, %[st]\n\t" \
                    "global_load_dwordx2 v[12:13], v[14:15], off\n\tv_lshl_add_u64 v[14:15], v[14:15], 0, %[st]\n\t" \
                    "global_load_dwordx2 v[14:15], v[14:15], off\n\t" \
                    W0 NOP C0 \
                    "v_pk_fma_f32 v[26:27], v[4:5], v[2:3], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[28:29], v[4:5], v[2:3], 0 op_sel_hi:[0,1,0]\n\t" \
                    "s_waitcnt vmcnt(3)\n\t" NOP C1 \
                    "v_pk_fma_f32 v[26:27], v[8:9], v[6:7], v[26:27] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[28:29], v[8:9], v[6:7], v[28:29] op_sel_hi:[1,0,1]\n\t" \
                    "s_waitcnt vmcnt(2)\n\t" NOP C2 \
                    "v_pk_fma_f32 v[26:27], v[10:11], v[16:17], v[26:27] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[28:29], v[10:11], v[16:17], v[28:29] op_sel_hi:[1,0,1]\n\t" \
                    "s_waitcnt vmcnt(1)\n\t" NOP C3 \
                    "v_pk_fma_f32 v[26:27], v[12:13], v[18:19], v[26:27] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[28:29], v[12:13], v[18:19], v[28:29] op_sel_hi:[1,0,1]\n\t" \
                    "s_waitcnt vmcnt(0)\n\t" NOP C4 \
                    "v_pk_fma_f32 v[30:31], v[14:15], v[0:1], v[26:27] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[28:29], v[14:15], v[0:1], v[28:29] op_sel_hi:[1,0,1]\n\t" \
                    "s_nop 0\n\tv_pk_add_f32 v[26:27], v[28:29], v[30:31] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_sub_f32_e32 v26, v28, v31"
#define DPDF_EXACT_OPS \
                    : "={v[4:5]}"(T0), "={v[6:7]}"(T1), "={v[16:17]}"(T2), "={v[18:19]}"(T3), "={v[0:1]}"(T4), \
                      "+{v[2:3]}"(r23), "={v[8:9]}"(L1), "={v[10:11]}"(L2), "={v[12:13]}"(L3), "={v[14:15]}"(L4), \
                      "={v[26:27]}"(A), "+{v[28:29]}"(r2829), "={v[30:31]}"(A2), \
                      "={v[40:41]}"(S0), "={v[42:43]}"(S1), "={v[44:45]}"(S2), "={v[46:47]}"(S3), "={v[48:49]}"(S4) \
                    : [st] "s"(stride) : "memory"
#define DPDF_EXACT_OPS2 \
                    : "={v[4:5]}"(T0), "={v[6:7]}"(T1), "={v[16:17]}"(T2), "={v[18:19]}"(T3), "={v[0:1]}"(T4), \
                      "+{v[2:3]}"(r23), "={v[8:9]}"(L1), "={v[10:11]}"(L2), "={v[12:13]}"(L3), "={v[14:15]}"(L4), \
                      "={v[26:27]}"(A), "+{v[28:29]}"(r2829), "={v[30:31]}"(A2), \
                      "={v[40:41]}"(S0), "={v[42:43]}"(S1), "={v[44:45]}"(S2), "={v[46:47]}"(S3), "={v[48:49]}"(S4) \
                    : [st] "s"(stride) : "memory", "v32", "v33", "v34", "v35"
            // scalar twins of the packed products (same registers, same order of accumulation): A' = v[32:33] = (sum ri, sum ii), B' = v[34:35] = (sum rr, sum ir)
#define DPDF_SC0 "v_mul_f32 v32, v5, v2\n\tv_mul_f32 v33, v5, v3\n\tv_mul_f32 v34, v4, v2\n\tv_mul_f32 v35, v4, v3\n\t"
#define DPDF_SCN(L, T) "v_fmac_f32 v32, v" #L ", v" #T "+1\n\t"
#define DPDF_SC(LX, LY, TR, TI) "v_fmac_f32 v32, " LX ", " TI "\n\tv_fmac_f32 v33, " LY ", " TI "\n\tv_fmac_f32 v34, " LX ", " TR "\n\tv_fmac_f32 v35, " LY ", " TR "\n\t"
            if (WAIT == 11)        // the packed stream as it is, each pair of packed products followed by its four scalar twins; the twins' sums go out beside the packed ones
                asm volatile(DPDF_EXACT("s_waitcnt vmcnt(4)\n\t", "", "", DPDF_SC0, DPDF_SC("v8", "v9", "v6", "v7"), DPDF_SC("v10", "v11", "v16", "v17"), DPDF_SC("v12", "v13", "v18", "v19"))
                             "\n\t" DPDF_SC("v14", "v15", "v0", "v1") "v_sub_f32 v40, v34, v33\n\tv_add_f32 v41, v35, v32" DPDF_EXACT_OPS2);
            else if (WAIT == 10)   // the scalar twins ALONE in the places of the packed products (waits in the same places)
                asm volatile(
                    "global_load_dwordx2 v[4:5], v[2:3], off\n\tglobal_load_dwordx2 v[6:7], v[2:3], off offset:8\n\t"
                    "global_load_dwordx2 v[16:17], v[2:3], off offset:16\n\tglobal_load_dwordx2 v[18:19], v[2:3], off offset:24\n\t"
                    "global_load_dwordx2 v[0:1], v[2:3], off offset:32\n\ts_waitcnt vmcnt(0)\n\t"
                    "global_load_dwordx2 v[2:3], v[28:29], off\n\tv_lshl_add_u64 v[10:11], v[28:29], 0, %[st]\n\t"
                    "global_load_dwordx2 v[8:9], v[10:11], off\n\tv_lshl_add_u64 v[12:13], v[10:11], 0, %[st]\n\t"
                    "global_load_dwordx2 v[10:11], v[12:13], off\n\tv_lshl_add_u64 v[14:15], v[12:13], 0, %[st]\n\t"
                    "global_load_dwordx2 v[12:13], v[14:15], off\n\tv_lshl_add_u64 v[14:15], v[14:15], 0, %[st]\n\t"
                    "global_load_dwordx2 v[14:15], v[14:15], off\n\t"
                    "s_waitcnt vmcnt(4)\n\t" DPDF_SC0 "s_waitcnt vmcnt(3)\n\t" DPDF_SC("v8", "v9", "v6", "v7") "s_waitcnt vmcnt(2)\n\t" DPDF_SC("v10", "v11", "v16", "v17")
                    "s_waitcnt vmcnt(1)\n\t" DPDF_SC("v12", "v13", "v18", "v19") "s_waitcnt vmcnt(0)\n\t" DPDF_SC("v14", "v15", "v0", "v1")
                    "v_sub_f32 v26, v34, v33\n\tv_add_f32 v27, v35, v32\n\tv_mov_b32 v40, v26\n\tv_mov_b32 v41, v27" DPDF_EXACT_OPS2);
            else
            if (WAIT == 6) asm volatile(DPDF_EXACT("s_waitcnt vmcnt(4)\n\t", "", "", "", "", "", "") DPDF_EXACT_OPS);
            else if (WAIT == 7) asm volatile(DPDF_EXACT("s_waitcnt vmcnt(4)\n\t", "s_nop 7\n\t", "", "", "", "", "") DPDF_EXACT_OPS);
            else if (WAIT == 8) asm volatile(DPDF_EXACT("s_waitcnt vmcnt(4)\n\t", "", "v_lshl_add_u64 v[40:41], v[2:3], 0, 0\n\t", "v_lshl_add_u64 v[42:43], v[8:9], 0, 0\n\t",
                                                        "v_lshl_add_u64 v[44:45], v[10:11], 0, 0\n\t", "v_lshl_add_u64 v[46:47], v[12:13], 0, 0\n\t",
                                                        "v_lshl_add_u64 v[48:49], v[14:15], 0, 0\n\t") DPDF_EXACT_OPS);
            else asm volatile(DPDF_EXACT("s_waitcnt vmcnt(0)\n\t", "", "", "", "", "", "") DPDF_EXACT_OPS);
#undef DPDF_EXACT
#undef DPDF_EXACT_OPS
#undef DPDF_EXACT_OPS2
#undef DPDF_SC0
#undef DPDF_SC
#undef DPDF_SCN
            (void)A2;
            cv[0] = T0.x; cv[1] = T0.y; cv[2] = T1.x; cv[3] = T1.y; cv[4] = T2.x; cv[5] = T2.y; cv[6] = T3.x; cv[7] = T3.y; cv[8] = T4.x; cv[9] = T4.y;
            sv[0] = __builtin_bit_cast(float2, r23); sv[1] = L1; sv[2] = L2; sv[3] = L3; sv[4] = L4;
            snap[0] = S0; snap[1] = S1; snap[2] = S2; snap[3] = S3; snap[4] = S4;
            rr = A.x; ri = A.y;           // (re = rr - ii, im = ri + ir below: exact)
        } else if (WAIT == 4 || WAIT == 5) {
            // the compiler's own sequence of the failing build by hand -- address arithmetic between the loads, every later load writing into the
            // ADDRESS registers of an earlier one -- with a copy behind each staged wait (WAIT 4) or one full wait in front of all copies (WAIT 5)
            unsigned long long l0, l1, r1, r2, r3, c0, c1, c2, c3, c4;
            const unsigned long long stride = (unsigned long long)fs * sizeof(float);
            if (WAIT == 4)
            asm volatile("global_load_dwordx2 %0, %10, off\n\tv_lshl_add_u64 %2, %10, 0, %11\n\tglobal_load_dwordx2 %1, %2, off\n\t"
                         "v_lshl_add_u64 %3, %2, 0, %11\n\tglobal_load_dwordx2 %2, %3, off\n\t"
                         "v_lshl_add_u64 %4, %3, 0, %11\n\tglobal_load_dwordx2 %3, %4, off\n\t"
                         "v_lshl_add_u64 %4, %4, 0, %11\n\tglobal_load_dwordx2 %4, %4, off\n\t"
                         "s_waitcnt vmcnt(4)\n\tv_lshl_add_u64 %5, %0, 0, 0\n\ts_waitcnt vmcnt(3)\n\tv_lshl_add_u64 %6, %1, 0, 0\n\t"
                         "s_waitcnt vmcnt(2)\n\tv_lshl_add_u64 %7, %2, 0, 0\n\ts_waitcnt vmcnt(1)\n\tv_lshl_add_u64 %8, %3, 0, 0\n\t"
                         "s_waitcnt vmcnt(0)\n\tv_lshl_add_u64 %9, %4, 0, 0"
                         : "=&v"(l0), "=&v"(l1), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
                         : "v"(xb), "s"(stride) : "memory");
            else
            asm volatile("global_load_dwordx2 %0, %10, off\n\tv_lshl_add_u64 %2, %10, 0, %11\n\tglobal_load_dwordx2 %1, %2, off\n\t"
                         "v_lshl_add_u64 %3, %2, 0, %11\n\tglobal_load_dwordx2 %2, %3, off\n\t"
                         "v_lshl_add_u64 %4, %3, 0, %11\n\tglobal_load_dwordx2 %3, %4, off\n\t"
                         "v_lshl_add_u64 %4, %4, 0, %11\n\tglobal_load_dwordx2 %4, %4, off\n\t"
                         "s_waitcnt vmcnt(0)\n\tv_lshl_add_u64 %5, %0, 0, 0\n\tv_lshl_add_u64 %6, %1, 0, 0\n\t"
                         "v_lshl_add_u64 %7, %2, 0, 0\n\tv_lshl_add_u64 %8, %3, 0, 0\n\t"
                         "v_lshl_add_u64 %9, %4, 0, 0"
                         : "=&v"(l0), "=&v"(l1), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
                         : "v"(xb), "s"(stride) : "memory");
            sv[0] = __builtin_bit_cast(float2, l0); sv[1] = __builtin_bit_cast(float2, l1); sv[2] = __builtin_bit_cast(float2, r1);
            sv[3] = __builtin_bit_cast(float2, r2); sv[4] = __builtin_bit_cast(float2, r3);
            snap[0] = __builtin_bit_cast(float2, c0); snap[1] = __builtin_bit_cast(float2, c1); snap[2] = __builtin_bit_cast(float2, c2);
            snap[3] = __builtin_bit_cast(float2, c3); snap[4] = __builtin_bit_cast(float2, c4);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float2 s = snap[n];
                const float cr = cv[2 * n], ci = cv[2 * n + 1];
                rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
            }
        } else {
#pragma unroll
            for (int n = 0; n < 5; ++n) sv[n] = *(const float2*)(xb + n * fs);
            // ONE full wait behind all history loads instead of the staged waits; 2: + 32 idle cycles behind it
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]) :: "memory");
            if (WAIT >= 2) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]) :: "memory");
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float2 s = sv[n];
                const float cr = cv[2 * n], ci = cv[2 * n + 1];
                rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
            }
        }
        re = rr - ii; im = ri + ir;
    } else {
        const float2 s = *(const float2*)(xb + 2 * fs);
#pragma unroll
        for (int n = 0; n < 5; ++n) sv[n] = s;
        re = s.x; im = s.y;
    }
    re *= a.inv_wnorm; im *= a.inv_wnorm;
    const int tg = a.out_t0 + t;
    float* o = a.out + (size_t)b * a.out_clip_stride + ((size_t)tg * a.F + f) * 2;
    if (a.raw) {
        float nr = 0.f, ni = 0.f;
        if (tg >= 4) {
            const float* r = a.raw + (size_t)b * a.out_clip_stride + ((size_t)(tg - 4) * a.F + f) * 2;
            nr = r[0]; ni = r[1];
        }
        re = a.alpha * nr + a.beta * re;
        im = a.alpha * ni + a.beta * im;
    }
    o[0] = re; o[1] = im;
    if (a.dump) {            // stores only, behind everything the kernel does otherwise: [B][T][F][36] = 10 taps | 5 x (re, im) history | out | XCC | time | out recomputed late | history copies taken right behind the staged waits
        unsigned* q = a.dump + (((size_t)b * a.dump_T + tg) * a.F + f) * 36;
#pragma unroll
        for (int j = 0; j < 10; ++j) q[j] = __float_as_uint(cv[j]);
#pragma unroll
        for (int n = 0; n < 5; ++n) { q[10 + 2 * n] = __float_as_uint(sv[n].x); q[11 + 2 * n] = __float_as_uint(sv[n].y); }
        q[20] = __float_as_uint(re); q[21] = __float_as_uint(im);
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        q[22] = xcc; q[23] = (unsigned)__builtin_amdgcn_s_memtime();
        // the same sum once more from the SAME registers, everything long arrived (the compiler cannot merge it with the first: the
        // values pass through an opaque asm): if this one is right where the first was wrong, the first read registers a load had not filled yet
        float re2 = 0.f, im2 = 0.f;
        if (LATE && f < a.D && !a.raw) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]) :: "memory");
#pragma unroll
            for (int j = 0; j < 10; ++j) asm volatile("" : "+v"(cv[j]));
            float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
#pragma unroll
            for (int n = 0; n < 5; ++n) { rr += sv[n].x * cv[2 * n]; ii += sv[n].y * cv[2 * n + 1]; ri += sv[n].x * cv[2 * n + 1]; ir += sv[n].y * cv[2 * n]; }
            re2 = (rr - ii) * a.inv_wnorm; im2 = (ri + ir) * a.inv_wnorm;
        }
        q[24] = __float_as_uint(re2); q[25] = __float_as_uint(im2);
        if (WAIT >= 3) {
#pragma unroll
            for (int n = 0; n < 5; ++n) { q[26 + 2 * n] = __float_as_uint(snap[n].x); q[27 + 2 * n] = __float_as_uint(snap[n].y); }
        }
    }
}
