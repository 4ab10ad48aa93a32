// gru_bf16x3.h -- OPT-IN precision mode for the GRU(64) scans: fp32 arithmetic emulated on the bf16 matrix pipe.
//
// Every fp32 operand is split into three bf16 limbs (a = a1 + a2 + a3, 24 significant bits) and each MFMA of the
// fp32 kernel becomes the six leading limb products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation:
//     a.b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)          (dropped terms < 2^-24 relative)
// Results agree with the fp32-MFMA kernels to fp32 rounding (tests: the same 2e-6 waveform tolerance as the default
// mode, 1.6e-7 max deviation on a 5e-2 signal in the scan micro-benchmark).  Why it is faster: the bf16 MFMA retires
// 16x the FLOPs of the fp32 one per cycle, so six products cost 6/16 of the matrix-pipe time, and -- unlike the fp32
// MFMA -- it overlaps with VALU work of other waves (tools/bf16_overlap.hip), so the gate math no longer comes out of
// the matrix pipe's budget.  Why it is NOT the default and never the headline dtype: it is a different instruction
// mix; its roofline is the bf16 peak / 6 (= 419 TFLOP/s fp32-equivalent), not the fp32 MFMA peak the headline is
// priced against.  Selected with dpdf_set_option("gru64_bf16x3", 1); the DPRNN then runs its plain (unfused) form:
// this scan kernel for both recurrences, fc + LayerNorm + residual as the fp32 gemm_rows passes.
#pragma once
#include "common.h"
#include "gru_scan.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__device__ __forceinline__ void split3(float a, __bf16& l1, __bf16& l2, __bf16& l3) {
    l1 = (__bf16)a; float r = a - (float)l1;
    l2 = (__bf16)r; r -= (float)l2;
    l3 = (__bf16)r;
}

// wlimb: [dir][wave 4][gate 3][kblock 4 (x 0..31, x 32..63, h 0..31, h 32..63)][limb 3][lane 64][8] bf16,
// element j of lane (q = lane>>4, cl = lane&15) = W_gate[col = 16 wave + cl][k = 32 (kblock & 1) + 8 q + j], exponent scales folded
// like the fp32 packing (build_gru64), so gru64_cell applies unchanged.  Same Gru64Args contract as gru64_scan_kernel.
__global__ __launch_bounds__(256, 2) void gru64_scan_bf16x3_kernel(Gru64Args a, const __bf16* wlimb) {
    __shared__ __attribute__((aligned(16))) __bf16 Hl[2][3][16][72];    // [buf][limb][row][k] (72: 16-byte aligned rows, conflict-light)
    __shared__ __attribute__((aligned(16))) __bf16 Xl[2][3][16][72];
    __shared__ __attribute__((aligned(16))) float Ho[16][68];           // fp32 h' for the row-contiguous global store
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = blockIdx.y, row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;
    bf16x8 wb[3][4][3];
    {
        const bf16x8* wp = (const bf16x8*)wlimb + ((size_t)(dir * 4 + w) * 3 * 4 * 3) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int l = 0; l < 3; ++l) wb[g][kb][l] = wp[(size_t)((g * 4 + kb) * 3 + l) * 64];
    }
    const float* bp = a.bias + (size_t)dir * 256 + 16 * w + cl;
    const float b_r = bp[0], b_z = bp[64], b_in = bp[128], b_hn = bp[192];
    // row addressing as in gru64_scan_kernel: wave-uniform tile base + 32-bit lane offsets
    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    const float* xbase = a.x + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    float* obase = a.out + (long)hi0 * a.o_hi + (long)lo0 * a.o_lo + dir * a.o_dir_off;
    const int srow = 4 * w + q, scol = 4 * cl;
    unsigned sx_off, so_off; bool so_ok;
    {
        int rs = row0 + srow;
        so_ok = rs < a.nrows;
        if (rs >= a.nrows) rs = a.nrows - 1;
        const int dh = rs / a.rdiv - hi0, dl = rs % a.rdiv - lo0;
        sx_off = (unsigned)((long)dh * a.x_hi + (long)dl * a.x_lo) + scol;
        so_off = (unsigned)((long)dh * a.o_hi + (long)dl * a.o_lo) + scol;
    }
    auto put_limbs = [&](__bf16 (*T)[16][72], int r, int c, const float (&v)[4]) __attribute__((always_inline)) {
        bf16x4 l1, l2, l3;
#pragma unroll
        for (int j = 0; j < 4; ++j) { __bf16 p1, p2, p3; split3(v[j], p1, p2, p3); l1[j] = p1; l2[j] = p2; l3[j] = p3; }
        *(bf16x4*)&T[0][r][c] = l1; *(bf16x4*)&T[1][r][c] = l2; *(bf16x4*)&T[2][r][c] = l3;
    };
    float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rc = row0 + q * 4 + i;
        if (rc >= a.nrows) rc = a.nrows - 1;
        const float hv = a.hstate ? a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] : 0.f;
        h_own[i] = hv;
        __bf16 p1, p2, p3; split3(hv, p1, p2, p3);
        Hl[1][0][q * 4 + i][16 * w + cl] = p1; Hl[1][1][q * 4 + i][16 * w + cl] = p2; Hl[1][2][q * 4 + i][16 * w + cl] = p3;
    }
    {
        const int p0 = dir ? a.nsteps - 1 : 0;
        const float4 v = *(const float4*)((xbase + (long)p0 * a.x_step) + sx_off);
        const float xv[4] = {v.x, v.y, v.z, v.w};
        put_limbs(Xl[0], srow, scol, xv);
    }
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < a.nsteps; ++s) {
        if (s > 0) {
            const int pp = dir ? a.nsteps - s : s - 1;
            const float4 hv4 = *(const float4*)&Ho[srow][scol];
            if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
        }
        const int sn = s + 1 < a.nsteps ? s + 1 : s;
        const float4 xnext = *(const float4*)((xbase + (long)(dir ? a.nsteps - 1 - sn : sn) * a.x_step) + sx_off);
        f32x4 ar = {b_r, b_r, b_r, b_r}, az = {b_z, b_z, b_z, b_z};
        f32x4 axn = {b_in, b_in, b_in, b_in}, ahn = {b_hn, b_hn, b_hn, b_hn};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 al[3];
#pragma unroll
            for (int l = 0; l < 3; ++l)
                al[l] = kb < 2 ? *(const bf16x8*)&Xl[buf][l][cl][32 * kb + 8 * q]
                               : *(const bf16x8*)&Hl[buf ^ 1][l][cl][32 * (kb - 2) + 8 * q];
            // the six leading limb products, small ones first
#define DPDF_LIMB6(acc, g)                                                                          \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[2], wb[g][kb][0], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], wb[g][kb][1], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][2], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], wb[g][kb][0], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][1], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][0], acc, 0, 0, 0);
            DPDF_LIMB6(ar, 0) DPDF_LIMB6(az, 1)
            if (kb < 2) { DPDF_LIMB6(axn, 2) } else { DPDF_LIMB6(ahn, 2) }
#undef DPDF_LIMB6
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
            h_own[i] = h;
            Ho[q * 4 + i][16 * w + cl] = h;
            __bf16 p1, p2, p3; split3(h, p1, p2, p3);
            Hl[buf][0][q * 4 + i][16 * w + cl] = p1; Hl[buf][1][q * 4 + i][16 * w + cl] = p2; Hl[buf][2][q * 4 + i][16 * w + cl] = p3;
        }
        {
            const float xv[4] = {xnext.x, xnext.y, xnext.z, xnext.w};
            put_limbs(Xl[buf ^ 1], srow, scol, xv);
        }
        __syncthreads();
        buf ^= 1;
    }
    if (a.nsteps > 0) {
        const int pp = dir ? 0 : a.nsteps - 1;
        const float4 hv4 = *(const float4*)&Ho[srow][scol];
        if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
    }
    if (a.hstate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rc = row0 + q * 4 + i;
            if (rc < a.nrows)
                a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] = h_own[i];
        }
    }
}
