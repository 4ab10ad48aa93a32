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

// ---------------------------------------------------------------------------------------------
// gru64_epi_bf16x3_kernel<EPI>: the bf16x3 scan with the DPRNN block's Linear (+ LayerNorm + residual) fused in, the
// counterpart of gru64_epi_kernel (gru_scan.h) -- without it the opt-in mode pays 29 ms per step for the fc + LN GEMM
// passes over HBM.  The fc product of step s-1 shares the A limbs of h'(s-1) with the h-part of step s (12 more bf16
// MFMAs), lands in an LDS tile in C layout and is finished one barrier later by the row-contiguous lanes.
//   EPI 0  intra-band FORWARD:   writes pf(p) = W_fc[:, 0:64] hf(p)  (the forward half of fc_intra, no bias) instead of hf
//   EPI 2  intra-band BACKWARD:  y(p) = x(p) + LN(W_fc[:, 64:128] hb(p) + pf(p) + b)      (pf read back as fp32: no split)
//   EPI 1  inter-band:           y(s) = x(s) + LN(W_fc h'(s) + b)
// (the fp32 kernel keeps both fc halves in the backward kernel; here that would not fit 256 registers beside the limbs)
// The residual x(s-2) is rebuilt exactly from its three limbs, which stay in a 4-slot LDS ring.
struct Gru64EpiBf3Args {
    Gru64Args g;            // g.out unused
    const __bf16* wlimb;    // GRU weights (layout above)
    const __bf16* fclimb;   // [wave 4][kblock 2][limb 3][lane 64][8]: W_fc[col = 16 wave + cl][k = 32 kblock + 8 q + j] of the half that h' feeds
    const float* fc_bias; const float* ln_g; const float* ln_b;   // [64] (unused by EPI 0)
    const float* extra;     // EPI 2: pf tensor, addressed like x
    float* y;               // output, addressed like x
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gru64_epi_bf16x3_kernel(Gru64EpiBf3Args ea) {
    const Gru64Args& a = ea.g;
    __shared__ __attribute__((aligned(16))) __bf16 Hl[2][3][16][72];
    __shared__ __attribute__((aligned(16))) __bf16 Xl[4][3][16][72];          // ring: the limbs of x(s-2) are the residual
    __shared__ __attribute__((aligned(16))) float Ys[2][16][68];
    __shared__ __attribute__((aligned(16))) float Es[EPI == 2 ? 4 : 1][EPI == 2 ? 16 : 1][EPI == 2 ? 68 : 4];
    __shared__ __attribute__((aligned(16))) float Lp[3][64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = EPI == 2 ? 1 : 0;
    const int row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;
    bf16x8 wb[3][4][3], wfc[2][3];
    {
        const bf16x8* wp = (const bf16x8*)ea.wlimb + ((size_t)(dir * 4 + w) * 3 * 4 * 3) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int l = 0; l < 3; ++l) wb[g][kb][l] = wp[(size_t)((g * 4 + kb) * 3 + l) * 64];
        const bf16x8* fp = (const bf16x8*)ea.fclimb + ((size_t)w * 2 * 3) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int l = 0; l < 3; ++l) wfc[kb][l] = fp[(size_t)(kb * 3 + l) * 64];
    }
    const float* bp = a.bias + (size_t)dir * 256 + 16 * w + cl;
    const float b_r = bp[0], b_z = bp[64], b_in = bp[128], b_hn = bp[192];
    if (EPI != 0 && tid < 64) { Lp[0][tid] = ea.fc_bias[tid]; Lp[1][tid] = ea.ln_g[tid]; Lp[2][tid] = ea.ln_b[tid]; }
    if (EPI == 0 && tid < 64) { Lp[0][tid] = 0.f; Lp[1][tid] = 1.f; Lp[2][tid] = 0.f; }

    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    const float* xbase = a.x + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    const float* ebase = EPI == 2 ? ea.extra + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo : nullptr;
    float* ybase = ea.y + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    const int srow = 4 * w + q, scol = 4 * cl;
    unsigned sx_off; bool so_ok;
    {
        int rs = row0 + srow;
        so_ok = rs < a.nrows;
        if (rs >= a.nrows) rs = a.nrows - 1;
        sx_off = (unsigned)((long)(rs / a.rdiv - hi0) * a.x_hi + (long)(rs % a.rdiv - lo0) * a.x_lo) + scol;
    }
    auto put_limbs = [&](__bf16 (*T)[16][72], int r, int c, const float4& v4) __attribute__((always_inline)) {
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
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
    const int n = a.nsteps;
    auto pos_of = [&](int s) { s = s < n ? s : n - 1; return dir ? n - 1 - s : s; };
    {
        put_limbs(Xl[0], srow, scol, *(const float4*)((xbase + (long)pos_of(0) * a.x_step) + sx_off));
        if (EPI == 2) *(float4*)&Es[0][srow][scol] = *(const float4*)((ebase + (long)pos_of(0) * a.x_step) + sx_off);
    }
    __syncthreads();

    for (int s = 0; s < n + 2; ++s) {
        const int hb = s & 1;
        // ---- finalize step s-2 on the row-contiguous pieces
        if (s >= 2) {
            float4 yv = *(const float4*)&Ys[hb][srow][scol];                       // fc(s-2) (+ bias), written during step s-1
            if (EPI == 2) { const float4 pv = *(const float4*)&Es[(s - 2) & 3][srow][scol]; yv.x += pv.x; yv.y += pv.y; yv.z += pv.z; yv.w += pv.w; }
            float4 o;
            if (EPI == 0) {
                o = yv;
            } else {
                const bf16x4 r1 = *(const bf16x4*)&Xl[(s - 2) & 3][0][srow][scol], r2 = *(const bf16x4*)&Xl[(s - 2) & 3][1][srow][scol],
                             r3 = *(const bf16x4*)&Xl[(s - 2) & 3][2][srow][scol];
                const float rx = ((float)r1[0] + (float)r2[0]) + (float)r3[0], ry = ((float)r1[1] + (float)r2[1]) + (float)r3[1];
                const float rz = ((float)r1[2] + (float)r2[2]) + (float)r3[2], rw = ((float)r1[3] + (float)r2[3]) + (float)r3[3];
                const float mean = row16_allreduce_sum(yv.x + yv.y + yv.z + yv.w) * (1.0f / 64.0f);
                const float d0 = yv.x - mean, d1 = yv.y - mean, d2 = yv.z - mean, d3 = yv.w - mean;
                const float s2 = row16_allreduce_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
                const float inv = rsqrtf(s2 * (1.0f / 64.0f) + 1e-5f);
                const float4 gg = *(const float4*)&Lp[1][scol], bb = *(const float4*)&Lp[2][scol];
                o.x = rx + d0 * inv * gg.x + bb.x; o.y = ry + d1 * inv * gg.y + bb.y;
                o.z = rz + d2 * inv * gg.z + bb.z; o.w = rw + d3 * inv * gg.w + bb.w;
            }
            if (so_ok) *(float4*)((ybase + (long)pos_of(s - 2) * a.x_step) + sx_off) = o;
        }
        // ---- loads for step s+1
        const float4 xnext = *(const float4*)((xbase + (long)pos_of(s + 1) * a.x_step) + sx_off);
        float4 enext = xnext;
        if (EPI == 2) enext = *(const float4*)((ebase + (long)pos_of(s + 1) * a.x_step) + sx_off);
        f32x4 ar = {b_r, b_r, b_r, b_r}, az = {b_z, b_z, b_z, b_z};
        f32x4 axn = {b_in, b_in, b_in, b_in}, ahn = {b_hn, b_hn, b_hn, b_hn};
        f32x4 ay = {0.f, 0.f, 0.f, 0.f};
#define DPDF_LIMB6(acc, W)                                                                   \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[2], (W)[0], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], (W)[1], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], (W)[2], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], (W)[0], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], (W)[1], acc, 0, 0, 0);           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], (W)[0], acc, 0, 0, 0);
        if (s < n) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 al[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) al[l] = *(const bf16x8*)&Xl[s & 3][l][cl][32 * kb + 8 * q];
                DPDF_LIMB6(ar, wb[0][kb]) DPDF_LIMB6(az, wb[1][kb]) DPDF_LIMB6(axn, wb[2][kb])
            }
        }
        if (s <= n) {
            // h-part of step s and fc of step s-1 share the A limbs of h'(s-1)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 al[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) al[l] = *(const bf16x8*)&Hl[hb ^ 1][l][cl][32 * kb + 8 * q];
                DPDF_LIMB6(ar, wb[0][2 + kb]) DPDF_LIMB6(az, wb[1][2 + kb]) DPDF_LIMB6(ahn, wb[2][2 + kb])
                DPDF_LIMB6(ay, wfc[kb])
            }
        }
#undef DPDF_LIMB6
        if (s < n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
                h_own[i] = h;
                __bf16 p1, p2, p3; split3(h, p1, p2, p3);
                Hl[hb][0][q * 4 + i][16 * w + cl] = p1; Hl[hb][1][q * 4 + i][16 * w + cl] = p2; Hl[hb][2][q * 4 + i][16 * w + cl] = p3;
            }
        }
        if (s >= 1 && s <= n) {
            const float fb = Lp[0][16 * w + cl];
#pragma unroll
            for (int i = 0; i < 4; ++i) Ys[hb ^ 1][q * 4 + i][16 * w + cl] = ay[i] + fb;    // fc(s-1): read at step s+1 from Ys[(s+1)&1]
        }
        if (s + 1 < n) {
            put_limbs(Xl[(s + 1) & 3], srow, scol, xnext);
            if (EPI == 2) *(float4*)&Es[(s + 1) & 3][srow][scol] = enext;
        }
        __syncthreads();
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
