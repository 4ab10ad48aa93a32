// dec_pyr.h -- the ERB decoder's three transposed-conv stages + the mask head as ONE launch for small launches.
//
// convt3 / convt2 / convt1 (reference onnx_model/dpdfnet.py:361-364, layers.py:895-916: pathway conv on the encoder skip +
// previous stage, S sub-pixel depthwise k(1,3) convs interleaved along the band axis, pointwise 64 x 64 + BN + ReLU) and the
// mask head's 64 -> 1 k(1,3) contraction (dpdfnet.py:320-323, 364-366) are three dependent launches of 9-17 us each on the
// tail of a streaming hop's chain (64 x 48 kHz streams), three of 4-6 us for one 16 kHz stream.  Here -- the mirror image of
// enc_seg.h -- a workgroup owns R3 positions of the embedding-side input of ONE frame and walks the pyramid up through LDS:
//   U3 = relu(ps3 e3 + pb3) + demb     (R3 + halo positions)              -> convt3 -> d3
//   U2 = relu(ps2 e2 + pb2) + d3       (formed in convt3's epilogue)      -> convt2 -> d2
//   U1 = relu(ps1 e1 + pb1) + d2                                          -> convt1 -> d1
//   U0 = relu(ps0 e0 + pb0) + d1  -> s_k[row] = sum_c w0[c][k] U0[row][c]  (k = 0..2: the mask head's taps)
// d3 / d2 / d1 never reach HBM.  Each stage: depthwise straight into the MFMA A-operand registers (sub-pixel phase k = output
// row mod S picks the tap set, staged in LDS), pointwise on the matrix cores (wave w = output channels [16w, 16w + 16)).
// Halo rows (k = 3 taps on every level) are recomputed by the neighbouring workgroups; positions outside [0, F) are the
// convolutions' zero padding (U = 0 there).  The tap sums are reduced in MaskSumEpi's order (per lane over the four column
// tiles, then across the 16 lanes), so the results equal the gemm_rows forms bit for bit.  Segmented (48 kHz): the sums go to
// HBM ([rows][4]) and mask_df_kernel finishes the mask; WHOLE (16 kHz: one workgroup holds the frame) the mask is finished here.
#pragma once
#include "common.h"

struct DecPyrArgs {
    const float* e3; const float* demb; const float* e2; const float* e1; const float* e0;   // [BT][F3 | F3 | F2 | F1 | Ec][64]
    float* ssum;            // [BT * Ec][4] (segmented) or null
    float* m;               // [BT][Em] (WHOLE) or null
    const float* ps3; const float* pb3; const float* dw3; const float* pw3; const float* bs3;      // pathway scale / shift [64], depthwise [S][64][3],
    const float* ps2; const float* pb2; const float* dw2; const float* pw2; const float* bs2;      // pointwise fragments (pack_frag 64 x 64, NT 4), BN shift [64]
    const float* ps1; const float* pb1; const float* dw1; const float* pw1; const float* bs1;
    const float* ps0; const float* pb0; const float* w0; float bias0;                              // mask head: pathway, conv0_out [64][3] (BN folded), shift
    int BT, F3, F2, F1, Ec, Em;
};

// rows of the next level's skip tensor that pair with this stage's output rows (row = rt * 16 + 4 q + i, column 16 w + cl), fetched
// at kernel start: they were written by another launch, usually on another XCD, and take ~1-2 us to arrive
template <int NT>
struct DecPyrSkip { float v[NT][4]; };
template <int NT>
__device__ __forceinline__ void dec_pyr_prefetch(DecPyrSkip<NT>& P, const float* __restrict__ e, int fo0, int nrows, int Fout) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cl = lane & 15, q = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rt * 16 + 4 * q + i, g = fo0 + row;
            P.v[rt][i] = (row < nrows && g >= 0 && g < Fout) ? e[(size_t)g * 64 + 16 * w + cl] : 0.f;
        }
}

// one stage: NF source positions (U rows 0 .. NF + 1 = positions flo - 1 .. flo + NF) -> NF * S output rows (row ro = position
// flo * S + ro).  Rows [OFF, OFF + NNEXT) become the next level's U (formed here from the prefetched skip rows); LAST: all rows
// are the frame's own d1 rows and U0 is parked for the tap sums.
template <int S, int NF, int OFF, int NNEXT>
__device__ __forceinline__ void dec_pyr_stage(const float (*U)[68], const float (*Dw)[3][64], const float* __restrict__ pw, const float* __restrict__ bs,
                                              int fo0, int Fout, const DecPyrSkip<(NF * S + 15) / 16>& skip, const float* __restrict__ psn,
                                              const float* __restrict__ pbn, float (*Unext)[68]) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    float frag[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) frag[k] = pw[(size_t)(((k >> 2) * 4 + w) * 4 + (k & 3)) * 64 + lane];
    const float bv = bs[16 * w + cl], sn = psn[16 * w + cl], bn = pbn[16 * w + cl];
    constexpr int NROW = NF * S, NT = (NROW + 15) / 16;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int ro = rt * 16 + cl;
        float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] = 0.f;
        if (ro < NROW) {
            const int fl = ro / S, k = ro - fl * S;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 x = *(const float4*)&U[fl + j][16 * c + 4 * q];
                    const float4 d = *(const float4*)&Dw[k][j][16 * c + 4 * q];
                    av[4 * c + 0] = __builtin_fmaf(d.x, x.x, av[4 * c + 0]); av[4 * c + 1] = __builtin_fmaf(d.y, x.y, av[4 * c + 1]);
                    av[4 * c + 2] = __builtin_fmaf(d.z, x.z, av[4 * c + 2]); av[4 * c + 3] = __builtin_fmaf(d.w, x.w, av[4 * c + 3]);
                }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = mfma16(av[k], frag[k], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rt * 16 + 4 * q + i, g = fo0 + row, su = row - OFF;
            if (row < NROW && su >= 0 && su < NNEXT) {
                float v = 0.f;
                if (g >= 0 && g < Fout) v = relu_f(acc[i] + bv) + relu_f(__builtin_fmaf(sn, skip.v[rt][i], bn));
                Unext[su][16 * w + cl] = v;
            }
        }
    }
}

template <int S1, int S2, int S3, int R3, bool WHOLE>
__global__ __launch_bounds__(256) void dec_pyr_kernel(DecPyrArgs a) {
    // level-3 sources [a3 - C3, ...): C3 = ceil(2 / S3) positions below the first owned one feed the two halo rows of d3
    constexpr int C3 = (2 + S3 - 1) / S3, OFF3 = C3 * S3 - 2, NF3 = R3 + C3 + (S3 == 1 ? 1 : 0) + 1;
    constexpr int NF2 = R3 * S3 + 2, OFF2 = S2 - 1, NF1 = R3 * S3 * S2, NR0 = NF1 * S1;
    static_assert(OFF3 + NF2 + 2 <= NF3 * S3 && OFF2 + NF1 + 2 <= NF2 * S2, "halo rows");
    __shared__ __attribute__((aligned(16))) float U3[NF3 + 2][68];
    __shared__ __attribute__((aligned(16))) float U2[NF2 + 2][68];
    __shared__ __attribute__((aligned(16))) float U1[NF1 + 2][68];
    __shared__ __attribute__((aligned(16))) float U0[NR0][68];
    __shared__ __attribute__((aligned(16))) float Dw3[S3][3][64];
    __shared__ __attribute__((aligned(16))) float Dw2[S2][3][64];
    __shared__ __attribute__((aligned(16))) float Dw1[S1][3][64];
    __shared__ __attribute__((aligned(16))) float Ss[WHOLE ? NR0 : 1][4];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int bt = blockIdx.y;
    const int a3 = blockIdx.x * R3, A2 = a3 * S3, A1 = A2 * S2, A0 = A1 * S1;
    const int flo3 = a3 - C3, flo2 = A2 - 1, flo1 = A1;
    // skip rows of every stage's epilogue, up front
    DecPyrSkip<(NF3 * S3 + 15) / 16> k2; DecPyrSkip<(NF2 * S2 + 15) / 16> k1; DecPyrSkip<(NR0 + 15) / 16> k0;
    dec_pyr_prefetch(k2, a.e2 + (size_t)bt * a.F2 * 64, flo3 * S3, NF3 * S3, a.F2);
    dec_pyr_prefetch(k1, a.e1 + (size_t)bt * a.F1 * 64, flo2 * S2, NF2 * S2, a.F1);
    dec_pyr_prefetch(k0, a.e0 + (size_t)bt * a.Ec * 64, A0, NR0, a.Ec);
    // depthwise taps [phase][tap][channel]
    for (int i = tid; i < S3 * 192; i += 256) { const int k = i / 192, r = i - k * 192, c = r / 3, j = r - c * 3; Dw3[k][j][c] = a.dw3[i]; }
    for (int i = tid; i < S2 * 192; i += 256) { const int k = i / 192, r = i - k * 192, c = r / 3, j = r - c * 3; Dw2[k][j][c] = a.dw2[i]; }
    for (int i = tid; i < S1 * 192; i += 256) { const int k = i / 192, r = i - k * 192, c = r / 3, j = r - c * 3; Dw1[k][j][c] = a.dw1[i]; }
    {   // U3 = relu(ps3 e3 + pb3) + demb on positions flo3 - 1 .. flo3 + NF3
        const int c4 = (tid & 15) * 4, r16 = tid >> 4;
        const float4 s4 = *(const float4*)(a.ps3 + c4), b4 = *(const float4*)(a.pb3 + c4);
        const float* e3 = a.e3 + (size_t)bt * a.F3 * 64;
        const float* de = a.demb + (size_t)bt * a.F3 * 64;
        for (int r = r16; r < NF3 + 2; r += 16) {
            const int f = flo3 - 1 + r;
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f >= 0 && f < a.F3) {
                const float4 e = *(const float4*)(e3 + (size_t)f * 64 + c4), p = *(const float4*)(de + (size_t)f * 64 + c4);
                u.x = relu_f(__builtin_fmaf(s4.x, e.x, b4.x)) + p.x; u.y = relu_f(__builtin_fmaf(s4.y, e.y, b4.y)) + p.y;
                u.z = relu_f(__builtin_fmaf(s4.z, e.z, b4.z)) + p.z; u.w = relu_f(__builtin_fmaf(s4.w, e.w, b4.w)) + p.w;
            }
            *(float4*)&U3[r][c4] = u;
        }
    }
    __syncthreads();
    dec_pyr_stage<S3, NF3, OFF3, NF2 + 2>(U3, Dw3, a.pw3, a.bs3, flo3 * S3, a.F2, k2, a.ps2, a.pb2, U2);
    __syncthreads();
    dec_pyr_stage<S2, NF2, OFF2, NF1 + 2>(U2, Dw2, a.pw2, a.bs2, flo2 * S2, a.F1, k1, a.ps1, a.pb1, U1);
    __syncthreads();
    dec_pyr_stage<S1, NF1, 0, NR0>(U1, Dw1, a.pw1, a.bs1, A0, a.Ec, k0, a.ps0, a.pb0, U0);
    __syncthreads();
    // mask head taps: s_k[row] = sum_c w0[c][k] U0[row][c], MaskSumEpi's order
    {
        float w0[4], w1[4], w2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { const int c = nt * 16 + cl; w0[nt] = a.w0[c * 3 + 0]; w1[nt] = a.w0[c * 3 + 1]; w2[nt] = a.w0[c * 3 + 2]; }
        for (int rt = w; rt < (NR0 + 15) / 16; rt += 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rt * 16 + 4 * q + i;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                if (row < NR0) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float u = U0[row][nt * 16 + cl];
                        s0 = __builtin_fmaf(w0[nt], u, s0); s1 = __builtin_fmaf(w1[nt], u, s1); s2 = __builtin_fmaf(w2[nt], u, s2);
                    }
                }
                s0 = row16_allreduce_sum(s0); s1 = row16_allreduce_sum(s1); s2 = row16_allreduce_sum(s2);
                if (cl == 0 && row < NR0) {
                    if (WHOLE) *(float4*)&Ss[row][0] = make_float4(s0, s1, s2, 0.f);
                    else *(float4*)(a.ssum + ((size_t)bt * a.Ec + A0 + row) * 4) = make_float4(s0, s1, s2, 0.f);
                }
            }
    }
    if (WHOLE) {    // m[f] = sigmoid(bias + s_0[f - 1] + s_1[f] + s_2[f + 1])  (mask_fin_kernel)
        __syncthreads();
        if (tid < NR0 && tid < a.Ec) {
            float acc = Ss[tid][1];
            if (tid > 0) acc += Ss[tid - 1][0];
            if (tid + 1 < a.Ec) acc += Ss[tid + 1][2];
            a.m[(size_t)bt * a.Em + tid] = sigmoid_f(acc + a.bias0);
        }
    }
}
