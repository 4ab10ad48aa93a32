// enc_seg.h -- the ERB encoder's four convolutions as ONE launch for small launches (streaming hops, short clips).
//
// erb_conv0 (dense 1 -> 64, k(3,3) over three frames of band features) and erb_conv1..3 (depthwise k(1,3) with stride S along
// the band axis + pointwise 64 x 64 + BN + ReLU) -- reference onnx_model/dpdfnet.py:74-95, 206-219 -- are four dependent
// launches of 8-13 us each in front of the ERB branch's DPRNN when a launch holds a few frames, and with 40 band positions at
// 48 kHz that branch is as long as the DF one: they are the head of the hop's critical path.  Here a workgroup owns R3 output
// positions of ONE frame and walks the whole pyramid for them through LDS: the e2 / e1 / e0 positions those R3 outputs depend
// on (k = 3 halos, recomputed by the neighbouring workgroups: 106 conv0 rows for 96 owned at 48 kHz), conv0 on the VALU from
// an LDS copy of the feature window, each separable layer as depthwise (straight into the MFMA A-operand registers) +
// pointwise on the matrix cores (wave w = output channels [16w, 16w + 16)).  Positions outside [0, F) are the convolutions'
// zero padding and are written as zeros into LDS; every e-tensor is stored to HBM by its owner (the decoder's skip inputs).
// Same arithmetic and summation order as conv0_erb_kernel + gemm_rows<DwConvA<S>> (explicit fmaf chains, fp32 MFMA 16x16x4 with k
// ascending): bit-identical results, so a clip enhanced alone equals the same clip inside a big batch.
#pragma once
#include "common.h"

struct ErbEncArgs {
    const float* feat;                  // [B][2 + Tc][E]
    float* e0; float* e1; float* e2; float* e3;     // [B*Tc][Ec | F1 | F2 | F3][64]
    const float* w0; const float* b0;   // erb_conv0 [64][9] (BN folded), shift [64]
    const float* dw1; const float* pw1; const float* bs1;      // depthwise [64][3], pointwise fragments (pack_frag 64 x 64, NT 4), BN shift
    const float* dw2; const float* pw2; const float* bs2;
    const float* dw3; const float* pw3; const float* bs3;
    int B, Tc, E, Ec, F1, F2, F3;
    float* gi; const float* ih; const float* ihb;      // (optional) the first DPRNN block's intra-band input projection: [B*Tc*F3][384], W_ih fragments, bias
};

// gi = W_ih x + b for one 16-row tile held in LDS (rows >= nrows are padding): 24 column tiles of 16 (group = direction * 3 + gate,
// 4 tiles each; fragments [6][chunk 4][nt 4][kb 4][lane]), 6 per wave; every load is issued before the first MFMA
__device__ __forceinline__ void enc_seg_gi(const float (*X)[68], const float* __restrict__ ih, const float* __restrict__ ihb, float* __restrict__ gi, int nrows) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    float frag[6][16], bv[6];
#pragma unroll
    for (int i6 = 0; i6 < 6; ++i6) {
        const int tt = w * 6 + i6, grp = tt >> 2, nt = tt & 3;
        const float* fp = ih + (size_t)grp * 4096 + (size_t)nt * 256 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) frag[i6][k] = fp[(size_t)(k >> 2) * 1024 + (k & 3) * 64];
        bv[i6] = ihb[grp * 64 + nt * 16 + cl];
    }
    float av[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 x = *(const float4*)&X[cl][16 * c + 4 * q];
        av[4 * c + 0] = x.x; av[4 * c + 1] = x.y; av[4 * c + 2] = x.z; av[4 * c + 3] = x.w;
    }
#pragma unroll
    for (int i6 = 0; i6 < 6; ++i6) {
        const int tt = w * 6 + i6, grp = tt >> 2, nt = tt & 3;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = mfma16(av[k], frag[i6][k], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            if (row < nrows) gi[(size_t)row * 384 + grp * 64 + nt * 16 + cl] = acc[i] + bv[i6];
        }
    }
}

// one separable layer over NROW local output rows: src row ro * S + j is input position (lo_out + ro) * S + j - 1
template <int S, int NROW, int NSRC>
__device__ __forceinline__ void enc_seg_layer(const float (*src)[68], const float* __restrict__ dw, const float* __restrict__ pw, const float* __restrict__ bs,
                                              int lo_out, int Fout, int own_lo, int own_hi, float* __restrict__ out, float (*dst)[68]) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    float frag[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) frag[k] = pw[(size_t)(((k >> 2) * 4 + w) * 4 + (k & 3)) * 64 + lane];
    float dq[16][3];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) dq[4 * c + k][j] = dw[(16 * c + 4 * q + k) * 3 + j];
    const float bv = bs[16 * w + cl];
    constexpr int NT = (NROW + 15) / 16;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int ro = rt * 16 + cl;
        float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] = 0.f;
        if (ro < NROW) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 x = *(const float4*)&src[ro * S + j][16 * c + 4 * q];
                    av[4 * c + 0] = __builtin_fmaf(dq[4 * c + 0][j], x.x, av[4 * c + 0]); av[4 * c + 1] = __builtin_fmaf(dq[4 * c + 1][j], x.y, av[4 * c + 1]);
                    av[4 * c + 2] = __builtin_fmaf(dq[4 * c + 2][j], x.z, av[4 * c + 2]); av[4 * c + 3] = __builtin_fmaf(dq[4 * c + 3][j], x.w, av[4 * c + 3]);
                }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = mfma16(av[k], frag[k], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rt * 16 + 4 * q + i, g = lo_out + row;
            const bool valid = row < NROW && g >= 0 && g < Fout;
            const float v = valid ? relu_f(acc[i] + bv) : 0.f;
            if (dst && row < NROW) dst[row][16 * w + cl] = v;
            if (valid && g >= own_lo && g < own_hi) out[(size_t)g * 64 + 16 * w + cl] = v;
        }
    }
    static_assert((NROW - 1) * S + 2 < NSRC, "source rows");
}

template <int S1, int S2, int S3, int R3>
__global__ __launch_bounds__(256) void erb_enc_seg_kernel(ErbEncArgs a) {
    // rows of each level this workgroup computes: from the lower k = 3 halo of its first output up to whichever is further, the
    // upper halo of its last output or the last row it OWNS (stride 3: the owned rows reach one past the halo).  H = end of the
    // range relative to the first owned row, N = row count including the lower halos
    constexpr int O2 = R3 * S3, H2 = ((R3 - 1) * S3 + 2 > O2) ? (R3 - 1) * S3 + 2 : O2, N2 = H2 + 1;
    constexpr int O1 = O2 * S2, H1 = ((H2 - 1) * S2 + 2 > O1) ? (H2 - 1) * S2 + 2 : O1, N1 = H1 + S2 + 1;
    constexpr int O0 = O1 * S1, H0 = ((H1 - 1) * S1 + 2 > O0) ? (H1 - 1) * S1 + 2 : O0, N0 = H0 + (S2 + 1) * S1 + 1;
    __shared__ __attribute__((aligned(16))) float E0[N0][68];
    __shared__ __attribute__((aligned(16))) float E1[N1][68];
    __shared__ __attribute__((aligned(16))) float E2[N2][68];
    __shared__ float Fw[3][N0 + 2];          // features of positions lo0 - 1 .. lo0 + N0, frames t .. t + 2 (zeros outside [0, Ec))
    __shared__ __attribute__((aligned(16))) float E3[16][68];
    static_assert(R3 <= 16, "one projection tile");
    const int tid = threadIdx.x;
    const int bt = blockIdx.y, b = bt / a.Tc, t = bt - b * a.Tc;
    const int a3 = blockIdx.x * R3, lo2 = a3 * S3 - 1, lo1 = lo2 * S2 - 1, lo0 = lo1 * S1 - 1;
    if (a.gi) for (int i = tid; i < (16 - R3) * 68; i += 256) (&E3[R3][0])[i] = 0.f;
    for (int i = tid; i < 3 * (N0 + 2); i += 256) {
        const int kt = i / (N0 + 2), r = i - kt * (N0 + 2), fi = lo0 - 1 + r;
        Fw[kt][r] = (fi >= 0 && fi < a.Ec) ? a.feat[((size_t)b * (a.Tc + 2) + t + kt) * a.E + fi] : 0.f;
    }
    const int c4 = (tid & 15) * 4, r16 = tid >> 4;
    float w0[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) w0[j][k] = a.w0[(c4 + j) * 9 + k];
    const float4 bias = *(const float4*)(a.b0 + c4);
    __syncthreads();
    // ---- erb_conv0 (as conv0_erb_kernel): local row r = band lo0 + r
    {
        float* e0 = a.e0 + (size_t)bt * a.Ec * 64;
        const int own_lo = a3 * S3 * S2 * S1, own_hi = own_lo + R3 * S3 * S2 * S1;
#pragma unroll 2
        for (int r = r16; r < N0; r += 16) {
            const int f = lo0 + r;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f >= 0 && f < a.Ec) {
                acc = bias;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int kf = 0; kf < 3; ++kf) {
                        const float x = Fw[kt][r + kf];      // band f + kf - 1
                        acc.x = __builtin_fmaf(w0[0][kt * 3 + kf], x, acc.x); acc.y = __builtin_fmaf(w0[1][kt * 3 + kf], x, acc.y);
                        acc.z = __builtin_fmaf(w0[2][kt * 3 + kf], x, acc.z); acc.w = __builtin_fmaf(w0[3][kt * 3 + kf], x, acc.w);
                    }
                acc.x = relu_f(acc.x); acc.y = relu_f(acc.y); acc.z = relu_f(acc.z); acc.w = relu_f(acc.w);
                if (f >= own_lo && f < own_hi) *(float4*)(e0 + (size_t)f * 64 + c4) = acc;
            }
            *(float4*)&E0[r][c4] = acc;
        }
    }
    __syncthreads();
    enc_seg_layer<S1, N1, N0>(E0, a.dw1, a.pw1, a.bs1, lo1, a.F1, a3 * S3 * S2, (a3 + R3) * S3 * S2, a.e1 + (size_t)bt * a.F1 * 64, E1);
    __syncthreads();
    enc_seg_layer<S2, N2, N1>(E1, a.dw2, a.pw2, a.bs2, lo2, a.F2, a3 * S3, (a3 + R3) * S3, a.e2 + (size_t)bt * a.F2 * 64, E2);
    __syncthreads();
    enc_seg_layer<S3, R3, N2>(E2, a.dw3, a.pw3, a.bs3, a3, a.F3, a3, a3 + R3, a.e3 + (size_t)bt * a.F3 * 64, a.gi ? E3 : nullptr);
    if (!a.gi) return;
    __syncthreads();
    enc_seg_gi(E3, a.ih, a.ihb, a.gi + ((size_t)bt * a.F3 + a3) * 384, a.F3 - a3 < R3 ? a.F3 - a3 : R3);
}

// ---------------------------------------------------------------------------------------------
// df_enc_seg_kernel: the DF branch's front end for small launches as ONE launch -- df_conv0 (the folded [18 -> 64] im2col GEMM,
// gemm_rows.h: Conv0DfA), df_conv1 (depthwise k(1,3) stride 2 + pointwise + BN + ReLU) and the first DPRNN block's intra-band
// input projection gi = W_ih c1 + b (both directions, 64 -> 384; run_dprnn's hoisted GEMM) -- reference
// onnx_model/dpdfnet.py:94-101, 221-234.  Three dependent launches (4-8 us each) at the head of the longest chain of a
// streaming hop become one.  A workgroup owns 16 c1 positions of one frame: 33 c0 rows (one halo row below) on the matrix
// cores straight from the feature window, c0 and c1 stored for their other consumers (the DF decoder's pathway conv, the
// DPRNN's residual), c1 handed to the projection through LDS.  Wave w = output columns [16w, 16w + 16) of the two convolutions
// and column tiles 6w .. 6w + 5 of the projection; k ascending as in the gemm_rows forms (bit-identical results).
struct DfEncArgs {
    const float* fs;        // feat_spec [B][2 + Tc][2][D]
    float* c0;              // [B][Tc + 4][D][64], frame t at index 4 + t
    float* c1;              // [B*Tc][Fd][64]
    float* gi;              // [B*Tc*Fd][384] or null
    const float* w0; const float* b0;       // df_conv0 fragments (pack_frag 32 x 64, NT 4), BN shift [64]
    const float* dw1; const float* pw1; const float* bs1;
    const float* ih; const float* ihb;      // W_ih fragments [6][chunk 4][nt 4][kb 4][lane], bias [6][64]
    int B, Tc, D, Fd;
    // Tc == 1 only (a streaming hop: the four older frames are the imported halo of c0, written before this launch): the DF decoder's
    // pathway conv df_convp -- grouped k(5,1) over the last five c0 frames + pointwise + BN + ReLU, folded into one [320 -> 10] matrix
    // (reference dpdfnet.py:424-431, 508-515) -- for the 32 bands this workgroup owns, as in df_ring_kernel (df_ring.h: same operand
    // order, so the same p): one launch less in stage 2, where df_out's epilogue then only ADDS p
    float* p; const float* cpfrag; const float* cpbias;     // p [B*Tc][D][10]; fragments [chunk 20][kb 4][lane]; bias [10]
};
// PCONV: the pathway conv is part of the launch (a.p != nullptr).
template <bool PCONV>
__global__ __launch_bounds__(256) void df_enc_seg_kernel(DfEncArgs a) {
    constexpr int R1 = 16, N0 = 2 * R1 + 1;
    __shared__ __attribute__((aligned(16))) float C0[N0][68];       // row r = band 2 a1 - 1 + r
    __shared__ __attribute__((aligned(16))) float C1[R1][68];
    __shared__ float Pz[2][4][64];                                  // pathway partial sums of the upper K half
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int bt = blockIdx.y, b = bt / a.Tc, t = bt - b * a.Tc;
    const int a1 = blockIdx.x * R1, lo0 = 2 * a1 - 1;
    // pathway conv operands, up front: wave (mt = row tile, kh = K half); chunk cg = 4 kt + cc <-> frame t - 4 + kt, channels 16 cc ..
    const int mt = w & 1, kh = w >> 1;
    float cp[PCONV ? 40 : 1]; float4 hist[PCONV ? 10 : 1];
    if (PCONV) {
#pragma unroll
        for (int i = 0; i < 40; ++i) cp[i] = a.cpfrag[(size_t)(40 * kh + i) * 64 + lane];
        const int band = 2 * a1 + mt * 16 + cl;
#pragma unroll
        for (int ci = 0; ci < 10; ++ci) {
            const int cg = 10 * kh + ci, kt = cg >> 2, cc = cg & 3;
            hist[ci] = (kt < 4 && band < a.D) ? *(const float4*)(a.c0 + (((size_t)b * (a.Tc + 4) + t + kt) * a.D + band) * 64 + cc * 16 + 4 * q)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // ---- df_conv0: column k = 16 c + 4 q + kb of the im2col row = frame t - 2 + kt, group g, band f + kb - 1 with kt = 2 c + (q >> 1), g = q & 1
    {
        float frag[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) frag[k] = a.w0[(size_t)(((k >> 2) * 4 + w) * 4 + (k & 3)) * 64 + lane];
        const float bv = a.b0[16 * w + cl];
        constexpr int NT = (N0 + 15) / 16;
        float av[NT][8];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            const int r = rt * 16 + cl, f = lo0 + r;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int kt = 2 * c + (q >> 1), g = q & 1;
                const float* p = a.fs + (((size_t)b * (a.Tc + 2) + t + kt) * 2 + g) * a.D + f;
                const bool row_ok = r < N0 && f >= 0 && f < a.D && kt < 3;
                av[rt][4 * c + 0] = (row_ok && f > 0) ? p[-1] : 0.f;
                av[rt][4 * c + 1] = row_ok ? p[0] : 0.f;
                av[rt][4 * c + 2] = (row_ok && f + 1 < a.D) ? p[1] : 0.f;
                av[rt][4 * c + 3] = 0.f;
            }
        }
        float* c0 = a.c0 + ((size_t)b * (a.Tc + 4) + 4 + t) * a.D * 64;
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = mfma16(av[rt][k], frag[k], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rt * 16 + 4 * q + i, f = lo0 + r;
                const bool valid = r < N0 && f >= 0 && f < a.D;
                const float v = valid ? relu_f(acc[i] + bv) : 0.f;
                if (r < N0) C0[r][16 * w + cl] = v;
                if (valid && f >= 2 * a1) c0[(size_t)f * 64 + 16 * w + cl] = v;
            }
        }
    }
    __syncthreads();
    // ---- pathway conv over frames t - 4 .. t of the owned bands (the newest frame from C0)
    if (PCONV) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < 10; ++ci) {
            const int cg = 10 * kh + ci, kt = cg >> 2, cc = cg & 3;
            // (two assignments, not `cond ? hist[ci] : *(lds)`: a conditional between two lvalues selects their ADDRESSES, a private
            // one and an LDS one, and that kept all of `hist` in scratch memory: 176 bytes per lane)
            float4 a4 = hist[ci];
            if (kt >= 4) a4 = *(const float4*)&C0[1 + mt * 16 + cl][cc * 16 + 4 * q];
            acc = mfma16(a4.x, cp[ci * 4 + 0], acc);
            acc = mfma16(a4.y, cp[ci * 4 + 1], acc);
            acc = mfma16(a4.z, cp[ci * 4 + 2], acc);
            acc = mfma16(a4.w, cp[ci * 4 + 3], acc);
        }
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Pz[mt][i][lane] = acc[i];
        }
        __syncthreads();
        if (kh == 0 && cl < 10) {
            const float cpb = a.cpbias[cl];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int band = 2 * a1 + mt * 16 + q * 4 + i;
                if (band < a.D) a.p[((size_t)bt * a.D + band) * 10 + cl] = relu_f(acc[i] + Pz[mt][i][lane] + cpb);
            }
        }
    }
    // ---- df_conv1 (stride 2): c1 position a1 + ro reads C0 rows 2 ro .. 2 ro + 2
    enc_seg_layer<2, R1, N0>(C0, a.dw1, a.pw1, a.bs1, a1, a.Fd, a1, a1 + R1, a.c1 + (size_t)bt * a.Fd * 64, C1);
    if (!a.gi) return;
    __syncthreads();
    enc_seg_gi(C1, a.ih, a.ihb, a.gi + ((size_t)bt * a.Fd + a1) * 384, a.Fd - a1 < R1 ? a.Fd - a1 : R1);
}
