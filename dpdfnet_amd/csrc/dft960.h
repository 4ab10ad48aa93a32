// dft960.h -- the 960-point real DFT of the 48 kHz models (analysis STFT / synthesis iSTFT of big launches) as TWO small matrix
// stages instead of one [960 x 962] operand (reference package/src/dpdfnet/audio.py:104-136, onnx_model/dpdfnet_48khz_hr.py:820-924
// run an FFT there; the one-GEMM form cost 0.92 M MAC per frame, 13 + 5 ms per 256 x 10 s step at 23 % of the matrix peak).
//
// Cooley-Tukey with 960 = 32 x 30:  n = 30 n1 + n2,  k = k1 + 32 k2  (n1, k1 < 32;  n2, k2 < 30):
//   X[k1 + 32 k2] = sum_n2 e^{-2 pi i n2 (k1 + 32 k2) / 960}  *  ( sum_n1 x[30 n1 + n2] e^{-2 pi i n1 k1 / 32} )
// forward stage 1: rows (frame, n2), K = 32 samples, N = 32 complex k1            -> Y [frame][n2][k1][re,im]   (61 k MAC / frame)
// forward stage 2: per k1, rows = frames, K = 30 complex n2, N = 16 complex k2    -> spec[k1 + 32 k2], k <= 480  (61 k MAC / frame);
//                  the twiddle e^{-2 pi i n2 k1 / 960} is folded into the per-k1 operand.
// The inverse is the mirror image (Hermitian extension, DC / Nyquist imaginary parts ignored as irfft does, 1 / 960 folded in):
// inverse stage A: per k1, rows = frames, K = 30 complex k2 (gathered from spec, k > 480 = conj of 960 - k, the conjugation in the
//                  operand's signs), N = 30 complex n2                             -> U [frame][n2][k1][re,im]
// inverse stage B: rows (frame, n2), K = 32 complex k1, N = 32 real n1            -> x[30 n1 + n2] * window
// 7.5 x fewer MACs; both intermediates make one HBM round trip (7.7 KB per frame).  All operands are packed on the host in MFMA
// B-fragment order (pack_frag); the strided ends (sample gather, k-interleaved spectrum rows) go through LDS tiles so that global
// traffic is whole lines.  Rows of a launch are the frames of one time chunk (RowSeg, gemm_rows.h).
#pragma once
#include "common.h"
#include "gemm_rows.h"

struct Dft960Args {
    // forward input / inverse output
    const float* wav; int N; int T; int hop; const float* window; const int* lens;   // forward: clips [B][N]
    float* frames;                     // inverse: [B][T][960] windowed synthesis frames
    float* spec;                       // [B][T][481][2]: forward output / inverse input
    float* mid;                        // [M][30][64] intermediate (Y or U)
    const float* frag_a;               // forward: stage 1 operand [2][4][4][64];   inverse: stage B operand [4][2][4][64]
    const float* frag_b;               // forward: stage 2 operands [32][4][2][4][64]; inverse: stage A operands [32][4][4][4][64]
    RowSeg seg; int M;                 // frames of this launch
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward stage 1: eight frames per workgroup; windowed samples (centre / reflect padding, per-clip lengths: as StftA) staged in
// LDS, 15 row tiles of (frame, n2) over the four waves.
__global__ __launch_bounds__(256) void dft960_fwd1_kernel(Dft960Args g) {
    __shared__ float xs[8][960];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * 8;
    for (int idx = tid; idx < 8 * 960; idx += 256) {
        const int f = idx / 960, n = idx - f * 960, fr = fr0 + f;
        float v = 0.f;
        if (fr < g.M) {
            const int b = fr / g.seg.Tc, t = g.seg.t0 + (fr - b * g.seg.Tc);
            const int nb_ = g.lens ? g.lens[b] : g.N, np_ = nb_ + 960;
            int j = t * g.hop + n - 480;
            if (j < 0) j = -j;
            if (j >= np_) j = 2 * (np_ - 1) - j;
            const bool live = !g.lens || t < 1 + np_ / g.hop;
            if (live && j >= 0 && j < nb_) v = g.wav[(size_t)b * g.N + j] * g.window[n];
        }
        xs[f][n] = v;
    }
    float bf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bf[i] = g.frag_a[(size_t)i * 64 + lane];
    __syncthreads();
    for (int rt = w; rt < 15; rt += 4) {
        const int row = rt * 16 + cl, f = row / 30, n2 = row - f * 30;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const float a = xs[f][30 * (16 * c + 4 * q + kb) + n2];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma16(a, bf[(c * 4 + nt) * 4 + kb], acc[nt]);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rt * 16 + 4 * q + i;
            if (fr0 + r / 30 < g.M) {
                float* o = g.mid + ((size_t)fr0 * 30 + r) * 64 + cl;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) o[nt * 16] = acc[nt][i];
            }
        }
    }
}

// forward stage 2: workgroup = (16 frames, k1 half); wave w takes k1 = 16 h + 4 w + j, j = 0..3.  A rows come straight from the
// intermediate (two 8-byte pieces per lane and K chunk; a 256-byte line (frame, n2) serves the 16 k1 of a half); results are
// gathered in an LDS tile [frame][k2][k1][re,im] and leave as contiguous 128-byte runs of spectrum bins.
__global__ __launch_bounds__(256) void dft960_fwd2_kernel(Dft960Args g) {
    __shared__ __attribute__((aligned(16))) float zs[16][16][32];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * 16, h = blockIdx.y;
    const int fra = min(fr0 + cl, g.M - 1);                   // A row of this lane (clamped; rows >= M are not stored)
    const float* yrow = g.mid + (size_t)fra * 30 * 64;
    for (int j = 0; j < 4; ++j) {
        const int k1 = 16 * h + 4 * w + j;
        const float* bp = g.frag_b + (size_t)k1 * 2048 + lane;
        float bfr[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) bfr[i] = bp[(size_t)i * 64];
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n2a = 8 * c + 2 * q;                    // K index kk = 16 c + 4 q + kb = (n2 = n2a + kb / 2, re / im = kb & 1)
            float2 p0 = make_float2(0.f, 0.f), p1 = p0;
            if (n2a < 30) p0 = *(const float2*)(yrow + (size_t)n2a * 64 + 2 * k1);
            if (n2a + 1 < 30) p1 = *(const float2*)(yrow + (size_t)(n2a + 1) * 64 + 2 * k1);
            const float av[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(av[kb], bfr[(c * 2 + nt) * 4 + kb], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = nt * 16 + cl;                 // (k2 = col >> 1, re / im = col & 1)
                zs[4 * q + i][col >> 1][(4 * w + j) * 2 + (col & 1)] = acc[nt][i];
            }
    }
    __syncthreads();
    // [16 frames][16 k2][16 k1] bins, 8 bytes each: a (frame, k2) is 16 consecutive bins k = 16 h + 32 k2 .. + 15
    for (int idx = tid; idx < 16 * 16 * 16; idx += 256) {
        const int k1l = idx & 15, k2 = (idx >> 4) & 15, fl = idx >> 8;
        const int fr = fr0 + fl, k = 16 * h + k1l + 32 * k2;
        if (fr < g.M && k <= 480)
            *(float2*)(g.spec + g.seg.map(fr) * 962 + 2 * k) = *(const float2*)&zs[fl][k2][2 * k1l];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// inverse stage A: workgroup = (16 frames, k1 half): per k1 the 30 bins k1 + 32 k2 of a frame (bins above 480 read their mirror;
// the conjugation sits in the operand) -> 30 complex U[n2]; gathered in LDS [frame][n2][k1][re,im], written as 128-byte runs.
__global__ __launch_bounds__(256) void dft960_invA_kernel(Dft960Args g) {
    __shared__ __attribute__((aligned(16))) float us[16][30][32];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * 16, h = blockIdx.y;
    const int fra = min(fr0 + cl, g.M - 1);
    const float* srow = g.spec + g.seg.map(fra) * 962;
    for (int j = 0; j < 4; ++j) {
        const int k1 = 16 * h + 4 * w + j;
        const float* bp = g.frag_b + (size_t)k1 * 4096 + lane;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k2a = 8 * c + 2 * q;                    // K index kk = (k2 = k2a + kb / 2, re / im = kb & 1)
            float2 p0 = make_float2(0.f, 0.f), p1 = p0;
            if (k2a < 30) { int k = k1 + 32 * k2a; if (k > 480) k = 960 - k; p0 = *(const float2*)(srow + 2 * k); }
            if (k2a + 1 < 30) { int k = k1 + 32 * (k2a + 1); if (k > 480) k = 960 - k; p1 = *(const float2*)(srow + 2 * k); }
            const float av[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma16(av[kb], bp[(size_t)((c * 4 + nt) * 4 + kb) * 64], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = nt * 16 + cl;                 // (n2 = col >> 1, re / im = col & 1), 60 live columns
                if (col < 60) us[4 * q + i][col >> 1][(4 * w + j) * 2 + (col & 1)] = acc[nt][i];
            }
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * 30 * 8; idx += 256) {     // float4 pieces: [frame][n2][8]
        const int v4 = idx & 7, rest = idx >> 3, n2 = rest % 30, fl = rest / 30;
        const int fr = fr0 + fl;
        if (fr < g.M) *(float4*)(g.mid + ((size_t)fr * 30 + n2) * 64 + 32 * h + 4 * v4) = *(const float4*)&us[fl][n2][4 * v4];
    }
}

// inverse stage B: eight frames per workgroup, rows (frame, n2): K = 32 complex k1 -> 32 real samples x[30 n1 + n2]; collected
// in an LDS tile [frame][960], written out times the synthesis window (the WindowStore epilogue's job in the one-GEMM form).
__global__ __launch_bounds__(256) void dft960_invB_kernel(Dft960Args g) {
    __shared__ __attribute__((aligned(16))) float xo[8][960];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * 8;
    float bf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bf[i] = g.frag_a[(size_t)i * 64 + lane];
    const size_t rows_total = (size_t)g.M * 30;
    for (int rt = w; rt < 15; rt += 4) {
        size_t grow = (size_t)fr0 * 30 + rt * 16 + cl;
        if (grow >= rows_total) grow = rows_total - 1;
        const float* urow = g.mid + grow * 64;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a4 = *(const float4*)(urow + 16 * c + 4 * q);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(av[kb], bf[(c * 2 + nt) * 4 + kb], acc[nt]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rt * 16 + 4 * q + i, f = r / 30, n2 = r - f * 30;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) xo[f][30 * (nt * 16 + cl) + n2] = acc[nt][i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 8 * 240; idx += 256) {         // float4 pieces of [frame][960]
        const int f = idx / 240, n4 = (idx - f * 240) * 4, fr = fr0 + f;
        if (fr < g.M) {
            const float4 v = *(const float4*)&xo[f][n4], wv = *(const float4*)(g.window + n4);
            *(float4*)(g.frames + g.seg.map(fr) * 960 + n4) = make_float4(v.x * wv.x, v.y * wv.y, v.z * wv.z, v.w * wv.w);
        }
    }
}

static inline void launch_dft960_forward(hipStream_t st, const Dft960Args& a) {
    if (a.M <= 0) return;
    hipLaunchKernelGGL(dft960_fwd1_kernel, dim3((a.M + 7) / 8), dim3(256), 0, st, a);
    hipLaunchKernelGGL(dft960_fwd2_kernel, dim3((a.M + 15) / 16, 2), dim3(256), 0, st, a);
}
static inline void launch_dft960_inverse(hipStream_t st, const Dft960Args& a) {
    if (a.M <= 0) return;
    hipLaunchKernelGGL(dft960_invA_kernel, dim3((a.M + 15) / 16, 2), dim3(256), 0, st, a);
    hipLaunchKernelGGL(dft960_invB_kernel, dim3((a.M + 7) / 8), dim3(256), 0, st, a);
}
