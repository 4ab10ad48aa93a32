// dft2stage.h -- the real DFT of the analysis STFT / synthesis iSTFT of big launches (N = 960 at 48 kHz, 320 at 16 kHz) as TWO
// small matrix stages instead of one [N x (N + 2)] operand (reference package/src/dpdfnet/audio.py:104-136,
// onnx_model/dpdfnet.py:854-873, dpdfnet_48khz_hr.py:820-924 run an FFT there; the one-GEMM form cost 0.92 M MAC per 48 kHz frame,
// 13 + 5 ms per 256 x 10 s step at 23 % of the matrix peak, and 1.7 + 0.9 ms on the critical stream at 16 kHz).
//
// Cooley-Tukey with N = 32 x N2 (N2 = 30 / 10):  n = N2 n1 + n2,  k = k1 + 32 k2  (n1, k1 < 32;  n2, k2 < N2):
//   X[k1 + 32 k2] = sum_n2 e^{-2 pi i n2 (k1 + 32 k2) / N}  *  ( sum_n1 x[N2 n1 + n2] e^{-2 pi i n1 k1 / 32} )
// forward stage 1: rows (frame, n2), K = 32 samples, N = 32 complex k1             -> Y [frame][n2][k1][re,im]
// forward stage 2: per k1, rows = frames, K = N2 complex n2, N = N2/2 + 1 complex k2 -> spec[k1 + 32 k2], k <= N/2;
//                  the twiddle e^{-2 pi i n2 k1 / N} is folded into the per-k1 operand.
// The inverse is the mirror image (Hermitian extension, DC / Nyquist imaginary parts ignored as irfft does, 1 / N folded in):
// inverse stage A: per k1, rows = frames, K = N2 complex k2 (gathered from spec, k > N/2 = conj of N - k, the conjugation in the
//                  operand's signs), N = N2 complex n2                              -> U [frame][n2][k1][re,im]
// inverse stage B: rows (frame, n2), K = 32 complex k1, N = 32 real n1             -> x[N2 n1 + n2] * window
// 960: 122 k MAC per frame and direction instead of 923 k; 320: 36 k instead of 103 k.  Both intermediates make one HBM round
// trip (256 N2 bytes per frame).  All operands are packed on the host in MFMA B-fragment order (pack_frag); the strided ends
// (sample gather, k-interleaved spectrum rows) go through LDS tiles so that global traffic is whole lines.  Rows of a launch
// are the frames of one time chunk (RowSeg, gemm_rows.h).
#pragma once
#include "common.h"
#include "gemm_rows.h"

template <int N2>
struct Dft2Cfg {
    static constexpr int N = 32 * N2, F = N / 2 + 1, SPEC = 2 * F;   // transform length, bins, floats per spectrum row
    static constexpr int FR = 240 / N2;                               // frames per workgroup of the (frame, n2)-row stages: 240 rows = 15 tiles
    static constexpr int KC = (2 * N2 + 15) / 16;                     // 16-wide K chunks of the per-k1 stages (2 N2 real values)
    static constexpr int NK2 = N2 / 2 + 1;                            // k2 values that reach bins <= N / 2
    static constexpr int NT2 = (2 * NK2 + 15) / 16;                   // column tiles of forward stage 2
    static constexpr int NTA = (2 * N2 + 15) / 16;                    // column tiles of inverse stage A
    static constexpr int F2 = KC * NT2 * 256, FA = KC * NTA * 256;    // floats per k1 of the packed operands
};

struct Dft2Args {
    // forward input / inverse output
    const float* wav; int N; int T; int hop; const float* window; const int* lens;   // forward: clips [B][N samples]
    float* frames;                     // inverse: [B][T][win] windowed synthesis frames
    float* spec;                       // [B][T][F][2]: forward output / inverse input
    float* mid;                        // [M][N2][64] intermediate (Y or U)
    const float* frag_a;               // forward: stage 1 operand [2][4][4][64];   inverse: stage B operand [4][2][4][64]
    const float* frag_b;               // forward: stage 2 operands [32][F2];        inverse: stage A operands [32][FA]
    RowSeg seg; int M;                 // frames of this launch
};

// (the analysis stages dft2_fwd1 / dft2_fwd2 of rounds 3-5 are gone: the analysis DFT is float64, dft64.h)
// ---------------------------------------------------------------------------------------------------------------------------
// inverse stage A: workgroup = (16 frames, k1 half): per k1 the N2 bins k1 + 32 k2 of a frame (bins above N / 2 read their mirror;
// the conjugation sits in the operand) -> N2 complex U[n2]; gathered in LDS [frame][n2][k1][re,im], written as 128-byte runs.
template <int N2>
__global__ __launch_bounds__(256) void dft2_invA_kernel(Dft2Args g) {
    using C = Dft2Cfg<N2>;
    __shared__ __attribute__((aligned(16))) float us[16][N2][32];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * 16, h = blockIdx.y;
    const int fra = min(fr0 + cl, g.M - 1);
    const float* srow = g.spec + g.seg.map(fra) * C::SPEC;
    for (int j = 0; j < 4; ++j) {
        const int k1 = 16 * h + 4 * w + j;
        const float* bp = g.frag_b + (size_t)k1 * C::FA + lane;
        f32x4 acc[C::NTA];
#pragma unroll
        for (int nt = 0; nt < C::NTA; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C::KC; ++c) {
            const int k2a = 8 * c + 2 * q;                    // K index kk = (k2 = k2a + kb / 2, re / im = kb & 1)
            float2 p0 = make_float2(0.f, 0.f), p1 = p0;
            if (k2a < N2) { int k = k1 + 32 * k2a; if (k > C::N / 2) k = C::N - k; p0 = *(const float2*)(srow + 2 * k); }
            if (k2a + 1 < N2) { int k = k1 + 32 * (k2a + 1); if (k > C::N / 2) k = C::N - k; p1 = *(const float2*)(srow + 2 * k); }
            const float av[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int nt = 0; nt < C::NTA; ++nt) acc[nt] = mfma16(av[kb], bp[(size_t)((c * C::NTA + nt) * 4 + kb) * 64], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < C::NTA; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = nt * 16 + cl;                 // (n2 = col >> 1, re / im = col & 1), 2 N2 live columns
                if (col < 2 * N2) us[4 * q + i][col >> 1][(4 * w + j) * 2 + (col & 1)] = acc[nt][i];
            }
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * N2 * 8; idx += 256) {     // float4 pieces: [frame][n2][8]
        const int v4 = idx & 7, rest = idx >> 3, n2 = rest % N2, fl = rest / N2;
        const int fr = fr0 + fl;
        if (fr < g.M) *(float4*)(g.mid + ((size_t)fr * N2 + n2) * 64 + 32 * h + 4 * v4) = *(const float4*)&us[fl][n2][4 * v4];
    }
}

// inverse stage B: FR frames per workgroup, rows (frame, n2): K = 32 complex k1 -> 32 real samples x[N2 n1 + n2]; collected
// in an LDS tile [frame][N], written out times the synthesis window (the WindowStore epilogue's job in the one-GEMM form).
template <int N2>
__global__ __launch_bounds__(256) void dft2_invB_kernel(Dft2Args g) {
    using C = Dft2Cfg<N2>;
    __shared__ __attribute__((aligned(16))) float xo[C::FR][C::N];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * C::FR;
    float bf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bf[i] = g.frag_a[(size_t)i * 64 + lane];
    const size_t rows_total = (size_t)g.M * N2;
    for (int rt = w; rt < 15; rt += 4) {
        size_t grow = (size_t)fr0 * N2 + rt * 16 + cl;
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
            const int r = rt * 16 + 4 * q + i, f = r / N2, n2 = r - f * N2;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) xo[f][N2 * (nt * 16 + cl) + n2] = acc[nt][i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < C::FR * (C::N / 4); idx += 256) {        // float4 pieces of [frame][N]
        const int f = idx / (C::N / 4), n4 = (idx - f * (C::N / 4)) * 4, fr = fr0 + f;
        if (fr < g.M) {
            const float4 v = *(const float4*)&xo[f][n4], wv = *(const float4*)(g.window + n4);
            *(float4*)(g.frames + g.seg.map(fr) * C::N + n4) = make_float4(v.x * wv.x, v.y * wv.y, v.z * wv.z, v.w * wv.w);
        }
    }
}

template <int N2>
static inline void launch_dft2_inverse(hipStream_t st, const Dft2Args& a) {
    if (a.M <= 0) return;
    constexpr int FR = Dft2Cfg<N2>::FR;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(dft2_invA_kernel<N2>), dim3((a.M + 15) / 16, 2), dim3(256), 0, st, a);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(dft2_invB_kernel<N2>), dim3((a.M + FR - 1) / FR), dim3(256), 0, st, a);
}
static inline void launch_dft2_inverse(hipStream_t st, const Dft2Args& a, int) { launch_dft2_inverse<30>(st, a); }   // 48 kHz only (win 960 = 30 x 32; the callers check): at 16 kHz the one-GEMM form is the faster one
