// small_fused.h -- the grouped linears around the GRU-256 cells for SMALL launches (streaming hops, single clips), one
// workgroup per row, several layers per launch.
//
// With few rows every kernel of stage 2 is a dependent launch that costs ~3 us of host time and ~5 us on the GPU's critical
// path whatever it computes (hop time against the number of DPRNN blocks, tools/hop_nb_sweep.py: 225 us of a one-stream
// 16 kHz hop and 353 us of a 64-stream 48 kHz hop are NOT the DPRNN chain), and the grouped linears (reference
// onnx_model/layers.py:1035-1046; instances dpdfnet.py:233-241, 343-360, 486-519) are a few thousand MACs per row.  So
// they are chained inside one workgroup per row on the VALU -- activations through LDS, weights k-major ([G][Ig][Og]: the
// lanes of a group read consecutive outputs) straight from L2, two fp32 FMA chains over k (equal to the MFMA forms to rounding):
//   emb_in_kernel:  c1d (+ e3d) -> df_fc_emb (+ erb_fc_emb at 48 kHz | copy at 16 kHz) -> concat -> emb_gru.linear_in  (3 -> 1)
//   emb_out_kernel: h_enc -> emb_gru.linear_out -> { df_gru.linear_in, erb_dec.linear_in, df_skip }                     (4 -> 1)
//   dec_in_kernel:  h_erb -> erb_dec.linear_out (-> erb_fc_emb at 48 kHz)                                              (2 -> 1)
// and the DF decoder's `c = gru(emb) + skip(emb)` is added by df_out's A producer (SumA) instead of a kernel of its own.
#pragma once
#include "common.h"

struct GlRow { const float* w; const float* b; int G, Og, Ig; };       // weight [G][Ig][Og] (k-major) + bias [G*Og]

// out[o] = act(b[o] + sum_k w[o][k] * x[(o / Og) * Ig + k]) for o = tid, tid + 256, ...;  x in LDS, out to LDS or global
template <bool RELU>
__device__ __forceinline__ void gl_row(const GlRow& g, const float* x, float* out) {
    const int n = g.G * g.Og;
    for (int o = threadIdx.x; o < n; o += 256) {
        // lanes o, o + 1, ... of one group read consecutive floats of row k: whole 64-byte runs per 16 lanes
        const int gi = o / g.Og, oo = o - gi * g.Og;
        const float* w = g.w + (size_t)gi * g.Ig * g.Og + oo;
        const float* xi = x + gi * g.Ig;
        float acc0 = g.b[o], acc1 = 0.f;
        int k = 0;
        for (; k + 7 < g.Ig; k += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k + u) * g.Og];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { acc0 = __builtin_fmaf(wv[u], xi[k + u], acc0); acc1 = __builtin_fmaf(wv[u + 1], xi[k + u + 1], acc1); }
        }
        for (; k < g.Ig; ++k) acc0 = __builtin_fmaf(w[(size_t)k * g.Og], xi[k], acc0);
        const float acc = acc0 + acc1;
        out[o] = RELU ? fmaxf(acc, 0.f) : acc;
    }
}

struct EmbInArgs {
    const float* c1d; int n_c1;         // [M][n_c1]   DF-branch encoder output, (f, c) flattened
    const float* e3d; int n_e3;         // [M][n_e3]   ERB-branch encoder output
    GlRow df_fc, erb_fc, lin_in;        // erb_fc.w == null: e3d is copied (16 kHz: n_e3 = 512)
    float* out;                         // [M][256]
    int M;
};
__global__ __launch_bounds__(256) void emb_in_kernel(EmbInArgs a) {
    __shared__ __attribute__((aligned(16))) float X[3072];
    __shared__ __attribute__((aligned(16))) float E[1024];
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int i = tid * 4; i < a.n_c1; i += 1024) *(float4*)&X[i] = *(const float4*)(a.c1d + (size_t)r * a.n_c1 + i);
    if (!a.erb_fc.w) { for (int i = tid * 4; i < a.n_e3; i += 1024) *(float4*)&E[i] = *(const float4*)(a.e3d + (size_t)r * a.n_e3 + i); }
    __syncthreads();
    gl_row<true>(a.df_fc, X, E + 512);
    if (a.erb_fc.w) {
        __syncthreads();
        for (int i = tid * 4; i < a.n_e3; i += 1024) *(float4*)&X[i] = *(const float4*)(a.e3d + (size_t)r * a.n_e3 + i);
        __syncthreads();
        gl_row<true>(a.erb_fc, X, E);
    }
    __syncthreads();
    gl_row<true>(a.lin_in, E, a.out + (size_t)r * 256);
}

struct EmbOutArgs {
    const float* h;                     // [M][256]
    GlRow lin_out, df_in, ed_in, skip;
    float* emb; float* df_x; float* ed_x; float* skip_out;      // [M][512], [M][256] x 3
    int M;
};
__global__ __launch_bounds__(256) void emb_out_kernel(EmbOutArgs a) {
    __shared__ __attribute__((aligned(16))) float H[256];
    __shared__ __attribute__((aligned(16))) float E[512];
    const int r = blockIdx.x, tid = threadIdx.x;
    H[tid] = a.h[(size_t)r * 256 + tid];
    __syncthreads();
    gl_row<true>(a.lin_out, H, E);
    __syncthreads();
    a.emb[(size_t)r * 512 + tid] = E[tid]; a.emb[(size_t)r * 512 + 256 + tid] = E[256 + tid];
    gl_row<true>(a.df_in, E, a.df_x + (size_t)r * 256);
    gl_row<true>(a.ed_in, E, a.ed_x + (size_t)r * 256);
    gl_row<false>(a.skip, E, a.skip_out + (size_t)r * 256);
}

struct DecInArgs {
    const float* h;                     // [M][256]
    GlRow lin_out, erb_fc;              // erb_fc.w == null: 16 kHz (out = demb [M][512])
    float* demb; float* demb2; int n2;  // 48 kHz: demb2 [M][n2]
    int M;
};
__global__ __launch_bounds__(256) void dec_in_kernel(DecInArgs a) {
    __shared__ __attribute__((aligned(16))) float H[256];
    __shared__ __attribute__((aligned(16))) float E[512];
    const int r = blockIdx.x, tid = threadIdx.x;
    H[tid] = a.h[(size_t)r * 256 + tid];
    __syncthreads();
    if (!a.erb_fc.w) { gl_row<true>(a.lin_out, H, a.demb + (size_t)r * 512); return; }
    gl_row<true>(a.lin_out, H, E);
    __syncthreads();
    gl_row<true>(a.erb_fc, E, a.demb2 + (size_t)r * a.n2);
}
