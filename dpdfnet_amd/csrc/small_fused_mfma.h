// small_fused_mfma.h -- the grouped linears around the GRU-256 cells for launches of up to a few hundred rows (streaming hops,
// single clips), chained per 16-row tile on the matrix cores, one launch per chain:
//   emb_in:  c1d (+ e3d) -> df_fc_emb (+ erb_fc_emb at 48 kHz | copy at 16 kHz) -> concat -> emb_gru.linear_in        (3 launches -> 1)
//   emb_out: h_enc -> emb_gru.linear_out -> { df_gru.linear_in, erb_dec.linear_in, df_skip }                           (4 -> 1)
//   dec_in:  h_erb -> erb_dec.linear_out (-> erb_fc_emb at 48 kHz)                                                    (2 -> 1)
// With few rows every kernel of stage 2 is a dependent launch that costs ~3 us of host time and ~5 us on the GPU's critical
// chain whatever it computes (tools/hop_nb_sweep.py: 225 us of a one-stream 16 kHz hop and 353 us of a 64-stream 48 kHz hop
// were NOT the DPRNN chain).  A workgroup is (64 rows, one group of the LAST layer of the chain): grouped linears are
// block-diagonal, so a group of the second layer needs only the few groups of the first layer that feed it; a wave owns 16
// rows, computes those first-layer tiles straight from global memory (the A fragment of lane (row, q) is four consecutive
// floats of the row), parks them in its own LDS rows and runs the second layer on them.  Weights in gemm_rows' fragment packing
// (pack_frag: [chunk][tile][kb][lane] per group), read once per wave.  The DF decoder's `c = gru(emb) + skip(emb)` is added by
// df_out's A producer (SumA) instead of a kernel of its own.  Measured (interleaved, tools/hop_ab2.py): 64 x 48 kHz streams
// -15 us, one 16 kHz stream -14 us, eight streams -18 us per hop.  Reference: onnx_model/layers.py:1035-1046; dpdfnet.py:233-241,
// 343-360, 486-519.
#pragma once
#include "common.h"

struct GlFrag { const float* frag; const float* bias; int G, Og, Ig, NT; };     // build_gl: frag of group g at g * ceil(Ig/16) * NT * 256

// acc[nt] (16 rows x 16 cols each) += A[16 rows][Ig] . W_g  with A rows at `arow` (this lane's row, + 4 q applied by the caller):
// arow points at element [row = lane & 15][k = 4 * (lane >> 4)] of the group's input slice; row pitch irrelevant (pointer per lane)
// NCH = ceil(Ig / 16) is a template parameter so that the loop unrolls completely: every A fragment and every weight fragment
// of the tile is requested before the first MFMA -- one memory latency per tile instead of one per 16-wide K chunk (a hop's
// kernels are chains of latencies: emb_in at 48 kHz is 4 x 5..6 chunks deep).
template <int NT, int NCH>
__device__ __forceinline__ void gl_load_w(const GlFrag& g, int grp, int lane, float (&wv)[NCH][NT][4]) {
    const float* wf = g.frag + (size_t)grp * NCH * NT * 256 + lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) wv[c][nt][kb] = wf[(size_t)((c * NT + nt) * 4 + kb) * 64];
}
template <int NCH>
__device__ __forceinline__ void gl_load_a(const float* arow, float4 (&a4)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) a4[c] = *(const float4*)(arow + 16 * c);
}
// the same through agent-scope loads: rows written by a kernel of another stream while this one was already running
__device__ __forceinline__ float4 ld4_agent(const float* p) {
    float4 v;
    v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
template <int NCH>
__device__ __forceinline__ void gl_load_a_agent(const float* arow, float4 (&a4)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) a4[c] = ld4_agent(arow + 16 * c);
}
template <int NT, int NCH>
__device__ __forceinline__ void gl_mma(const float4 (&a4)[NCH], const float (&wv)[NCH][NT][4], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float av[4] = {a4[c].x, a4[c].y, a4[c].z, a4[c].w};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[nt] = mfma16(av[kb], wv[c][nt][kb], acc[nt]);
    }
}
// A hop's kernels are chains of memory latencies (a launch finds nothing in its L2: every operand comes from the memory side,
// 1-2 us), so these kernels request EVERYTHING that does not depend on a previous phase at entry -- the weight fragments of the
// second layer and every bias too -- instead of where it is used (a bias loaded behind its matrix block cost one such latency
// per group: emb_in 10.3 -> ~6 us at 64 x 48 kHz streams).  Same products in the same order: bit-identical to the per-phase form.
template <int NT, int NCH, bool FROM_LDS>
__device__ __forceinline__ void gl_tile(const GlFrag& g, int grp, const float* arow, f32x4 (&acc)[NT], int lane) {
    float4 a4[NCH];
    gl_load_a<NCH>(arow, a4);
    float wv[NCH][NT][4];
    gl_load_w<NT, NCH>(g, grp, lane, wv);
    gl_mma<NT, NCH>(a4, wv, acc);
}

struct EmbInMArgs {
    const float* c1d; int n_c1; const float* e3d; int n_e3;
    GlFrag df_fc, erb_fc, lin_in;       // erb_fc.frag == null: the first 512 inputs of linear_in are e3d itself (16 kHz)
    float* out; int M;                  // [M][256]
    // optional: e3d comes from a kernel of ANOTHER stream that may still be running (the ERB stack's last one-launch block of a
    // streaming hop, dprnn_hop_block.h): the workgroups that read it wait until `wait_ctr` has reached `wait_target` (tiles out)
    const unsigned* wait_ctr; unsigned wait_target; int* err;
};
// grid (ceil(M / 64), 16 groups of linear_in): group g2 reads embin[64 g2, 64 g2 + 64) = four 16-wide first-layer groups
// first layer of emb_in for the four groups gg0 .. gg0 + 3 of one half (block-uniform), two groups per round of loads
template <int NCH, bool AGENT>
__device__ __forceinline__ void emb_in_first(const GlFrag& g, const float* xrow, int gg0, float (*Es)[68], int lane) {
    const int cl = lane & 15, q = lane >> 4;
    float b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b1[j] = g.bias[(gg0 + j) * 16 + cl];
#pragma unroll
    for (int jj = 0; jj < 4; jj += 2) {
        float4 a4[2][NCH]; float wv[2][NCH][1][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (AGENT) gl_load_a_agent<NCH>(xrow + (size_t)(gg0 + jj + u) * g.Ig + 4 * q, a4[u]);
            else gl_load_a<NCH>(xrow + (size_t)(gg0 + jj + u) * g.Ig + 4 * q, a4[u]);
            gl_load_w<1, NCH>(g, gg0 + jj + u, lane, wv[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            gl_mma<1, NCH>(a4[u], wv[u], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) Es[4 * q + i][16 * (jj + u) + cl] = relu_f(acc[0][i] + b1[jj + u]);
        }
    }
}
__global__ __launch_bounds__(256) void emb_in_mfma_kernel(EmbInMArgs a) {
    __shared__ __attribute__((aligned(16))) float Es[4][16][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int g2 = blockIdx.y;
    const int row0 = blockIdx.x * 64 + 16 * w;
    int row = row0 + cl; if (row >= a.M) row = a.M - 1;
    // second layer: operand and bias requested now
    float w2[4][1][4];
    gl_load_w<1, 4>(a.lin_in, g2, lane, w2);
    const float b2 = a.lin_in.bias[g2 * 16 + cl];
    const bool erb = g2 < 8;                              // groups 4 g2 .. 4 g2 + 3 of the first layer: 0..31 ERB half, 32..63 DF half
    if (erb && a.wait_ctr) {
        if (tid == 0) {
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(a.wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.wait_target) < 0) {
                if (++spins > (1u << 20)) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (erb && !a.erb_fc.frag) {                          // 16 kHz: copy e3d[row][64 g2 ..]
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* src = a.e3d + (size_t)row * a.n_e3 + 64 * g2 + 4 * i;
                *(float4*)&Es[w][cl][4 * i] = a.wait_ctr ? ld4_agent(src) : *(const float4*)src;
            }
        }
    } else if (erb) {                                                                                      // 2560 / 32 = 80 inputs per group
        if (a.wait_ctr) emb_in_first<5, true>(a.erb_fc, a.e3d + (size_t)row * a.n_e3, 4 * g2, Es[w], lane);
        else emb_in_first<5, false>(a.erb_fc, a.e3d + (size_t)row * a.n_e3, 4 * g2, Es[w], lane);
    } else emb_in_first<6, false>(a.df_fc, a.c1d + (size_t)row * a.n_c1, 4 * g2 - 32, Es[w], lane);        // 3072 / 32 = 96
    __syncthreads();
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    float4 a4[4];
    gl_load_a<4>(&Es[w][cl][4 * q], a4);
    gl_mma<1, 4>(a4, w2, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * q + i;
        if (r < a.M) a.out[(size_t)r * 256 + g2 * 16 + cl] = relu_f(acc[0][i] + b2);
    }
}

struct EmbOutMArgs {
    const float* h;                     // [M][256]
    GlFrag lin_out, df_in, ed_in, skip; // 16 x (16 -> 32), 8 x (64 -> 32), 16 x (32 -> 16), 16 x (32 -> 16)
    float* emb; float* df_x; float* ed_x; float* skip_out;
    int M;
};
// grid (ceil(M / 64), 8): block g owns linear_out groups 2g, 2g + 1 (emb columns [64 g, 64 g + 64)), the ERB decoder's
// linear_in / df_skip groups 2g, 2g + 1 and the DF decoder's linear_in group g
__global__ __launch_bounds__(256) void emb_out_mfma_kernel(EmbOutMArgs a) {
    __shared__ __attribute__((aligned(16))) float Es[4][16][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int row0 = blockIdx.x * 64 + 16 * w;
    int row = row0 + cl; if (row >= a.M) row = a.M - 1;
    // every operand and bias of both layers, requested at entry (see gl_tile)
    float4 a1[2][1]; float w1[2][1][2][4]; float b1[2][2];
    float we[2][2][1][4], ws[2][2][1][4], wd[4][2][4];
    float be[2], bs[2], bd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g1 = 2 * g + j;
        gl_load_a<1>(a.h + (size_t)row * 256 + 16 * g1 + 4 * q, a1[j]);
        gl_load_w<2, 1>(a.lin_out, g1, lane, w1[j]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b1[j][nt] = a.lin_out.bias[g1 * 32 + nt * 16 + cl];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g1 = 2 * g + j;
        gl_load_w<1, 2>(a.ed_in, g1, lane, we[j]);
        gl_load_w<1, 2>(a.skip, g1, lane, ws[j]);
        be[j] = a.ed_in.bias[g1 * 16 + cl]; bs[j] = a.skip.bias[g1 * 16 + cl];
    }
    gl_load_w<2, 4>(a.df_in, g, lane, wd);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bd[nt] = a.df_in.bias[g * 32 + nt * 16 + cl];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g1 = 2 * g + j;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        gl_mma<2, 1>(a1[j], w1[j], acc);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = relu_f(acc[nt][i] + b1[j][nt]);
                Es[w][4 * q + i][32 * j + 16 * nt + cl] = v;
                const int r = row0 + 4 * q + i;
                if (r < a.M) a.emb[(size_t)r * 512 + g1 * 32 + nt * 16 + cl] = v;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g1 = 2 * g + j;
        f32x4 ae[1] = {{0.f, 0.f, 0.f, 0.f}}, as[1] = {{0.f, 0.f, 0.f, 0.f}};
        float4 x2[2];
        gl_load_a<2>(&Es[w][cl][32 * j + 4 * q], x2);
        gl_mma<1, 2>(x2, we[j], ae);
        gl_mma<1, 2>(x2, ws[j], as);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + 4 * q + i;
            if (r < a.M) {
                a.ed_x[(size_t)r * 256 + g1 * 16 + cl] = relu_f(ae[0][i] + be[j]);
                a.skip_out[(size_t)r * 256 + g1 * 16 + cl] = as[0][i] + bs[j];
            }
        }
    }
    f32x4 ad[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 x4[4];
    gl_load_a<4>(&Es[w][cl][4 * q], x4);
    gl_mma<2, 4>(x4, wd, ad);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + 4 * q + i;
            if (r < a.M) a.df_x[(size_t)r * 256 + g * 32 + nt * 16 + cl] = relu_f(ad[nt][i] + bd[nt]);
        }
    }
}

struct DecInMArgs {
    const float* h; GlFrag lin_out, erb_fc;     // 16 x (16 -> 32); 48 kHz: 32 x (16 -> 80), NT = 5.  erb_fc.frag == null: demb is the output
    float* demb; float* demb2; int n2; int M;
    // grid.y rows 16 .. 31 (optional; the decoders in series on one stream): the DF decoder's df_out for group y - 16 --
    // coefs[b, 2 + t, 60 g + n] = tanh(W (gc + skip) + bias) + p[row, 60 g + n] (gemm_rows.h: SumA producer + DfOutEpi, same operand
    // order) -- it only shares the launch, not data, with the ERB decoder's linears: one dependent launch less per hop
    GlFrag df_out; const float* gc; const float* skipb; const float* p; float* coefs; int Tc;
};
// grid (ceil(M / 64), 16): block g owns linear_out group g (demb columns [32 g, 32 g + 32)) and erb_fc_emb groups 2g, 2g + 1
__global__ __launch_bounds__(256) void dec_in_mfma_kernel(DecInMArgs a) {
    __shared__ __attribute__((aligned(16))) float Es[4][16][36];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int row0 = blockIdx.x * 64 + 16 * w;
    int row = row0 + cl; if (row >= a.M) row = a.M - 1;
    if (g >= 16) {
        const int gd = g - 16;
        const float* wf = a.df_out.frag + (size_t)gd * 4 * 256 + lane;
        float wv[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) wv[nt][kb] = wf[(size_t)(nt * 4 + kb) * 64];
        const float4 x4 = *(const float4*)(a.gc + (size_t)row * 256 + 16 * gd + 4 * q), y4 = *(const float4*)(a.skipb + (size_t)row * 256 + 16 * gd + 4 * q);
        const float av[4] = {x4.x + y4.x, x4.y + y4.y, x4.z + y4.z, x4.w + y4.w};
        // bias and the pathway term of every output element, requested before the matrix block (see gl_tile)
        float bv[4], pv[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nt * 16 + cl < 60 ? nt * 16 + cl : 59;
            bv[nt] = a.df_out.bias[gd * 60 + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = row0 + 4 * q + i; if (r >= a.M) r = a.M - 1;
                pv[nt][i] = a.p[(size_t)r * 960 + gd * 60 + col];
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[nt] = mfma16(av[kb], wv[nt][kb], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nt * 16 + cl;
            if (col >= 60) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row0 + 4 * q + i;
                if (r >= a.M) continue;
                const int b = r / a.Tc, t = r - b * a.Tc;
                a.coefs[((size_t)b * (a.Tc + 2) + 2 + t) * 960 + gd * 60 + col] = tanh_f(acc[nt][i] + bv[nt]) + pv[nt][i];
            }
        }
        return;
    }
    // second layer (48 kHz): operands and biases requested at entry (see gl_tile)
    float w2[2][1][5][4], b2[2][5];
    if (a.erb_fc.frag) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gl_load_w<5, 1>(a.erb_fc, 2 * g + j, lane, w2[j]);
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) b2[j][nt] = a.erb_fc.bias[(2 * g + j) * 80 + nt * 16 + cl];
        }
    }
    {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float4 a1[1]; float w1[1][2][4];
        gl_load_a<1>(a.h + (size_t)row * 256 + 16 * g + 4 * q, a1);
        gl_load_w<2, 1>(a.lin_out, g, lane, w1);
        const float b1[2] = {a.lin_out.bias[g * 32 + cl], a.lin_out.bias[g * 32 + 16 + cl]};
        gl_mma<2, 1>(a1, w1, acc);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bv = b1[nt];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = relu_f(acc[nt][i] + bv);
                Es[w][4 * q + i][16 * nt + cl] = v;
                const int r = row0 + 4 * q + i;
                if (!a.erb_fc.frag && r < a.M) a.demb[(size_t)r * 512 + g * 32 + nt * 16 + cl] = v;
            }
        }
    }
    if (!a.erb_fc.frag) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g2 = 2 * g + j;
        f32x4 acc[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 x1[1];
        gl_load_a<1>(&Es[w][cl][16 * j + 4 * q], x1);
        gl_mma<5, 1>(x1, w2[j], acc);
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
            const float bv = b2[j][nt];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row0 + 4 * q + i;
                if (r < a.M) a.demb2[(size_t)r * a.n2 + g2 * 80 + nt * 16 + cl] = relu_f(acc[nt][i] + bv);
            }
        }
    }
}
