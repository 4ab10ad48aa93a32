// fcln_gi_kernel<K1>: small-batch DPRNN glue -- the block's Linear + LayerNorm + residual (reference
// onnx_model/layers.py:178-181, 190-193) AND the input projection W_ih y of the NEXT recurrence in one launch.
//
// With few streams (single-hop streaming, one clip through enhance()) every kernel of the DPRNN chain is a dependent
// launch of 5-10 us however little it computes; the unfused chain of a block is
//   gi GEMM -> intra scan -> fc+LN GEMM -> gi GEMM -> inter scan -> fc+LN GEMM -> (next block) gi GEMM -> ...
// and each fc+LN GEMM is followed by a gi GEMM over the very rows it has just produced.  Both are row-local, so one
// workgroup per 16-row tile does them back to back: A tile [16][K1] -> fc on the matrix cores -> bias -> LayerNorm in the
// row-contiguous layout (DPP row reductions) -> + residual -> y (stored, and kept in LDS as the next A operand) ->
// gi = y . W_ih^T + b for NG groups of 64 columns (3: inter-band cell, 6: both directions of the intra-band GRU).
// Two launches per block fewer on the critical chain (64 streams x 48 kHz dpdfnet8, one hop: 1.34 -> docs/HISTORY.md).
// Operand packings are the ones gemm_rows uses (pack_frag: [k-chunk][col tile][kb][lane]); results equal the two-launch
// form to rounding (the same MFMA order per output).
#pragma once
#include "common.h"
#include "gru_scan.h"

struct FclnGiArgs {
    const float* a; int lda;          // fc input rows [M][K1]
    const float* res;                 // residual x [M][64]
    float* y;                         // [M][64]
    const float* fc_frag; const float* fc_bias; const float* ln_g; const float* ln_b;
    float* gi; int ldg;               // [M][64 * NG]
    const float* ih_frag; const float* ih_bias;
    int M;
};

template <int K1, int NG>
__global__ __launch_bounds__(256) void fcln_gi_kernel(FclnGiArgs g) {
    __shared__ __attribute__((aligned(16))) float As[16][K1 + 4];
    __shared__ __attribute__((aligned(16))) float Fs[16][68];
    __shared__ __attribute__((aligned(16))) float Ys[16][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    // ---- A tile
    for (int f = tid; f < 16 * (K1 / 4); f += 256) {
        const int r = f / (K1 / 4), c4 = f % (K1 / 4);
        int row = row0 + r; if (row >= g.M) row = g.M - 1;
        *(float4*)&As[r][4 * c4] = *(const float4*)(g.a + (size_t)row * g.lda + 4 * c4);
    }
    // residual piece of this thread (row-contiguous layout: row tid >> 4, columns 4 (tid & 15) .. +3), early
    const int rr = tid >> 4, rc4 = 4 * (tid & 15);
    const int grow = row0 + rr < g.M ? row0 + rr : g.M - 1;
    const float4 rv = *(const float4*)(g.res + (size_t)grow * 64 + rc4);
    // all operands up front: the kernel is a chain of short dependent phases, a load issued inside one costs its L2 latency there
    float ffc[K1 / 4], fih[NG * 16];
#pragma unroll
    for (int k = 0; k < K1 / 4; ++k) ffc[k] = g.fc_frag[(size_t)((((k >> 2) * 4 + w) * 4 + (k & 3)) * 64) + lane];
#pragma unroll
    for (int gp = 0; gp < NG; ++gp)
#pragma unroll
        for (int k = 0; k < 16; ++k) fih[gp * 16 + k] = g.ih_frag[(size_t)gp * 4096 + (size_t)((((k >> 2) * 4 + w) * 4 + (k & 3)) * 64) + lane];
    const float bfc = g.fc_bias[16 * w + cl];
    __syncthreads();
    // ---- fc: wave w owns output columns [16 w, 16 w + 16)
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;      // two chains: a dependent MFMA costs its full latency
#pragma unroll
        for (int c = 0; c < K1 / 16; c += 2) {
            const float4 a4 = *(const float4*)&As[cl][16 * c + 4 * q], b4 = *(const float4*)&As[cl][16 * c + 16 + 4 * q];
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                acc0 = mfma16(av[kb], ffc[c * 4 + kb], acc0);
                acc1 = mfma16(bv4[kb], ffc[(c + 1) * 4 + kb], acc1);
            }
        }
        f32x4 acc = acc0 + acc1;
        const float bv = bfc;
#pragma unroll
        for (int i = 0; i < 4; ++i) Fs[4 * q + i][16 * w + cl] = acc[i] + bv;
    }
    __syncthreads();
    // ---- LayerNorm + residual on the row-contiguous pieces
    {
        const float4 v = *(const float4*)&Fs[rr][rc4];
        const float mean = row16_allreduce_sum(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        const float s2 = row16_allreduce_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
        const float inv = rsqrtf(s2 * (1.0f / 64.0f) + 1e-5f);
        const float4 gg = *(const float4*)(g.ln_g + rc4), bb = *(const float4*)(g.ln_b + rc4);
        float4 o;
        o.x = rv.x + d0 * inv * gg.x + bb.x; o.y = rv.y + d1 * inv * gg.y + bb.y;
        o.z = rv.z + d2 * inv * gg.z + bb.z; o.w = rv.w + d3 * inv * gg.w + bb.w;
        *(float4*)&Ys[rr][rc4] = o;
        if (row0 + rr < g.M) *(float4*)(g.y + (size_t)(row0 + rr) * 64 + rc4) = o;
    }
    __syncthreads();
    // ---- gi = y . W_ih^T + b
    float4 y4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y4[c] = *(const float4*)&Ys[cl][16 * c + 4 * q];
    {   // wave w owns column tile w of every group: NG independent accumulator chains
        f32x4 acc[NG];
#pragma unroll
        for (int gp = 0; gp < NG; ++gp) acc[gp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float yv[4] = {y4[c].x, y4[c].y, y4[c].z, y4[c].w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int gp = 0; gp < NG; ++gp) acc[gp] = mfma16(yv[kb], fih[gp * 16 + c * 4 + kb], acc[gp]);
        }
#pragma unroll
        for (int gp = 0; gp < NG; ++gp) {
            const int col = gp * 64 + w * 16 + cl;
            const float bv = g.ih_bias[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * q + i;
                if (row < g.M) g.gi[(size_t)row * g.ldg + col] = acc[gp][i] + bv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dprnn_hop_glue_kernel<NEXT>: single-hop streaming (one frame per stream and call).  Everything of a DPRNN block that
// lies between two intra-band scans is row-local when there is ONE frame -- the inter-band "scan" is one GRUCell step per
// (stream, band) row on its carried hidden state -- so it runs as one launch per 16-row tile:
//   hcat (fwd | bwd intra outputs) -> fc_intra + LayerNorm + residual -> y1
//   -> inter-band GRUCell step (W_ih y1 + W_hh h, gates, h' -> state) -> fc_inter + LayerNorm + residual -> y2 (block output)
//   -> [NEXT] input projection of the next block's intra-band bi-GRU: gi = y2 . W_ih^T + b (both directions)
// instead of fc+LN GEMM, fused inter kernel and gi GEMM: the hop is a chain of dependent launches of 5-10 us each, and
// this takes two per block out of it.  Weights come straight from L2 (one step: nothing to keep resident).  Reference:
// onnx_model/layers.py:159-196 (DPRNN block), :278-302 (streaming inter GRUCell).
struct HopGlueArgs {
    const float* hcat;                // [M][128]
    const float* x;                   // block input [M][64] (residual of the intra half)
    float* y2;                        // block output [M][64]
    const float* fci_frag; const float* fci_b; const float* lni_g; const float* lni_b;
    const float* wfrag; const float* bias;      // inter-band GRU, packed as for gru64_scan_kernel (dir 0)
    float* hstate; long h_hi, h_lo; int rdiv;   // carried state of row r: hstate[(r / rdiv) * h_hi + (r % rdiv) * h_lo + unit]
    const float* fce_frag; const float* fce_b; const float* lne_g; const float* lne_b;
    float* gi; const float* ih_frag; const float* ih_bias;      // NEXT: [M][384]
    int M;
};

// (the four-wave form dprnn_hop_glue_kernel -- 240 MFMAs and 240 weight registers per wave -- was replaced by the eight-wave form below in
// round 3 and removed in round 6.)

// ---------------------------------------------------------------------------------------------
// dprnn_hop_glue8_kernel<NEXT>: the same launch on EIGHT waves per 16-row tile.  A hop's glue launch is a chain of short
// dependent phases on one workgroup per tile -- with four waves 240 MFMAs and 240 weight registers per wave, loaded up front;
// here wave (wc = w & 3, wk = w >> 2) owns output columns [16 wc, 16 wc + 16) of every phase and half of its work:
//   fc_intra: K half wk (16 MFMAs), halves summed by the LayerNorm pass;   inter-band GRU: wk = 0 the x part, wk = 1 the h part,
//   the h part's three accumulators handed over through LDS;   fc_inter: K half wk (8 MFMAs);   next gi: three of the 24
//   (direction x gate, column tile) pairs (48 MFMAs) -- 120 MFMAs and 120 weight registers per wave.
// Same operands, same packings; sums are taken in a different order than the four-wave form (equal to rounding).
#ifdef DPDF_PHASE_TRACE
#define DPDF_STAMP(i) do { if (NEXT && g.rdiv >= 48 && tile == 0 && threadIdx.x == 0) dpdf_trace_buf[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DPDF_STAMP(i) do {} while (0)
#endif
// HANDOFF (dprnn_hop_block.h): the tile's hcat rows come from scan workgroups of the SAME launch -- everything that does not depend
// on them (operands, residual rows, carried state) is fetched first, then the tile waits for its scans' flags and reads the rows with
// agent-scope loads
struct HopHandoff { const unsigned* flags; unsigned epoch; int nscan_x, Fp; int* err; unsigned* done; };   // done (last block of a stack, optional): tiles whose output is out -- a consumer on another stream waits for it instead of for an event
template <bool NEXT, bool HANDOFF>
__device__ __forceinline__ void dprnn_hop_glue8_body(const HopGlueArgs& g, int tile, const HopHandoff& ho) {
    __shared__ __attribute__((aligned(16))) float As[16][132];
    __shared__ __attribute__((aligned(16))) float Fs[2][16][68];
    __shared__ __attribute__((aligned(16))) float Ys[16][68];      // y1, later y2
    __shared__ __attribute__((aligned(16))) float Hs[16][68];      // h, later h'
    __shared__ __attribute__((aligned(16))) float Gs[4][3][4][64]; // h-part accumulators of the wk = 1 waves
    __shared__ __attribute__((aligned(16))) float Ln[4][64];       // LayerNorm gains / shifts (intra g, b; inter g, b): fetched at entry, read from LDS in their phases
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int wc = w & 3, wk = w >> 2;
    const int cl = lane & 15, q = lane >> 4;
    const int row0 = tile * 16;
    DPDF_STAMP(0);
    if (!HANDOFF) {
        const int r = tid >> 5, c4 = tid & 31;
        int row = row0 + r; if (row >= g.M) row = g.M - 1;
        *(float4*)&As[r][4 * c4] = *(const float4*)(g.hcat + (size_t)row * 128 + 4 * c4);
    }
    if (tid < 256) {                                                 // (a load issued inside a phase costs its L2 -- or, first touch on this XCD, memory -- latency there)
        const float* src = tid < 64 ? g.lni_g : (tid < 128 ? g.lni_b : (tid < 192 ? g.lne_g : g.lne_b));
        Ln[tid >> 6][tid & 63] = src[tid & 63];
    }
    const bool ln_role = tid < 256;                                  // row-contiguous pieces: row tid >> 4, columns 4 (tid & 15) .. + 3
    const int rr = (tid >> 4) & 15, rc4 = 4 * (tid & 15);
    const bool rok = row0 + rr < g.M;
    const int grow = rok ? row0 + rr : g.M - 1;
    float4 xres = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ln_role) {
        xres = *(const float4*)(g.x + (size_t)grow * 64 + rc4);
        const float* hp = g.hstate + (long)(grow / g.rdiv) * g.h_hi + (long)(grow % g.rdiv) * g.h_lo + rc4;
        *(float4*)&Hs[rr][rc4] = *(const float4*)hp;
    }
    // operands of this wave, all up front
    float wg[3][16];                                                 // wk = 0: W_ih, wk = 1: W_hh (units [16 wc, 16 wc + 16))
    {
        const float* wp = g.wfrag + ((size_t)wc * 2 + wk) * 3 * 16 * 64 + lane;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int j = 0; j < 16; ++j) wg[gt][j] = wp[(size_t)(gt * 16 + j) * 64];
    }
    const float b_r = g.bias[16 * wc + cl], b_z = g.bias[64 + 16 * wc + cl], b_in = g.bias[128 + 16 * wc + cl], b_hn = g.bias[192 + 16 * wc + cl];
    float ffi[16], ffe[8], fih[NEXT ? 48 : 1];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int kk = 16 * wk + k; ffi[k] = g.fci_frag[(size_t)((((kk >> 2) * 4 + wc) * 4 + (kk & 3)) * 64) + lane]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int kk = 8 * wk + k; ffe[k] = g.fce_frag[(size_t)((((kk >> 2) * 4 + wc) * 4 + (kk & 3)) * 64) + lane]; }
    if (NEXT) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int pr = w + 8 * p, gp = pr >> 2, tile = pr & 3;
#pragma unroll
            for (int k = 0; k < 16; ++k) fih[p * 16 + k] = g.ih_frag[(size_t)gp * 4096 + (size_t)((((k >> 2) * 4 + tile) * 4 + (k & 3)) * 64) + lane];
        }
    }
    const float bfi = g.fci_b[16 * wc + cl], bfe = g.fce_b[16 * wc + cl];
    float bih[3] = {0.f, 0.f, 0.f};
    if (NEXT) {
#pragma unroll
        for (int p = 0; p < 3; ++p) { const int pr = w + 8 * p; bih[p] = g.ih_bias[(pr >> 2) * 64 + (pr & 3) * 16 + cl]; }
    }
    // inter-band GRUCell step, one operand: wk = 0 W_ih . y1 (+ input-side biases), wk = 1 W_hh . h (+ b_hn) -- three accumulators
    f32x4 a0, a1, a2;
    auto gru_part = [&](const float (*src)[68]) {
        if (wk == 0) { a0 = (f32x4){b_r, b_r, b_r, b_r}; a1 = (f32x4){b_z, b_z, b_z, b_z}; a2 = (f32x4){b_in, b_in, b_in, b_in}; }
        else { a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; a2 = (f32x4){b_hn, b_hn, b_hn, b_hn}; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 x4 = *(const float4*)&src[cl][16 * c + 4 * q];
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                a0 = mfma16(xv[kb], wg[0][c * 4 + kb], a0);
                a1 = mfma16(xv[kb], wg[1][c * 4 + kb], a1);
                a2 = mfma16(xv[kb], wg[2][c * 4 + kb], a2);
            }
        }
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { Gs[wc][0][i][lane] = a0[i]; Gs[wc][1][i][lane] = a1[i]; Gs[wc][2][i][lane] = a2[i]; }
        }
    };
    if (HANDOFF) {
        // the h part of the cell step depends on the carried state only: its 48 matrix instructions per wave run while the scans do
        // (they share the SIMDs' matrix pipes with the x part otherwise: 3 072 of the phase's 3 600 cycles)
        __syncthreads();
        if (wk == 1) gru_part(Hs);
        // scan workgroup x = stream / 4 of either direction has published flag[dir * nscan_x + x] = epoch behind its last row
        const int last = (row0 + 15 < g.M ? row0 + 15 : g.M - 1);
        const int x_lo = (row0 / ho.Fp) >> 2, x_hi = (last / ho.Fp) >> 2, nx = x_hi - x_lo + 1;
        if (tid < 2 * nx) {
            const unsigned* f = ho.flags + (tid / nx) * ho.nscan_x + x_lo + tid % nx;
            unsigned spins = 0; bool dead = false;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ho.epoch) < 0) {
                if (cluster_spin_expired(spins, ho.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 512 * k, r = idx >> 7, c = idx & 127;
            int row = row0 + r; if (row >= g.M) row = g.M - 1;
            As[r][c] = __hip_atomic_load(g.hcat + (size_t)row * 128 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    DPDF_STAMP(1);
    auto layer_norm_res = [&](const float4 v, const float4 res, const float* gam, const float* bet) {
        const float mean = row16_allreduce_sum(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        const float s2 = row16_allreduce_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
        const float inv = rsqrtf(s2 * (1.0f / 64.0f) + 1e-5f);
        const float4 gg = *(const float4*)(gam + rc4), bb = *(const float4*)(bet + rc4);
        float4 o;
        o.x = res.x + d0 * inv * gg.x + bb.x; o.y = res.y + d1 * inv * gg.y + bb.y;
        o.z = res.z + d2 * inv * gg.z + bb.z; o.w = res.w + d3 * inv * gg.w + bb.w;
        return o;
    };
    // ---- fc_intra: K half wk
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            const float4 a4 = *(const float4*)&As[cl][64 * wk + 16 * c + 4 * q], b4 = *(const float4*)&As[cl][64 * wk + 16 * c + 16 + 4 * q];
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                acc0 = mfma16(av[kb], ffi[c * 4 + kb], acc0);
                acc1 = mfma16(bv4[kb], ffi[(c + 1) * 4 + kb], acc1);
            }
        }
        const float bv = wk ? 0.f : bfi;
#pragma unroll
        for (int i = 0; i < 4; ++i) Fs[wk][4 * q + i][16 * wc + cl] = acc0[i] + acc1[i] + bv;
    }
    __syncthreads();
    DPDF_STAMP(2);
    float4 y1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ln_role) {
        const float4 u = *(const float4*)&Fs[0][rr][rc4], v2 = *(const float4*)&Fs[1][rr][rc4];
        y1 = layer_norm_res(make_float4(u.x + v2.x, u.y + v2.y, u.z + v2.z, u.w + v2.w), xres, Ln[0], Ln[1]);
        *(float4*)&Ys[rr][rc4] = y1;
    }
    __syncthreads();
    DPDF_STAMP(3);
    // ---- inter-band GRUCell step: wk = 0 the x part, wk = 1 the h part
    float hn[4];
    {
        if (!HANDOFF || wk == 0) gru_part(wk ? Hs : Ys);
        __syncthreads();
        DPDF_STAMP(4);
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                hn[i] = gru64_cell(a0[i] + Gs[wc][0][i][lane], a1[i] + Gs[wc][1][i][lane], a2[i], Gs[wc][2][i][lane], Hs[4 * q + i][16 * wc + cl]);
        }
        __syncthreads();                      // every wave has read h
        DPDF_STAMP(5);
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Hs[4 * q + i][16 * wc + cl] = hn[i];
        }
    }
    __syncthreads();
    DPDF_STAMP(6);
    if (ln_role && rok) {      // h' back into the carried state (row-contiguous pieces)
        float* hp = g.hstate + (long)(grow / g.rdiv) * g.h_hi + (long)(grow % g.rdiv) * g.h_lo + rc4;
        *(float4*)hp = *(const float4*)&Hs[rr][rc4];
    }
    // ---- fc_inter on h': K half wk
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float4 a4 = *(const float4*)&Hs[cl][32 * wk + 4 * q], b4 = *(const float4*)&Hs[cl][32 * wk + 16 + 4 * q];
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            acc0 = mfma16(av[kb], ffe[kb], acc0);
            acc1 = mfma16(bv4[kb], ffe[4 + kb], acc1);
        }
        const float bv = wk ? 0.f : bfe;
#pragma unroll
        for (int i = 0; i < 4; ++i) Fs[wk][4 * q + i][16 * wc + cl] = acc0[i] + acc1[i] + bv;
    }
    __syncthreads();
    DPDF_STAMP(7);
    if (ln_role) {
        const float4 u = *(const float4*)&Fs[0][rr][rc4], v2 = *(const float4*)&Fs[1][rr][rc4];
        const float4 y2 = layer_norm_res(make_float4(u.x + v2.x, u.y + v2.y, u.z + v2.z, u.w + v2.w), y1, Ln[2], Ln[3]);
        if (rok) {
            float* yo = g.y2 + (size_t)(row0 + rr) * 64 + rc4;
            if (HANDOFF && !NEXT && ho.done) {          // read by a kernel that is already running on another stream: write-through
                __hip_atomic_store(yo + 0, y2.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(yo + 1, y2.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(yo + 2, y2.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(yo + 3, y2.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else *(float4*)yo = y2;
        }
        if (NEXT) *(float4*)&Ys[rr][rc4] = y2;   // the x-part waves read y1 out of Ys three barriers ago
    }
    if (HANDOFF && !NEXT && ho.done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ho.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (NEXT) {
        __syncthreads();
        DPDF_STAMP(8);
        float4 y4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) y4[c] = *(const float4*)&Ys[cl][16 * c + 4 * q];
        f32x4 acc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float yv[4] = {y4[c].x, y4[c].y, y4[c].z, y4[c].w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[p] = mfma16(yv[kb], fih[NEXT ? p * 16 + c * 4 + kb : 0], acc[p]);
        }
#ifdef DPDF_PHASE_TRACE
        if (g.rdiv >= 48 && tile == 0 && threadIdx.x == 0) { asm volatile("s_nop 7\n s_nop 7" ::: "memory"); dpdf_trace_buf[10] = __builtin_amdgcn_s_memtime() + (unsigned long long)(acc[0][0] == 12345.f); }
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int pr = w + 8 * p, col = (pr >> 2) * 64 + (pr & 3) * 16 + cl;
            const float bv = bih[p];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * q + i;
                if (row < g.M) g.gi[(size_t)row * 384 + col] = acc[p][i] + bv;
            }
        }
    }
    DPDF_STAMP(9);
}

template <bool NEXT>
__global__ __launch_bounds__(512) void dprnn_hop_glue8_kernel(HopGlueArgs g) {
    dprnn_hop_glue8_body<NEXT, false>(g, blockIdx.x, HopHandoff{nullptr, 0u, 0, 1, nullptr, nullptr});
}
