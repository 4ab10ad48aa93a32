// dprnn_hop_stack.h -- single-hop streaming: a whole DPRNN STACK (all nb blocks of the DF or of the ERB branch) as ONE persistent launch.
//
// A hop's DPRNN block was one launch (dprnn_hop_block.h): scans (48 dependent steps, 21.8 us) | hand-off by flag 2.6 us | glue tile 4.7 us,
// with the launch ramp (~2 us), the operand fetch (~2 us) and the drain (~2 us) around it: 35 us per block, 8 blocks per stack.  What is
// not recurrence in there is paid once per LAUNCH, so here the launch covers the stack:
//   * workgroups [0, 2 nx): the intra-band scans, one per (4 streams, direction), persistent over the blocks.  h' leaves as 8-byte
//     {epoch, value} granules (one agent-scope store per lane and step, as before -- nothing is waited for): a row is valid for a reader
//     the moment its epochs are, so nobody publishes "the scan is done";
//   * workgroups [2 nx, 2 nx + S): the glue of ONE stream each, persistent over the blocks.  A stream's Fp band positions are three 16-row
//     tiles ordered by WHEN the two scans have passed them: A = the middle positions (both directions are through after ~2/3 of the
//     steps), B = the next ring, C = the outer positions (complete with the last step).  Tiles A and B are computed UNDER the scans; only
//     C -- the same 16-row tile as before -- lies behind the last step.  The block output never leaves the workgroup: it is the next
//     block's residual input of the same rows (LDS); only the input projection of the next block's scans goes through memory
//     (agent-scope stores, one flag per stream and block), and the last block's output rows;
//   * operands of block n + 1 (64 weight registers per scan lane, 120 per glue lane) are fetched while the other role works on block n.
// MEASURED (64 x dpdfnet8_48khz_hr, s_memrealtime stamps inside the kernel: tools/stack_trace.py, profiles/r6_stack_kernel_timeline.txt): a block
// takes 34.7 us -- what the per-block launches take.  The scans run their 48 steps in 21.2 us; tile A's rows are valid at 15.6 us and the
// tile is through at 20.3; tile B 20.5 -> 26.8; tile C 26.9 -> 33.7; the next block's first step at 35.3.  A 16-row tile is 1.97 MFLOP: 3.2 us
// of ONE CU's fp32 matrix rate (8 waves share 4 SIMDs) + ~1.3 us of barriers + ~1.9 us of set-up (carried state, residual rows, h part of
// the cell step, granule reads): three tiles per stream on one CU are 19 us that start when tile A's rows are there.  The per-block launches put
// every tile on a CU of its own BEHIND the scans (4.7 us for all of them: the chip's matrix rate) and pay the launch boundary instead; with
// 256 CUs and 2 x (32 + 64) resident workgroups there is no third way.  OPT-IN therefore (dpdf_set_option "hop_stack" = 1): no gain, one more
// co-residency assumption.  2 nx + S workgroups of one CU each must be co-resident (64 streams: 96 per stack, both stacks 192 of 256 CUs):
// the host asks for it only while they fit, every wait has the time-out of the GRU-256 clusters behind it (device error flag -> snapshot
// recovery on the plain launches), and the engine counts recoveries (dpdf_recovery_count).
// Arithmetic: the per-row operations and their order are those of dprnn_hop_glue8_body / gru64_scan4_body -- a row's result does not
// depend on which rows share its tile -- so the outputs are bit-identical to the per-block launches (tests/test_gpu_parity.py::test_streaming_hop_forms_equal_the_plain_chain).
// Reference: onnx_model/layers.py:159-196 (block), :278-302 (streaming inter-band GRUCell).
#pragma once
#include <type_traits>
#include "gru_scan4.h"
#include "fcln_gi.h"

#ifdef DPDF_PHASE_TRACE
// timing builds only (tools/stack_trace.py): s_memrealtime stamps (100 MHz, one clock for the whole chip) of the DF stack's workgroups of stream
// group 0: [0 + 4 n + e] forward scan of block n (e: flags seen | first step | behind the last step), [64 + 4 n + e] backward scan,
// [128 + 16 n + 4 k + e] glue of stream 0, tile k (e: tile entered | granules valid | tile computed | flag out)
__device__ unsigned long long dpdf_stack_trace[512];
#define DPDF_KSTAMP(cond, i) do { if ((cond) && threadIdx.x == 0) dpdf_stack_trace[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DPDF_KSTAMP(cond, i) do {} while (0)
#endif
struct HopStackBlock {
    const float* hh4; const float* intra_bias;                    // this block's intra-band scans (gru64_scan4: [dir][wave][64][64], [dir][4][64])
    const float* fci_frag; const float* fci_b; const float* lni_g; const float* lni_b;
    const float* wfrag; const float* bias;                        // inter-band GRU (packed as for gru64_scan_kernel, dir 0)
    float* hstate;                                                // carried state of this block: hstate[stream * h_hi + pos * 64 + unit]
    const float* fce_frag; const float* fce_b; const float* lne_g; const float* lne_b;
    const float* ih_frag; const float* ih_bias;                   // NEXT block's intra input projection (null: last block)
};
constexpr int HOP_STACK_MAX_BLOCKS = 8;
struct HopStackArgs {
    HopStackBlock blk[HOP_STACK_MAX_BLOCKS]; int nb;
    int S, Fp; long h_hi;
    const float* x0;                 // [S * Fp][64] block-0 input rows (previous launch)
    const float* gi0;                // [S * Fp][384] block-0 input projection (previous launch)
    float* gi;                       // [2][S * Fp][384]: blocks 1.. by parity
    unsigned long long* hcat;        // [2][S * Fp][128] granules {epoch, h}, by block parity
    float* y_out;                    // [S * Fp][64] last block's output
    unsigned* gi_flags;              // [S][4] (first word used): the epoch of the block whose NEXT-block projection stream s has published
    unsigned epoch0;                 // epoch of block 0 of this launch (block n: epoch0 + n); the host advances it by nb per launch
    int* err; unsigned* done;        // time-out flag; optional counter the glue workgroups bump behind the last block (stage 2's first kernel waits for it)
};

// band position of local row r (0..15) of tile k of a stream with Fp positions (< 0 or >= Fp: no such row)
__device__ __forceinline__ int hop_stack_pos(int Fp, int k, int r) { const int h = Fp >> 1; return r < 8 ? h - 8 * (k + 1) + r : h + 8 * k + (r - 8); }

// ---- scan role: gru64_scan4_body over the blocks of the stack, gi of blocks >= 1 behind the glue's flags, h' as granules
__device__ __forceinline__ void hop_stack_scan(const HopStackArgs& a, int bx, int dir) {
    __shared__ __attribute__((aligned(16))) float Hs[2][4][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int u = lane >> 2, j = lane & 3;
    const int unit = 16 * w + u;
    int rj = bx * 4 + j; const bool ok = rj < a.S; if (!ok) rj = a.S - 1;
    const int Fp = a.Fp;
    constexpr int gw = 384;
    float wk[64];
    auto load_w = [&](int n) {
        const float* wp = a.blk[n].hh4 + ((size_t)(dir * 4 + w) * 64) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 64; ++t) wk[t] = wp[(size_t)t * 64];
    };
    load_w(0);
    float* hw = &Hs[0][j][(unit & 3) * 16 + (unit >> 2)];
    const float* hr = &Hs[0][j][(((lane >> 4) + u) & 3) * 16];
    const unsigned g_off = (unsigned)((long)rj * Fp * gw + dir * 192 + unit);
    const long gdelta = dir ? -(long)gw : (long)gw;
    const unsigned o_off = (unsigned)((long)rj * Fp * 128 + dir * 64 + unit);
    const long odelta = dir ? -128L : 128L;
    unsigned spins = 0; bool dead = false;
    for (int n = 0; n < a.nb; ++n) {
        const float b_hn = a.blk[n].intra_bias[(size_t)dir * 256 + 192 + unit];
        const unsigned epoch = a.epoch0 + (unsigned)n;
        if (n > 0) {        // the glue of block n - 1 has published this block's projection, tile by tile (its last tile is the last to come)
            if (tid < 4) {
                const int s = bx * 4 + tid;
                if (s < a.S) {
                    const unsigned* f = a.gi_flags + (size_t)s * 4;
                    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (epoch - 1u)) < 0) {
                        if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            __syncthreads();
        }
        DPDF_KSTAMP(Fp >= 48 && bx == 0, dir * 64 + 4 * n + 0);
        const float* gi = n == 0 ? a.gi0 : a.gi + (size_t)(n & 1) * a.S * Fp * gw;
        unsigned long long* oc = a.hcat + (size_t)(n & 1) * a.S * Fp * 128 + (dir ? (long)(Fp - 1) * 128 : 0);
        float h_own = 0.f;
        *hw = 0.f;
        constexpr int PF = 8;
        float g[PF][3];
        const float* gnext = gi + (dir ? (long)(Fp - 1) * gw : 0);
        // (blocks >= 1: written by other workgroups of this launch a moment ago -- agent-scope loads, for block 0 as well: one code path)
#define HS_LDG(P) __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            g[d][0] = HS_LDG(gnext + g_off); g[d][1] = HS_LDG(gnext + g_off + 64); g[d][2] = HS_LDG(gnext + g_off + 128);
            if (d + 1 < Fp) gnext += gdelta;
        }
        __syncthreads();
        DPDF_KSTAMP(Fp >= 48 && bx == 0, dir * 64 + 4 * n + 1);
        int buf = 0;
        auto step = [&](int s, float (&gs)[3]) {
            f32x4 acc0 = {gs[0], gs[1], b_hn, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float xn = gs[2];
            gs[0] = HS_LDG(gnext + g_off); gs[1] = HS_LDG(gnext + g_off + 64); gs[2] = HS_LDG(gnext + g_off + 128);
            if (s + PF + 1 < Fp) gnext += gdelta;
            __builtin_amdgcn_sched_barrier(0);
            const float4* h4p = (const float4*)(hr + buf * (4 * 68));
            const float4 q0 = h4p[0], q1 = h4p[1], q2 = h4p[2], q3 = h4p[3];
            DPDF_M4R(0, q0) DPDF_M4R(1, q1) DPDF_M4R(2, q2) DPDF_M4R(3, q3)
            const f32x4 acc = acc0 + acc1;
            const float h = gru64_cell(acc[0], acc[1], xn, acc[2], h_own);
            h_own = h;
            hw[(buf ^ 1) * (4 * 68)] = h;
            if (ok) __hip_atomic_store(oc + o_off, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            oc += odelta;
            __syncthreads();
            buf ^= 1;
        };
        int s = 0;
        for (; s + PF <= Fp; s += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) step(s + d, g[d]);
        }
#pragma unroll
        for (int d = 0; d < PF - 1; ++d)
            if (s + d < Fp) step(s + d, g[d]);
#undef HS_LDG
        DPDF_KSTAMP(Fp >= 48 && bx == 0, dir * 64 + 4 * n + 2);
        if (n + 1 < a.nb) load_w(n + 1);          // in flight while the glue finishes tile C of this block
        __syncthreads();
    }
}

// ---- glue role: dprnn_hop_glue8_body for the three tiles of ONE stream, over the blocks of the stack.
// A block is two PASSES: tile A alone (ready after ~2/3 of the steps: it runs under the scans) and tiles B + C together -- the phases of a
// tile are latency (a barrier, an LDS round trip and a dependent MFMA chain each), so two tiles walked through the same phases cost ~1.4 x
// one, and only this pass lies behind the last step.  Everything of a pass that does not depend on the scans (carried state, residual rows,
// the h part of the cell step, tile B's rows) is done while it waits for the rows that do.
__device__ __forceinline__ void hop_stack_glue(const HopStackArgs& a, int s) {
    __shared__ __attribute__((aligned(16))) float As[2][16][132];
    __shared__ __attribute__((aligned(16))) float Fs[2][2][16][68];
    __shared__ __attribute__((aligned(16))) float Ys[2][16][68];
    __shared__ __attribute__((aligned(16))) float Hs[2][16][68];
    __shared__ __attribute__((aligned(16))) float Gs[2][4][3][4][64];
    __shared__ __attribute__((aligned(16))) float Ln[4][64];
    __shared__ __attribute__((aligned(16))) float Xs[3][16][68];      // the block's input rows of the three tiles (= the previous block's output)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int wc = w & 3, wk = w >> 2;
    const int cl = lane & 15, q = lane >> 4;
    const int Fp = a.Fp, nt = (Fp + 15) / 16;
    const bool ln_role = tid < 256;
    const int rr = (tid >> 4) & 15, rc4 = 4 * (tid & 15);
    unsigned spins = 0; bool dead = false;
    // operands of a block: this wave's share (see dprnn_hop_glue8_body)
    float wg[3][16], ffi[16], ffe[8], fih[48], bih[3];
    float b_r, b_z, b_in, b_hn, bfi, bfe;
    auto load_ops = [&](int n) {
        const HopStackBlock& g = a.blk[n];
        const float* wp = g.wfrag + ((size_t)wc * 2 + wk) * 3 * 16 * 64 + lane;
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) wg[gt][jj] = wp[(size_t)(gt * 16 + jj) * 64];
        b_r = g.bias[16 * wc + cl]; b_z = g.bias[64 + 16 * wc + cl]; b_in = g.bias[128 + 16 * wc + cl]; b_hn = g.bias[192 + 16 * wc + cl];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int kk = 16 * wk + k; ffi[k] = g.fci_frag[(size_t)((((kk >> 2) * 4 + wc) * 4 + (kk & 3)) * 64) + lane]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int kk = 8 * wk + k; ffe[k] = g.fce_frag[(size_t)((((kk >> 2) * 4 + wc) * 4 + (kk & 3)) * 64) + lane]; }
        if (g.ih_frag) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int pr = w + 8 * p, gp = pr >> 2, tile = pr & 3;
#pragma unroll
                for (int k = 0; k < 16; ++k) fih[p * 16 + k] = g.ih_frag[(size_t)gp * 4096 + (size_t)((((k >> 2) * 4 + tile) * 4 + (k & 3)) * 64) + lane];
                bih[p] = g.ih_bias[(pr >> 2) * 64 + (pr & 3) * 16 + cl];
            }
        }
        bfi = g.fci_b[16 * wc + cl]; bfe = g.fce_b[16 * wc + cl];
    };
    load_ops(0);
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
    auto granule_index = [&](int k, int e) -> size_t {       // granule e of this thread in tile k: (row idx >> 7, column idx & 127), rows clamped
        const int idx = tid + 512 * e, r = idx >> 7, c = idx & 127;
        const int p = hop_stack_pos(Fp, k, r);
        const int pp = p < 0 ? 0 : (p >= Fp ? Fp - 1 : p);
        return ((size_t)s * Fp + pp) * 128 + c;
    };
    for (int n = 0; n < a.nb; ++n) {
        const HopStackBlock& g = a.blk[n];
        const bool NEXT = g.ih_frag != nullptr;
        const unsigned epoch = a.epoch0 + (unsigned)n;
        const unsigned long long* hc = a.hcat + (size_t)(n & 1) * a.S * Fp * 128;
        float* gi_out = a.gi + (size_t)((n + 1) & 1) * a.S * Fp * 384;
        __syncthreads();                 // (the previous block's last pass is through with the shared tiles)
        if (tid < 256) {
            const float* src = tid < 64 ? g.lni_g : (tid < 128 ? g.lni_b : (tid < 192 ? g.lne_g : g.lne_b));
            Ln[tid >> 6][tid & 63] = src[tid & 63];
        }
        // the scans' rows of tile k into As[u]: polled until every epoch is this block's
        auto fetch_rows = [&](int k, int u) {
            unsigned long long v[4];
            const unsigned long long* src[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) src[e] = hc + granule_index(k, e);
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __hip_atomic_load(src[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); all_in &= (unsigned)(v[e] >> 32) == epoch; }
                if (__syncthreads_and(all_in ? 1 : 0)) break;
                if (!dead && cluster_spin_expired(spins, a.err, dead)) dead = true;
                if (__syncthreads_or(dead ? 1 : 0)) { dead = true; break; }      // a time-out reaches every thread within 256 rounds (the error flag): leave together
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int idx = tid + 512 * e; As[u][idx >> 7][idx & 127] = __uint_as_float((unsigned)v[e]); }
        };
        // one pass over CNT tiles k0, k0 + 1
        auto pass = [&](auto cnt_c, int k0) {
            constexpr int CNT = decltype(cnt_c)::value;
            DPDF_KSTAMP(Fp >= 48 && s == 0, 128 + 16 * n + 4 * k0 + 0);
            bool rok[CNT]; size_t grow[CNT]; int pc[CNT]; float4 xres[CNT], y1[CNT];
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                // rows of tile k0 + u: band position of local row rr, clamped for the loads (rows that do not exist compute on a copy and store nothing)
                const int p_ln = hop_stack_pos(Fp, k0 + u, rr); rok[u] = p_ln >= 0 && p_ln < Fp;
                pc[u] = p_ln < 0 ? 0 : (p_ln >= Fp ? Fp - 1 : p_ln);
                grow[u] = (size_t)s * Fp + pc[u];
                xres[u] = make_float4(0.f, 0.f, 0.f, 0.f); y1[u] = xres[u];
                if (ln_role) {
                    if (n == 0) xres[u] = *(const float4*)(a.x0 + grow[u] * 64 + rc4);
                    else xres[u] = *(const float4*)&Xs[k0 + u][rr][rc4];
                    *(float4*)&Hs[u][rr][rc4] = *(const float4*)(g.hstate + (long)s * a.h_hi + (long)pc[u] * 64 + rc4);
                }
            }
            f32x4 a0[CNT], a1[CNT], a2[CNT];
            auto gru_part = [&](int u, const float (*src)[68]) {
                if (wk == 0) { a0[u] = (f32x4){b_r, b_r, b_r, b_r}; a1[u] = (f32x4){b_z, b_z, b_z, b_z}; a2[u] = (f32x4){b_in, b_in, b_in, b_in}; }
                else { a0[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; a1[u] = a0[u]; a2[u] = (f32x4){b_hn, b_hn, b_hn, b_hn}; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 x4 = *(const float4*)&src[cl][16 * c + 4 * q];
                    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        a0[u] = mfma16(xv[kb], wg[0][c * 4 + kb], a0[u]);
                        a1[u] = mfma16(xv[kb], wg[1][c * 4 + kb], a1[u]);
                        a2[u] = mfma16(xv[kb], wg[2][c * 4 + kb], a2[u]);
                    }
                }
                if (wk == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { Gs[u][wc][0][i][lane] = a0[u][i]; Gs[u][wc][1][i][lane] = a1[u][i]; Gs[u][wc][2][i][lane] = a2[u][i]; }
                }
            };
            __syncthreads();
            if (wk == 1) {                   // the h part of the cell step depends on the carried state only: under the scans
#pragma unroll
                for (int u = 0; u < CNT; ++u) gru_part(u, Hs[u]);
            }
            // ---- the tiles' scan outputs (the earlier tile of a pair has long been complete; the later one is what the pass waits for)
#pragma unroll
            for (int u = 0; u < CNT; ++u) fetch_rows(k0 + u, u);
            __syncthreads();
            DPDF_KSTAMP(Fp >= 48 && s == 0, 128 + 16 * n + 4 * k0 + 1);
            // ---- fc_intra: K half wk
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    const float4 a4 = *(const float4*)&As[u][cl][64 * wk + 16 * c + 4 * q], b4 = *(const float4*)&As[u][cl][64 * wk + 16 * c + 16 + 4 * q];
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        acc0 = mfma16(av[kb], ffi[c * 4 + kb], acc0);
                        acc1 = mfma16(bv4[kb], ffi[(c + 1) * 4 + kb], acc1);
                    }
                }
                const float bv = wk ? 0.f : bfi;
#pragma unroll
                for (int i = 0; i < 4; ++i) Fs[u][wk][4 * q + i][16 * wc + cl] = acc0[i] + acc1[i] + bv;
            }
            __syncthreads();
            if (ln_role) {
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    const float4 uu = *(const float4*)&Fs[u][0][rr][rc4], v2 = *(const float4*)&Fs[u][1][rr][rc4];
                    y1[u] = layer_norm_res(make_float4(uu.x + v2.x, uu.y + v2.y, uu.z + v2.z, uu.w + v2.w), xres[u], Ln[0], Ln[1]);
                    *(float4*)&Ys[u][rr][rc4] = y1[u];
                }
            }
            __syncthreads();
            // ---- inter-band GRUCell step: wk = 0 the x part (the h part is in Gs already)
            float hn[CNT][4];
            if (wk == 0) {
#pragma unroll
                for (int u = 0; u < CNT; ++u) gru_part(u, Ys[u]);
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int u = 0; u < CNT; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        hn[u][i] = gru64_cell(a0[u][i] + Gs[u][wc][0][i][lane], a1[u][i] + Gs[u][wc][1][i][lane], a2[u][i], Gs[u][wc][2][i][lane], Hs[u][4 * q + i][16 * wc + cl]);
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int u = 0; u < CNT; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) Hs[u][4 * q + i][16 * wc + cl] = hn[u][i];
            }
            __syncthreads();
            if (ln_role) {
#pragma unroll
                for (int u = 0; u < CNT; ++u)
                    if (rok[u]) *(float4*)(g.hstate + (long)s * a.h_hi + (long)pc[u] * 64 + rc4) = *(const float4*)&Hs[u][rr][rc4];
            }
            // ---- fc_inter on h': K half wk
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                const float4 a4 = *(const float4*)&Hs[u][cl][32 * wk + 4 * q], b4 = *(const float4*)&Hs[u][cl][32 * wk + 16 + 4 * q];
                const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    acc0 = mfma16(av[kb], ffe[kb], acc0);
                    acc1 = mfma16(bv4[kb], ffe[4 + kb], acc1);
                }
                const float bv = wk ? 0.f : bfe;
#pragma unroll
                for (int i = 0; i < 4; ++i) Fs[u][wk][4 * q + i][16 * wc + cl] = acc0[i] + acc1[i] + bv;
            }
            __syncthreads();
            if (ln_role) {
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    const float4 uu = *(const float4*)&Fs[u][0][rr][rc4], v2 = *(const float4*)&Fs[u][1][rr][rc4];
                    const float4 y2 = layer_norm_res(make_float4(uu.x + v2.x, uu.y + v2.y, uu.z + v2.z, uu.w + v2.w), y1[u], Ln[2], Ln[3]);
                    if (NEXT) { *(float4*)&Xs[k0 + u][rr][rc4] = y2; *(float4*)&Ys[u][rr][rc4] = y2; }
                    else if (rok[u]) {
                        float* yo = a.y_out + grow[u] * 64 + rc4;
                        if (a.done) {           // read by a kernel that is already running on another stream: write-through
                            __hip_atomic_store(yo + 0, y2.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(yo + 1, y2.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(yo + 2, y2.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(yo + 3, y2.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else *(float4*)yo = y2;
                    }
                }
            }
            if (NEXT) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    float4 y4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) y4[c] = *(const float4*)&Ys[u][cl][16 * c + 4 * q];
                    f32x4 acc[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float yv[4] = {y4[c].x, y4[c].y, y4[c].z, y4[c].w};
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int p = 0; p < 3; ++p) acc[p] = mfma16(yv[kb], fih[p * 16 + c * 4 + kb], acc[p]);
                    }
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const int pr = w + 8 * p, col = (pr >> 2) * 64 + (pr & 3) * 16 + cl;
                        const float bv = bih[p];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int pos = hop_stack_pos(Fp, k0 + u, 4 * q + i);
                            if (pos >= 0 && pos < Fp)
                                __hip_atomic_store(gi_out + ((size_t)s * Fp + pos) * 384 + col, acc[p][i] + bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                DPDF_KSTAMP(Fp >= 48 && s == 0, 128 + 16 * n + 4 * k0 + 2);
                if (k0 + CNT == nt) {
                    // the stream's rows of the next block's projection are out (the first pass's stores have long been acknowledged): ONE flag per
                    // stream and block behind the acknowledged stores -- the scans need every tile before their first step anyway
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(a.gi_flags + (size_t)s * 4, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                DPDF_KSTAMP(Fp >= 48 && s == 0, 128 + 16 * n + 4 * k0 + 3);
            }
            __syncthreads();
        };
        // (two tiles per pass -- pass(integral_constant<int, 2>, 1) for B + C -- was measured: the second tile's accumulators and rows push the
        // wave over its 256 registers (83 spilled), the pass takes 15.6 us instead of 2 x 6.5: one tile per pass)
        for (int k = 0; k < nt; ++k) pass(std::integral_constant<int, 1>{}, k);
        if (n + 1 < a.nb) load_ops(n + 1);       // in flight while the scans of the next block run
    }
    if (a.done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(512) void dprnn_hop_stack_kernel(HopStackArgs a) {
    const int nx = (a.S + 3) / 4, nscan = 2 * nx;
    if ((int)blockIdx.x < nscan) {
        if (threadIdx.x >= 256) return;
        hop_stack_scan(a, blockIdx.x >> 1, blockIdx.x & 1);
    } else {
        hop_stack_glue(a, (int)blockIdx.x - nscan);
    }
}
