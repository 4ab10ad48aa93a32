// gru_scan4.h -- the GRU(64) recurrence for FEW rows (streaming hops, a single clip): 4-row tiles on the 4x4x1 matrix shape.
//
// A GRU-64 step on the 16x16x4 fp32 MFMA shape costs 48 instructions x 32 cycles per SIMD whether the tile holds 16 live
// rows or one (gru64_scan_gi_kernel: ~1.0 us per dependent step), and with few rows nothing else runs on those SIMDs --
// the step time IS the call's latency (8 blocks x 48 band positions per streaming hop).  The fp32 matrix rate is per
// output element, so the way to a shorter step is a smaller tile spread over more CUs: v_mfma_f32_4x4x1_16b_f32 computes
// sixteen independent 4x4 outer products per instruction (block b: D[i][j] += A[lane 4b + i] * B[lane 4b + j], D register i of
// lane 4b + j), 8.4 cycles each (tools/mfma4_probe.hip).  A workgroup owns 4 rows (and one direction); wave w owns hidden units
// [16w, 16w + 16), block b = unit 16w + b:
//   * A operand = the WEIGHTS: lane 4b + i of instruction t holds W_hh[gate i][unit 16w + b][k(t, b)] for i = r, z, n (i = 3: zero)
//     -- 64 VGPRs, resident for the whole scan;
//   * B operand = h(s-1), lane-group broadcast: BLGP = 4 + g feeds all four 16-lane groups the B values of group g, so block b
//     sees lanes 16g + 4(b & 3) + j.  Lane (g, c, j) keeps h[row j][4m + ((g + c) & 3)], m = 0..15, in 16 registers (four
//     ds_read_b128 of one 64-byte run per lane: LDS holds h as [row][k & 3][k >> 2]); instruction t = (m, g) then gives block b
//     the product over k(t, b) = 4m + ((g + b) & 3): every block meets every k once in 64 instructions;
//   * D: register i = GATE i, lane 4b + j = (unit b, row j): a lane ends up with r, z and the candidate's h part of ONE
//     (row, unit) in its own registers -- no exchange between lanes in front of the gate math (the first form of this kernel had
//     h as a CBSZ / ABID-broadcast A operand and the weights as B: gates across the lanes of a quad, a 4x4 DPP transpose of
//     24 instructions = 120 cycles of every 1090-cycle step, tools/scan4_bench.hip).  The accumulator starts from the hoisted
//     input-side pre-activations (r, z, b_hn); the candidate's x part stays in a register; two k-interleaved chains (a dependent
//     4x4x1 costs 12.5 cycles, an independent one 8.4);
//   * h' goes back to LDS (one dword per lane, conflict-free) and straight to HBM (one dword per lane, 64-byte runs).
// Four times the workgroups of the 16-row form, so it is used while they still spread over idle CUs (run_dprnn); at saturation
// the 16-row forms win (a quarter of every instruction is the zero row).  gi: see gru64_scan_gi_kernel.  Results equal that
// kernel to rounding (k is summed in two interleaved chains here, in MFMA-internal groups of four there).
#pragma once
#include "common.h"
#include "gru_scan.h"

template <int G>
__device__ __forceinline__ f32x4 mfma4g(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 4 + G); }

// the four lane groups of h register M (instructions t = 4M .. 4M + 3), alternating between two accumulator chains
#define DPDF_M4Q(M, HV) \
    acc0 = mfma4g<0>(wk[4 * (M) + 0], HV, acc0); acc1 = mfma4g<1>(wk[4 * (M) + 1], HV, acc1); \
    acc0 = mfma4g<2>(wk[4 * (M) + 2], HV, acc0); acc1 = mfma4g<3>(wk[4 * (M) + 3], HV, acc1);
#define DPDF_M4R(Q, H4) DPDF_M4Q(4 * (Q) + 0, H4.x) DPDF_M4Q(4 * (Q) + 1, H4.y) DPDF_M4Q(4 * (Q) + 2, H4.z) DPDF_M4Q(4 * (Q) + 3, H4.w)

// wfrag4: [dir][wave 4][instruction t = 4m + g][lane 4b + i] (build_gru64), bias: the 16-row kernels' [dir][4][64]
// HANDOFF (dprnn_hop_block.h): h' leaves through agent-scope (write-through) stores and the caller publishes a flag behind them
#ifdef DPDF_PHASE_TRACE
#define DPDF_SSTAMP(i) do { if (HANDOFF && a.nsteps >= 48 && bx == 0 && dir == 0 && threadIdx.x == 0) dpdf_trace_buf[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DPDF_SSTAMP(i) do {} while (0)
#endif
template <bool HANDOFF>
__device__ __forceinline__ void gru64_scan4_body(const Gru64Args& a, const float* wfrag4, const float* gi, int gw, int bx, int dir) {
    __shared__ __attribute__((aligned(16))) float Hs[2][4][68];      // [buffer][row][(k & 3) * 16 + (k >> 2)], rows padded: conflict-free b128 reads
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int row0 = bx * 4;
    DPDF_SSTAMP(16);
    const int u = lane >> 2, j = lane & 3;                           // D / B role: (unit 16w + u, row j); A role: (block u, gate j)
    float wk[64];
    {
        const float* wp = wfrag4 + ((size_t)(dir * 4 + w) * 64) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 64; ++t) wk[t] = wp[(size_t)t * 64];
    }
    const int unit = 16 * w + u;
    const float b_hn = a.bias[(size_t)dir * 256 + 192 + unit];
    // Addressing: one wave-uniform base pointer per tensor, advanced by a scalar add per step, + a 32-bit lane offset (the host
    // keeps these launches below 2^30 elements) -- recomputing 64-bit positions per step cost ~70 scalar instructions in front of
    // the MFMA block of every step, a fifth of the step with one wave per SIMD and nothing to hide them under.
    int rj = row0 + j; const bool ok = rj < a.nrows; if (!ok) rj = a.nrows - 1;
    const unsigned g_off = (unsigned)(((long)(rj / a.rdiv) * a.x_hi + (long)(rj % a.rdiv) * a.x_lo) / 64 * gw + dir * 192 + unit);   // + 0 / 64 / 128: r, z, n
    const long gstep = a.x_step / 64 * gw;
    const long gdelta = dir ? -gstep : gstep;
    const unsigned o_off = (unsigned)((long)(rj / a.rdiv) * a.o_hi + (long)(rj % a.rdiv) * a.o_lo + dir * a.o_dir_off + unit);
    const long odelta = dir ? -a.o_step : a.o_step;
    float* ocur = a.out + (dir ? (long)(a.nsteps - 1) * a.o_step : 0);                  // wave-uniform
    float* hp = a.hstate ? a.hstate + (long)(rj / a.rdiv) * a.h_hi + (long)(rj % a.rdiv) * a.h_lo + unit : nullptr;
    float h_own = hp ? *hp : 0.f;
    float* hw = &Hs[0][j][(unit & 3) * 16 + (unit >> 2)];                              // k = unit: [k & 3][k >> 2]
    const float* hr = &Hs[0][j][(((lane >> 4) + u) & 3) * 16];                         // lane group g = lane >> 4, c = u & 3: the run of k & 3 = (g + c) & 3
    *hw = h_own;
    // input-side pre-activations: a register ring PF steps deep.  gi was written by the previous launch, usually on another
    // XCD: the loads miss this XCD's L2 and come back from the memory side in 1-2 us -- several steps -- so a short
    // lookahead leaves the scan waiting on them at the top of every step
    constexpr int PF = 8;          // (eight steps = 3.3 us of cover: in a hop the rows come from the other XCDs' side of the fabric; four left the 0.41-us step waiting)
    float g[PF][3];
    const float* gnext = gi + (dir ? (long)(a.nsteps - 1) * gstep : 0);                 // wave-uniform: rows of the step being fetched
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        g[d][0] = gnext[g_off]; g[d][1] = gnext[g_off + 64]; g[d][2] = gnext[g_off + 128];
        if (d + 1 < a.nsteps) gnext += gdelta;                                          // (clamped at the last step)
    }
    __syncthreads();
    DPDF_SSTAMP(17);
    int buf = 0;
    auto step = [&](int s, float (&gs)[3]) {
        f32x4 acc0 = {gs[0], gs[1], b_hn, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float xn = gs[2];
        // refill this ring slot for step s + PF, in front of the MFMA block (see gru64_scan_gi_kernel)
        gs[0] = gnext[g_off]; gs[1] = gnext[g_off + 64]; gs[2] = gnext[g_off + 128];
        if (s + PF + 1 < a.nsteps) gnext += gdelta;
        __builtin_amdgcn_sched_barrier(0);
        const float4* h4p = (const float4*)(hr + buf * (4 * 68));
        const float4 q0 = h4p[0], q1 = h4p[1], q2 = h4p[2], q3 = h4p[3];
        DPDF_M4R(0, q0) DPDF_M4R(1, q1) DPDF_M4R(2, q2) DPDF_M4R(3, q3)
        const f32x4 acc = acc0 + acc1;              // r, z, hn pre-activations of (row j, unit)
        const float h = gru64_cell(acc[0], acc[1], xn, acc[2], h_own);
        h_own = h;
        hw[(buf ^ 1) * (4 * 68)] = h;
        if (ok) {
            if (HANDOFF) __hip_atomic_store(ocur + o_off, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ocur[o_off] = h;
        }
        ocur += odelta;
        __syncthreads();
        buf ^= 1;
    };
    int s = 0;
    for (; s + PF <= a.nsteps; s += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(s + d, g[d]);
    }
#pragma unroll
    for (int d = 0; d < PF - 1; ++d)
        if (s + d < a.nsteps) step(s + d, g[d]);
    DPDF_SSTAMP(18);
    if (hp && ok) *hp = h_own;
}

__global__ __launch_bounds__(256) void gru64_scan4_gi_kernel(Gru64Args a, const float* wfrag4, const float* gi, int gw) {
    gru64_scan4_body<false>(a, wfrag4, gi, gw, blockIdx.x, blockIdx.y);
}
