// gru_scan4.h -- the GRU(64) recurrence for FEW rows (streaming hops, a single clip): 4-row tiles on the 4x4x1 matrix shape.
//
// A GRU-64 step on the 16x16x4 fp32 MFMA shape costs 48 instructions x 32 cycles per SIMD whether the tile holds 16 live
// rows or one (gru64_scan_gi_kernel: ~1.0 us per dependent step), and with few rows nothing else runs on those SIMDs --
// the step time IS the call's latency (8 blocks x 48 band positions per streaming hop).  The fp32 matrix rate is per
// output element, so the way to a shorter step is a smaller tile spread over more CUs: v_mfma_f32_4x4x1_16b_f32 computes
// sixteen independent 4x4 outer products per instruction, and its broadcast modifiers (CBSZ = 4, ABID = b) feed all
// sixteen blocks the A operand of block b -- one instruction = 4 rows x 64 columns x one k, 8.4 cycles (measured,
// tools/mfma4_probe.hip).  A workgroup owns 4 rows (and one direction); wave w owns hidden units [16w, 16w+16):
//   * B operand of step k: lane 4u + j holds W_hh[gate j][unit 16w + u][k] for j = r, z, n (j = 3: zero) -- 64 VGPRs,
//     resident for the whole scan;
//   * A operand: h(s-1) as ONE float4 per lane from LDS: lane 4b + i, component v = h[row i][k = 16v + b], so that
//     ABID = b picks k inside register v;
//   * D: register i = row i, lane 4u + j = (unit u, gate j): the accumulator starts from the hoisted input-side
//     pre-activations (r, z, b_hn, and the candidate's x part parked in the zero column j = 3), two k-interleaved chains
//     (a dependent 4x4x1 costs 12.5 cycles, an independent one 8.4);
//   * a 4x4 transpose inside every lane quad (DPP quad_perm + select) hands lane j the four gate values of ROW j, so the
//     gate math -- 6 transcendentals -- runs once per lane instead of four times;
//   * h' goes back to LDS in the A layout (one dword per lane) and straight to HBM (one dword per lane, 64-byte runs).
// 64 MFMAs x 8.4 cycles + one LDS round trip + ~30 VALU per step: ~0.4 us.  Four times the workgroups of the 16-row
// form, so it is used while they still spread over idle CUs (run_dprnn); at saturation the 16-row forms win (the zero
// column and the transposes are overhead there).  gi: see gru64_scan_gi_kernel.  Results equal that kernel to rounding
// (k is summed in two interleaved chains here, in MFMA-internal groups of four there).
#pragma once
#include "common.h"
#include "gru_scan.h"

template <int AB>
__device__ __forceinline__ f32x4 mfma4b(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0); }

#define DPDF_QUAD_X 0xB1   // quad_perm [1,0,3,2]
#define DPDF_QUAD_Y 0x4E   // quad_perm [2,3,0,1]
__device__ __forceinline__ float quad_x(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPDF_QUAD_X, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_y(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPDF_QUAD_Y, 0xf, 0xf, false)); }

// v[i] at lane j of a quad  ->  v[g] at lane j = old v[j] at lane g   (4x4 transpose across registers x quad lanes)
__device__ __forceinline__ void quad_transpose4(f32x4& v, bool odd, bool hi) {
    const float t0 = quad_x(v[1]), u0 = quad_x(v[0]), t1 = quad_x(v[3]), u1 = quad_x(v[2]);
    const float a0 = odd ? t0 : v[0], a1 = odd ? v[1] : u0, a2 = odd ? t1 : v[2], a3 = odd ? v[3] : u1;
    const float p0 = quad_y(a2), q0 = quad_y(a0), p1 = quad_y(a3), q1 = quad_y(a1);
    v[0] = hi ? p0 : a0; v[2] = hi ? a2 : q0; v[1] = hi ? p1 : a1; v[3] = hi ? a3 : q1;
}

// 16 k-steps out of register component V of the A operand, alternating between two accumulator chains
#define DPDF_M4(V, B, ACC) ACC = mfma4b<B>(hv[V], wk[16 * (V) + (B)], ACC);
#define DPDF_M4x16(V) \
    DPDF_M4(V, 0, acc0) DPDF_M4(V, 1, acc1) DPDF_M4(V, 2, acc0) DPDF_M4(V, 3, acc1) DPDF_M4(V, 4, acc0) DPDF_M4(V, 5, acc1) DPDF_M4(V, 6, acc0) DPDF_M4(V, 7, acc1) \
    DPDF_M4(V, 8, acc0) DPDF_M4(V, 9, acc1) DPDF_M4(V, 10, acc0) DPDF_M4(V, 11, acc1) DPDF_M4(V, 12, acc0) DPDF_M4(V, 13, acc1) DPDF_M4(V, 14, acc0) DPDF_M4(V, 15, acc1)

// wfrag4: [dir][wave 4][k 64][lane 64] (build_gru64), bias: the 16-row kernels' [dir][4][64]
// HANDOFF (dprnn_hop_block.h): h' leaves through agent-scope (write-through) stores and the caller publishes a flag behind them
#ifdef DPDF_PHASE_TRACE
#define DPDF_SSTAMP(i) do { if (HANDOFF && a.nsteps >= 48 && bx == 0 && dir == 0 && threadIdx.x == 0) dpdf_trace_buf[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DPDF_SSTAMP(i) do {} while (0)
#endif
template <bool HANDOFF>
__device__ __forceinline__ void gru64_scan4_body(const Gru64Args& a, const float* wfrag4, const float* gi, int gw, int bx, int dir) {
    __shared__ __attribute__((aligned(16))) float Hs[2][256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int row0 = bx * 4;
    DPDF_SSTAMP(16);
    const int u = lane >> 2, j = lane & 3;
    const bool odd = j & 1, hi = j & 2;
    float wk[64];
    {
        const float* wp = wfrag4 + ((size_t)(dir * 4 + w) * 64) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 64; ++k) wk[k] = wp[(size_t)k * 64];
    }
    const int unit = 16 * w + u;
    const float b_hn = a.bias[(size_t)dir * 256 + 192 + unit];
    // pre-transpose role of this lane: gate j of `unit` for rows 0..3 (register i = row i); j = 3 carries the candidate's x part.
    // Addressing: one wave-uniform base pointer per tensor, advanced by a scalar add per step, + a 32-bit lane offset (the host
    // keeps these launches below 2^30 elements) -- recomputing 64-bit positions per step cost ~70 scalar instructions in front of
    // the MFMA block of every step, a fifth of the step with one wave per SIMD and nothing to hide them under.
    const int gcol = dir * 192 + (j == 0 ? 0 : (j == 1 ? 64 : 128)) + unit;
    unsigned g_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rc = row0 + i; if (rc >= a.nrows) rc = a.nrows - 1;
        g_off[i] = (unsigned)(((long)(rc / a.rdiv) * a.x_hi + (long)(rc % a.rdiv) * a.x_lo) / 64 * gw + gcol);
    }
    const long gstep = a.x_step / 64 * gw;
    const long gdelta = dir ? -gstep : gstep;
    // post-transpose role: (row j, unit)
    int rj = row0 + j; const bool ok = rj < a.nrows; if (!ok) rj = a.nrows - 1;
    const unsigned o_off = (unsigned)((long)(rj / a.rdiv) * a.o_hi + (long)(rj % a.rdiv) * a.o_lo + dir * a.o_dir_off + unit);
    const long odelta = dir ? -a.o_step : a.o_step;
    float* ocur = a.out + (dir ? (long)(a.nsteps - 1) * a.o_step : 0);                  // wave-uniform
    float* hp = a.hstate ? a.hstate + (long)(rj / a.rdiv) * a.h_hi + (long)(rj % a.rdiv) * a.h_lo + unit : nullptr;
    float h_own = hp ? *hp : 0.f;
    Hs[0][lane * 4 + w] = h_own;
    // input-side pre-activations: a register ring PF steps deep.  gi was written by the previous launch, usually on another
    // XCD: the loads miss this XCD's L2 and come back from the memory side in ~1 us -- longer than a step -- so one step of
    // lookahead leaves the scan waiting on them at the top of every step
    constexpr int PF = 4;
    float g[PF][4];
    const float* gnext = gi + (dir ? (long)(a.nsteps - 1) * gstep : 0);                 // wave-uniform: rows of the step being fetched
#pragma unroll
    for (int d = 0; d < PF; ++d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[d][i] = gnext[g_off[i]];
        if (d + 1 < a.nsteps) gnext += gdelta;                                          // (clamped at the last step)
    }
    __syncthreads();
    DPDF_SSTAMP(17);
    int buf = 0;
    auto step = [&](int s, float (&gs)[4]) {
        f32x4 acc0, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc0[i] = j == 2 ? b_hn : gs[i];
        // refill this ring slot for step s + PF, in front of the MFMA block (see gru64_scan_gi_kernel)
#pragma unroll
        for (int i = 0; i < 4; ++i) gs[i] = gnext[g_off[i]];
        if (s + PF + 1 < a.nsteps) gnext += gdelta;
        __builtin_amdgcn_sched_barrier(0);
        const float4 h4 = *(const float4*)&Hs[buf][lane * 4];
        const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
        DPDF_M4x16(0) DPDF_M4x16(1) DPDF_M4x16(2) DPDF_M4x16(3)
        f32x4 acc = acc0 + acc1;
        quad_transpose4(acc, odd, hi);              // acc[0..3] = r, z, hn, xn pre-activations of (row j, unit)
        const float h = gru64_cell(acc[0], acc[1], acc[3], acc[2], h_own);
        h_own = h;
        Hs[buf ^ 1][lane * 4 + w] = h;
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
