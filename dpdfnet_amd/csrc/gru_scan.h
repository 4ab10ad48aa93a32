// gru_scan.h -- recurrent scans on the CDNA4 matrix cores.
//
// gru64_scan_kernel: the DPRNN work-horse (85 % of the model's FLOPs).  One kernel serves both
// recurrences of a DPRNN block (reference onnx_model/layers.py:159-196):
//   * intra-band bi-GRU: rows = frames (clip,t), steps = frequency positions, h0 = 0, 2 directions;
//   * inter-band GRUCell: rows = (clip, band position), steps = frames, h carried in the state.
// A 256-thread workgroup (4 waves) owns 16 rows; wave w owns hidden units [16w,16w+16) of all
// three gates.  Both W_ih and W_hh slices (96 VGPRs of MFMA B fragments) stay in registers for the
// whole scan; per step the wave issues 96 v_mfma_f32_16x16x4_f32 (x-part + h-part), does the
// gate math lane-locally in the MFMA C layout, and exchanges its 16x16 slice of h' with the
// other three waves through a double-buffered 4 KB LDS tile (one barrier per step).  Nothing
// but x (read once) and h' (written once) touches HBM: the [rows,192] gate pre-activations of
// the reference's formulation never exist.
//
// gru256_scan_kernel: the five 256-wide GRU cells (emb/erb-decoder/df-decoder stacks,
// reference onnx_model/layers.py:1168-1188, 1235-1259).  Input projections W_ih x (+biases) are
// hoisted out of the recurrence into one big MFMA GEMM over all frames; the scan only carries
// h W_hh^T.  16 rows per workgroup, 16 waves (one per 16 hidden units), W_hh fragments streamed
// from L2 each step, h exchanged through LDS.
#pragma once
#include "common.h"
#ifndef GRU64_VARIANT
#define GRU64_VARIANT 0   // tools/gru64_bench.hip ablations: 1 no gate math, 2 no global stores, 4 no barrier, 8 no x loads, 16 no LDS h reads
#endif

struct Gru64Args {
    const float* x;        // inputs, channels-last rows of 64
    float* out;            // h' for every step
    const float* wfrag;    // [dir][wave][part ih/hh][gate r,z,n][chunk][kb][lane]
    const float* bias;     // [dir][4][64]: b_ir+b_hr | b_iz+b_hz | b_in | b_hn
    float* hstate;         // optional carried state (h0 in, final h out), else h0 = 0
    int nrows, nsteps, ndirs;
    int rdiv;              // row r -> hi = r / rdiv, lo = r % rdiv
    long x_hi, x_lo, x_step;
    long o_hi, o_lo, o_step;
    int o_dir_off;         // column offset of direction d in the output row (d * 64)
    long h_hi, h_lo;
};

__global__ __launch_bounds__(256, 3) void gru64_scan_kernel(Gru64Args a) {
    // Hs: h exchange tiles, Xs: x staging tiles; both double-buffered, rows padded to 68 floats
    __shared__ __attribute__((aligned(16))) float Hs[2][16][68];
    __shared__ __attribute__((aligned(16))) float Xs[2][16][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = blockIdx.y;
    const int row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;

    // --- weights: registers for the whole scan --------------------------------------------
    float wih[3][16], whh[3][16];
    {
        const float* wp = a.wfrag + ((size_t)(dir * 4 + w) * 2) * 3 * 16 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                wih[g][j] = wp[(size_t)((0 * 3 + g) * 16 + j) * 64];
                whh[g][j] = wp[(size_t)((1 * 3 + g) * 16 + j) * 64];
            }
    }
    const float* bp = a.bias + (size_t)dir * 256 + 16 * w + cl;
    const float b_r = bp[0], b_z = bp[64], b_in = bp[128], b_hn = bp[192];

    // --- row addressing ----------------------------------------------------------------------
    // Tile base is wave-uniform (64-bit, scalar); per-lane offsets are 32-bit deltas from it
    // (a tile spans at most two `hi` groups, so deltas are bounded by one clip stride).
    // Global I/O is cooperative and row-contiguous: lane (w, q, cl) moves the 16-byte piece
    // [row 4w+q][cols 4cl..4cl+3] of the 16x64 tile -- ONE dwordx4 load of x and ONE dwordx4 store
    // of h' per lane per step, whole 256-byte rows per 16 lanes.
    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    const float* xbase = a.x + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    float* obase = a.out + (long)hi0 * a.o_hi + (long)lo0 * a.o_lo + dir * a.o_dir_off;
    const int srow = 4 * w + q, scol = 4 * cl;
    unsigned sx_off, so_off; bool so_ok;     // non-negative lane offsets: address = scalar per-step base + VGPR offset
    {
        int rs = row0 + srow;
        so_ok = rs < a.nrows;
        if (rs >= a.nrows) rs = a.nrows - 1;
        const int dh = rs / a.rdiv - hi0, dl = rs % a.rdiv - lo0;
        sx_off = (unsigned)((long)dh * a.x_hi + (long)dl * a.x_lo) + scol;
        so_off = (unsigned)((long)dh * a.o_hi + (long)dl * a.o_lo) + scol;
    }
    float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rc = row0 + q * 4 + i;
        if (rc >= a.nrows) rc = a.nrows - 1;
        float hv = a.hstate ? a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] : 0.f;
        h_own[i] = hv;
        Hs[1][q * 4 + i][16 * w + cl] = hv;      // h0 in the exchange tile, as every later h'
    }
    {
        const int p0 = dir ? a.nsteps - 1 : 0;
        *(float4*)&Xs[0][srow][scol] = *(const float4*)((xbase + (long)p0 * a.x_step) + sx_off);
    }
    __syncthreads();

    int buf = 0;
    for (int s = 0; s < a.nsteps; ++s) {
        // Vector-memory traffic of the step, issued back to back at its very top and waited for only
        // at its very end (vmcnt retires in order and counts stores on CDNA4):
        //   store h'(s-1): this lane's 16-byte piece of the tile completed by the last barrier;
        //   load  x(s+1):  this lane's 16-byte piece of the next input tile.
        if (s > 0 && !(GRU64_VARIANT & 2)) {
            const int pp = dir ? a.nsteps - s : s - 1;
            const float4 hv4 = *(const float4*)&Hs[buf ^ 1][srow][scol];
            if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
        }
        float4 xnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(GRU64_VARIANT & 8)) {
            const int sn = s + 1 < a.nsteps ? s + 1 : s;
            const int pn = dir ? a.nsteps - 1 - sn : sn;
            xnext = *(const float4*)((xbase + (long)pn * a.x_step) + sx_off);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 ar = {b_r, b_r, b_r, b_r}, az = {b_z, b_z, b_z, b_z};
        f32x4 axn = {b_in, b_in, b_in, b_in}, ahn = {b_hn, b_hn, b_hn, b_hn};
        const float* xrow = &Xs[buf][cl][4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 x4 = *(const float4*)(xrow + 16 * c);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ar = mfma16(xv[kb], wih[0][c * 4 + kb], ar);
                az = mfma16(xv[kb], wih[1][c * 4 + kb], az);
                axn = mfma16(xv[kb], wih[2][c * 4 + kb], axn);
            }
        }
        const float* hrow = &Hs[buf ^ 1][cl][4 * q];       // h(s-1): written last step (or h0)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#if GRU64_VARIANT & 16
            const float4 h4 = make_float4(h_own[0], h_own[1], h_own[2], h_own[3]);
#else
            const float4 h4 = *(const float4*)(hrow + 16 * c);
#endif
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ar = mfma16(hv[kb], whh[0][c * 4 + kb], ar);
                az = mfma16(hv[kb], whh[1][c * 4 + kb], az);
                ahn = mfma16(hv[kb], whh[2][c * 4 + kb], ahn);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if GRU64_VARIANT & 1
            float h = 0.01f * (ar[i] + az[i] + axn[i] + ahn[i]) + 0.5f * h_own[i];
#else
            float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
#endif
            h_own[i] = h;
            Hs[buf][q * 4 + i][16 * w + cl] = h;
        }
        __builtin_amdgcn_sched_barrier(0);
        *(float4*)&Xs[buf ^ 1][srow][scol] = xnext;        // the step's only vmcnt wait: both ops are a step old
#if !(GRU64_VARIANT & 4)
        __syncthreads();
#endif
        buf ^= 1;
    }
    if (a.nsteps > 0 && !(GRU64_VARIANT & 2)) {
        const int pp = dir ? 0 : a.nsteps - 1;
        const float4 hv4 = *(const float4*)&Hs[buf ^ 1][srow][scol];
        if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
    }
    if (a.hstate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int rc = row0 + q * 4 + i;
            if (rc < a.nrows)
                a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] = h_own[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gru64_scan_gi_kernel: the small-batch form of the scan.  When streams x frames is too small to put several
// workgroups on every CU (single-hop streaming, one clip through enhance()), a scan step is pure latency: 96
// dependent-issue MFMAs per wave, nothing to overlap them with.  Half of them (W_ih x) do not depend on the
// recurrence, so they are hoisted into one gemm_rows launch over ALL (row, step) pairs -- which spreads over the idle
// CUs -- and the scan keeps only the 48 h-part MFMAs per step: 1.9 -> 1.3 us per step.  gi holds, per memory row of x,
// [dir][gate r,z,n][64] pre-activations incl. the input-side biases, already in the exponent scaling of gru64_cell.
// At large batch the plain / fused kernels are used instead: there the MFMA count is what matters, not its latency,
// and gi would only add HBM traffic.
__global__ __launch_bounds__(256, 3) void gru64_scan_gi_kernel(Gru64Args a, const float* gi, int gw) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][68];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = blockIdx.y;
    const int row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;
    float whh[3][16];
    {
        const float* wp = a.wfrag + ((size_t)(dir * 4 + w) * 2) * 3 * 16 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 16; ++j) whh[g][j] = wp[(size_t)((1 * 3 + g) * 16 + j) * 64];
    }
    const float b_hn = a.bias[(size_t)dir * 256 + 192 + 16 * w + cl];
    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    float* obase = a.out + (long)hi0 * a.o_hi + (long)lo0 * a.o_lo + dir * a.o_dir_off;
    // gi is addressed like x: memory row of x = x offset / 64, gw floats per row
    const float* gbase = gi + ((long)hi0 * a.x_hi + (long)lo0 * a.x_lo) / 64 * gw + dir * 192 + 16 * w + cl;
    const long gstep = a.x_step / 64 * gw;
    const int srow = 4 * w + q, scol = 4 * cl;
    unsigned so_off; bool so_ok;
    {
        int rs = row0 + srow;
        so_ok = rs < a.nrows;
        if (rs >= a.nrows) rs = a.nrows - 1;
        so_off = (unsigned)((long)(rs / a.rdiv - hi0) * a.o_hi + (long)(rs % a.rdiv - lo0) * a.o_lo) + scol;
    }
    unsigned g_off[4];
    float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rc = row0 + q * 4 + i;
        if (rc >= a.nrows) rc = a.nrows - 1;
        g_off[i] = (unsigned)(((long)(rc / a.rdiv - hi0) * a.x_hi + (long)(rc % a.rdiv - lo0) * a.x_lo) / 64 * gw);
        float hv = a.hstate ? a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] : 0.f;
        h_own[i] = hv;
        Hs[1][q * 4 + i][16 * w + cl] = hv;
    }
    float gr[4], gz[4], gn[4];
    {
        const float* gp = gbase + (long)(dir ? a.nsteps - 1 : 0) * gstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) { gr[i] = gp[g_off[i]]; gz[i] = gp[g_off[i] + 64]; gn[i] = gp[g_off[i] + 128]; }
    }
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < a.nsteps; ++s) {
        if (s > 0) {
            const int pp = dir ? a.nsteps - s : s - 1;
            const float4 hv4 = *(const float4*)&Hs[buf ^ 1][srow][scol];
            if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
        }
        f32x4 ar = {gr[0], gr[1], gr[2], gr[3]}, az = {gz[0], gz[1], gz[2], gz[3]};
        f32x4 axn = {gn[0], gn[1], gn[2], gn[3]}, ahn = {b_hn, b_hn, b_hn, b_hn};
        {   // next step's input-side pre-activations (independent of h)
            const int sn = s + 1 < a.nsteps ? s + 1 : s;
            const float* gp = gbase + (long)(dir ? a.nsteps - 1 - sn : sn) * gstep;
#pragma unroll
            for (int i = 0; i < 4; ++i) { gr[i] = gp[g_off[i]]; gz[i] = gp[g_off[i] + 64]; gn[i] = gp[g_off[i] + 128]; }
        }
        // keep the loads HERE, in front of the MFMA block: left alone the scheduler sinks them into the gate math at the
        // end of the step and then waits for them there -- their L2 round trip in the dependent chain of every step
        // (scan launch of 48 steps 61 -> ~50 us; 1 clip x 10 s 9.9 -> 9.4 ms)
        __builtin_amdgcn_sched_barrier(0);
        const float* hrow = &Hs[buf ^ 1][cl][4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ar = mfma16(hv[kb], whh[0][c * 4 + kb], ar);
                az = mfma16(hv[kb], whh[1][c * 4 + kb], az);
                ahn = mfma16(hv[kb], whh[2][c * 4 + kb], ahn);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
            h_own[i] = h;
            Hs[buf][q * 4 + i][16 * w + cl] = h;
        }
        __syncthreads();
        buf ^= 1;
    }
    if (a.nsteps > 0) {
        const int pp = dir ? 0 : a.nsteps - 1;
        const float4 hv4 = *(const float4*)&Hs[buf ^ 1][srow][scol];
        if (so_ok) *(float4*)((obase + (long)pp * a.o_step) + so_off) = hv4;
    }
    if (a.hstate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int rc = row0 + q * 4 + i;
            if (rc < a.nrows)
                a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] = h_own[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct Gru256Args {
    const float* gi;       // [B*Tc][768]: W_ih x + (b_ir+b_hr | b_iz+b_hz | b_in)
    float* out;            // [B*Tc][256]
    const float* whh_frag; // [wave 16][gate 3][chunk 16][kb 4][lane 64]
    const float* b_hn;     // [256]
    float* hstate;         // h[b*h_stride + j] (flat reference state), in/out
    long h_stride;
    int B, Tc;
};

__global__ __launch_bounds__(1024) void gru256_scan_kernel(Gru256Args a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const float* wf = a.whh_frag + (size_t)w * 3 * 64 * 64 + lane;
    const float bhn = a.b_hn[16 * w + cl];

    int rc[4]; bool ok[4]; float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = row0 + q * 4 + i;
        ok[i] = r < a.B;
        rc[i] = ok[i] ? r : a.B - 1;
        h_own[i] = a.hstate[(long)rc[i] * a.h_stride + 16 * w + cl];
        Hs[0][q * 4 + i][16 * w + cl] = h_own[i];
    }
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < a.Tc; ++t) {
        f32x4 ar, az, axn, ahn = {bhn, bhn, bhn, bhn};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* g = a.gi + ((size_t)rc[i] * a.Tc + t) * 768 + 16 * w + cl;
            ar[i] = g[0]; az[i] = g[256]; axn[i] = g[512];
        }
        const float* hrow = &Hs[buf][cl][4 * q];
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ar = mfma16(hv[kb], wf[(size_t)((0 * 16 + c) * 4 + kb) * 64], ar);
                az = mfma16(hv[kb], wf[(size_t)((1 * 16 + c) * 4 + kb) * 64], az);
                ahn = mfma16(hv[kb], wf[(size_t)((2 * 16 + c) * 4 + kb) * 64], ahn);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = sigmoid_f(ar[i]);
            float z = sigmoid_f(az[i]);
            float n = gru_candidate(r, ahn[i], axn[i]);
            float h = gru_blend(z, n, h_own[i]);
            h_own[i] = h;
            Hs[buf ^ 1][q * 4 + i][16 * w + cl] = h;
            if (ok[i]) a.out[((size_t)rc[i] * a.Tc + t) * 256 + 16 * w + cl] = h;
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ok[i]) a.hstate[(long)rc[i] * a.h_stride + 16 * w + cl] = h_own[i];
}

// ---------------------------------------------------------------------------------------------
// gru256_cluster_kernel: the same recurrence spread over a CLUSTER of 4 workgroups per 16-row tile
// so that W_hh never moves: workgroup j of a cluster owns hidden units [64j, 64j+64) of all three
// gates; each of its 4 waves keeps its 16 units' W_hh slice (3 x 256 x 16 floats = 192 VGPRs) in
// registers for the whole scan (one wave per SIMD).  Per step the only cross-workgroup traffic is
// h' itself: 16 x 64 floats per workgroup, published as 8-byte {epoch, value} granules with
// write-through (sc1) agent-scope stores and swept by the three peers with relaxed agent-scope
// loads -- the data IS the flag (MI355X_MICROARCH.md "handoff-1to1", cdna_hip_programming.md
// Guideline 16 recipe R2): placement-independent, no fences, no separate flag.  Epochs grow
// monotonically across launches (host-supplied base), so the granule buffer is never re-zeroed.
// Two slots (step parity) suffice: a peer can only be one step ahead.
// Spin-wait bookkeeping of the cluster scans.  A sweep that is still stale after ~1 s of polling (2^20 rounds) gives
// up: it raises the device error flag and the workgroup stops waiting for the rest of the launch (`dead`), and every
// other workgroup polls the flag every 256 rounds and does the same -- a lost peer turns into DPDF_E_RUNTIME within
// about a second instead of ~10 s per remaining step (the host reads the flag at the next synchronisation point).
__device__ __forceinline__ bool cluster_spin_expired(unsigned& spins, int* err, bool& dead) {
    ++spins;
    if ((spins & 255u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = true; return true; }
    if (spins > (1u << 20)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; return true; }
    return false;
}

struct Gru256CArgs {
    const float* gi;       // [B*Tc][768]
    float* out;            // [B*Tc][256]
    const float* whh_frag; // [j 4][wave 4][gate 3][chunk 16][kb 4][lane 64]
    const float* b_hn;     // [256]
    float* hstate; long h_stride;
    int B, Tc;
    unsigned long long* xbuf;   // [tiles][2][16][256] granules
    unsigned epoch_base;
    int* err;                   // set to 1 on spin timeout
};

__global__ __launch_bounds__(256, 1) void gru256_cluster_kernel(Gru256CArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int ntiles = gridDim.x >> 2;
    int rt, j;
    if ((ntiles & 7) == 0) {       // keep a cluster on one XCD (block b -> XCD b % 8): speed only
        rt = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5);
        j = (blockIdx.x >> 3) & 3;
    } else {
        rt = blockIdx.x >> 2; j = blockIdx.x & 3;
    }
    const int row0 = rt * 16;
    const int u0 = 64 * j + 16 * w;            // first hidden unit of this wave

    float wr[64], wz[64], wn[64];
    {
        const float* wf = a.whh_frag + ((size_t)(j * 4 + w) * 3) * 64 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            wr[k] = wf[(size_t)(0 * 64 + k) * 64];
            wz[k] = wf[(size_t)(1 * 64 + k) * 64];
            wn[k] = wf[(size_t)(2 * 64 + k) * 64];
        }
    }
    const float bhn = a.b_hn[u0 + cl];
    int rc[4]; bool ok[4]; float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = row0 + q * 4 + i;
        ok[i] = r < a.B;
        rc[i] = ok[i] ? r : a.B - 1;
        h_own[i] = a.hstate[(long)rc[i] * a.h_stride + u0 + cl];
    }
    // full h of the tile into LDS (every workgroup of the cluster reads the carried state itself)
    for (int idx = tid; idx < 16 * 256; idx += 256) {
        int r = idx >> 8, u = idx & 255;
        int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        Hs[0][r][u] = a.hstate[(long)rr * a.h_stride + u];
    }
    __syncthreads();

    unsigned long long* xb = a.xbuf + (size_t)rt * 2 * 16 * 256;
    float gr[4], gz[4], gn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* g = a.gi + ((size_t)rc[i] * a.Tc) * 768 + u0 + cl;
        gr[i] = g[0]; gz[i] = g[256]; gn[i] = g[512];
    }
    int cur = 0;
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    for (int t = 0; t < a.Tc; ++t) {
        f32x4 ar = {gr[0], gr[1], gr[2], gr[3]}, az = {gz[0], gz[1], gz[2], gz[3]};
        f32x4 axn = {gn[0], gn[1], gn[2], gn[3]}, ahn = {bhn, bhn, bhn, bhn};
        if (t + 1 < a.Tc) {        // prefetch next step's input projections (independent of h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* g = a.gi + ((size_t)rc[i] * a.Tc + t + 1) * 768 + u0 + cl;
                gr[i] = g[0]; gz[i] = g[256]; gn[i] = g[512];
            }
        }
        const float* hrow = &Hs[cur][cl][4 * q];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ar = mfma16(hv[kb], wr[c * 4 + kb], ar);
                az = mfma16(hv[kb], wz[c * 4 + kb], az);
                ahn = mfma16(hv[kb], wn[c * 4 + kb], ahn);
            }
        }
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = xb + (size_t)(t & 1) * 16 * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = sigmoid_f(ar[i]);
            float z = sigmoid_f(az[i]);
            float n = gru_candidate(r, ahn[i], axn[i]);
            h_own[i] = gru_blend(z, n, h_own[i]);
        }
        // publish first (the peers' next step waits on these), then the local copies
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __hip_atomic_store(slot + (q * 4 + i) * 256 + u0 + cl,
                               ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h_own[i]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Hs[nxt][q * 4 + i][u0 + cl] = h_own[i];
            if (ok[i]) a.out[((size_t)rc[i] * a.Tc + t) * 256 + u0 + cl] = h_own[i];
        }
        // sweep the three peers' slices (3 x 16 rows x 64 units = 12 granules per thread): all 12 loads go
        // out back to back and are checked together -- one L2 round trip per sweep instead of twelve
        // dependent ones; a sweep that finds a stale granule is simply repeated.  Also after the LAST
        // step: a workgroup may overwrite the carried state (below) only once every peer has provably
        // consumed the old one, i.e. has published its own last step.
        {
            unsigned long long xv[12];
            unsigned spins = 0;
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const int idx = tid + 256 * k;
                    const int s = idx >> 10, r = (idx >> 6) & 15, u = 64 * ((j + 1 + s) & 3) + (idx & 63);
                    xv[k] = __hip_atomic_load(slot + r * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 12; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int idx = tid + 256 * k;
                const int s = idx >> 10, r = (idx >> 6) & 15, u = 64 * ((j + 1 + s) & 3) + (idx & 63);
                Hs[nxt][r][u] = __uint_as_float((unsigned)xv[k]);
            }
        }
        __syncthreads();
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ok[i]) a.hstate[(long)rc[i] * a.h_stride + u0 + cl] = h_own[i];
}

// ---------------------------------------------------------------------------------------------
// gru256_cluster8_kernel: the same cluster scan on EIGHT workgroups per 16-row tile, for launches that leave most of
// the chip idle anyway (<= 4 tiles = 64 streams; beyond that the doubled CU footprint costs more than it buys).  A step of the 4-workgroup form is 2.6 us of dependent MFMAs (192 per wave) plus
// one L2 exchange; here workgroup j owns 32 hidden units and its four waves split them 2 (unit halves) x 2 (K
// halves): 96 MFMAs per wave per step, the two K-halves are summed through LDS, and each wave finishes two of the four
// C-layout rows of its 16x16 block.  Same granule protocol, same exchange buffer layout, same W_hh fragment packing.
__global__ __launch_bounds__(256, 1) void gru256_cluster8_kernel(Gru256CArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    __shared__ float Ps[4][3][4][64];            // per wave: partial pre-activations [gate][C-layout row i][lane]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int uh = w & 1, kh = w >> 1;
    const int ntiles = gridDim.x >> 3;
    int rt, j;
    if ((ntiles & 7) == 0) {       // keep a cluster on one XCD (block b -> XCD b % 8): speed only
        rt = (blockIdx.x & 7) + 8 * (blockIdx.x >> 6);
        j = (blockIdx.x >> 3) & 7;
    } else {
        rt = blockIdx.x >> 3; j = blockIdx.x & 7;
    }
    const int row0 = rt * 16;
    const int u0 = 32 * j + 16 * uh;           // first hidden unit of this wave's column block

    float wr[32], wz[32], wn[32];              // K half kh: chunks [8 kh, 8 kh + 8)
    {
        const float* wf = a.whh_frag + ((size_t)(2 * j + uh) * 3) * 64 * 64 + (size_t)(32 * kh) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            wr[k] = wf[(size_t)(0 * 64 + k) * 64];
            wz[k] = wf[(size_t)(1 * 64 + k) * 64];
            wn[k] = wf[(size_t)(2 * 64 + k) * 64];
        }
    }
    const float bhn = a.b_hn[u0 + cl];
    // this wave finalises C-layout rows i = 2 kh + {0, 1} of its block: tile rows q*4 + 2 kh + e
    int rc[2]; bool ok[2]; float h_own[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int r = row0 + q * 4 + 2 * kh + e;
        ok[e] = r < a.B;
        rc[e] = ok[e] ? r : a.B - 1;
        h_own[e] = a.hstate[(long)rc[e] * a.h_stride + u0 + cl];
    }
    for (int idx = tid; idx < 16 * 256; idx += 256) {
        int r = idx >> 8, u = idx & 255;
        int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        Hs[0][r][u] = a.hstate[(long)rr * a.h_stride + u];
    }
    __syncthreads();

    unsigned long long* xb = a.xbuf + (size_t)rt * 2 * 16 * 256;
    float gr[2], gz[2], gn[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float* g = a.gi + ((size_t)rc[e] * a.Tc) * 768 + u0 + cl;
        gr[e] = g[0]; gz[e] = g[256]; gn[e] = g[512];
    }
    int cur = 0;
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    for (int t = 0; t < a.Tc; ++t) {
        f32x4 pr = {0.f, 0.f, 0.f, 0.f}, pz = pr, pn = pr;
        const float* hrow = &Hs[cur][cl][128 * kh + 4 * q];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(hv[kb], wr[c * 4 + kb], pr);
                pz = mfma16(hv[kb], wz[c * 4 + kb], pz);
                pn = mfma16(hv[kb], wn[c * 4 + kb], pn);
            }
        }
        // hand the two rows the partner wave (same units, other K half) finalises over to it
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * (1 - kh) + e;
            Ps[w][0][i][lane] = pr[i]; Ps[w][1][i][lane] = pz[i]; Ps[w][2][i][lane] = pn[i];
        }
        const float xr[2] = {gr[0], gr[1]}, xz[2] = {gz[0], gz[1]}, xn[2] = {gn[0], gn[1]};
        if (t + 1 < a.Tc) {        // prefetch next step's input projections (independent of h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float* g = a.gi + ((size_t)rc[e] * a.Tc + t + 1) * 768 + u0 + cl;
                gr[e] = g[0]; gz[e] = g[256]; gn[e] = g[512];
            }
        }
        __syncthreads();
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = xb + (size_t)(t & 1) * 16 * 256;
        const int pw = w ^ 2;                   // partner: same unit half, other K half
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * kh + e;
            const float ar = xr[e] + (pr[i] + Ps[pw][0][i][lane]);
            const float az = xz[e] + (pz[i] + Ps[pw][1][i][lane]);
            const float ahn = bhn + (pn[i] + Ps[pw][2][i][lane]);
            float r = sigmoid_f(ar);
            float z = sigmoid_f(az);
            float n = gru_candidate(r, ahn, xn[e]);
            h_own[e] = gru_blend(z, n, h_own[e]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
            __hip_atomic_store(slot + (q * 4 + 2 * kh + e) * 256 + u0 + cl,
                               ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h_own[e]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Hs[nxt][q * 4 + 2 * kh + e][u0 + cl] = h_own[e];
            if (ok[e]) a.out[((size_t)rc[e] * a.Tc + t) * 256 + u0 + cl] = h_own[e];
        }
        // sweep the seven peers' slices (7 x 16 rows x 32 units = 14 granules per thread), all loads in one batch;
        // also after the LAST step (see gru256_cluster_kernel)
        {
            unsigned long long xv[14];
            unsigned spins = 0;
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 14; ++k) {
                    const int idx = tid + 256 * k;
                    const int s = idx >> 9, r = (idx >> 5) & 15, u = 32 * ((j + 1 + s) & 7) + (idx & 31);
                    xv[k] = __hip_atomic_load(slot + r * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 14; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                const int idx = tid + 256 * k;
                const int s = idx >> 9, r = (idx >> 5) & 15, u = 32 * ((j + 1 + s) & 7) + (idx & 31);
                Hs[nxt][r][u] = __uint_as_float((unsigned)xv[k]);
            }
        }
        __syncthreads();
        cur = nxt;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (ok[e]) a.hstate[(long)rc[e] * a.h_stride + u0 + cl] = h_own[e];
}

// ---------------------------------------------------------------------------------------------
// gru256_cluster16_kernel: SIXTEEN workgroups per 16-row tile, for launches of one or two tiles (a single clip through
// enhance(), <= 32 streams): there the step is pure latency and the cross-CU exchange (~2 us) is a fixed price, so the
// dependent MFMA chain is cut as short as the tile shape allows.  Workgroup j owns 16 hidden units (one MFMA column
// tile); its four waves split K four ways -- 48 MFMAs per wave per step (0.64 us) instead of 96 (cluster8) or 192
// (cluster) -- the four partial sums are combined through LDS, wave w finishing C-layout row w of the 16x16 block.
// Same granule protocol, exchange buffer layout and W_hh packing as gru256_cluster_kernel.
__global__ __launch_bounds__(256, 1) void gru256_cluster16_kernel(Gru256CArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    __shared__ float Ps[4][3][4][64];            // per wave: partial pre-activations [gate][C-layout row i][lane]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int rt = blockIdx.x >> 4, j = blockIdx.x & 15;
    const int row0 = rt * 16;
    const int u0 = 16 * j;                      // the workgroup's hidden units

    float wr[16], wz[16], wn[16];               // K quarter w: chunks [4 w, 4 w + 4)
    {
        const float* wf = a.whh_frag + ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            wr[k] = wf[(size_t)(0 * 64 + k) * 64];
            wz[k] = wf[(size_t)(1 * 64 + k) * 64];
            wn[k] = wf[(size_t)(2 * 64 + k) * 64];
        }
    }
    const float bhn = a.b_hn[u0 + cl];
    // this wave finalises C-layout row i = w of the block: tile row q*4 + w
    const int r_own = row0 + q * 4 + w;
    const bool ok = r_own < a.B;
    const int rc = ok ? r_own : a.B - 1;
    float h_own = a.hstate[(long)rc * a.h_stride + u0 + cl];
    for (int idx = tid; idx < 16 * 256; idx += 256) {
        int r = idx >> 8, u = idx & 255;
        int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        Hs[0][r][u] = a.hstate[(long)rr * a.h_stride + u];
    }
    __syncthreads();

    unsigned long long* xb = a.xbuf + (size_t)rt * 2 * 16 * 256;
    float gr, gz, gn;
    {
        const float* g = a.gi + ((size_t)rc * a.Tc) * 768 + u0 + cl;
        gr = g[0]; gz = g[256]; gn = g[512];
    }
    int cur = 0;
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    for (int t = 0; t < a.Tc; ++t) {
        f32x4 pr = {0.f, 0.f, 0.f, 0.f}, pz = pr, pn = pr;
        const float* hrow = &Hs[cur][cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(hv[kb], wr[c * 4 + kb], pr);
                pz = mfma16(hv[kb], wz[c * 4 + kb], pz);
                pn = mfma16(hv[kb], wn[c * 4 + kb], pn);
            }
        }
        // every wave's K-quarter partials through LDS (its own row too: the quarters are then added in the SAME order for
        // every row, so that identical clips in different batch slots stay bit-identical)
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ps[w][0][i][lane] = pr[i]; Ps[w][1][i][lane] = pz[i]; Ps[w][2][i][lane] = pn[i]; }
        const float xr = gr, xz = gz, xn = gn;
        if (t + 1 < a.Tc) {        // prefetch next step's input projections (independent of h)
            const float* g = a.gi + ((size_t)rc * a.Tc + t + 1) * 768 + u0 + cl;
            gr = g[0]; gz = g[256]; gn = g[512];
        }
        __syncthreads();
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = xb + (size_t)(t & 1) * 16 * 256;
        {
            float sr = Ps[0][0][w][lane], sz = Ps[0][1][w][lane], sn = Ps[0][2][w][lane];
#pragma unroll
            for (int k = 1; k < 4; ++k) { sr += Ps[k][0][w][lane]; sz += Ps[k][1][w][lane]; sn += Ps[k][2][w][lane]; }
            const float r = sigmoid_f(xr + sr);
            const float z = sigmoid_f(xz + sz);
            const float n = gru_candidate(r, bhn + sn, xn);
            h_own = gru_blend(z, n, h_own);
        }
        __hip_atomic_store(slot + (q * 4 + w) * 256 + u0 + cl,
                           ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h_own),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Hs[nxt][q * 4 + w][u0 + cl] = h_own;
        if (ok) a.out[((size_t)rc * a.Tc + t) * 256 + u0 + cl] = h_own;
        // sweep the fifteen peers' slices (15 x 16 rows x 16 units = 15 granules per thread), all loads in one batch;
        // also after the LAST step (see gru256_cluster_kernel)
        {
            unsigned long long xv[15];
            unsigned spins = 0;
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    const int r = tid >> 4, u = 16 * ((j + 1 + k) & 15) + (tid & 15);
                    xv[k] = __hip_atomic_load(slot + r * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 15; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const int r = tid >> 4, u = 16 * ((j + 1 + k) & 15) + (tid & 15);
                Hs[nxt][r][u] = __uint_as_float((unsigned)xv[k]);
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (ok) a.hstate[(long)rc * a.h_stride + u0 + cl] = h_own;
}

// ---------------------------------------------------------------------------------------------
// gru64_epi_kernel<EPI>: the same scan with the DPRNN block's Linear + LayerNorm + residual
// (reference onnx_model/layers.py:178-181, 190-193) fused in, so neither the GRU outputs (h
// sequences) of the inter-band scan nor the fwd|bwd concatenation of the intra-band scan make a
// round trip through HBM, and the separate fc+LN kernels disappear.
//   EPI = 1  inter-band scan:      y(s) = x(s) + LN(W_fc h'(s) + b)
//   EPI = 2  intra-band BACKWARD:  y(p) = x(p) + LN(W_f hf(p) + W_b hb(p) + b), hf from the forward
//            scan (run first, plain gru64_scan_kernel), p = position of step s
// Everything is row-local.  The fc product rides on the h-part MFMAs of the NEXT step (same A
// fragments of h'), lands in an LDS tile in MFMA C layout, and is normalised one barrier later by
// the row-contiguous lanes (16 lanes x 4 values = one 64-channel row: LayerNorm statistics are two
// 4-step butterflies inside a 16-lane group), which also hold the residual piece and issue the one
// coalesced 16-byte store per lane.  Output therefore lags the recurrence by two steps; the loop
// runs nsteps + 2 iterations with wave-uniform guards.
struct Gru64EpiArgs {
    Gru64Args g;            // g.out unused
    const float* fc_frag;   // [part][wave 4][16][lane 64]: part 0 = W_fc columns fed by h' (inter: the only part;
                            //                               intra: the bwd half), part 1 = the fwd half (EPI 2)
    const float* fc_bias;   // [64]
    const float* ln_g; const float* ln_b;   // [64]
    const float* extra;     // EPI 2: hf tensor, addressed like x
    float* y;               // output, addressed like x
};

#ifndef EPI1_WF_LDS
#define EPI1_WF_LDS 1     // 1: inter-band fc fragments in LDS, 3 WG/CU; 0: in registers, 2 WG/CU
#endif
template <int EPI>
__global__ __launch_bounds__(256, (EPI == 1 && EPI1_WF_LDS) ? 3 : 2) void gru64_epi_kernel(Gru64EpiArgs ea) {
    const Gru64Args& a = ea.g;
    __shared__ __attribute__((aligned(16))) float Hs[2][16][68];
    __shared__ __attribute__((aligned(16))) float Xs[4][16][68];      // ring: x(s-2) must outlive x(s+1)'s staging
    __shared__ __attribute__((aligned(16))) float Ys[2][16][68];
    __shared__ __attribute__((aligned(16))) float Es[EPI == 2 ? 4 : 1][EPI == 2 ? 16 : 1][EPI == 2 ? 68 : 4];   // hf ring (3 live slots; 4 so that every ring index is s & const)
    constexpr bool WF_LDS = EPI == 1 && EPI1_WF_LDS;
    __shared__ float Wf[WF_LDS ? 4 : 1][16][64];                      // EPI 1: fc B fragments in LDS (keeps 3 WG/CU)
    __shared__ __attribute__((aligned(16))) float Lp[3][64];          // fc bias | ln gamma | ln beta
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = EPI == 2 ? 1 : 0;                                   // EPI 2 is the backward direction
    const int row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;

    float wih[3][16], whh[3][16];
    {
        const float* wp = a.wfrag + ((size_t)(dir * 4 + w) * 2) * 3 * 16 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                wih[g][j] = wp[(size_t)((0 * 3 + g) * 16 + j) * 64];
                whh[g][j] = wp[(size_t)((1 * 3 + g) * 16 + j) * 64];
            }
    }
    const float* bp = a.bias + (size_t)dir * 256 + 16 * w + cl;
    const float b_r = bp[0], b_z = bp[64], b_in = bp[128], b_hn = bp[192];
    float wfc[EPI == 2 ? 2 : 1][WF_LDS ? 1 : 16];                     // otherwise in registers (2 WG/CU anyway)
    if (WF_LDS) {
        for (int i = tid; i < 4 * 16 * 64; i += 256) (&Wf[0][0][0])[i] = ea.fc_frag[i];
    } else {
#pragma unroll
        for (int pt = 0; pt < (EPI == 2 ? 2 : 1); ++pt)
#pragma unroll
            for (int j = 0; j < 16; ++j) wfc[pt][j] = ea.fc_frag[((size_t)(pt * 4 + w) * 16 + j) * 64 + lane];
    }
    if (tid < 64) { Lp[0][tid] = ea.fc_bias[tid]; Lp[1][tid] = ea.ln_g[tid]; Lp[2][tid] = ea.ln_b[tid]; }

    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    const float* xbase = a.x + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    const float* ebase = EPI == 2 ? ea.extra + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo : nullptr;
    float* ybase = ea.y + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    const int srow = 4 * w + q, scol = 4 * cl;
    // non-negative 32-bit lane offset: every global address is (wave-uniform base advanced on the scalar unit) + sx_off,
    // so the step costs no 64-bit VALU address arithmetic -- VALU cycles come straight out of the MFMA budget (docs/HISTORY.md 3)
    unsigned sx_off; bool so_ok;
    {
        int rs = row0 + srow;
        so_ok = rs < a.nrows;
        if (rs >= a.nrows) rs = a.nrows - 1;
        sx_off = (unsigned)((long)(rs / a.rdiv - hi0) * a.x_hi + (long)(rs % a.rdiv - lo0) * a.x_lo) + scol;
    }
    float h_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rc = row0 + q * 4 + i;
        if (rc >= a.nrows) rc = a.nrows - 1;
        float hv = a.hstate ? a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] : 0.f;
        h_own[i] = hv;
        Hs[1][q * 4 + i][16 * w + cl] = hv;
    }
    const int n = a.nsteps;
    auto pos_of = [&](int s) { s = s < n ? s : n - 1; return dir ? n - 1 - s : s; };    // clamped: the tail re-reads the last tile instead of branching
    {
        *(float4*)&Xs[0][srow][scol] = *(const float4*)((xbase + (long)pos_of(0) * a.x_step) + sx_off);
        if (EPI == 2) *(float4*)&Es[0][srow][scol] = *(const float4*)((ebase + (long)pos_of(0) * a.x_step) + sx_off);   // Es(s) lives in slot s & 3
    }
    __syncthreads();

    for (int s = 0; s < n + 2; ++s) {
        const int hb = s & 1;                 // Hs/Ys/Es slot written this step
        // ---- finalize step s-2: LayerNorm + residual on the row-contiguous pieces, one store per lane
        auto finalize = [&]() __attribute__((always_inline)) {
        if (s >= 2) {
                const float4 yv = *(const float4*)&Ys[hb][srow][scol];            // fc(s-2) + bias, written during step s-1
                const float4 rv = *(const float4*)&Xs[(s - 2) & 3][srow][scol];   // residual x(s-2)
                const float mean = row16_allreduce_sum(yv.x + yv.y + yv.z + yv.w) * (1.0f / 64.0f);
                const float d0 = yv.x - mean, d1 = yv.y - mean, d2 = yv.z - mean, d3 = yv.w - mean;
                const float s2 = row16_allreduce_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
                const float inv = rsqrtf(s2 * (1.0f / 64.0f) + 1e-5f);
                const float4 gg = *(const float4*)&Lp[1][scol], bb = *(const float4*)&Lp[2][scol];
                float4 o;
                o.x = rv.x + d0 * inv * gg.x + bb.x; o.y = rv.y + d1 * inv * gg.y + bb.y;
                o.z = rv.z + d2 * inv * gg.z + bb.z; o.w = rv.w + d3 * inv * gg.w + bb.w;
                if (so_ok) *(float4*)((ybase + (long)pos_of(s - 2) * a.x_step) + sx_off) = o;
            }
        };
        if (EPI == 1) finalize();
        // ---- loads for step s+1
        const float4 xnext = *(const float4*)((xbase + (long)pos_of(s + 1) * a.x_step) + sx_off);
        float4 enext = xnext;
        if (EPI == 2) enext = *(const float4*)((ebase + (long)pos_of(s + 1) * a.x_step) + sx_off);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 ar = {b_r, b_r, b_r, b_r}, az = {b_z, b_z, b_z, b_z};
        f32x4 axn = {b_in, b_in, b_in, b_in}, ahn = {b_hn, b_hn, b_hn, b_hn};
        f32x4 ay = {0.f, 0.f, 0.f, 0.f};
        if (s < n) {
            const float* xrow = &Xs[s & 3][cl][4 * q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 x4 = *(const float4*)(xrow + 16 * c);
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    ar = mfma16(xv[kb], wih[0][c * 4 + kb], ar);
                    az = mfma16(xv[kb], wih[1][c * 4 + kb], az);
                    axn = mfma16(xv[kb], wih[2][c * 4 + kb], axn);
                }
            }
        }
        if (s <= n) {
            // h-part of step s and fc of step s-1 share the A fragments of h'(s-1)
            const float* hrow = &Hs[hb ^ 1][cl][4 * q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 h4 = *(const float4*)(hrow + 16 * c);
                const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    ar = mfma16(hv[kb], whh[0][c * 4 + kb], ar);
                    az = mfma16(hv[kb], whh[1][c * 4 + kb], az);
                    ahn = mfma16(hv[kb], whh[2][c * 4 + kb], ahn);
                    ay = mfma16(hv[kb], WF_LDS ? Wf[w][c * 4 + kb][lane] : wfc[0][WF_LDS ? 0 : c * 4 + kb], ay);
                }
            }
            if (EPI == 2) {
                const float* erow = &Es[(s + 3) & 3][cl][4 * q];             // slot (s-1) & 3: hf at the position of step s-1
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 e4 = *(const float4*)(erow + 16 * c);
                    const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) ay = mfma16(ev[kb], wfc[EPI == 2 ? 1 : 0][c * 4 + kb], ay);
                }
            }
        }
        if (EPI == 2) finalize();          // behind the MFMA block: its latency hides under it (registers allow it at 2 WG/CU)
        if (s < n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
                h_own[i] = h;
                Hs[hb][q * 4 + i][16 * w + cl] = h;
            }
        }
        if (s >= 1 && s <= n) {
            const float fb = Lp[0][16 * w + cl];
#pragma unroll
            for (int i = 0; i < 4; ++i) Ys[hb ^ 1][q * 4 + i][16 * w + cl] = ay[i] + fb;    // fc(s-1): read at step s+1 from Ys[(s+1)&1]
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < n) {
            *(float4*)&Xs[(s + 1) & 3][srow][scol] = xnext;
            if (EPI == 2) *(float4*)&Es[(s + 1) & 3][srow][scol] = enext;  // read during step s+2; slot (s-1)&3 is the one in use now
        }
        __syncthreads();
    }
    if (a.hstate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int rc = row0 + q * 4 + i;
            if (rc < a.nrows)
                a.hstate[(long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + 16 * w + cl] = h_own[i];
        }
    }
}
