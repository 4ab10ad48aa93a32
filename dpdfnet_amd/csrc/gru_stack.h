// gru256_stack16_kernel: TWO stacked GRUCell(256) layers (reference onnx_model/layers.py:1168-1259: the decoders'
// `emb_gru` / `df_gru` stacks, where the second cell's input IS the first cell's hidden state of the same frame) as ONE
// launch for small launches (one or two 16-row tiles: a single clip through enhance(), <= 32 live streams).
//
// There the five GRU-256 scans of stage 2 are pure dependent-step latency, three scans deep (embedding cell -> decoder
// cell 0 -> decoder cell 1): one 10 s clip = 3 x 1003 steps x 2.7 us.  Run as a wavefront the two decoder cells cost the
// steps of ONE scan plus a lag of two frames: while cell A works on frame t+1, cell B works on frame t.
//   * workgroups [0, 16) of a tile are cell A -- gru256_cluster16_kernel's scheme unchanged (16 hidden units per
//     workgroup, K split over the four waves, W_hh quarter in registers, input projection hoisted into a GEMM) except
//     that h_A(t) is published into a ring with one slot PER FRAME of the chunk, so that A never waits for B;
//   * workgroups [16, 32) are cell B: W_ih^B and W_hh^B quarters in registers (96 VGPRs).  Per frame t a B wave adds
//     the recurrent partial W_hh^B h_B(t-1) onto the input partial W_ih^B h_A(t) it computed one iteration earlier --
//     behind the ISSUE of its granule sweep and in front of its completion, i.e. under the exchange round trip -- so a
//     B step costs what an A step costs.
// Granules are 8-byte {epoch, value} words written with relaxed agent-scope stores and swept with relaxed agent-scope
// loads (the data is the flag; gru_scan.h).  A's ring: [tile][Tc][16 rows][256]; B's exchange: [tile][2][16][256].
// Partial sums of the four K quarters go through LDS and are added in one fixed order, so identical clips in different
// batch slots stay bit-identical.  All 32 workgroups of a tile must be co-resident (<= 64 workgroups on 256 CUs).
#pragma once
#include "gru_scan.h"

struct Gru256SArgs {
    const float* gi;            // cell A: hoisted input projection [B*Tc][768] (biases folded in)
    float* out_a; float* out_b; // [B*Tc][256] hidden sequences (out_a may be null)
    const float* whh_a; const float* bhn_a;                       // cell A recurrent fragments / b_hn
    const float* wih_b; const float* whh_b;                       // cell B fragments, both packed like whh ([j 16][gate 3][k-chunk 64][lane 64])
    const float* bias_b;        // [768]: b_ih + b_hh (r, z) | b_in (n)
    const float* bhn_b;         // [256]
    float* hstate_a; float* hstate_b; long h_stride;
    int B, Tc;
    unsigned long long* ring_a; // [tiles][Tc][16][256]
    unsigned long long* xbuf_b; // [tiles][2][16][256]
    unsigned epoch_base;
    int* err;
};

__global__ __launch_bounds__(256, 1) void gru256_stack16_kernel(Gru256SArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];      // own cell's h (double buffer)
    __shared__ __attribute__((aligned(16))) float Ha[2][16][260];      // cell B: h_A of frames t+1 / t+2 (by frame parity)
    __shared__ float Ps[4][4][4][64];           // per wave: partial pre-activations [r | z | n_x | n_h][C-layout row i][lane]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int rt = blockIdx.x >> 5, role_b = (blockIdx.x >> 4) & 1, j = blockIdx.x & 15;
    const int row0 = rt * 16;
    const int u0 = 16 * j;                      // the workgroup's hidden units
    const int Tc = a.Tc;

    float wr[16], wz[16], wn[16];               // recurrent fragments, K quarter w
    float xr_[16], xz_[16], xn_[16];            // cell B: input fragments, K quarter w
    {
        const float* wf = (role_b ? a.whh_b : a.whh_a) + ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            wr[k] = wf[(size_t)(0 * 64 + k) * 64];
            wz[k] = wf[(size_t)(1 * 64 + k) * 64];
            wn[k] = wf[(size_t)(2 * 64 + k) * 64];
        }
        const float* xf = a.wih_b + ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            xr_[k] = role_b ? xf[(size_t)(0 * 64 + k) * 64] : 0.f;
            xz_[k] = role_b ? xf[(size_t)(1 * 64 + k) * 64] : 0.f;
            xn_[k] = role_b ? xf[(size_t)(2 * 64 + k) * 64] : 0.f;
        }
    }
    float* hstate = role_b ? a.hstate_b : a.hstate_a;
    float* out = role_b ? a.out_b : a.out_a;
    const float bhn = (role_b ? a.bhn_b : a.bhn_a)[u0 + cl];
    const float b_r = role_b ? a.bias_b[u0 + cl] : 0.f, b_z = role_b ? a.bias_b[256 + u0 + cl] : 0.f, b_n = role_b ? a.bias_b[512 + u0 + cl] : 0.f;
    // this wave finalises C-layout row i = w of the block: tile row q*4 + w
    const int r_own = row0 + q * 4 + w;
    const bool ok = r_own < a.B;
    const int rc = ok ? r_own : a.B - 1;
    float h_own = hstate[(long)rc * a.h_stride + u0 + cl];
    for (int idx = tid; idx < 16 * 256; idx += 256) {
        int r = idx >> 8, u = idx & 255;
        int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        Hs[0][r][u] = hstate[(long)rr * a.h_stride + u];
    }
    unsigned long long* ring = a.ring_a + (size_t)rt * Tc * 16 * 256;
    unsigned long long* xb = a.xbuf_b + (size_t)rt * 2 * 16 * 256;
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    const int sr_ = tid >> 4, su_ = tid & 15;           // sweep assignment: row, unit-in-slice

    // blocking sweep of all 16 slices of h_A(frame f) into Ha[f & 1]
    auto fetch_a = [&](int f) {
        const unsigned ep = a.epoch_base + (unsigned)f + 1u;
        const unsigned long long* slot = ring + (size_t)f * 16 * 256;
        unsigned long long xv[16];
        unsigned spins = 0;
        for (;;) {
            bool all_in = true;
#pragma unroll
            for (int k = 0; k < 16; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 16; ++k) all_in &= (unsigned)(xv[k] >> 32) == ep;
            if (all_in) break;
            if (dead || cluster_spin_expired(spins, a.err, dead)) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) Ha[f & 1][sr_][16 * k + su_] = __uint_as_float((unsigned)xv[k]);
    };
    // cell B: input partials of frame f from Ha[f & 1] (K quarter of this wave)
    f32x4 pi_r = {0.f, 0.f, 0.f, 0.f}, pi_z = pi_r, pi_n = pi_r;
    auto ih_part = [&](int f) {
        pi_r = f32x4{0.f, 0.f, 0.f, 0.f}; pi_z = pi_r; pi_n = pi_r;
        const float* arow = &Ha[f & 1][cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h4 = *(const float4*)(arow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pi_r = mfma16(hv[kb], xr_[c * 4 + kb], pi_r);
                pi_z = mfma16(hv[kb], xz_[c * 4 + kb], pi_z);
                pi_n = mfma16(hv[kb], xn_[c * 4 + kb], pi_n);
            }
        }
    };

    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (!role_b) {
        const float* g = a.gi + ((size_t)rc * Tc) * 768 + u0 + cl;
        gr = g[0]; gz = g[256]; gn = g[512];
    } else {
        fetch_a(0);
        if (Tc > 1) fetch_a(1);
    }
    __syncthreads();
    if (role_b) ih_part(0);

    unsigned long long av[16];        // cell B: granules of h_A two frames ahead, in flight across a whole step
#pragma unroll
    for (int k = 0; k < 16; ++k) av[k] = 0ull;
    if (role_b && Tc > 2) {
        const unsigned long long* slot_a = ring + (size_t)2 * 16 * 256;
#pragma unroll
        for (int k = 0; k < 16; ++k) av[k] = __hip_atomic_load(slot_a + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int cur = 0;
    for (int t = 0; t < Tc; ++t) {
        // ---- recurrent part of frame t (cell B: on top of the input partials)
        f32x4 pr = pi_r, pz = pi_z, pn = {0.f, 0.f, 0.f, 0.f};
        const float* hrow = &Hs[cur][cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(hv[kb], wr[c * 4 + kb], pr);
                pz = mfma16(hv[kb], wz[c * 4 + kb], pz);
                pn = mfma16(hv[kb], wn[c * 4 + kb], pn);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ps[w][0][i][lane] = pr[i]; Ps[w][1][i][lane] = pz[i]; Ps[w][2][i][lane] = pi_n[i]; Ps[w][3][i][lane] = pn[i]; }
        const float xr = role_b ? b_r : gr, xz = role_b ? b_z : gz, xn = role_b ? b_n : gn;
        if (!role_b && t + 1 < Tc) {        // cell A: prefetch next step's input projections (independent of h)
            const float* g = a.gi + ((size_t)rc * Tc + t + 1) * 768 + u0 + cl;
            gr = g[0]; gz = g[256]; gn = g[512];
        }
        __syncthreads();
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = role_b ? xb + (size_t)(t & 1) * 16 * 256 : ring + (size_t)t * 16 * 256;
        {
            float sr = Ps[0][0][w][lane], sz = Ps[0][1][w][lane], sx = Ps[0][2][w][lane], sn = Ps[0][3][w][lane];
#pragma unroll
            for (int k = 1; k < 4; ++k) { sr += Ps[k][0][w][lane]; sz += Ps[k][1][w][lane]; sx += Ps[k][2][w][lane]; sn += Ps[k][3][w][lane]; }
            const float r = sigmoid_f(xr + sr);
            const float z = sigmoid_f(xz + sz);
            const float n = gru_candidate(r, bhn + sn, xn + sx);
            h_own = gru_blend(z, n, h_own);
        }
        __hip_atomic_store(slot + (q * 4 + w) * 256 + u0 + cl,
                           ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h_own),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Hs[nxt][q * 4 + w][u0 + cl] = h_own;
        if (ok && out) out[((size_t)rc * Tc + t) * 256 + u0 + cl] = h_own;
        // ---- sweep: the fifteen peers' slices of this cell's h(t), loads issued in one batch; cell B computes the input
        // partials of frame t+1 (h_A(t+1) is already in LDS) between their issue and their completion.
        {
            unsigned long long xv[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * ((j + 1 + k) & 15) + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
            if (role_b && t + 1 < Tc) ih_part(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            unsigned spins = 0;
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 15; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < 15; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * ((j + 1 + k) & 15) + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) Hs[nxt][sr_][16 * ((j + 1 + k) & 15) + su_] = __uint_as_float((unsigned)xv[k]);
        }
        // ---- cell B: h_A(t+2), whose loads were issued a whole step ago (cell A is far ahead; its ring lines are cold --
        // another XCD's L2 or HBM -- and a fetch-and-wait here cost 1.5 us per step), into Ha[(t+2) & 1]; then the loads
        // for frame t+3 go out, to be consumed at the end of the next step.
        if (role_b) {
            if (t + 2 < Tc) {
                const unsigned ep_a = a.epoch_base + (unsigned)t + 3u;
                bool a_in = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) a_in &= (unsigned)(av[k] >> 32) == ep_a;
                if (a_in) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) Ha[t & 1][sr_][16 * k + su_] = __uint_as_float((unsigned)av[k]);
                } else {
                    fetch_a(t + 2);            // cell A not that far ahead yet (start of the launch): blocking sweep
                }
            }
            if (t + 3 < Tc) {
                const unsigned long long* slot_a = ring + (size_t)(t + 3) * 16 * 256;
#pragma unroll
                for (int k = 0; k < 16; ++k) av[k] = __hip_atomic_load(slot_a + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (ok) hstate[(long)rc * a.h_stride + u0 + cl] = h_own;
}

// ---------------------------------------------------------------------------------------------
// gru256_step_kernel: ONE GRUCell(256) step for B rows -- single-hop streaming, where a "scan" is one step and the
// hoisted input projection plus the cluster scan would be two dependent launches (~10 us each) for 0.4 MFLOP per row.
// 16 workgroups per 16-row tile as in gru256_cluster16_kernel (16 hidden units each, K split over the four waves), but
// with BOTH operands' quarters in registers (W_ih, W_hh: 96 VGPRs) and no exchange: every workgroup reads the full x and
// h rows of its tile itself.  The carried state is updated in place, so a workgroup may store its slice of h' only after
// all sixteen of the tile have read h: an arrival counter per tile (one relaxed agent-scope atomic add after the loads,
// one poll before the stores -- by then the peers have long arrived).
struct Gru256StepArgs {
    const float* x;             // [B][256]
    float* out;                 // [B][256]
    const float* wih; const float* whh;      // packed like hh_frag ([j 16][gate 3][k-chunk 64][lane 64])
    const float* bias;          // [768]: b_ih + b_hh (r, z) | b_in (n)
    const float* bhn;           // [256]
    float* hstate; long h_stride;
    int B;
    unsigned* arrive;           // [tiles] arrival counters (zeroed once; every launch adds 16 to the counters of its tiles)
    int* err;
};

__device__ __forceinline__ void gru256_step_body(const Gru256StepArgs& a, const int bx) {
    __shared__ __attribute__((aligned(16))) float Xs[16][260];
    __shared__ __attribute__((aligned(16))) float Hs[16][260];
    __shared__ float Ps[4][4][4][64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int rt = bx >> 4, j = bx & 15;
    const int row0 = rt * 16, u0 = 16 * j;
    for (int idx = tid; idx < 16 * 64; idx += 256) {
        const int r = idx >> 6, c4 = (idx & 63) * 4;
        const int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        *(float4*)&Xs[r][c4] = *(const float4*)(a.x + (size_t)rr * 256 + c4);
        *(float4*)&Hs[r][c4] = *(const float4*)(a.hstate + (long)rr * a.h_stride + c4);
    }
    float xr_[16], xz_[16], xn_[16], wr[16], wz[16], wn[16];
    {
        const size_t o = ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            xr_[k] = a.wih[o + (size_t)(0 * 64 + k) * 64]; xz_[k] = a.wih[o + (size_t)(1 * 64 + k) * 64]; xn_[k] = a.wih[o + (size_t)(2 * 64 + k) * 64];
            wr[k] = a.whh[o + (size_t)(0 * 64 + k) * 64]; wz[k] = a.whh[o + (size_t)(1 * 64 + k) * 64]; wn[k] = a.whh[o + (size_t)(2 * 64 + k) * 64];
        }
    }
    const float b_r = a.bias[u0 + cl], b_z = a.bias[256 + u0 + cl], b_n = a.bias[512 + u0 + cl], bhn = a.bhn[u0 + cl];
    __syncthreads();
    // this workgroup has read h.  The tile's round is complete at the next multiple of 16 above the value found (launches
    // that use a counter are stream-ordered, so the sixteen arrivals of one launch see 16 k ... 16 k + 15): no host-side target,
    // tiles that take part in different numbers of launches stay consistent
    unsigned target = 0;
    if (tid == 0) target = (__hip_atomic_fetch_add(a.arrive + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~15u) + 16u;
    f32x4 pr = {0.f, 0.f, 0.f, 0.f}, pz = pr, px = pr, pn = pr;
    {
        const float* xrow = &Xs[cl][64 * w + 4 * q];
        const float* hrow = &Hs[cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 x4 = *(const float4*)(xrow + 16 * c), h4 = *(const float4*)(hrow + 16 * c);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(xv[kb], xr_[c * 4 + kb], pr);
                pz = mfma16(xv[kb], xz_[c * 4 + kb], pz);
                px = mfma16(xv[kb], xn_[c * 4 + kb], px);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(hv[kb], wr[c * 4 + kb], pr);
                pz = mfma16(hv[kb], wz[c * 4 + kb], pz);
                pn = mfma16(hv[kb], wn[c * 4 + kb], pn);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { Ps[w][0][i][lane] = pr[i]; Ps[w][1][i][lane] = pz[i]; Ps[w][2][i][lane] = px[i]; Ps[w][3][i][lane] = pn[i]; }
    __syncthreads();
    // wave w finalises C-layout row i = w of the block: tile row q*4 + w; the four K quarters are added in one fixed order
    const int r_own = row0 + q * 4 + w;
    const bool ok = r_own < a.B;
    float sr = Ps[0][0][w][lane], sz = Ps[0][1][w][lane], sx = Ps[0][2][w][lane], sn = Ps[0][3][w][lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) { sr += Ps[k][0][w][lane]; sz += Ps[k][1][w][lane]; sx += Ps[k][2][w][lane]; sn += Ps[k][3][w][lane]; }
    const float r = sigmoid_f(b_r + sr);
    const float z = sigmoid_f(b_z + sz);
    const float n = gru_candidate(r, bhn + sn, b_n + sx);
    const float h_new = gru_blend(z, n, Hs[q * 4 + w][u0 + cl]);
    if (ok) a.out[(size_t)r_own * 256 + u0 + cl] = h_new;
    // in-place state: wait until all sixteen workgroups of the tile have read h
    if (tid == 0) {
        unsigned spins = 0; bool dead = false;
        while ((int)(__hip_atomic_load(a.arrive + rt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (cluster_spin_expired(spins, a.err, dead)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (ok) a.hstate[(long)r_own * a.h_stride + u0 + cl] = h_new;
}
__global__ __launch_bounds__(256, 1) void gru256_step_kernel(Gru256StepArgs a) { gru256_step_body(a, blockIdx.x); }
// Two independent cells' steps (the DF and the ERB decoder's first -- then second -- cell of a streaming hop; each with its own
// arrival counters) as ONE launch: the decoders' four dependent step launches become two.
__global__ __launch_bounds__(256, 1) void gru256_step_dual_kernel(Gru256StepArgs a0, Gru256StepArgs a1, int n0) {
    if ((int)blockIdx.x < n0) gru256_step_body(a0, blockIdx.x);
    else gru256_step_body(a1, blockIdx.x - n0);
}
