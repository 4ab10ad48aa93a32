// gru_clusterx.h -- the GRU-256 cluster scan WITH its input projection inside (big batches).
//
// gru256_cluster_kernel (gru_scan.h) is latency-bound: a step is 2.6 us of dependent MFMAs (192 per wave) and ~2.5 us of
// waiting for the three peers' h' granules; its 64 workgroups hold 64 CUs for 25 ms of a 256 x 10 s step at half of their
// matrix rate.  The input projection W_ih x + b of the same cells ran as a chip-wide GEMM over all frames in front of the scan
// (gemm_rows_wn<PlainA<64>>, 5.6 ms serial, 4.9 ms of the pipelined step: tools/skip_probe.sh) and its result went through HBM
// (3 KB per row written, read back by the scan).  The projection of step t+1 does not depend on h: here each wave computes its
// own 16 units' x part (192 more MFMAs) right AFTER it has published h'(t) -- in the time it would otherwise spend polling for the
// peers' granules -- so the chip-wide GEMM and its HBM round trip disappear and the scan step grows by what does not fit into
// the wait.  Weights: a wave now needs 2 x 192 fragment registers.  W_hh lives in AccVGPRs and feeds its MFMAs from there (srcB =
// AGPR, inline asm: the allocator would copy every weight through a VGPR first), W_ih's r and z thirds in VGPRs and its candidate
// third in LDS (64 KB per workgroup, read as B operands; builtin MFMAs); the kernel runs one
// wave per SIMD (512 registers per lane).  x(t+2) is staged through LDS two steps ahead (double buffer of the tile's 16 rows x 256 inputs).
// Granule protocol, exchange buffer, fragment packing (W_ih packed like W_hh: Gru256W::ih_as_hh) and time-out handling are those
// of gru256_cluster_kernel.  Results equal the hoisted form to rounding (the same products, summed in the same k order, the bias
// added first here and last there).
#pragma once
#include "common.h"
#include "gru_scan.h"

struct Gru256XArgs {
    const float* x;        // [B*Tc][256] input rows (row b * Tc + t)
    float* out;            // [B*Tc][256]
    const float* whh_frag; // [j 4][wave 4][gate 3][chunk 16][kb 4][lane 64]
    const float* wih_frag; // the same packing of W_ih
    const float* ih_bias;  // [768]: b_ih + b_hh (r, z), b_ih (n)
    const float* b_hn;     // [256]
    float* hstate; long h_stride;
    int B, Tc;
    unsigned long long* xbuf;   // [tiles][2][16][256] granules
    unsigned epoch_base;
    int* err;
};

// D(a[]) += A(v) * B(a): fp32 16x16x4 MFMA with the B operand taken from an AccVGPR
__device__ __forceinline__ void mfma16_accb(f32x4& c, float a, float b_acc) {
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ float to_acc(float v) {       // park a value in an AccVGPR
    float r;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
    return r;
}

__global__ __launch_bounds__(256, 1) void gru256_clusterx_kernel(Gru256XArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    __shared__ __attribute__((aligned(16))) float Xs[2][16][260];          // x(t) in Xs[t & 1]
    __shared__ float Wn[4][64][64];                                        // W_ih fragments of the candidate gate, [wave][k][lane]: the
                                                                           // 64 registers per lane that do not fit next to the other 320
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    float hr_[64], hz_[64], hn_[64];           // W_hh, AccVGPR-resident
    float ir_[64], iz_[64];                    // W_ih (r, z), VGPRs; the candidate's third in LDS (Wn)
    {
        const float* wf = a.whh_frag + ((size_t)(j * 4 + w) * 3) * 64 * 64 + lane;
        const float* xf = a.wih_frag + ((size_t)(j * 4 + w) * 3) * 64 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            hr_[k] = to_acc(wf[(size_t)(0 * 64 + k) * 64]);
            hz_[k] = to_acc(wf[(size_t)(1 * 64 + k) * 64]);
            hn_[k] = to_acc(wf[(size_t)(2 * 64 + k) * 64]);
            ir_[k] = xf[(size_t)(0 * 64 + k) * 64];
            iz_[k] = xf[(size_t)(1 * 64 + k) * 64];
            Wn[w][k][lane] = xf[(size_t)(2 * 64 + k) * 64];
        }
    }
    const float bhn = a.b_hn[u0 + cl];
    const float bir = a.ih_bias[u0 + cl], biz = a.ih_bias[256 + u0 + cl], bin = a.ih_bias[512 + u0 + cl];
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
    // x tile of a step: 16 rows x 256 inputs = 1024 float4 pieces, four per thread: row (tid >> 6) + 4 i, float4 column tid & 63
    const unsigned xlane = (unsigned)(tid & 63) * 4u;
    const int xrow_l = tid >> 6;
    const float* xp0; const float* xp1; const float* xp2; const float* xp3;
    {
        auto rowp = [&](int i) { int rr = row0 + xrow_l + 4 * i; if (rr >= a.B) rr = a.B - 1; return a.x + (size_t)rr * a.Tc * 256 + xlane; };
        xp0 = rowp(0); xp1 = rowp(1); xp2 = rowp(2); xp3 = rowp(3);
    }
#define DPDF_X_LOAD(T) const float4 x0_ = *(const float4*)(xp0 + (size_t)(T) * 256), x1_ = *(const float4*)(xp1 + (size_t)(T) * 256), \
                                    x2_ = *(const float4*)(xp2 + (size_t)(T) * 256), x3_ = *(const float4*)(xp3 + (size_t)(T) * 256)
#define DPDF_X_STAGE(BUF) do { *(float4*)&Xs[BUF][xrow_l][xlane] = x0_; *(float4*)&Xs[BUF][xrow_l + 4][xlane] = x1_; \
                               *(float4*)&Xs[BUF][xrow_l + 8][xlane] = x2_; *(float4*)&Xs[BUF][xrow_l + 12][xlane] = x3_; } while (0)
    // x part of a step from its staged inputs: W_ih x + b for this wave's 16 units
    f32x4 xr, xz, xn;
// (A operand in groups of four 16-wide K chunks, the next group's LDS reads issued in front of the current group's 48 MFMAs and
// pinned there: left alone, the scheduler hoists all sixteen reads of a block to its top -- 64 registers this kernel does not have)
#define DPDF_X_PART(BUF) do { \
        xr = (f32x4){bir, bir, bir, bir}; xz = (f32x4){biz, biz, biz, biz}; xn = (f32x4){bin, bin, bin, bin}; \
        const float* xrw = &Xs[BUF][cl][4 * q]; \
        float4 xa[2], xb_[2]; \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) xa[c] = *(const float4*)(xrw + 16 * c); \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) { \
            if (g < 7) { _Pragma("unroll") for (int c = 0; c < 2; ++c) xb_[c] = *(const float4*)(xrw + 16 * (2 * g + 2 + c)); } \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int c = 0; c < 2; ++c) { \
                const float xv_[4] = {xa[c].x, xa[c].y, xa[c].z, xa[c].w}; \
                _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) { \
                    xr = mfma16(xv_[kb], ir_[(2 * g + c) * 4 + kb], xr); xz = mfma16(xv_[kb], iz_[(2 * g + c) * 4 + kb], xz); \
                    xn = mfma16(xv_[kb], Wn[w][(2 * g + c) * 4 + kb][lane], xn); \
                } } \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int c = 0; c < 2; ++c) xa[c] = xb_[c]; \
        } } while (0)
    {
        DPDF_X_LOAD(0);
        DPDF_X_STAGE(0);
    }
    if (a.Tc > 1) {
        DPDF_X_LOAD(1);
        DPDF_X_STAGE(1);
    }
    __syncthreads();
    DPDF_X_PART(0);                            // step 0

    unsigned long long* xb = a.xbuf + (size_t)rt * 2 * 16 * 256;
    int cur = 0;
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    for (int t = 0; t < a.Tc; ++t) {
        // invariant: xr / xz / xn = x part of step t; Xs[(t + 1) & 1] = x(t + 1); Xs[t & 1] is free (its last readers are
        // behind the previous step's barrier): x(t + 2) goes there, its loads in flight under the h part
        const int tl = t + 2 < a.Tc ? t + 2 : a.Tc - 1;
        DPDF_X_LOAD(tl);
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, ahn = {bhn, bhn, bhn, bhn};
        const float* hrow = &Hs[cur][cl][4 * q];
        {
            float4 ha[4], hb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) ha[c] = *(const float4*)(hrow + 16 * c);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) hb[c] = *(const float4*)(hrow + 16 * (4 * g + 4 + c));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float hv[4] = {ha[c].x, ha[c].y, ha[c].z, ha[c].w};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        mfma16_accb(ar, hv[kb], hr_[(4 * g + c) * 4 + kb]);
                        mfma16_accb(az, hv[kb], hz_[(4 * g + c) * 4 + kb]);
                        mfma16_accb(ahn, hv[kb], hn_[(4 * g + c) * 4 + kb]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) ha[c] = hb[c];
            }
        }
        // the asm MFMAs are opaque to the compiler's hazard recogniser: XDL write -> VALU read of the accumulators needs 11 wait
        // states after an 8-pass MFMA (CDNA3 ISA); 24 given, between the last MFMA and the first read
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(ar), "+a"(az), "+a"(ahn));
        DPDF_X_STAGE(t & 1);                   // (the last step stages x(Tc - 1) once more: harmless, nobody reads it)
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = xb + (size_t)(t & 1) * 16 * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = sigmoid_f(ar[i] + xr[i]);
            const float z = sigmoid_f(az[i] + xz[i]);
            const float n = gru_candidate(r, ahn[i], xn[i]);
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
        // sweep the three peers' slices (see gru256_cluster_kernel), also after the LAST step.  The first round of loads goes
        // out BEFORE the next step's x part (the peers publish at about the same time: it usually comes back fresh) and is
        // checked behind it -- the sweep's L2 round trip and the peers' skew lie under 192 MFMAs
        {
            unsigned long long xv[12];
            unsigned spins = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int idx = tid + 256 * k;
                const int s = idx >> 10, r = (idx >> 6) & 15, u = 64 * ((j + 1 + s) & 3) + (idx & 63);
                xv[k] = __hip_atomic_load(slot + r * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t + 1 < a.Tc) DPDF_X_PART((t + 1) & 1);
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 12; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const int idx = tid + 256 * k;
                    const int s = idx >> 10, r = (idx >> 6) & 15, u = 64 * ((j + 1 + s) & 3) + (idx & 63);
                    xv[k] = __hip_atomic_load(slot + r * 256 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int idx = tid + 256 * k;
                const int s = idx >> 10, r = (idx >> 6) & 15, u = 64 * ((j + 1 + s) & 3) + (idx & 63);
                Hs[nxt][r][u] = __uint_as_float((unsigned)xv[k]);
            }
        }
        __syncthreads();                       // h(t) of the whole tile is in Hs[nxt]; x(t + 2) is staged; every wave is done with x(t + 1)
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ok[i]) a.hstate[(long)rc[i] * a.h_stride + u0 + cl] = h_own[i];
#undef DPDF_X_LOAD
#undef DPDF_X_STAGE
#undef DPDF_X_PART
}
