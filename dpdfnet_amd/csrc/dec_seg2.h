// dec_seg2.h -- the 48 kHz decoder stages (dec_last.h: dec_seg_kernel) as a software pipeline over tiles inside one 512-thread workgroup.
//
// dec_seg_kernel runs a tile through four barrier-separated phases (u -> LDS, sub-pixel depthwise, pointwise on the matrix cores, read-out)
// with two workgroups per CU; it reaches 2.9 TB/s and 32 % of the fp32 matrix rate: a 5- or 6-row-tile pointwise dealt to four waves leaves
// one SIMD with twice the matrix work, and every phase waits for the slowest wave of the one before.  Here the four phases of FOUR
// consecutive tiles run in the same tick, one barrier per tick, every buffer between them doubled:
//   tick n:  pointwise of tile n (As[n & 1] -> Os[n & 1]) | read-out of tile n - 1 (Os) | u of tile n + 2 (registers loaded two ticks ago
//            -> U1[n & 1], then the loads of tile n + 4 are issued) | depthwise of tile n + 1 (U1 -> As).
// The pointwise is split by (row tile, 16 output columns): wave w owns columns [16 (w & 3), + 16) -- 16 B fragments instead of 64 -- and
// the first (w < 4) or second half of the row tiles, so the two waves of a SIMD together issue a quarter of the tile's matrix
// instructions whatever the number of row tiles; waves 0-3 start a tick with their matrix work, waves 4-7 end with it.  Every output
// element of the pointwise is accumulated in the order dec_seg_kernel uses: stages 3 and 2 (LAST = false) are bit-identical to it.
// LAST (convt1 + the mask head): e0 passes through a ring of THREE LDS buffers -- relu(ps0 e0 + pb0) written a tick ahead, u0 = relu(d1) + that
// formed in place by the lane that owns the element, the three tap sums of conv0_out taken a tick later by four threads per row (sixteen
// channels each, two quad DPP steps) instead of dec_seg_kernel's DPP row reductions on the accumulators: equal to rounding (4e-8 RMS).
// Frames are dealt to workgroups (frame b0 + k G: all its tiles), 110 KB (LAST: 149 KB) of LDS, one workgroup per CU.
// Measured: DESIGN.md section 3b, tools/dec_seg_bench.hip, docs/HISTORY.md (round-5 ledger: what was tried around it).
#pragma once
#include "common.h"
#include "dec_last.h"
#include <type_traits>
#ifndef DS2_VAR
#define DS2_VAR 0      // tools/dec_seg_bench.hip timing builds: 1 no matrix work, 2 no depthwise, 3 no loads behind the first, 4 no stores
#endif

template <int S, int R, bool LAST>
constexpr int dec_seg2_lds_floats() { return (2 * (R / S + 2) + 2 * R + (LAST ? 3 * R : 2 * R)) * 68; }

// One stage over this workgroup's frames (b0, b0 + G, ...: every tile of a frame belongs to the frame's workgroup, so a stage's input frame is
// the same workgroup's output of the stage before -- dec_seg2_all_kernel runs the three stages back to back without leaving its CU).
template <int S, int R, bool LAST>
__device__ __forceinline__ void dec_seg2_body(const DecSegArgs& a, float* lds) {
    constexpr int RI = R / S, NRT = R / 16, NIN = RI + 2;
    constexpr int NP = (NIN * 16 + 511) / 512;                           // float4 pieces per thread of e / prev (incl. the halo bands)
    constexpr int NO = (R * 16 + 511) / 512;                             // ... of the depthwise panel / the output tile / e0 (LAST)
    constexpr int RT_SPLIT = (NRT + 1) / 2;
    static_assert(R % 16 == 0 && R % S == 0, "tile = whole MFMA row tiles and whole input bands");
    float (*U1)[NIN][68] = reinterpret_cast<float (*)[NIN][68]>(lds);                                    // [2]
    float (*As)[R][68] = reinterpret_cast<float (*)[R][68]>(lds + 2 * NIN * 68);                         // [2]
    float (*Os)[R][68] = reinterpret_cast<float (*)[R][68]>(lds + (2 * NIN + 2 * R) * 68);               // [2] (!LAST)
    float (*E0)[R][68] = Os;                                                                             // [3] (LAST): relu(ps0 e0 + pb0), then u0 in place; a tile's buffer lives three ticks
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, q = lane >> 4;
    const int c4 = (tid & 15) * 4, r32 = tid >> 4;
    const int nt = w & 3;
    float breg[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) breg[c * 4 + kb] = a.pwfrag[(size_t)((c * 4 + nt) * 4 + kb) * 64 + lane];
    const float bvn = a.bias[nt * 16 + cl];
    const float4 s1 = *(const float4*)(a.ps + c4), b1 = *(const float4*)(a.pb + c4);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), b0v = s0;
    // LAST: conv0_out's weights for this thread's quarter of a row (taps: thread = (row tid >> 2, channels [16 (tid & 3), + 16)))
    float w0q[LAST ? 48 : 1];
    if (LAST) {
        s0 = *(const float4*)(a.ps0 + c4); b0v = *(const float4*)(a.pb0 + c4);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float4 v = *(const float4*)(a.w0 + 48 * (tid & 3) + 4 * k);
            w0q[4 * k] = v.x; w0q[4 * k + 1] = v.y; w0q[4 * k + 2] = v.z; w0q[4 * k + 3] = v.w;
        }
    }
    // depthwise taps of this thread's rows (r = r32 + 32 i: conv k = r % S is fixed per thread and i -- no select per tap as in dec_seg_kernel)
    float dwr[NO][4][3];
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int k = (r32 + 32 * i) % S;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) dwr[i][j][t] = a.dw[((size_t)k * 64 + c4 + j) * 3 + t];
    }
    const int FI = a.FO / S, nseg = a.FO / R;
    const int G = gridDim.x, b0 = blockIdx.x;
    if (b0 >= a.BT) return;
    const int cnt = (a.BT - b0 + G - 1) / G * nseg;                      // this workgroup's tiles: segment j % nseg of frame b0 + (j / nseg) G, j < cnt

    // A tile's input run starts one band below its first output's source band; the halo band is outside the frame for the first / last
    // segment of a frame (its u stays zero: the pathway term too) -- those threads read their neighbour row instead, p1 drops it.
    // Addresses are a wave-uniform base + a per-thread byte offset fixed for the whole launch.
    // No control flow around the loads: behind a branch that may skip them the compiler can no longer count the loads in flight and waits
    // for ALL of them where a tile's registers are first used -- i.e. for the loads it has just issued (measured: +0.95 ms on the 3.15 ms of
    // the mask-head stage).  Rows past the run read the last row again, tiles past the workgroup's last read the last tile again.
    unsigned in_off[NP];
    bool in_lo[NP], in_hi[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = r32 + 32 * i < NIN ? r32 + 32 * i : NIN - 1;
        in_off[i] = (unsigned)(row * 64 + c4) * 4u;              // against the halo band below the tile
        in_lo[i] = row == 0; in_hi[i] = row == NIN - 1;
    }
    auto load_tile = [&](int j, float4 (&ve)[NP], float4 (&vp)[NP]) __attribute__((always_inline)) {
        if (DS2_VAR == 3 && j > 1) return;
        const int jc = j < cnt ? j : cnt - 1, fr = jc / nseg, bt = b0 + fr * G, seg = jc - fr * nseg;
        const char* ep = (const char*)(a.e + ((size_t)bt * FI + (size_t)seg * RI) * 64) - 256;
        const char* pp = (const char*)(a.prev + ((size_t)bt * FI + (size_t)seg * RI) * 64) - 256;
        const bool first = seg == 0, last = seg == nseg - 1;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            unsigned off = in_off[i];
            off = first && in_lo[i] ? off + 256u : off;
            off = last && in_hi[i] ? off - 256u : off;
            ve[i] = *(const float4*)(ep + off);
            vp[i] = *(const float4*)(pp + off);
        }
    };
    auto p1 = [&](int j, int ub, const float4 (&ve)[NP], const float4 (&vp)[NP]) __attribute__((always_inline)) {
        if (j < 0 || j >= cnt) return;
        const int seg = j % nseg;
        const bool first = seg == 0, last = seg == nseg - 1;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = r32 + 32 * i;
            if (row < NIN) {
                float4 u;
                u.x = relu_f(__builtin_fmaf(s1.x, ve[i].x, b1.x)) + vp[i].x; u.y = relu_f(__builtin_fmaf(s1.y, ve[i].y, b1.y)) + vp[i].y;
                u.z = relu_f(__builtin_fmaf(s1.z, ve[i].z, b1.z)) + vp[i].z; u.w = relu_f(__builtin_fmaf(s1.w, ve[i].w, b1.w)) + vp[i].w;
                if ((first && in_lo[i]) || (last && in_hi[i])) u = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)&U1[ub][row][c4] = u;
            }
        }
    };
    // sub-pixel depthwise: output band fo = S f + k <- u bands f - 1 .. f + 1 with conv k
    auto p2 = [&](int j, int ub, int ab) __attribute__((always_inline)) {
        if (j < 0 || j >= cnt || DS2_VAR == 2) return;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int r = r32 + 32 * i;
            if (r < R) {
                const int f = r / S;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 x = *(const float4*)&U1[ub][f + t][c4];
                    v.x += dwr[i][0][t] * x.x; v.y += dwr[i][1][t] * x.y; v.z += dwr[i][2][t] * x.z; v.w += dwr[i][3][t] * x.w;
                }
                *(float4*)&As[ab][r][c4] = v;
            }
        }
    };
    // pointwise 64 x 64, this wave's (row tile, column tile) units
    auto mma = [&](int j, int ab, int eb) __attribute__((always_inline)) {
        if (j < 0 || j >= cnt || DS2_VAR == 1) return;
        const int rt0 = w < 4 ? 0 : RT_SPLIT, rt1 = w < 4 ? RT_SPLIT : NRT;
        for (int rt = rt0; rt < rt1; ++rt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* arow = &As[ab][16 * rt + cl][4 * q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 a4 = *(const float4*)(arow + 16 * c);
                acc = mfma16(a4.x, breg[c * 4 + 0], acc);
                acc = mfma16(a4.y, breg[c * 4 + 1], acc);
                acc = mfma16(a4.z, breg[c * 4 + 2], acc);
                acc = mfma16(a4.w, breg[c * 4 + 3], acc);
            }
            if (!LAST) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Os[ab][16 * rt + 4 * q + i][nt * 16 + cl] = relu_f(acc[i] + bvn);
            } else {            // u0 = relu(d1) + relu(ps0 e0 + pb0), in place (this lane is the element's only reader and writer)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* ep = &E0[eb][16 * rt + 4 * q + i][nt * 16 + cl];
                    *ep = relu_f(acc[i] + bvn) + *ep;
                }
            }
        }
    };
    // LAST: relu(ps0 e0 + pb0) of tile j into its ring buffer
    auto p1e = [&](int j, int eb, const float4 (&ve0)[NO]) __attribute__((always_inline)) {
        if (j < 0 || j >= cnt || DS2_VAR == 6) return;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int row = r32 + 32 * i;
            if (row < R) {
                float4 u;
                u.x = relu_f(__builtin_fmaf(s0.x, ve0[i].x, b0v.x)); u.y = relu_f(__builtin_fmaf(s0.y, ve0[i].y, b0v.y));
                u.z = relu_f(__builtin_fmaf(s0.z, ve0[i].z, b0v.z)); u.w = relu_f(__builtin_fmaf(s0.w, ve0[i].w, b0v.w));
                *(float4*)&E0[eb][row][c4] = u;
            }
        }
    };
    auto load_e0 = [&](int j, float4 (&ve0)[NO]) __attribute__((always_inline)) {
        if (DS2_VAR == 3 && j > 1) return;
        const int jc = j < cnt ? j : cnt - 1, fr = jc / nseg, bt = b0 + fr * G, seg = jc - fr * nseg;
        const char* e0p = (const char*)(a.e0 + ((size_t)bt * a.FO + (size_t)seg * R) * 64);
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int row = r32 + 32 * i < R ? r32 + 32 * i : R - 1;
            ve0[i] = *(const float4*)(e0p + (unsigned)(row * 64 + c4) * 4u);
        }
    };
    // LAST: the three tap sums of conv0_out per row from the finished u0 rows: four threads per row, sixteen channels each, summed over
    // the quad by two DPP steps; thread 0 of the quad stores [t0 t1 t2 0] (dec_seg_kernel sums in another order: equal to rounding)
    auto taps = [&](int j, int eb) __attribute__((always_inline)) {
        if (j < 0 || j >= cnt || DS2_VAR == 4 || DS2_VAR == 5) return;
        const int fr = j / nseg, bt = b0 + fr * G, seg = j - fr * nseg;
        char* sp = (char*)(a.ssum + ((size_t)bt * a.FO + (size_t)seg * R) * 4);
        const int r = tid >> 2, pq = tid & 3;
        if (r < R) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 u = *(const float4*)&E0[eb][r][16 * pq + 4 * k];
                const float uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    t0 = __builtin_fmaf(w0q[(4 * k + m) * 3 + 0], uv[m], t0); t1 = __builtin_fmaf(w0q[(4 * k + m) * 3 + 1], uv[m], t1);
                    t2 = __builtin_fmaf(w0q[(4 * k + m) * 3 + 2], uv[m], t2);
                }
            }
            auto quad_sum = [](float v) {
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2, 3, 0, 1]
                return v;
            };
            t0 = quad_sum(t0); t1 = quad_sum(t1); t2 = quad_sum(t2);
            if (pq == 0) *(float4*)(sp + (unsigned)r * 16u) = make_float4(t0, t1, t2, 0.f);
        }
    };
    auto p4 = [&](int j, int ob) __attribute__((always_inline)) {
        if (LAST || j < 0 || j >= cnt || DS2_VAR == 4) return;
        const int fr = j / nseg, bt = b0 + fr * G, seg = j - fr * nseg;
        char* op = (char*)(a.out + ((size_t)bt * a.FO + (size_t)seg * R) * 64);
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int r = r32 + 32 * i;
            if (r < R) *(float4*)(op + (unsigned)(r * 64 + c4) * 4u) = *(const float4*)&Os[ob][r][c4];
        }
    };

    float4 ve[2][NP], vp[2][NP], ve0[2][NO];
    // (in the order of the steady state, pinned: the wait in front of a tile's first use counts the loads issued behind it)
    load_tile(0, ve[0], vp[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (LAST) load_e0(0, ve0[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (LAST) load_e0(1, ve0[1]);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(1, ve[1], vp[1]);
    __builtin_amdgcn_sched_barrier(0);
    int k3 = 1;                                                          // (n + 3) % 3: tile n's E0 ring slot
    auto tick = [&](auto par, int n) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::value;
        const int e_cur = k3, e_prev = k3 == 0 ? 2 : k3 - 1, e_next = k3 == 2 ? 0 : k3 + 1;
        if (w < 4) mma(n, P, e_cur);
        if (LAST) taps(n - 1, e_prev); else p4(n - 1, P ^ 1);
        p1(n + 2, P, ve[P], vp[P]);
        load_tile(n + 4, ve[P], vp[P]);
        if (LAST) { p1e(n + 1, e_next, ve0[P ^ 1]); load_e0(n + 3, ve0[P ^ 1]); }
        p2(n + 1, P ^ 1, P ^ 1);
        if (w >= 4) mma(n, P, e_cur);
        __syncthreads();
        k3 = e_next;
    };
    for (int n = -2; n <= cnt; n += 2) {
        tick(std::integral_constant<int, 0>{}, n);
        tick(std::integral_constant<int, 1>{}, n + 1);
    }
}

template <int S, int R, bool LAST>
__global__ __launch_bounds__(512) void dec_seg2_kernel(DecSegArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[dec_seg2_lds_floats<S, R, LAST>()];
    dec_seg2_body<S, R, LAST>(a, lds);
}

// The three stages of the 48 kHz decoder in ONE launch (option dec_seg = 3; NOT the default).  Stage 2 of the pipeline runs beside the next
// chunk's stage 1, whose GRU-64 workgroups hold their CUs for a millisecond or two, and each of the three launches waits for CUs again (traced:
// 2.35 ms per launch for stages that take 0.49 / 0.87 ms alone).  With frames dealt to workgroups, a stage reads what the same workgroup
// wrote: nothing crosses workgroups, the stages follow each other behind a drained store queue and a barrier.  Bit-identical to the three
// launches, 3.92 against 4.00 ms alone -- and 3 ms per step SLOWER in the pipeline (dpdfnet2_48khz_hr 131.0 -> 134.1): holding all 256 CUs for
// the whole decoder costs stage 1 more than the two re-acquisitions cost stage 2.
__global__ __launch_bounds__(512) void dec_seg2_all_kernel(DecSegArgs a3, DecSegArgs a2, DecSegArgs a1) {
    constexpr int NF = dec_seg2_lds_floats<3, 96, true>() > dec_seg2_lds_floats<2, 80, false>() ? dec_seg2_lds_floats<3, 96, true>() : dec_seg2_lds_floats<2, 80, false>();
    __shared__ __attribute__((aligned(16))) float lds[NF];
    dec_seg2_body<2, 80, false>(a3, lds);
    __threadfence(); __syncthreads();
    dec_seg2_body<2, 80, false>(a2, lds);
    __threadfence(); __syncthreads();
    dec_seg2_body<3, 96, true>(a1, lds);
}
