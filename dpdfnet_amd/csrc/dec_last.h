// dec_last.h -- the ERB decoder's last stage and the mask head as one kernel (16 kHz geometry: 16 -> 32 bands).
//
// reference onnx_model/dpdfnet.py:361-366: d1 = convt1(conv1p(e1) + d2); m = conv0_out(conv0p(e0) + d1), with
// convt1 = sub-pixel (2 depthwise k(1,3) convs interleaved along frequency, layers.py:895-916) + pointwise 64x64 + BN +
// ReLU, conv{1,0}p = per-channel scale + BN + ReLU, conv0_out = dense 64 -> 1 k(1,3) + BN + Sigmoid.
// As a gemm_rows launch (SubpixA<2> producer + MaskSumEpi) this stage took 0.41 ms per chunk of the headline
// workload -- 1.3 TB/s: every thread fetched the three taps of both inputs itself (6x redundant through the L1) and
// recomputed the pathway term per tap, with a runtime modulo per row.  Here a 64-row tile is exactly two frames, so a
// workgroup loads each of its three input tiles ONCE, fully coalesced (they are contiguous in memory), forms
// u1 = relu(ps1 e1 + pb1) + d2 once per element into LDS, builds the depthwise panel from LDS, runs the pointwise
// GEMM on the matrix cores, and finishes the mask in place: 64 -> 1 contraction on the accumulators (DPP row
// reductions), the three-band sum and the sigmoid through a 1 KB LDS array.  Only m [frames][32] is written.
#pragma once
#include "common.h"

struct DecLastArgs {
    const float* e1; const float* d2;   // [BT][16][64]
    const float* e0;                    // [BT][32][64]
    float* m;                           // [BT][32]
    const float* ps1; const float* pb1; // conv1p folded [64]
    const float* dw;                    // convt1 depthwise [2][64][3]
    const float* pwfrag;                // convt1 pointwise (BN folded), B fragments [chunk 4][tile 4][kb 4][lane 64]
    const float* bias;                  // convt1 BN shift [64]
    const float* ps0; const float* pb0; // conv0p folded [64]
    const float* w0;                    // conv0_out [64][3] (BN folded)
    float bias0;
    int BT;                             // frames (even or odd; a trailing single frame is handled)
};

__global__ __launch_bounds__(256, 2) void dec_last_kernel(DecLastArgs a) {
    __shared__ __attribute__((aligned(16))) float U1[2][18][68];    // u1 per frame, rows 0 and 17 = zero padding bands
    __shared__ __attribute__((aligned(16))) float E0[64][68];       // relu(ps0 e0 + pb0)
    __shared__ __attribute__((aligned(16))) float As[64][68];       // depthwise panel
    __shared__ float Ss[64][4];                                     // tap sums per output row
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, q = lane >> 4;
    const int c4 = (tid & 15) * 4;

    float breg[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) breg[i] = a.pwfrag[(size_t)i * 64 + lane];
    float4 s1 = *(const float4*)(a.ps1 + c4), b1 = *(const float4*)(a.pb1 + c4);
    float4 s0 = *(const float4*)(a.ps0 + c4), b0 = *(const float4*)(a.pb0 + c4);
    float dwv[2][4][3];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) dwv[k][j][t] = a.dw[((size_t)k * 64 + c4 + j) * 3 + t];
    float bv[4], w0v[4][3];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = nt * 16 + cl;
        bv[nt] = a.bias[c];
#pragma unroll
        for (int t = 0; t < 3; ++t) w0v[nt][t] = a.w0[c * 3 + t];
    }
    // zero the padding bands once
    if (tid < 4 * 17) {
        const int fr = tid / 34, rem = tid - fr * 34, row = rem / 17 ? 17 : 0, cc = (rem % 17) * 4;
        *(float4*)&U1[fr][row][cc] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntiles = (a.BT + 1) >> 1;
    auto load_tile = [&](int tile, float4 (&ve1)[2], float4 (&vd2)[2], float4 (&ve0)[4]) __attribute__((always_inline)) {
        const size_t bt0 = (size_t)tile * 2;
        const bool two = bt0 + 1 < (size_t)a.BT;                     // second frame of the tile exists
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = i == 0 || two;                           // piece i of e1/d2 = frame i (256 float4 per frame)
            ve1[i] = ok ? *(const float4*)(a.e1 + bt0 * 1024 + (size_t)(tid + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            vd2[i] = ok ? *(const float4*)(a.d2 + bt0 * 1024 + (size_t)(tid + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = i < 2 || two;                            // pieces 0,1 = frame 0; 2,3 = frame 1
            ve0[i] = ok ? *(const float4*)(a.e0 + bt0 * 2048 + (size_t)(tid + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    float4 ve1[2], vd2[2], ve0[4];
    load_tile(tile, ve1, vd2, ve0);
    for (; tile < ntiles; tile += gridDim.x) {
        // ---- stage the tile: u1 and the pathway term of e0, once per element
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int band = tid >> 4;                               // piece i: frame i, band tid>>4
            float4 u;
            u.x = relu_f(__builtin_fmaf(s1.x, ve1[i].x, b1.x)) + vd2[i].x;
            u.y = relu_f(__builtin_fmaf(s1.y, ve1[i].y, b1.y)) + vd2[i].y;
            u.z = relu_f(__builtin_fmaf(s1.z, ve1[i].z, b1.z)) + vd2[i].z;
            u.w = relu_f(__builtin_fmaf(s1.w, ve1[i].w, b1.w)) + vd2[i].w;
            *(float4*)&U1[i][1 + band][c4] = u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            float4 u;
            u.x = relu_f(__builtin_fmaf(s0.x, ve0[i].x, b0.x)); u.y = relu_f(__builtin_fmaf(s0.y, ve0[i].y, b0.y));
            u.z = relu_f(__builtin_fmaf(s0.z, ve0[i].z, b0.z)); u.w = relu_f(__builtin_fmaf(s0.w, ve0[i].w, b0.w));
            *(float4*)&E0[row][c4] = u;
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, ve1, vd2, ve0);           // in flight under the rest of this tile
        __syncthreads();
        // ---- sub-pixel depthwise: output band fo = 2 f + k <- u1 bands f-1 .. f+1 with conv k
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, fr = r >> 5, fo = r & 31, f = fo >> 1, k = fo & 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float4 x = *(const float4*)&U1[fr][f + t][c4];
                v.x += (k ? dwv[1][0][t] : dwv[0][0][t]) * x.x; v.y += (k ? dwv[1][1][t] : dwv[0][1][t]) * x.y;
                v.z += (k ? dwv[1][2][t] : dwv[0][2][t]) * x.z; v.w += (k ? dwv[1][3][t] : dwv[0][3][t]) * x.w;
            }
            *(float4*)&As[r][c4] = v;
        }
        __syncthreads();
        // ---- pointwise 64 x 64 on the matrix cores: wave w = rows 16 w .. 16 w + 15
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* arow = &As[16 * w + cl][4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a4 = *(const float4*)(arow + 16 * c);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int bi = (c * 4 + nt) * 4;
                acc[nt] = mfma16(a4.x, breg[bi + 0], acc[nt]);
                acc[nt] = mfma16(a4.y, breg[bi + 1], acc[nt]);
                acc[nt] = mfma16(a4.z, breg[bi + 2], acc[nt]);
                acc[nt] = mfma16(a4.w, breg[bi + 3], acc[nt]);
            }
        }
        // ---- mask head on the accumulators: u0 = relu(d1) + pathway(e0); three tap sums per row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * w + 4 * q + i;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float u = relu_f(acc[nt][i] + bv[nt]) + E0[row][nt * 16 + cl];
                t0 = __builtin_fmaf(w0v[nt][0], u, t0); t1 = __builtin_fmaf(w0v[nt][1], u, t1); t2 = __builtin_fmaf(w0v[nt][2], u, t2);
            }
            t0 = row16_allreduce_sum(t0); t1 = row16_allreduce_sum(t1); t2 = row16_allreduce_sum(t2);
            if (cl == 0) { Ss[row][0] = t0; Ss[row][1] = t1; Ss[row][2] = t2; }
        }
        __syncthreads();
        if (tid < 64) {
            const int fo = tid & 31;
            float s = Ss[tid][1];
            if (fo > 0) s += Ss[tid - 1][0];
            if (fo < 31) s += Ss[tid + 1][2];
            const size_t bt = (size_t)tile * 2 + (tid >> 5);
            if (bt < (size_t)a.BT) a.m[bt * 32 + fo] = sigmoid_f(s + a.bias0);
        }
        // the next iteration's staging writes U1 / E0: every wave has passed the barrier above after its last read of
        // them (U1: depthwise, E0: mask head); As and Ss are rewritten only after the next iteration's first barrier
    }
}

// ---------------------------------------------------------------------------------------------
// dec_stage_kernel<S, FO>: an inner ERB-decoder stage d_out = convt(conv_p(e) + d_in) (reference
// onnx_model/dpdfnet.py:361-364; sub-pixel S = 2 or plain S = 1 depthwise k(1,3) + pointwise + BN + ReLU) for
// geometries where a 64-row tile is a whole number of frames (16 kHz: FO = 8 or 16 output bands).  Same plan as
// dec_last_kernel: the two input tiles are contiguous and loaded once, coalesced; u = relu(ps e + pb) + d_in once
// per element into LDS; depthwise panel from LDS; pointwise GEMM; the output tile leaves through LDS as whole rows.
struct DecStageArgs {
    const float* e; const float* prev;  // [BT][FO/S][64]
    float* out;                         // [BT][FO][64]
    const float* ps; const float* pb;   // pathway conv folded [64]
    const float* dw;                    // [S][64][3]
    const float* pwfrag; const float* bias;
    int BT;
};
template <int S, int FO>
__global__ __launch_bounds__(256, 2) void dec_stage_kernel(DecStageArgs a) {
    constexpr int FI = FO / S, NF = 64 / FO, NIN = 64 / S;              // input bands, frames per tile, input rows per tile
    constexpr int NL = NIN * 16 / 256;                                   // float4 pieces per thread and tensor (S=1: 4, S=2: 2)
    static_assert(64 % FO == 0 && FO % S == 0 && NL >= 1, "tile must hold whole frames");
    __shared__ __attribute__((aligned(16))) float U1[NF][FI + 2][68];
    __shared__ __attribute__((aligned(16))) float As[64][68];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, q = lane >> 4;
    const int c4 = (tid & 15) * 4;
    float breg[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) breg[i] = a.pwfrag[(size_t)i * 64 + lane];
    const float4 s1 = *(const float4*)(a.ps + c4), b1 = *(const float4*)(a.pb + c4);
    float dwv[S][4][3];
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) dwv[k][j][t] = a.dw[((size_t)k * 64 + c4 + j) * 3 + t];
    float bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = a.bias[nt * 16 + cl];
    for (int i = tid; i < NF * 2 * 17; i += 256) {                       // zero the padding bands once
        const int fr = i / 34, rem = i - fr * 34;
        *(float4*)&U1[fr][rem / 17 ? FI + 1 : 0][(rem % 17) * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntiles = (a.BT + NF - 1) / NF;
    auto load_tile = [&](int tile, float4 (&ve)[NL], float4 (&vp)[NL]) __attribute__((always_inline)) {
        const size_t row0 = (size_t)tile * NIN;                          // first input row (frame-major, FI rows per frame)
        const size_t nrows = (size_t)a.BT * FI;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 256 * i;
            const bool ok = row0 + (idx >> 4) < nrows;
            ve[i] = ok ? *(const float4*)(a.e + row0 * 64 + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            vp[i] = ok ? *(const float4*)(a.prev + row0 * 64 + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    float4 ve[NL], vp[NL];
    load_tile(tile, ve, vp);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = (tid >> 4) + 16 * i, fr = row / FI, band = row - fr * FI;
            float4 u;
            u.x = relu_f(__builtin_fmaf(s1.x, ve[i].x, b1.x)) + vp[i].x; u.y = relu_f(__builtin_fmaf(s1.y, ve[i].y, b1.y)) + vp[i].y;
            u.z = relu_f(__builtin_fmaf(s1.z, ve[i].z, b1.z)) + vp[i].z; u.w = relu_f(__builtin_fmaf(s1.w, ve[i].w, b1.w)) + vp[i].w;
            *(float4*)&U1[fr][1 + band][c4] = u;
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, ve, vp);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, fr = r / FO, fo = r - fr * FO, f = fo / S, k = fo - f * S;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float4 x = *(const float4*)&U1[fr][f + t][c4];
                v.x += (S > 1 && k ? dwv[S - 1][0][t] : dwv[0][0][t]) * x.x; v.y += (S > 1 && k ? dwv[S - 1][1][t] : dwv[0][1][t]) * x.y;
                v.z += (S > 1 && k ? dwv[S - 1][2][t] : dwv[0][2][t]) * x.z; v.w += (S > 1 && k ? dwv[S - 1][3][t] : dwv[0][3][t]) * x.w;
            }
            *(float4*)&As[r][c4] = v;
        }
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* arow = &As[16 * w + cl][4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a4 = *(const float4*)(arow + 16 * c);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int bi = (c * 4 + nt) * 4;
                acc[nt] = mfma16(a4.x, breg[bi + 0], acc[nt]);
                acc[nt] = mfma16(a4.y, breg[bi + 1], acc[nt]);
                acc[nt] = mfma16(a4.z, breg[bi + 2], acc[nt]);
                acc[nt] = mfma16(a4.w, breg[bi + 3], acc[nt]);
            }
        }
        // out tile through LDS: wave w overwrites only the rows it has just read (As[16 w ..]) -- no barrier needed before
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) As[16 * w + 4 * q + i][nt * 16 + cl] = relu_f(acc[nt][i] + bv[nt]);
        __syncthreads();
        {
            const size_t orow0 = (size_t)tile * 64, nrows = (size_t)a.BT * FO;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 4) + 16 * i;
                if (orow0 + r < nrows) *(float4*)(a.out + (orow0 + r) * 64 + c4) = *(const float4*)&As[r][c4];
            }
        }
        // U1 is rewritten at the top of the next iteration: its last readers (depthwise) are two barriers back; As is
        // rewritten after the next iteration's first barrier, i.e. after every thread has finished the read-out above
    }
}

// ---------------------------------------------------------------------------------------------
// dec_seg_kernel<S, R, LAST>: the same plan for geometries in which a 64-row tile is NOT a whole number of frames (the
// 48 kHz decoder: 40 -> 80 -> 160 -> 480 bands, strides 2 / 2 / 3; reference onnx_model/dpdfnet_48khz_hr.py:371-432).
// A tile is R consecutive OUTPUT bands of one frame (FO % R == 0, R % 16 == 0, R % S == 0): its R / S input bands plus
// one halo band on each side (zero at the frame's edges) are one contiguous run of each input tensor -- loaded once,
// coalesced; u = relu(ps e + pb) + d_in once per element into LDS; sub-pixel depthwise from LDS; pointwise 64 x 64 on the
// matrix cores, the R / 16 row tiles dealt round-robin to the four waves.  LAST = false: the output tile leaves through
// LDS as whole rows.  LAST = true (convt1 + mask head): u0 = relu(d1) + relu(ps0 e0 + pb0) on the accumulators, the three
// tap sums of conv0_out per row go out as [rows][4] (16 B instead of the 256-B d1 row) for mask_fin_kernel -- the taps
// of neighbouring bands cross tile borders, so the sum over them stays in that kernel.
// As gemm_rows launches (SubpixA producers) these three stages cost 67 ms per 256 x 10 s step of either 48 kHz model --
// three times their HBM bound: every thread fetched the taps of both inputs itself and recomputed the pathway term per tap.
struct DecSegArgs {
    const float* e; const float* prev;  // [BT][FO/S][64]
    float* out;                         // [BT][FO][64]                 (LAST = false)
    const float* ps; const float* pb;   // pathway conv folded [64]
    const float* dw;                    // [S][64][3]
    const float* pwfrag; const float* bias;
    const float* e0; float* ssum;       // LAST: e0 [BT][FO][64], tap sums [BT * FO][4]
    const float* ps0; const float* pb0; const float* w0;   // LAST: conv0p folded, conv0_out [64][3]
    int BT, FO;
};
// (dec_seg_kernel itself -- one tile through four barrier-separated phases, two 256-thread workgroups per CU -- was superseded by the
// tile-pipelined dec_seg2.h in round 5 and removed in round 6; its argument block and tile plan above are dec_seg2's.)
