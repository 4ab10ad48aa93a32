// df_ring.h -- the DF branch's two consumers of c0 in ONE pass over it.
//
// c0 = df_conv0(feat_spec) [B][T][96 bands][64 ch] (reference onnx_model/dpdfnet.py:94-101) feeds
//   * df_conv1: depthwise k(1,3) stride 2 over bands + pointwise 64x64 + BN + ReLU -> c1 [B][T][48][64]
//     (dpdfnet.py:102, layers.py:761-834), and
//   * the DF decoder's pathway conv df_convp: grouped k(5,1) conv over the last FIVE frames + pointwise 10x10 + BN
//     + ReLU -> 10 values per (frame, band) (dpdfnet.py:424-431, 508-515; folded on the host into one [320 -> 10]
//     matrix), added to df_out's taps in stage 2.
// As two gemm_rows launches c0 was read 1x by the first and ~1.5x by the second (five row tiles touch every c0 row, the
// L2 catches most) -- 15 GB per step of the headline workload, the pathway GEMM latency-bound at 1.6 TB/s with five
// dependent K panels per tile.  Here a workgroup owns (clip, 32-band tile) and walks TIME: the last five c0 frames of
// its tile live in an LDS ring, every frame is loaded from HBM exactly once (plus one halo band for the stride-2 conv),
// and both products come out of the ring:
//     pathway  [32 rows x K=320] . [320 x 16]   waves = 2 row tiles x 2 K halves, halves summed through LDS
//     df_conv1 depthwise on the VALU from the newest ring slot -> [16 rows x 64] . [64 x 64], wave = 16-column tile
// The pathway result p = relu(. + BN shift) does not depend on the GRUs, so it is produced here in stage 1 and only
// ADDED to tanh(df_out) in stage 2 (DfOutEpi).  Used when clips x 3 workgroups fill the chip; small batches keep the
// two gemm_rows launches (time-parallel, latency-friendly).
#pragma once
#include "common.h"

struct DfRingArgs {
    float* c0;             // [B][4 + Tc][96][64]: 4 halo frames (imported from the state FIFO) in front of each clip.
                           // CONV0: only the halo is read and only the last five frames of the chunk are written (state export)
    float* c1;             // [B*Tc][48][64]
    float* p;              // [B*Tc][96][10]
    const float* dw;       // df_conv1 depthwise [64][3]
    const float* pwfrag;   // df_conv1 pointwise (BN folded), MFMA B fragments [chunk 4][tile 4][kb 4][lane 64]
    const float* pwbias;   // [64]
    const float* cpfrag;   // pathway [320 -> 16], fragments [chunk 20][kb 4][lane 64]
    const float* cpbias;   // [10]
    int B, Tc;
    // CONV0 = true: df_conv0 itself runs here as well (c0 is never written to HBM except the frames the state FIFO needs)
    const float* fs;       // feat_spec [B][2 + Tc][2][96]: halo 2
    const float* c0frag;   // df_conv0 folded [32 -> 64] (gemm_rows.h Conv0DfA: k = kt*8 + g*4 + kf), fragments [chunk 2][tile 4][kb 4][lane 64]
    const float* c0bias;   // [64]
};

// CONV0: c0[t] of the tile is computed from the feature frames t-2..t (kept in a 3-slot LDS ring) straight into the ring
// slot -- [33 (48) rows x K = 32] . [32 x 64], wave = 16-column tile, 24 MFMAs -- instead of being loaded: the 805 MB per
// launch that df_conv0 wrote and this kernel read back disappear, and so does the df_conv0 launch (HBM-write-bound at
// 3.5 TB/s, 1.8 ms per step).
template <bool CONV0>
__global__ __launch_bounds__(256, 3) void df_ring_kernel(DfRingArgs a) {
    constexpr int D = 96, FD = 48;
    __shared__ __attribute__((aligned(16))) float R[5][33][68];     // ring of frames; row 0 = halo band f0-1, rows 1..32 = bands f0..f0+31
    __shared__ __attribute__((aligned(16))) float A1[16][68];       // depthwise output rows of the newest frame
    __shared__ float Pz[2][4][64];                                  // pathway partial sums of the upper K half
    __shared__ float FSr[CONV0 ? 3 : 1][2][CONV0 ? 52 : 1];         // feature frames t..t+2 (buffer index), bands f0-2 .. f0+49 (zero outside 0..95)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / 3, jt = blockIdx.x - 3 * b;
    const int f0 = 32 * jt, fo0 = 16 * jt;
    const int mt = w & 1, kh = w >> 1;

    float cp[40], pw[16];
#pragma unroll
    for (int i = 0; i < 40; ++i) cp[i] = a.cpfrag[(size_t)(40 * kh + i) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) pw[c * 4 + kb] = a.pwfrag[(size_t)((c * 4 + w) * 4 + kb) * 64 + lane];
    const float pwb = a.pwbias[16 * w + cl];
    const float cpb = cl < 10 ? a.cpbias[cl] : 0.f;
    const int fo_l = tid >> 4, c4 = (tid & 15) * 4;                 // depthwise: this thread's output row / channel quad
    float dwv[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) dwv[j][k] = a.dw[(c4 + j) * 3 + k];

    // frame tile loader: 33 rows x 16 float4 = 528 pieces over 256 threads (3 rounds, the last one 16 threads wide)
    const float* cbase = a.c0 + ((size_t)b * (a.Tc + 4) + 4) * D * 64;           // frame 0 of this clip
    auto load_frame = [&](int t, float4 (&v)[3]) __attribute__((always_inline)) {
        const float* fb = cbase + (long)t * D * 64;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, band = f0 - 1 + row;
            v[i] = (idx < 528 && band >= 0) ? *(const float4*)(fb + (size_t)band * 64 + (idx & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_frame = [&](int slot, const float4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 528) *(float4*)&R[slot][idx >> 4][(idx & 15) * 4] = v[i];
        }
    };
    float4 fr[3];
#pragma unroll
    for (int h = 0; h < 4; ++h) {             // the four halo frames t = -4..-1 -> slots 1..4
        load_frame(h - 4, fr);
        store_frame(h + 1, fr);
    }
    // ---- CONV0 state: feature ring + df_conv0 operand
    float c0w[8]; float c0b = 0.f;
    const float* fsb = nullptr;
    auto load_feat = [&](int bf) __attribute__((always_inline)) {      // buffer frame bf -> FSr slot bf % 3
        if (tid < 104) {
            const int g = tid / 52, i = tid - 52 * g, band = f0 - 2 + i;
            FSr[bf % 3][g][i] = (band >= 0 && band < D && bf < a.Tc + 2) ? fsb[((size_t)bf * 2 + g) * D + band] : 0.f;
        }
    };
    if (CONV0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) c0w[c * 4 + kb] = a.c0frag[(size_t)((c * 4 + w) * 4 + kb) * 64 + lane];
        c0b = a.c0bias[16 * w + cl];
        fsb = a.fs + (size_t)b * (a.Tc + 2) * 2 * D;
        load_feat(0); load_feat(1); load_feat(2);
        __syncthreads();
    } else {
        load_frame(0, fr);
    }

    int slot = 0;                              // slot of frame t: (t mod 5)
    for (int t = 0; t < a.Tc; ++t) {
        if (CONV0) {
            // df_conv0 for tile rows r = 0..32 (band f0 - 1 + r): A[r][k = kt*8 + g*4 + kf] = feat[t + kt][g][band + kf - 1]
            f32x4 c0acc[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) c0acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int kt = 2 * c + (q >> 1), g = q & 1;
                const bool live = kt < 3;                               // kt = 3: zero-padded K rows (zero weights)
                const float* fp = &FSr[live ? (t + kt) % 3 : 0][g][cl];
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const float x0 = live ? fp[mt * 16] : 0.f, x1 = live ? fp[mt * 16 + 1] : 0.f, x2 = live ? fp[mt * 16 + 2] : 0.f;
                    c0acc[mt] = mfma16(x0, c0w[c * 4 + 0], c0acc[mt]);
                    c0acc[mt] = mfma16(x1, c0w[c * 4 + 1], c0acc[mt]);
                    c0acc[mt] = mfma16(x2, c0w[c * 4 + 2], c0acc[mt]);
                    // (kb = 3 is the zero padding of the three band taps to four: its product is skipped, not issued)
                }
            }
            const bool keep = t + 5 >= a.Tc;                            // the state FIFO exports the chunk's last five frames
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = mt * 16 + q * 4 + i;
                    if (r < 33) {
                        const float v = (r == 0 && f0 == 0) ? 0.f : relu_f(c0acc[mt][i] + c0b);
                        R[slot][r][16 * w + cl] = v;
                        if (keep && r >= 1) a.c0[(((size_t)b * (a.Tc + 4) + 4 + t) * D + f0 + r - 1) * 64 + 16 * w + cl] = v;
                    }
                }
        } else {
            store_frame(slot, fr);
            if (t + 1 < a.Tc) load_frame(t + 1, fr);
        }
        __syncthreads();                       // ring holds frames t-4 .. t
        if (CONV0) load_feat(t + 3);           // slot t % 3 is free: its last readers are behind the barrier above
        // ---- pathway conv: rows mt*16.., K chunks [10 kh, 10 kh + 10): chunk cg = 4 kt + cc <-> frame t-4+kt, channels 16 cc..
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < 10; ++ci) {
            const int cg = 10 * kh + ci, kt = cg >> 2, cc = cg & 3;
            int s = slot + 1 + kt; s = s >= 5 ? s - 5 : s;           // frame t-4+kt = slot (slot + 1 + kt) mod 5
            const float4 a4 = *(const float4*)&R[s][1 + mt * 16 + cl][cc * 16 + 4 * q];
            acc = mfma16(a4.x, cp[ci * 4 + 0], acc);
            acc = mfma16(a4.y, cp[ci * 4 + 1], acc);
            acc = mfma16(a4.z, cp[ci * 4 + 2], acc);
            acc = mfma16(a4.w, cp[ci * 4 + 3], acc);
        }
        // ---- df_conv1 depthwise (stride 2, zero pad 1): output band fo <- c0 bands 2 fo - 1 .. 2 fo + 1 = ring rows 2 fo_l + {0,1,2}
        {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 x = *(const float4*)&R[slot][2 * fo_l + k][c4];
                v.x += dwv[0][k] * x.x; v.y += dwv[1][k] * x.y; v.z += dwv[2][k] * x.z; v.w += dwv[3][k] * x.w;
            }
            *(float4*)&A1[fo_l][c4] = v;
        }
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Pz[mt][i][lane] = acc[i];
        }
        __syncthreads();
        // ---- df_conv1 pointwise: [16 x 64] . [64 x 16 w ..]
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a4 = *(const float4*)&A1[cl][c * 16 + 4 * q];
            a1 = mfma16(a4.x, pw[c * 4 + 0], a1);
            a1 = mfma16(a4.y, pw[c * 4 + 1], a1);
            a1 = mfma16(a4.z, pw[c * 4 + 2], a1);
            a1 = mfma16(a4.w, pw[c * 4 + 3], a1);
        }
        const size_t bt = (size_t)b * a.Tc + t;
        {
            float* dst = a.c1 + (bt * FD + fo0 + q * 4) * 64 + 16 * w + cl;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(size_t)i * 64] = relu_f(a1[i] + pwb);
        }
        if (kh == 0 && cl < 10) {
            float* dst = a.p + (bt * D + f0 + mt * 16 + q * 4) * 10 + cl;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(size_t)i * 10] = relu_f(acc[i] + Pz[mt][i][lane] + cpb);
        }
        slot = slot == 4 ? 0 : slot + 1;
    }
}
