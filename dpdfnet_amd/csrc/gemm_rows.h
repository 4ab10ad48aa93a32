// gemm_rows.h -- the feed-forward workhorse: out[r, :] = epi(A[r, 0:K] . W[K, N]) for millions of
// short rows (r = (clip, frame[, band]) ), K <= 1008, N <= 768.  fp32 in / fp32 accumulate on the
// CDNA4 matrix cores (v_mfma_f32_16x16x4_f32, bit-exact fp32 FMA chains).
//
//   * 256-thread workgroup = 4 wavefronts; a workgroup owns a 64-row tile, each wave 16 rows x
//     NT 16-column MFMA tiles (NT <= 8).  blockIdx.y selects the weight group / column block.
//   * A is never read "as stored": an A-producer functor stages a [64 x KP] panel in LDS --
//     this is where depthwise convs, sub-pixel upsampling, pathway convs, FIFO (time-halo)
//     gathers, STFT framing/windowing are fused, so those intermediates never touch HBM.
//     Producers are split into load() (global -> registers, issued BEFORE the MFMA block of the
//     current panel) and store() (registers -> other LDS buffer, AFTER it): HBM latency of the
//     next panel hides under the current panel's matrix work, one barrier per panel.
//   * W is pre-packed on the host into MFMA B-fragment order ([group][chunk][tile][kb][lane]),
//     so every weight load is one coalesced 256-byte wave transaction served from L2; small
//     panels (K*NT*16 <= 128 VGPR) are hoisted into registers once per workgroup (PERSIST_B) and
//     reused across a grid-stride loop over row tiles.
//   * The epilogue functor sees the accumulators in MFMA C layout (row = (lane>>4)*4+i,
//     col = tile*16 + (lane&15)): bias/BN shift, activation, LayerNorm over the 64 channels
//     (in-register butterfly over the 16 column lanes), residual add, scatter to halo'd tensors.
//     Its prefetch() (residual / skip operands) is issued before the MFMA block as well.
#pragma once
#include "common.h"

constexpr int GEMM_BM = 64;

// ----- tensor view: [B][Tt][Fp][C] with a time halo of H frames in front of each clip ---------
struct TView {
    float* p;
    int Tt, H, Fp, C;      // frames per clip in the buffer (halo + chunk), halo, bands, channels
    __device__ __forceinline__ float* at(int b, int t, int f) const {
        return p + (((size_t)b * Tt + H + t) * Fp + f) * C;
    }
};

// n / d for 0 <= n < 2^31 with a host-precomputed multiplier: q = umulhi(n, m) >> sh
// (Granlund-Montgomery with s = ceil(log2 d), m = floor(2^(31+s)/d) + 1 < 2^32, sh = s - 1).
struct FastDiv {
    unsigned m; int sh; int d;
    __host__ static FastDiv make(int d_) {
        FastDiv f; f.d = d_; f.m = 0; f.sh = -1;
        if (d_ <= 1) return f;
        int s = 0; while ((1ll << s) < d_) ++s;
        f.m = (unsigned)(((1ull << (31 + s)) / (unsigned long long)d_) + 1ull);
        f.sh = s - 1;
        return f;
    }
    __device__ __forceinline__ int div(int n) const { return sh < 0 ? n : (int)(__umulhi((unsigned)n, m) >> sh); }
};

struct RowMap {            // flat row r over (b, t, f): t in [0,Tc), f in [0,Fp)
    int Tc, Fp; FastDiv dT, dF;
    __host__ static RowMap make(int Tc_, int Fp_) { return RowMap{Tc_, Fp_, FastDiv::make(Tc_), FastDiv::make(Fp_)}; }
    __device__ __forceinline__ void split(int r, int& b, int& t, int& f) const {
        int bt = dF.div(r); f = r - bt * Fp; b = dT.div(bt); t = bt - b * Tc;
    }
};

struct NoRegs {};

// ================================= A producers ================================================
// load(R&, row0, kp, grp, M): issue the global loads of panel rows [row0,row0+64) x K cols [kp,kp+KP)
// store(As, R&, row0, kp, grp, M): finish the computation and write As[64][KP+4]

template <int KP>
struct PlainA {            // A[r][k] = src[r*lda + grp*gstride + k], zero for k >= kmax or r >= M
    const float* src; size_t lda; int gstride; int kmax;
    static constexpr int V = KP / 4, NI = (GEMM_BM * V + 255) / 256;
    struct Regs { float4 v[NI]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int grp, int M) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            int row = row0 + r, k = kp + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < GEMM_BM * V && row < M) {
                const float* s = src + (size_t)row * lda + (size_t)grp * gstride + k;
                if (k + 3 < kmax) v = *(const float4*)s;
                else {
                    if (k < kmax) v.x = s[0];
                    if (k + 1 < kmax) v.y = s[1];
                    if (k + 2 < kmax) v.z = s[2];
                }
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[KP + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            if (idx < GEMM_BM * V) *(float4*)&As[r][c4] = R.v[i];
        }
    }
};

// A[r][k] = a[r*lda + grp*gstride + k] + b[same]  (DF decoder: c = df_gru(emb) + df_skip(emb), reference onnx_model/dpdfnet.py:503-506,
// added on the way into df_out instead of by a kernel of its own; small launches)
template <int KP>
struct SumA {
    const float* src; const float* src2; size_t lda; int gstride; int kmax;
    static constexpr int V = KP / 4, NI = (GEMM_BM * V + 255) / 256;
    struct Regs { float4 v[NI]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int grp, int M) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            int row = row0 + r, k = kp + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < GEMM_BM * V && row < M && k + 3 < kmax) {
                const size_t o = (size_t)row * lda + (size_t)grp * gstride + k;
                const float4 x = *(const float4*)(src + o), y = *(const float4*)(src2 + o);
                v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[KP + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            if (idx < GEMM_BM * V) *(float4*)&As[r][c4] = R.v[i];
        }
    }
};

// depthwise k(1,3) conv, zero pad 1, frequency stride S, fused in front of the pointwise GEMM
// (reference Conv2dNormAct separable, onnx_model/layers.py:761-834).  K = C = 64.
template <int S>
struct DwConvA {
    TView x; RowMap rm;    // rm.Fp = output bands; x.Fp = input bands
    const float* dw;       // [64][3]
    struct Regs { float4 v[4][3]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int, int, int M) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx >> 4, c4 = (idx & 15) * 4;
            int row = row0 + r;
            int b = 0, t = 0, fo = 0;
            if (row < M) rm.split(row, b, t, fo);
            const float* xr = x.at(b, t, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int fi = fo * S + j - 1;
                R.v[i][j] = (row < M && fi >= 0 && fi < x.Fp) ? *(const float4*)(xr + (size_t)fi * 64 + c4)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(float (*As)[64 + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx >> 4, c4 = (idx & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                v.x += dw[(c4 + 0) * 3 + j] * R.v[i][j].x; v.y += dw[(c4 + 1) * 3 + j] * R.v[i][j].y;
                v.z += dw[(c4 + 2) * 3 + j] * R.v[i][j].z; v.w += dw[(c4 + 3) * 3 + j] * R.v[i][j].w;
            }
            *(float4*)&As[r][c4] = v;
        }
    }
};

// decoder stage: u = relu(ps*e + pb) + prev   (pathway conv = per-channel scale + BN + ReLU,
// SURVEY appendix A.1; reference onnx_model/dpdfnet.py:361-364), then sub-pixel upsampling:
// S depthwise k(1,3) convs on u interleaved along frequency (layers.py:895-916), S=1 = plain.
template <int S>
struct SubpixA {
    TView e, prev; RowMap rm;          // rm.Fp = output bands = e.Fp * S
    const float* ps; const float* pb;  // folded pathway scale / shift [64]
    const float* dw;                   // [S][64][3]
    struct Regs { float4 ev[4][3], pv[4][3]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int, int, int M) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx >> 4, c4 = (idx & 15) * 4;
            int row = row0 + r;
            int b = 0, t = 0, fo = 0;
            if (row < M) rm.split(row, b, t, fo);
            int f = fo / S;
            const float* er = e.at(b, t, 0);
            const float* pr = prev.at(b, t, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int fi = f + j - 1;
                bool okk = row < M && fi >= 0 && fi < e.Fp;
                R.ev[i][j] = okk ? *(const float4*)(er + (size_t)fi * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                R.pv[i][j] = okk ? *(const float4*)(pr + (size_t)fi * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(float (*As)[64 + 4], const Regs& R, int row0, int, int, int M) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx >> 4, c4 = (idx & 15) * 4;
            int row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                int fo = row % rm.Fp;
                int f = fo / S, k = fo - f * S;
                const float* w = dw + (size_t)k * 64 * 3;
                float4 s4 = *(const float4*)(ps + c4), b4 = *(const float4*)(pb + c4);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    int fi = f + j - 1;
                    if (fi >= 0 && fi < e.Fp) {
                        float ux = relu_f(s4.x * R.ev[i][j].x + b4.x) + R.pv[i][j].x;
                        float uy = relu_f(s4.y * R.ev[i][j].y + b4.y) + R.pv[i][j].y;
                        float uz = relu_f(s4.z * R.ev[i][j].z + b4.z) + R.pv[i][j].z;
                        float uw = relu_f(s4.w * R.ev[i][j].w + b4.w) + R.pv[i][j].w;
                        v.x += w[(c4 + 0) * 3 + j] * ux; v.y += w[(c4 + 1) * 3 + j] * uy;
                        v.z += w[(c4 + 2) * 3 + j] * uz; v.w += w[(c4 + 3) * 3 + j] * uw;
                    }
                }
            }
            *(float4*)&As[r][c4] = v;
        }
    }
};

// df_conv0 (reference onnx_model/dpdfnet.py:94-101, layers.py:761-834, 1083-1114): GroupedConv2D(2 groups, 1->32 each,
// k(3,3)) -> pointwise 64x64 -> BatchNorm -> ReLU.  Nothing non-linear sits between the grouped and the pointwise
// conv, so the host folds the chain into ONE [18 -> 64] matrix (+ BN shift) and the kernel is an im2col GEMM with
// K = 32: column k = kt*8 + g*4 + kf holds feat_spec[frame t-2+kt][group g][band f+kf-1] (kf = 3 and kt = 3 are zero
// padding).  The first version computed the grouped conv on the VALU in front of a K = 64 pointwise GEMM: 860 VALU
// instructions per wave-tile against 64 MFMAs -- and VALU cycles are MFMA cycles on this chip.  Now: ~40 and 32.
struct Conv0DfA {
    const float* fs;       // feat_spec [B][2+Tc][2][D]
    int Tt, D; RowMap rm;  // rm.Fp = D
    struct Regs { float v[8]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int, int, int M) const {
        const int r = threadIdx.x >> 2, kt = threadIdx.x & 3, row = row0 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) R.v[j] = 0.f;
        if (row < M && kt < 3) {
            int b, t, f; rm.split(row, b, t, f);
            const float* p = fs + (((size_t)b * Tt + t + kt) * 2) * D + f;      // frame t-2+kt (+halo 2), group 0
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (f > 0) R.v[4 * g + 0] = p[g * D - 1];
                R.v[4 * g + 1] = p[g * D];
                if (f + 1 < D) R.v[4 * g + 2] = p[g * D + 1];
            }
        }
    }
    __device__ __forceinline__ void store(float (*As)[32 + 4], const Regs& R, int, int, int, int) const {
        const int r = threadIdx.x >> 2, kt = threadIdx.x & 3;
        *(float4*)&As[r][kt * 8] = make_float4(R.v[0], R.v[1], R.v[2], R.v[3]);
        *(float4*)&As[r][kt * 8 + 4] = make_float4(R.v[4], R.v[5], R.v[6], R.v[7]);
    }
};

// df_convp gather: A[(b,t,f)][kt*64 + c] = c0[b, t-4+kt, f, c]  (K = 5*64, panel kp/64 = kt)
struct ConvpA {
    TView c0; RowMap rm;   // c0.H = 4
    struct Regs { float4 v[4]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int, int M) const {
        const int kt = kp >> 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx >> 4, c4 = (idx & 15) * 4;
            int row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                int b, t, f; rm.split(row, b, t, f);
                v = *(const float4*)(c0.at(b, t - 4 + kt, f) + c4);
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[64 + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = threadIdx.x + i * 256;
            *(float4*)&As[idx >> 4][(idx & 15) * 4] = R.v[i];
        }
    }
};

// STFT framing: A[(b,t)][k] = window[k] * x_pad[b][t*hop + k - win/2], x_pad = reflect(np.pad(wav,(0,win)))
// (reference package/src/dpdfnet/audio.py:104-117 + api.py:88)
template <int KP>
struct StftA {
    const float* wav; int N; int T; int win, hop; const float* window;
    int causal = 0;        // 1: frame t = x[t*hop : t*hop+win] (StreamEnhancer), no centre/reflect padding
    const int* lens = nullptr;   // ragged batch: clip b holds lens[b] <= N valid samples (row stride stays N); its own tail pad,
                                 // reflection point and frame count T_b = 1 + (lens[b] + win) / hop; frames t >= T_b are zero
    const float* tail = nullptr; // causal streaming: the stream's signal is [tail[b][0:hop] | wav[b][0:N - hop]] (N counts both) --
                                 // the analysis buffer of the previous call and the new samples, read in place (no staging copy)
    static constexpr int NI = GEMM_BM * KP / 256;
    struct Regs { float v[NI]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int, int M) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / KP, k = idx - r * KP;
            int row = row0 + r;
            float v = 0.f;
            if (row < M) {
                int b = row / T, t = row - b * T;
                const int nb_ = lens ? lens[b] : N;
                const int np_ = nb_ + win;
                int kk = kp + k;
                int j = t * hop + kk;
                if (!causal) {
                    j -= win / 2;
                    if (j < 0) j = -j;
                    if (j >= np_) j = 2 * (np_ - 1) - j;
                }
                const bool live = !lens || t < 1 + np_ / hop;
                if (live && j >= 0 && j < nb_ && kk < win)
                    v = (tail ? (j < hop ? tail[(size_t)b * hop + j] : wav[(size_t)b * (N - hop) + (j - hop)]) : wav[(size_t)b * N + j]) * window[kk];
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[KP + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / KP, k = idx - r * KP;
            As[r][k] = R.v[i];
        }
    }
};

// ================================= epilogues ==================================================
// prefetch(P&, row0, lane, grp, M) is issued before the MFMA block;
// call(acc, P, row0, lane, grp, M): acc[nt][i] = C[row0 + (lane>>4)*4 + i][nt*16 + (lane&15)]

template <int NT>
struct BiasActStore {      // out[r*ldo + grp*gostride + col] = act(acc + bias[grp*gbstride + col]), col < N
    float* out; size_t ldo; int gostride; const float* bias; int gbstride; int N; int act;
    int ncol_total = 0;    // >0: also bound the global column grp*gostride+col (partial last group)
    using Pref = NoRegs;
    __device__ __forceinline__ void prefetch(Pref&, int, int, int, int) const {}
    __device__ __forceinline__ void call(f32x4 (&acc)[NT], const Pref&, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int col = nt * 16 + cl;
            if (col >= N) continue;
            if (ncol_total > 0 && grp * gostride + col >= ncol_total) continue;
            float bv = bias ? bias[grp * gbstride + col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = row0 + rq + i;
                if (row < M) out[(size_t)row * ldo + (size_t)grp * gostride + col] = apply_act(acc[nt][i] + bv, act);
            }
        }
    }
};

// conv output (+folded BN shift, ReLU) into a halo'd [B][Tt][Fp][64] tensor
struct BiasReluToView {
    TView o; RowMap rm; const float* bias;
    using Pref = NoRegs;
    __device__ __forceinline__ void prefetch(Pref&, int, int, int, int) const {}
    __device__ __forceinline__ void call(f32x4 (&acc)[4], const Pref&, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
        float bv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt] = bias[nt * 16 + cl];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = row0 + rq + i;
            if (row >= M) continue;
            int b, t, f; rm.split(row, b, t, f);
            float* dst = o.at(b, t, f);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) dst[nt * 16 + cl] = relu_f(acc[nt][i] + bv[nt]);
        }
    }
};

// Last decoder stage + mask head, fused (reference onnx_model/dpdfnet.py:320-323, 364-366): the GEMM produces
// d1 = relu(convt1(..) + BN shift); the mask head needs u = relu(ps*e0 + pb) + d1 (conv0p pathway + skip) and then
// conv0_out = dense 64->1 k(1,3) over frequency + BN + sigmoid.  The 64->1 contraction is done HERE, on the
// accumulators: s_k[row] = sum_c w[c][k] * u[row][c] for the three taps (one DPP row all-reduce each), 16 bytes per
// row go to HBM instead of the 256-byte d1 row (which the stand-alone mask kernel then read back together with e0:
// 1.4x its algorithmic traffic and no MFMA).  mask_fin_kernel adds the three neighbours and applies the sigmoid.
struct MaskSumEpi {
    const float* e0;       // [rows][64], same flat row index as the GEMM
    float* s;              // [rows][4]: s_0, s_1, s_2, unused
    const float* bias;     // convt1 BN shift [64]
    const float* ps; const float* pb;   // conv0p folded scale / shift [64]
    const float* w;        // conv0_out [64][3], BN folded
    struct Pref { float r[4][4]; };
    __device__ __forceinline__ void prefetch(Pref& P, int row0, int lane, int, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = row0 + rq + i;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) P.r[nt][i] = row < M ? e0[(size_t)row * 64 + nt * 16 + cl] : 0.f;
        }
    }
    __device__ __forceinline__ void call(f32x4 (&acc)[4], const Pref& P, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
        float bv[4], sc[4], sh[4], w0[4], w1[4], w2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = nt * 16 + cl;
            bv[nt] = bias[c]; sc[nt] = ps[c]; sh[nt] = pb[c];
            w0[nt] = w[c * 3 + 0]; w1[nt] = w[c * 3 + 1]; w2[nt] = w[c * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float u = relu_f(acc[nt][i] + bv[nt]) + relu_f(__builtin_fmaf(sc[nt], P.r[nt][i], sh[nt]));
                s0 = __builtin_fmaf(w0[nt], u, s0); s1 = __builtin_fmaf(w1[nt], u, s1); s2 = __builtin_fmaf(w2[nt], u, s2);
            }
            s0 = row16_allreduce_sum(s0); s1 = row16_allreduce_sum(s1); s2 = row16_allreduce_sum(s2);
            const int row = row0 + rq + i;
            if (cl == 0 && row < M) *(float4*)(s + (size_t)row * 4) = make_float4(s0, s1, s2, 0.f);
        }
    }
};

// Linear bias + LayerNorm(64, eps 1e-5, biased var) + residual  (DPRNNBlock fc_intra/ln_intra and
// fc_inter/ln_inter, reference onnx_model/layers.py:178-193).  Rows are contiguous [M][64].
struct LnResStore {
    float* out; const float* res; const float* bias; const float* g; const float* be;
    struct Pref { float r[4][4]; };
    __device__ __forceinline__ void prefetch(Pref& P, int row0, int lane, int, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = row0 + rq + i;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) P.r[nt][i] = row < M ? res[(size_t)row * 64 + nt * 16 + cl] : 0.f;
        }
    }
    __device__ __forceinline__ void call(f32x4 (&acc)[4], const Pref& P, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
        float v[4][4], gg[4], bb[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float bv = bias[nt * 16 + cl];
            gg[nt] = g[nt * 16 + cl]; bb[nt] = be[nt * 16 + cl];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[nt][i] = acc[nt][i] + bv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = v[0][i] + v[1][i] + v[2][i] + v[3][i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
            float mean = s * (1.0f / 64.0f);
            float d0 = v[0][i] - mean, d1 = v[1][i] - mean, d2 = v[2][i] - mean, d3 = v[3][i] - mean;
            float q = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) q += __shfl_xor(q, off, 64);
            float inv = rsqrtf(q * (1.0f / 64.0f) + 1e-5f);
            int row = row0 + rq + i;
            if (row < M) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float y = (v[nt][i] - mean) * inv * gg[nt] + bb[nt];
                    out[(size_t)row * 64 + nt * 16 + cl] = P.r[nt][i] + y;
                }
            }
        }
    }
};

// df_convp epilogue: coefs[b, t, f, k] = relu(acc + bias[k]) + df_out[b, t, f*10 + k]   (k < 10)
// (reference onnx_model/dpdfnet.py:508-515; pointwise 10->10 + BN are folded into the GEMM weights)
struct ConvpEpi {
    float* coefs; int Tt; RowMap rm;   // coefs [B][2+Tc][D][10], halo 2
    const float* dfo;                  // [B*Tc][D*10]
    const float* bias;
    struct Pref { float d[4]; };
    __device__ __forceinline__ void prefetch(Pref& P, int row0, int lane, int, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = row0 + rq + i;
            P.d[i] = (cl < 10 && row < M) ? dfo[(size_t)row * 10 + cl] : 0.f;
        }
    }
    __device__ __forceinline__ void call(f32x4 (&acc)[1], const Pref& P, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
        if (cl >= 10) return;
        float bv = bias[cl];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = row0 + rq + i;
            if (row >= M) continue;
            int b, t, f; rm.split(row, b, t, f);
            float v = relu_f(acc[0][i] + bv) + P.d[i];
            coefs[((((size_t)b * Tt + 2 + t) * rm.Fp) + f) * 10 + cl] = v;
        }
    }
};

// df_out epilogue when the pathway conv ran in stage 1 (df_ring.h): coefs[b, 2 + t, col] = tanh(acc + bias) + p[row, col],
// col = grp * 60 + n  (reference onnx_model/dpdfnet.py:508-515: c = df_out(..).tanh() ; c + pathway)
struct DfOutEpi {
    float* coefs; int Tc; FastDiv dT;      // coefs [B][2 + Tc][960], halo 2
    const float* p;                        // [B*Tc][960]
    const float* bias; int Og;             // 60 valid columns per group
    struct Pref { float d[4][4]; };
    __device__ __forceinline__ void prefetch(Pref& P, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nt * 16 + cl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + rq + i;
                P.d[nt][i] = (col < Og && row < M) ? p[(size_t)row * 960 + grp * Og + col] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void call(f32x4 (&acc)[4], const Pref& P, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nt * 16 + cl;
            if (col >= Og) continue;
            const float bv = bias[grp * Og + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + rq + i;
                if (row >= M) continue;
                const int b = dT.div(row), t = row - b * Tc;
                coefs[((size_t)b * (Tc + 2) + 2 + t) * 960 + grp * Og + col] = tanh_f(acc[nt][i] + bv) + P.d[nt][i];
            }
        }
    }
};

// iSTFT frame epilogue: frames[r][n] = acc * window[n]
template <int NT>
struct WindowStore {
    float* out; int win; const float* window;
    using Pref = NoRegs;
    __device__ __forceinline__ void prefetch(Pref&, int, int, int, int) const {}
    __device__ __forceinline__ void call(f32x4 (&acc)[NT], const Pref&, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int col = grp * NT * 16 + nt * 16 + cl;
            if (col >= win) continue;
            float wv = window[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = row0 + rq + i;
                if (row < M) out[(size_t)row * win + col] = acc[nt][i] * wv;
            }
        }
    }
};

// ============================ time-slice forms (pipelined host I/O) ============================
// A host-pointer batch call is pipelined over time slices (dpdf_model.hip, enhance_host_pipelined): the PCM of time chunk k+1 is
// uploaded and the PCM of chunk k-1 downloaded under the compute of chunk k, so the STFT and the iSTFT run PER CHUNK on the
// rows (clip b, frames t0 .. t0+Tc) of tensors laid out [B][T][...].  RowSeg maps the launch's dense row index onto them.
struct RowSeg {
    int Tc, T, t0;
    __device__ __forceinline__ size_t map(int row) const { const int b = row / Tc; return (size_t)b * T + t0 + (row - b * Tc); }
};
// StftA on the rows of one time chunk (same arithmetic per element: window[k] * x_pad[...])
template <int KP>
struct StftSegA {
    const float* wav; int N; int T; int win, hop; const float* window; const int* lens; RowSeg seg;
    static constexpr int NI = GEMM_BM * KP / 256;
    struct Regs { float v[NI]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int, int M) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / KP, k = idx - r * KP;
            int row = row0 + r;
            float v = 0.f;
            if (row < M) {
                const int b = row / seg.Tc, t = seg.t0 + (row - b * seg.Tc);
                const int nb_ = lens ? lens[b] : N;
                const int np_ = nb_ + win;
                int kk = kp + k;
                int j = t * hop + kk - win / 2;
                if (j < 0) j = -j;
                if (j >= np_) j = 2 * (np_ - 1) - j;
                const bool live = !lens || t < 1 + np_ / hop;
                if (live && j >= 0 && j < nb_ && kk < win) v = wav[(size_t)b * N + j] * window[kk];
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[KP + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / KP, k = idx - r * KP;
            As[r][k] = R.v[i];
        }
    }
};
template <int KP>
struct PlainSegA {         // PlainA whose row r is row seg.map(r) of src
    const float* src; size_t lda; int kmax; RowSeg seg;
    static constexpr int V = KP / 4, NI = (GEMM_BM * V + 255) / 256;
    struct Regs { float4 v[NI]; };
    __device__ __forceinline__ void load(Regs& R, int row0, int kp, int, int M) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            int row = row0 + r, k = kp + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < GEMM_BM * V && row < M) {
                const float* s = src + seg.map(row) * lda + k;
                if (k + 3 < kmax) v = *(const float4*)s;
                else {
                    if (k < kmax) v.x = s[0];
                    if (k + 1 < kmax) v.y = s[1];
                    if (k + 2 < kmax) v.z = s[2];
                }
            }
            R.v[i] = v;
        }
    }
    __device__ __forceinline__ void store(float (*As)[KP + 4], const Regs& R, int, int, int, int) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = threadIdx.x + i * 256;
            int r = idx / V, c4 = (idx - r * V) * 4;
            if (idx < GEMM_BM * V) *(float4*)&As[r][c4] = R.v[i];
        }
    }
};
template <int NT>
struct SegStore {          // out[seg.map(r)*ldo + grp*gostride + col] = acc (the STFT's store, no bias / activation)
    float* out; size_t ldo; int gostride; int N; int ncol_total; RowSeg seg;
    using Pref = NoRegs;
    __device__ __forceinline__ void prefetch(Pref&, int, int, int, int) const {}
    __device__ __forceinline__ void call(f32x4 (&acc)[NT], const Pref&, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int col = nt * 16 + cl;
            if (col >= N) continue;
            if (grp * gostride + col >= ncol_total) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = row0 + rq + i;
                if (row < M) out[seg.map(row) * ldo + (size_t)grp * gostride + col] = acc[nt][i];
            }
        }
    }
};
template <int NT>
struct WindowSegStore {    // WindowStore onto rows seg.map(r)
    float* out; int win; const float* window; RowSeg seg;
    using Pref = NoRegs;
    __device__ __forceinline__ void prefetch(Pref&, int, int, int, int) const {}
    __device__ __forceinline__ void call(f32x4 (&acc)[NT], const Pref&, int row0, int lane, int grp, int M) const {
        const int cl = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int col = grp * NT * 16 + nt * 16 + cl;
            if (col >= win) continue;
            float wv = window[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = row0 + rq + i;
                if (row < M) out[seg.map(row) * win + col] = acc[nt][i] * wv;
            }
        }
    }
};

// ================================= the kernel =================================================
// wfrag layout: [grp][chunk c][tile nt][kb][lane]  (chunk = 16 K values; value = W[kperm(c,lane>>4,kb)][nt*16+(lane&15)])
template <int NT, int KP, bool PERSIST_B, class AProd, class Epi>
__global__ __launch_bounds__(256) void gemm_rows_kernel(AProd ap, const float* __restrict__ wfrag, Epi ep,
                                                        int M, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][GEMM_BM][KP + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = blockIdx.y;
    const int nchunks = K / 16, npanels = K / KP;
    const float* wf = wfrag + (size_t)grp * nchunks * NT * 256 + lane;
    const int ntiles = (M + GEMM_BM - 1) / GEMM_BM;
    constexpr int PB = (KP / 16) * NT * 4;
    float breg[PB];
    if (PERSIST_B) {
#pragma unroll
        for (int i = 0; i < PB; ++i) breg[i] = wf[(size_t)i * 64];
    }
    // XCD-aware tile order: the hardware deals consecutive workgroups round-robin onto the 8 XCDs (block b -> XCD
    // b % 8), each with a private L2.  Row tiles that are neighbours in (clip, frame, band) order share input rows
    // through the time/frequency halos of the conv producers (up to 5x for the DF pathway conv), so each sweep of
    // gridDim.x tiles is cut into 8 contiguous slabs, one per XCD, instead of being interleaved across all of them.
    // K split (gridDim.z > 1: few-row launches whose K loop is a chain of load latencies, e.g. the STFT of one streaming hop):
    // block z accumulates panels [z, z + 1) * npanels / gridDim.z and hands its partial sums to the epilogue under the virtual
    // group index grp + z * gridDim.y; the consumer adds the partials in a fixed order (stream_ola_ksplit_kernel).
    const int pper = npanels / (int)gridDim.z, p0 = (int)blockIdx.z * pper, p1 = p0 + pper;
    const int vgrp = grp + (int)blockIdx.z * (int)gridDim.y;
    int tile = blockIdx.x, panel = p0, cur = 0;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    typename AProd::Regs R;
    ap.load(R, tile * GEMM_BM, p0 * KP, grp, M);
    ap.store(As[0], R, tile * GEMM_BM, p0 * KP, grp, M);
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    while (true) {
        // next (tile, panel) in the flattened sequence
        int ntile = tile, npanel = panel + 1;
        if (npanel == p1) { npanel = p0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < ntiles;
        if (!PERSIST_B) {
            // this panel's B fragments first: vmcnt retires in order, so the A prefetch issued after
            // them stays in flight while the MFMA block waits only for these L2-resident loads
            const float* wp = wf + (size_t)panel * PB * 64;
#pragma unroll
            for (int i = 0; i < PB; ++i) breg[i] = wp[(size_t)i * 64];
        }
        if (has_next) ap.load(R, ntile * GEMM_BM, npanel * KP, grp, M);
        typename Epi::Pref P;
        const bool last_panel = panel == p1 - 1;
        if (last_panel) ep.prefetch(P, tile * GEMM_BM + wave * 16, lane, vgrp, M);
        const float* arow = &As[cur][wave * 16 + (lane & 15)][(lane >> 4) * 4];
#pragma unroll
        for (int c = 0; c < KP / 16; ++c) {
            float4 a4 = *(const float4*)(arow + c * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int bi = (c * NT + nt) * 4;
                acc[nt] = mfma16(a4.x, breg[bi + 0], acc[nt]);
                acc[nt] = mfma16(a4.y, breg[bi + 1], acc[nt]);
                acc[nt] = mfma16(a4.z, breg[bi + 2], acc[nt]);
                acc[nt] = mfma16(a4.w, breg[bi + 3], acc[nt]);
            }
        }
        if (last_panel) {
            ep.call(acc, P, tile * GEMM_BM + wave * 16, lane, vgrp, M);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!has_next) break;
        ap.store(As[cur ^ 1], R, ntile * GEMM_BM, npanel * KP, grp, M);
        __syncthreads();
        cur ^= 1; tile = ntile; panel = npanel;
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_rows_wn_kernel: the same 64-row tile, waves split over COLUMNS instead of rows.  In gemm_rows_kernel every wave
// owns 16 rows and all NT column tiles, so the four waves each fetch the SAME B fragments of a panel (128 loads per
// lane for NT = 8): on wide, non-persistent operands that redundant traffic through the vector-memory path takes as
// long as the MFMAs.  Here wave w owns all 64 rows and the NTW column tiles of column group (grp*4 + w): a quarter of
// the B loads, none of them redundant; the A fragments of the four 16-row tiles come from the shared LDS panel.
// wfrag / epilogue are those of a gemm_rows launch with NT = NTW and 4x the groups.
template <int NTW, int KP, class AProd, class Epi>
__global__ __launch_bounds__(256) void gemm_rows_wn_kernel(AProd ap, const float* __restrict__ wfrag, Epi ep, int M, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][GEMM_BM][KP + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = blockIdx.y, cgrp = grp * 4 + wave;
    const int nchunks = K / 16, npanels = K / KP;
    const float* wf = wfrag + (size_t)cgrp * nchunks * NTW * 256 + lane;
    const int ntiles = (M + GEMM_BM - 1) / GEMM_BM;
    constexpr int PB = (KP / 16) * NTW * 4;
    float breg[PB];
    int tile = blockIdx.x, panel = 0, cur = 0;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    typename AProd::Regs R;
    ap.load(R, tile * GEMM_BM, 0, grp, M);
    ap.store(As[0], R, tile * GEMM_BM, 0, grp, M);
    __syncthreads();
    f32x4 acc[4][NTW];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    while (true) {
        int ntile = tile, npanel = panel + 1;
        if (npanel == npanels) { npanel = 0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < ntiles;
        {
            const float* wp = wf + (size_t)panel * PB * 64;
#pragma unroll
            for (int i = 0; i < PB; ++i) breg[i] = wp[(size_t)i * 64];
        }
        if (has_next) ap.load(R, ntile * GEMM_BM, npanel * KP, grp, M);
        const bool last_panel = panel == npanels - 1;
        const float* arow = &As[cur][lane & 15][(lane >> 4) * 4];
#pragma unroll
        for (int c = 0; c < KP / 16; ++c) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                float4 a4 = *(const float4*)(arow + rt * 16 * (KP + 4) + c * 16);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int bi = (c * NTW + nt) * 4;
                    acc[rt][nt] = mfma16(a4.x, breg[bi + 0], acc[rt][nt]);
                    acc[rt][nt] = mfma16(a4.y, breg[bi + 1], acc[rt][nt]);
                    acc[rt][nt] = mfma16(a4.z, breg[bi + 2], acc[rt][nt]);
                    acc[rt][nt] = mfma16(a4.w, breg[bi + 3], acc[rt][nt]);
                }
            }
        }
        if (last_panel) {
            typename Epi::Pref P;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                ep.call(acc[rt], P, tile * GEMM_BM + rt * 16, lane, cgrp, M);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!has_next) break;
        ap.store(As[cur ^ 1], R, ntile * GEMM_BM, npanel * KP, grp, M);
        __syncthreads();
        cur ^= 1; tile = ntile; panel = npanel;
    }
}

// groups4 = number of 4-wave column-group quadruples (the gemm_rows launch with NT = NTW would use 4*groups4 groups)
template <int NTW, int KP, class AProd, class Epi>
static inline void launch_gemm_rows_wn(hipStream_t st, const AProd& ap, const float* wfrag, const Epi& ep,
                                       int M, int K, int groups4, int max_blocks_x = 2048) {
    int ntiles = (M + GEMM_BM - 1) / GEMM_BM;
    if (ntiles <= 0) return;
    int gx = ntiles < max_blocks_x ? ntiles : max_blocks_x;
    hipLaunchKernelGGL((gemm_rows_wn_kernel<NTW, KP, AProd, Epi>), dim3(gx, groups4, 1), dim3(256), 0, st, ap, wfrag, ep, M, K);
}

template <int NT, int KP, bool PERSIST_B, class AProd, class Epi>
static inline void launch_gemm_rows(hipStream_t st, const AProd& ap, const float* wfrag, const Epi& ep,
                                    int M, int K, int groups, int max_blocks_x = 2048, int ksplit = 1) {
    int ntiles = (M + GEMM_BM - 1) / GEMM_BM;
    if (ntiles <= 0) return;
    int gx = ntiles < max_blocks_x ? ntiles : max_blocks_x;
    dim3 grid(gx, groups, ksplit);      // ksplit must divide K / KP
    hipLaunchKernelGGL((gemm_rows_kernel<NT, KP, PERSIST_B, AProd, Epi>), grid, dim3(256), 0, st, ap, wfrag, ep, M, K);
}

