// dft64.h -- the ANALYSIS transform in double precision: the float32 product frame * window (exactly the reference's operand,
// package/src/dpdfnet/stream.py:119 `windowed = in_buf[:win] * window`, audio.py:104-117) goes through a float64 DFT and is
// rounded to float32 once, which is what the reference's np.fft.rfft delivers (stream.py:120-126: float64 under its pinned
// numpy 1.26.4, and the same values to the last float32 bit under numpy 2.x) and what the CPU checker under tests restates.
//
// Why: at 48 kHz the features are 10 log10(|X| + 1e-10) PER BIN (onnx_model/dpdfnet_48khz_hr.py:887-924).  A bin that holds no
// signal (band-limited speech, 16-bit sources, DC) carries only the quantisation residue of the windowed frame, ~1e-7 of the
// loud bins -- and an fp32 accumulation chain (f32 MFMA = fmaf chain) or fp32 twiddles leave their own 1e-7 there, so that
// every fp32 DFT form (one GEMM, two stages, K split) fed the network different -60 .. -100 dB features: 1.6e-3 RMS on the
// enhanced waveform for band-limited audio, 16 x the parity budget.  In float64 the transform's own error is 1e-16 of the
// loud bins; the three call sites (offline batch, offline time chunks, streaming hops) now share ONE kernel and one result.
//
// Shape: the same Cooley-Tukey split as dft2stage.h (N = 32 x N2, n = N2 n1 + n2, k = k1 + 32 k2), both stages in one launch on
// v_mfma_f64_16x16x4_f64, the intermediate kept in float64 in LDS (rounding it to float32 would put the 1e-7 back):
//   workgroup = (16 frames, 8 of the 32 k1)
//   stage 1: rows (frame, n2), K = 32 samples n1, 16 columns = 8 complex k1, times the middle twiddle e^{-2 pi i n2 k1 / N}
//            (re / im of a value sit in neighbouring lanes: one DPP swap)            -> Y[frame][n2][k1]        (LDS, f64)
//   stage 2: rows (frame, k1), K = 2 N2 (n2, re/im), columns (k2, re/im): the plain N2-point DFT, ONE operand for every row,
//            held in registers                 -> X[k1 + 32 k2], k <= N / 2, gathered in an LDS tile, written as 64-byte runs.
// 960: 120 f64 MFMAs per frame (246 kFLOP); the f64 matrix rate is half the f32 one, so 256 x 1003 frames cost ~1 ms of the
// chip -- under 1 % of the 48 kHz models.  A operands of stage 1 are read straight from the clip (L2-resident, lanes of a
// quad read 64 contiguous bytes), windowed in float32, converted exactly.
#pragma once
#include "common.h"
#include "gemm_rows.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: lane l supplies A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15];
// D reg i holds D[row = (l >> 4) + 4 i][col = l & 15]  (NOT the f32 shape's 4 (l >> 4) + i).
__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int N2>
struct Dft64Cfg {
    static constexpr int N = 32 * N2, F = N / 2 + 1, SPEC = 2 * F;
    static constexpr int NK2 = N2 / 2 + 1;                 // k2 values that reach bins <= N / 2
    static constexpr int KS = (2 * N2 + 3) / 4;            // K steps of stage 2 (2 N2 real values, 4 per MFMA)
    static constexpr int NT = (2 * NK2 + 15) / 16;         // column tiles of stage 2
    static constexpr int MIDW = N2 * 16 + 2;               // doubles per frame of the intermediate (+2: 16 frames land on 16 distinct bank groups)
    static constexpr size_t TW1 = (size_t)4 * 8 * 64;      // doubles: stage-1 operand [k1 group 4][K step 8][lane 64]
    static constexpr size_t TWM = (size_t)N2 * 32 * 2;     // doubles: middle twiddle e^{-2 pi i n2 k1 / N} as [n2][k1][cos, sin]
    static constexpr size_t TW2 = (size_t)KS * NT * 64;    // doubles: stage-2 operand (the N2-point DFT) [K step][column tile][lane 64]
};

struct Dft64Args {
    const float* wav; int N; int T; int hop; const float* window; const int* lens;
    RowSeg seg;            // rows of this launch -> (clip, frame): b = row / seg.Tc, t = seg.t0 + row % seg.Tc; spectrum row = seg.map(row)
    int causal;            // 1: StreamEnhancer analysis, frame t = x[t hop : t hop + win], no padding; x = [tail | wav] when tail != null
    const float* tail;     // [clips][hop] or null (then wav holds the (T + 1) hop samples of a stream itself)
    float* spec;           // [..][F][2]
    int M;                 // rows
    const double* tw1; const double* twm; const double* tw2;
};

// the other half of a (re, im) lane pair: quad_perm [1, 0, 3, 2] on both dwords
__device__ __forceinline__ double dft64_pair_swap(double v) {
    const long long u = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffll), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0xB1, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// FT = frames per workgroup: 16 for big launches; 4 for the few frames of a streaming hop (64 streams = 64 workgroups of ~2 row
// tiles per wave instead of 16 of ~8: the launch is one workgroup's latency either way)
template <int N2, int FT>
__global__ __launch_bounds__(256) void dft64_fwd_kernel(Dft64Args g) {
    using C = Dft64Cfg<N2>;
    static_assert(FT == 16 || FT == 4, "stage 2 packs two frames per 16-row tile");
    __shared__ __attribute__((aligned(16))) double mid[FT][C::MIDW];
    __shared__ __attribute__((aligned(16))) float outs[FT][C::NK2][16];
    __shared__ int f_t[FT], f_nb[FT]; __shared__ long long f_base[FT];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, cl = lane & 15, q = lane >> 4;
    const int fr0 = blockIdx.x * FT, grp = blockIdx.y;
    if (tid < FT) {       // per frame of the tile: first sample index, clip base, clip length (-1: a frame beyond a short clip's last)
        const int fr = min(fr0 + tid, g.M - 1);
        const int b = fr / g.seg.Tc, t = g.seg.t0 + (fr - b * g.seg.Tc);
        const int nb_ = g.lens ? g.lens[b] : g.N;
        const bool live = g.causal || !g.lens || t < 1 + (nb_ + C::N) / g.hop;
        f_t[tid] = t * g.hop; f_nb[tid] = live ? nb_ : -1; f_base[tid] = b;
    }
    __syncthreads();

    // ---- stage 1: rows (frame, n2), one row tile ahead in flight ----
    double b1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) b1[kk] = g.tw1[((size_t)grp * 8 + kk) * 64 + lane];
    // (loads only: the products are formed in run_tile, one tile later, so that nothing here waits for memory)
    struct Tile { float x[8], wv[8]; unsigned ok; double c[4], s[4]; };
    auto load_tile = [&](int rt, Tile& tl) {
        const int r = min(rt * 16 + cl, FT * N2 - 1), f = r / N2, n2 = r - f * N2;      // (FT = 4: the last tile is half rows)
        const int t0 = f_t[f], nb_ = f_nb[f], np_ = nb_ + C::N;
        const size_t b = (size_t)f_base[f];
        tl.ok = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int n = N2 * (4 * kk + q) + n2;
            int j = t0 + n;
            const float* p;
            bool ok = true;
            if (g.causal) {
                p = g.tail ? (j < g.hop ? g.tail + b * g.hop + j : g.wav + b * (g.N - g.hop) + (j - g.hop)) : g.wav + b * g.N + j;
            } else {
                j -= C::N / 2;
                if (j < 0) j = -j;
                if (j >= np_) j = 2 * (np_ - 1) - j;
                ok = j >= 0 && j < nb_;
                p = g.wav + b * g.N + (ok ? j : 0);
            }
            tl.x[kk] = *p; tl.wv[kk] = g.window[n];
            tl.ok |= (unsigned)ok << kk;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                   // middle twiddles of the four result rows of this lane
            const int ro = min(rt * 16 + q + 4 * i, FT * N2 - 1), no = ro % N2;
            const double2 cs = *(const double2*)(g.twm + ((size_t)no * 32 + 8 * grp + (cl >> 1)) * 2);
            tl.c[i] = cs.x; tl.s[i] = cs.y;
        }
    };
    auto run_tile = [&](int rt, const Tile& tl) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float xw = ((tl.ok >> kk) & 1) ? tl.x[kk] * tl.wv[kk] : 0.f;     // the float32 product, as the reference forms it
            acc = mfma64((double)xw, b1[kk], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (re + i im) e^{-i th} = (re c + im s) + i (im c - re s); this lane holds re (even column) or im (odd), its neighbour the other
            const double v = acc[i], p = dft64_pair_swap(v);
            const double o = (cl & 1) ? (v * tl.c[i] - p * tl.s[i]) : (v * tl.c[i] + p * tl.s[i]);
            const int ro = rt * 16 + q + 4 * i, fo = ro / N2, no = ro - fo * N2;
            if (fo < FT) mid[fo][no * 16 + cl] = o;
        }
    };
    const int nf = min(FT, g.M - fr0);                       // live frames of this tile (a streaming hop of a few streams: 1 .. 15)
    const int nt1 = (nf * N2 + 15) / 16;                     // row tiles of stage 1 that hold a live frame
    if (w < nt1) {
        Tile ta, tb;
        int rt = w;
        load_tile(rt, ta);
        for (; rt + 4 < nt1; rt += 8) {
            load_tile(rt + 4, tb);
            run_tile(rt, ta);
            if (rt + 8 < nt1) load_tile(rt + 8, ta);
            run_tile(rt + 4, tb);
        }
        if (rt < nt1) run_tile(rt, ta);
    }
    // stage-2 operand: the N2-point DFT, the same for every row
    double b2[C::KS * C::NT];
#pragma unroll
    for (int i = 0; i < C::KS * C::NT; ++i) b2[i] = g.tw2[(size_t)i * 64 + lane];
    __syncthreads();

    // ---- stage 2: rows (frame, k1), FT / 2 row tiles of two frames; wave w takes tiles w, w + 4 ----
#pragma unroll
    for (int j = 0; j < (FT + 7) / 8; ++j) {
        const int tile = w + 4 * j, fa = min(2 * tile + (cl >> 3), FT - 1), k1a = cl & 7;
        if (2 * tile >= nf) break;                           // (frames of dead tiles: rows of mid that stage 1 did not write)
        f64x4 acc[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[nt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < C::KS; ++kk) {
            const int n2 = 2 * kk + (q >> 1);           // K index 4 kk + q = (n2, re / im = q & 1)
            const double a = n2 < N2 ? mid[fa][n2 * 16 + 2 * k1a + (q & 1)] : 0.0;
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) acc[nt] = mfma64(a, b2[kk * C::NT + nt], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = q + 4 * i, col = nt * 16 + cl, k2 = col >> 1;
                if (k2 < C::NK2 && 2 * tile + (row >> 3) < FT) outs[2 * tile + (row >> 3)][k2][2 * (row & 7) + (col & 1)] = (float)acc[nt][i];
            }
    }
    __syncthreads();
    // [16 frames][k2][8 k1] bins of 8 bytes: a (frame, k2) is 8 consecutive bins k = 8 grp + 32 k2 ..
    for (int idx = tid; idx < nf * C::NK2 * 8; idx += 256) {
        const int k1l = idx & 7, rest = idx >> 3, k2 = rest % C::NK2, fl = rest / C::NK2;
        const int fr = fr0 + fl, k = 8 * grp + k1l + 32 * k2;
        if (fr < g.M && k < C::F)
            *(float2*)(g.spec + g.seg.map(fr) * C::SPEC + 2 * k) = *(const float2*)&outs[fl][k2][2 * k1l];
    }
}

// host: the operand tables, doubles in MFMA B-fragment order
template <int N2>
static inline void dft64_tables(std::vector<double>& tw1, std::vector<double>& twm, std::vector<double>& tw2) {
    using C = Dft64Cfg<N2>;
    auto ang = [](long num, int den) { return 2.0 * M_PI * (double)(num % den) / den; };
    tw1.assign(C::TW1, 0.0); twm.assign(C::TWM, 0.0); tw2.assign(C::TW2, 0.0);
    for (int grp = 0; grp < 4; ++grp)
        for (int kk = 0; kk < 8; ++kk)
            for (int lane = 0; lane < 64; ++lane) {
                const int n1 = 4 * kk + (lane >> 4), col = lane & 15, k1 = 8 * grp + (col >> 1);
                const double a = ang((long)n1 * k1, 32);
                tw1[((size_t)grp * 8 + kk) * 64 + lane] = (col & 1) ? -std::sin(a) : std::cos(a);
            }
    for (int n2 = 0; n2 < N2; ++n2)
        for (int k1 = 0; k1 < 32; ++k1) {
            const double a = ang((long)n2 * k1, C::N);
            twm[((size_t)n2 * 32 + k1) * 2] = std::cos(a); twm[((size_t)n2 * 32 + k1) * 2 + 1] = std::sin(a);
        }
    for (int kk = 0; kk < C::KS; ++kk)
        for (int nt = 0; nt < C::NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = 4 * kk + (lane >> 4), n2 = k >> 1, cc = k & 1;       // input (n2, re / im)
                const int col = nt * 16 + (lane & 15), k2 = col >> 1, cp = col & 1; // output (k2, re / im)
                double v = 0.0;
                if (n2 < N2 && k2 < C::NK2) {
                    const double th = ang((long)n2 * k2, N2);                      // times e^{-i th}
                    v = (cc == cp) ? std::cos(th) : (cc ? std::sin(th) : -std::sin(th));
                }
                tw2[((size_t)kk * C::NT + nt) * 64 + lane] = v;
            }
}

static inline void launch_dft64_forward(hipStream_t st, const Dft64Args& a, int win) {
    if (a.M <= 0) return;
    if (a.M <= 256) {           // a few frames (streaming hops, tiny clips): four frames per workgroup
        const dim3 grid((a.M + 3) / 4, 4);
        if (win == 960) hipLaunchKernelGGL(HIP_KERNEL_NAME(dft64_fwd_kernel<30, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(dft64_fwd_kernel<10, 4>), grid, dim3(256), 0, st, a);
        return;
    }
    const dim3 grid((a.M + 15) / 16, 4);
    if (win == 960) hipLaunchKernelGGL(HIP_KERNEL_NAME(dft64_fwd_kernel<30, 16>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(dft64_fwd_kernel<10, 16>), grid, dim3(256), 0, st, a);
}
