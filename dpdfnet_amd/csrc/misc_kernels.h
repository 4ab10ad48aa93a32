// misc_kernels.h -- the HBM-bound, non-GEMM stages of the frame function and the I/O glue:
// feature extraction + EMA norms, erb_conv0, mask-decoder tail, mask / deep-filter apply,
// overlap-add, and the import/export between the reference's flat state vector and the
// time-halo form the batched kernels use.
#pragma once
#include "common.h"
#include "gemm_rows.h"

// ---------------------------------------------------------------------------------------------
// features A (parallel over frames): raw spec -> scaled spec (x wnorm), ERB log-power (16 kHz) or
// per-bin log-magnitude (48 kHz).  Reference onnx_model/dpdfnet.py:831-834 (16k),
// onnx_model/dpdfnet_48khz_hr.py:903-906 (48k), export wrapper export_dpdfnet_to_onnx.py:22.
// One workgroup per frame.
struct FeatAArgs {
    const float* raw;      // [B][Tc][F][2] unnormalised (chunk view: frame stride given)
    size_t raw_clip_stride;  // floats between clips in raw
    float* xs;             // [B][2+Tc][F][2] scaled spec (halo 2)
    float* feat_erb;       // [B][2+Tc][E]  (dB values here; normalised in place by features B)
    const int* band_start; // [33] (16k) band edges, or null for 48k
    int B, Tc, F, E, is48; float wnorm;
};
__global__ __launch_bounds__(256) void feat_a_kernel(FeatAArgs a) {
    __shared__ float pw[512];
    const int bt = blockIdx.x;
    const int b = bt / a.Tc, t = bt - b * a.Tc;
    const float* src = a.raw + (size_t)b * a.raw_clip_stride + (size_t)t * a.F * 2;
    float* xd = a.xs + ((size_t)b * (a.Tc + 2) + 2 + t) * a.F * 2;
    float* fe = a.feat_erb + ((size_t)b * (a.Tc + 2) + 2 + t) * a.E;
    for (int f = threadIdx.x; f < a.F; f += 256) {
        float2 v = *(const float2*)(src + 2 * f);
        v.x *= a.wnorm; v.y *= a.wnorm;
        *(float2*)(xd + 2 * f) = v;
        float p = v.x * v.x + v.y * v.y;
        if (a.is48) fe[f] = 10.0f * log10f(sqrtf(p) + 1e-10f);
        else pw[f] = p;
    }
    if (!a.is48) {
        __syncthreads();
        const int e = threadIdx.x;
        if (e < a.E) {
            const int s = a.band_start[e], n = a.band_start[e + 1] - s;
            const float inv = 1.0f / (float)n;
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += pw[s + j] * inv;
            fe[e] = 10.0f * log10f(acc + 1e-10f);
        }
    }
}

// features B (scan over frames): exponential mean normalisation of the ERB / magnitude features
// and unit-normalisation of the complex DF bins.  Reference onnx_model/layers.py:498-506
// (ErbNorm), 637-661 (MagNorm48, fixed sigma 40), 561-572 (SpecNorm/SpecNorm48).
// Thread (b, j): j < E -> erb band j; E <= j < E + D -> DF bin j - E.  State read/written in the
// reference flat state vector.
struct FeatBArgs {
    float* feat_erb;       // [B][2+Tc][E] in place
    const float* xs;       // [B][2+Tc][F][2]
    float* feat_spec;      // [B][2+Tc][2][D]
    float* state; long S; int off_erb, off_spec;
    int B, Tc, F, E, D;
};
__global__ void feat_b_kernel(FeatBArgs a) {
    const int b = blockIdx.x;
    const int j = threadIdx.x;
    const float al = 0.98f, be = (float)(1.0 - 0.98);
    const int Tt = a.Tc + 2;
    // The EMA recurrences are one dependent FMA per frame; what costs time is the load in front of each one (an L2
    // round trip, and the compiler may not move it above the previous frame's store to the same tensor).  Frames
    // are therefore taken eight at a time: eight independent loads, then the recurrence, then eight stores.
    constexpr int U = 8;
    if (j < a.E) {
        float mu = a.state[b * a.S + a.off_erb + j];
        float* p = a.feat_erb + ((size_t)b * Tt + 2) * a.E + j;
        for (int t0 = 0; t0 < a.Tc; t0 += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = t0 + u < a.Tc ? p[(size_t)(t0 + u) * a.E] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < a.Tc) {
                    mu = al * mu + be * x[u];
                    p[(size_t)(t0 + u) * a.E] = (x[u] - mu) / 40.0f;
                }
            }
        }
        a.state[b * a.S + a.off_erb + j] = mu;
    } else if (j < a.E + a.D) {
        const int f = j - a.E;
        float s = a.state[b * a.S + a.off_spec + f];
        const float* xp = a.xs + ((size_t)b * Tt + 2) * a.F * 2 + 2 * f;
        float* o = a.feat_spec + ((size_t)b * Tt + 2) * 2 * a.D + f;
        for (int t0 = 0; t0 < a.Tc; t0 += U) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = t0 + u < a.Tc ? *(const float2*)(xp + (size_t)(t0 + u) * a.F * 2) : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < a.Tc) {
                    float mag = sqrtf(v[u].x * v[u].x + v[u].y * v[u].y);
                    s = al * s + be * mag;
                    float den = sqrtf(s + 1e-12f);
                    o[(size_t)(t0 + u) * 2 * a.D] = v[u].x / den;
                    o[(size_t)(t0 + u) * 2 * a.D + a.D] = v[u].y / den;
                }
            }
        }
        a.state[b * a.S + a.off_spec + f] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// feat_hop_kernel: features A + B for ONE frame per stream (single-hop streaming) as one launch, one workgroup per stream,
// with the two chores around the analysis STFT of a hop folded in: the sum over the K-split partial spectra (part != null:
// raw[f] = sum over z, in order, of part[row][z][2 f ..]) and the hand-over of the analysis buffer (in_tail <- the new hop,
// after the STFT has read the old one; snap_in keeps the pre-call copy for the recovery path).  Same arithmetic, in the same
// order, as feat_a_kernel + feat_b_kernel at Tc = 1.  A hop is a chain of dependent launches, each ~3 us of host time and
// ~5 us on the GPU's critical path: three fewer.
struct FeatHopArgs {
    FeatAArgs a; FeatBArgs b;
    const float* part; int ks, W;               // K-split STFT partials [row][ks][W] or null
    const float* pcm_new; float* in_tail; float* snap_in; int hop;   // null pcm_new: no hand-over
};
__global__ __launch_bounds__(256) void feat_hop_kernel(FeatHopArgs h) {
    __shared__ float pw[512];
    __shared__ float2 xd_s[128];                 // scaled spectrum of the D lowest bins (SpecNorm input)
    const FeatAArgs& a = h.a; const FeatBArgs& fb = h.b;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (h.pcm_new) {
        for (int i = tid; i < h.hop; i += 256) {
            const size_t o = (size_t)b * h.hop + i;
            if (h.snap_in) h.snap_in[o] = h.in_tail[o];
            h.in_tail[o] = h.pcm_new[o];
        }
    }
    const float* src = a.raw + (size_t)b * a.raw_clip_stride;
    float* xd = a.xs + ((size_t)b * 3 + 2) * a.F * 2;
    float* fe = a.feat_erb + ((size_t)b * 3 + 2) * a.E;
    for (int f = tid; f < a.F; f += 256) {
        float2 v;
        if (h.part) {
            const float* p = h.part + (size_t)b * h.ks * h.W + 2 * f;
            v = *(const float2*)p;
            for (int z = 1; z < h.ks; ++z) { const float2 q = *(const float2*)(p + (size_t)z * h.W); v.x += q.x; v.y += q.y; }
        } else v = *(const float2*)(src + 2 * f);
        v.x *= a.wnorm; v.y *= a.wnorm;
        *(float2*)(xd + 2 * f) = v;
        if (f < fb.D) xd_s[f] = v;
        const float p = v.x * v.x + v.y * v.y;
        pw[f] = a.is48 ? 10.0f * log10f(sqrtf(p) + 1e-10f) : p;
    }
    __syncthreads();
    const float al = 0.98f, be = (float)(1.0 - 0.98);
    for (int j = tid; j < a.E + fb.D; j += 256) {
        if (j < a.E) {
            float x;
            if (a.is48) x = pw[j];
            else {
                const int s = a.band_start[j], n = a.band_start[j + 1] - s;
                const float inv = 1.0f / (float)n;
                float acc = 0.f;
                for (int k = 0; k < n; ++k) acc += pw[s + k] * inv;
                x = 10.0f * log10f(acc + 1e-10f);
            }
            float mu = fb.state[b * fb.S + fb.off_erb + j];
            mu = al * mu + be * x;
            fe[j] = (x - mu) / 40.0f;
            fb.state[b * fb.S + fb.off_erb + j] = mu;
        } else {
            const int f = j - a.E;
            float s = fb.state[b * fb.S + fb.off_spec + f];
            const float2 v = xd_s[f];
            const float mag = sqrtf(v.x * v.x + v.y * v.y);
            s = al * s + be * mag;
            const float den = sqrtf(s + 1e-12f);
            float* o = fb.feat_spec + ((size_t)b * 3 + 2) * 2 * fb.D + f;
            o[0] = v.x / den; o[fb.D] = v.y / den;
            fb.state[b * fb.S + fb.off_spec + f] = s;
        }
    }
}
// the hand-over of the analysis buffers alone, for calls of several hops (after the STFT has read the old tails)
__global__ void stream_tail_update_kernel(const float* pcm_new, float* in_tail, float* snap_in, int n_hops, int hop) {
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < hop; i += blockDim.x) {
        const size_t o = (size_t)s * hop + i;
        if (snap_in) snap_in[o] = in_tail[o];
        in_tail[o] = pcm_new[(size_t)s * n_hops * hop + (size_t)(n_hops - 1) * hop + i];
    }
}

// ---------------------------------------------------------------------------------------------
// erb_conv0: dense Conv2d 1->64 k(3,3) over (3 frames, bands) + folded BN + ReLU
// (reference onnx_model/dpdfnet.py:74-81, 206-211; 48 kHz runs on bins [0,480) hr.py:263).
struct Conv0ErbArgs {
    const float* feat;     // [B][2+Tc][E]
    float* e0;             // [B][Tc][Ec][64]
    const float* w;        // [64][9] (BN folded)
    const float* bias;     // [64]
    int B, Tc, E, Ec;
};
__global__ __launch_bounds__(256) void conv0_erb_kernel(Conv0ErbArgs a) {
    // thread = 4 output channels (their 36 taps + bias in registers) x a strided set of rows
    const int c4 = (threadIdx.x & 15) * 4;
    float w[4][9]; 
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) w[j][k] = a.w[(c4 + j) * 9 + k];
    const float4 bias = *(const float4*)(a.bias + c4);
    const size_t rows = (size_t)a.B * a.Tc * a.Ec;
    for (size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < rows; row += (size_t)gridDim.x * 16) {
        const int f = (int)(row % a.Ec);
        const size_t bt = row / a.Ec;
        const int b = (int)(bt / a.Tc), t = (int)(bt - (size_t)b * a.Tc);
        float4 acc = bias;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const float* src = a.feat + ((size_t)b * (a.Tc + 2) + t + kt) * a.E;
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                int fi = f + kf - 1;
                float x = (fi >= 0 && fi < a.Ec) ? src[fi] : 0.f;
                acc.x += w[0][kt * 3 + kf] * x; acc.y += w[1][kt * 3 + kf] * x;
                acc.z += w[2][kt * 3 + kf] * x; acc.w += w[3][kt * 3 + kf] * x;
            }
        }
        acc.x = relu_f(acc.x); acc.y = relu_f(acc.y); acc.z = relu_f(acc.z); acc.w = relu_f(acc.w);
        *(float4*)(a.e0 + row * 64 + c4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// mask-decoder tail: u = relu(ps*e0 + pb) + d1 ; m = sigmoid(bias + sum_{c,k} w[c][k] u[f+k-1][c])
// (conv0p pathway + conv0_out dense 64->1 k(1,3) + BN + Sigmoid, reference
// onnx_model/dpdfnet.py:320-323, 364).  One wavefront per output band, lane = channel.
struct MaskOutArgs {
    const float* e0; const float* d1;  // [B][Tc][Ec][64]
    float* m;                          // [B][Tc][Em]
    const float* ps; const float* pb;  // [64]
    const float* w;                    // [64][3] (BN folded)
    float bias;
    int rows;                          // B*Tc*Ec
    int Ec, Em, is48;
};
__global__ __launch_bounds__(256) void mask_out_kernel(MaskOutArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)a.rows) return;
    const int f = (int)(row % a.Ec);
    const size_t bt = row / a.Ec;
    const float ps = a.ps[lane], pb = a.pb[lane];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int fi = f + k - 1;
        if (fi >= 0 && fi < a.Ec) {
            size_t o = (bt * a.Ec + fi) * 64 + lane;
            float u = relu_f(ps * a.e0[o] + pb) + a.d1[o];
            acc += a.w[lane * 3 + k] * u;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        float mv = sigmoid_f(acc + a.bias);
        a.m[bt * a.Em + f] = mv;
        if (a.is48 && f == a.Ec - 2) a.m[bt * a.Em + a.Ec] = mv;   // F.pad reflect (0,1): m[480] = m[478]
    }
}

// mask head, second half (first half: MaskSumEpi in gemm_rows.h): m[f] = sigmoid(bias + s_0[f-1] + s_1[f] + s_2[f+1])
// = conv0_out k(1,3), zero pad 1, + BN + Sigmoid (reference onnx_model/dpdfnet.py:320-323).  One thread per band.
struct MaskFinArgs {
    const float* s;        // [rows][4]
    float* m;              // [B*Tc][Em]
    float bias;
    int rows, Ec, Em, is48;
};
__global__ __launch_bounds__(256) void mask_fin_kernel(MaskFinArgs a) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (size_t)a.rows) return;
    const int f = (int)(row % a.Ec);
    const size_t bt = row / a.Ec;
    float acc = a.s[row * 4 + 1];
    if (f > 0) acc += a.s[(row - 1) * 4 + 0];
    if (f + 1 < a.Ec) acc += a.s[(row + 1) * 4 + 2];
    const float mv = sigmoid_f(acc + a.bias);
    a.m[bt * a.Em + f] = mv;
    if (a.is48 && f == a.Ec - 2) a.m[bt * a.Em + a.Ec] = mv;   // F.pad reflect (0,1): m[480] = m[478]
}

// ---------------------------------------------------------------------------------------------
// Mask.forward / MagnitudeMask.forward: masked spec of the frame from two steps ago
// (reference onnx_model/layers.py:414-445, dpdfnet_48khz_hr.py:55-69).
struct MaskApplyArgs {
    const float* xs;       // [B][2+Tc][F][2]
    const float* m;        // [B][Tc][Em]
    float* xm;             // [B][4+Tc][F][2]
    const int* band_of;    // [F] (16k) or null
    int B, Tc, F, Em;
};
__global__ void mask_apply_kernel(MaskApplyArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.B * a.Tc * a.F;
    if (idx >= total) return;
    const int f = (int)(idx % a.F);
    const size_t bt = idx / a.F;
    const int b = (int)(bt / a.Tc), t = (int)(bt - (size_t)b * a.Tc);
    float g = a.m[bt * a.Em + (a.band_of ? a.band_of[f] : f)];
    float2 v = *(const float2*)(a.xs + (((size_t)b * (a.Tc + 2) + t) * a.F + f) * 2);   // frame t-2 (+halo 2)
    v.x *= g; v.y *= g;
    *(float2*)(a.xm + (((size_t)b * (a.Tc + 4) + 4 + t) * a.F + f) * 2) = v;
}

// DF.forward + df_real (reference onnx_model/multiframe.py:200-232, 140-154), x 1/wnorm
// (export wrapper) and, for the offline path, apply_attn_limit (package/src/dpdfnet/audio.py:41-76).
struct DfApplyArgs {
    const float* xm;       // [B][4+Tc][F][2]
    const float* coefs;    // [B][2+Tc][D][10]
    float* out;            // [B][out_T][F][2] written at frame out_t0 + t
    size_t out_clip_stride; int out_t0;
    const float* raw;      // attn-limit noisy reference (same geometry as out) or null
    float alpha, beta;
    int B, Tc, F, D; float inv_wnorm;
#ifdef DPDF_HAZARD_PROBE
    unsigned* dump; int dump_T;   // [B][dump_T][F][24]: what this thread consumed and produced (bit patterns), XCC id, s_memtime low word
#endif
};
#ifdef DPDF_HAZARD_PROBE
#include "hazard_probe.h"     // df_apply_probe_kernel<...>: tools/hazard_probe.py, DESIGN.md section 6
#endif
__global__ void df_apply_kernel(DfApplyArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.B * a.Tc * a.F;
    if (idx >= total) return;
    const int f = (int)(idx % a.F);
    const size_t bt = idx / a.F;
    const int b = (int)(bt / a.Tc), t = (int)(bt - (size_t)b * a.Tc);
    const float* xb = a.xm + (((size_t)b * (a.Tc + 4) + t) * a.F + f) * 2;     // frame t-4
    const size_t fs = (size_t)a.F * 2;
    float re, im;
    if (f < a.D) {
        const float* c = a.coefs + (((size_t)b * (a.Tc + 2) + t) * a.D + f) * 10;   // frame t-2 (+halo 2)
        float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            float2 s = *(const float2*)(xb + n * fs);
            const float cr = c[2 * n], ci = c[2 * n + 1];
            rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
        }
        re = rr - ii; im = ri + ir;
    } else {
        float2 s = *(const float2*)(xb + 2 * fs);
        re = s.x; im = s.y;
    }
    re *= a.inv_wnorm; im *= a.inv_wnorm;
    const int tg = a.out_t0 + t;
    float* o = a.out + (size_t)b * a.out_clip_stride + ((size_t)tg * a.F + f) * 2;
    if (a.raw) {
        float nr = 0.f, ni = 0.f;
        if (tg >= 4) {
            const float* r = a.raw + (size_t)b * a.out_clip_stride + ((size_t)(tg - 4) * a.F + f) * 2;
            nr = r[0]; ni = r[1];
        }
        re = a.alpha * nr + a.beta * re;
        im = a.alpha * ni + a.beta * im;
    }
    o[0] = re; o[1] = im;
}

// mask_apply_kernel + df_apply_kernel as ONE launch for small launches (streaming hops, single clips: a dependent launch is
// ~3 us of host time and ~5 us on the GPU's critical path, the arithmetic is nothing): thread (b, t, f) masks its own frame,
// stores it (the masked-spectrum FIFO is exported from xm) and takes the five deep-filter taps from the imported history
// (frames before the chunk) or by masking the chunk's frames again itself -- same products, same order as the two kernels.
// ssum != null (48 kHz, small launches): the mask itself is finished here as well (mask_fin_kernel's sum of the three taps +
// sigmoid, reference dpdfnet_48khz_hr.py:428 incl. the reflect-padded last bin) and written to mk.m for the frame's own thread.
struct MaskDfArgs { MaskApplyArgs mk; DfApplyArgs df; const float* ssum; float bias0; int Ec; };
__global__ void mask_df_kernel(MaskDfArgs a) {
    const MaskApplyArgs& k = a.mk; const DfApplyArgs& d = a.df;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)k.B * k.Tc * k.F;
    if (idx >= total) return;
    const int f = (int)(idx % k.F);
    const size_t bt = idx / k.F;
    const int b = (int)(bt / k.Tc), t = (int)(bt - (size_t)b * k.Tc);
    const int band = k.band_of ? k.band_of[f] : f;
    auto gain = [&](int tt) {
        if (!a.ssum) return k.m[((size_t)b * k.Tc + tt) * k.Em + band];
        const int fe = f == a.Ec ? a.Ec - 2 : f;                         // F.pad reflect (0, 1): m[Ec] = m[Ec - 2]
        const size_t row = ((size_t)b * k.Tc + tt) * a.Ec + fe;
        float acc = a.ssum[row * 4 + 1];
        if (fe > 0) acc += a.ssum[(row - 1) * 4 + 0];
        if (fe + 1 < a.Ec) acc += a.ssum[(row + 1) * 4 + 2];
        return sigmoid_f(acc + a.bias0);
    };
    auto masked = [&](int tt) {                 // masked frame tt of the chunk (what mask_apply_kernel stores at xm index 4 + tt)
        const float g = gain(tt);
        if (a.ssum && tt == t) const_cast<float*>(k.m)[((size_t)b * k.Tc + tt) * k.Em + band] = g;
        float2 v = *(const float2*)(k.xs + (((size_t)b * (k.Tc + 2) + tt) * k.F + f) * 2);
        v.x *= g; v.y *= g;
        return v;
    };
    const float2 own = masked(t);
    *(float2*)(k.xm + (((size_t)b * (k.Tc + 4) + 4 + t) * k.F + f) * 2) = own;
    auto tap = [&](int n) {                     // xm index t + n
        const int j = t + n;
        if (n == 4) return own;
        if (j < 4) return *(const float2*)(d.xm + (((size_t)b * (d.Tc + 4) + j) * d.F + f) * 2);
        return masked(j - 4);
    };
    float re, im;
    if (f < d.D) {
        const float* c = d.coefs + (((size_t)b * (d.Tc + 2) + t) * d.D + f) * 10;
        float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const float2 s = tap(n);
            const float cr = c[2 * n], ci = c[2 * n + 1];
            rr += s.x * cr; ii += s.y * ci; ri += s.x * ci; ir += s.y * cr;
        }
        re = rr - ii; im = ri + ir;
    } else {
        const float2 s = tap(2);
        re = s.x; im = s.y;
    }
    re *= d.inv_wnorm; im *= d.inv_wnorm;
    const int tg = d.out_t0 + t;
    float* o = d.out + (size_t)b * d.out_clip_stride + ((size_t)tg * d.F + f) * 2;
    if (d.raw) {
        float nr = 0.f, ni = 0.f;
        if (tg >= 4) {
            const float* r = d.raw + (size_t)b * d.out_clip_stride + ((size_t)(tg - 4) * d.F + f) * 2;
            nr = r[0]; ni = r[1];
        }
        re = d.alpha * nr + d.beta * re;
        im = d.alpha * ni + d.beta * im;
    }
    o[0] = re; o[1] = im;
}


// ---------------------------------------------------------------------------------------------
// overlap-add + window-sum-square normalisation + centre trim + 2*win alignment shift + fit to N
// (reference package/src/dpdfnet/audio.py:120-136, 30-38)
struct OlaArgs {
    const float* frames;   // [B][T][win] windowed synthesis frames
    const float* window;
    float* out;            // [B][N]
    int B, T, N, win, hop;
    const int* lens;       // ragged batch (else nullptr): clip b is lens[b] samples / T_b = 1 + (lens[b] + win) / hop frames long;
                           // its own 2*win shift, zero tail and fit_length; out[b][lens[b]:N] = 0
    int n0 = 0, nw = 0;    // nw > 0: only the samples [n0, n0 + nw) of every clip (time slice of a pipelined host call: they need
                           // frames < (n0 + nw) / hop + 5 only)
};
__global__ void ola_kernel(OlaArgs a) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int span = a.nw > 0 ? a.nw : a.N;
    if (idx >= (size_t)a.B * span) return;
    const int b = (int)(idx / span), i = a.n0 + (int)(idx - (size_t)b * span);
    idx = (size_t)b * a.N + i;
    const int nb = a.lens ? a.lens[b] : a.N;
    const int Tb = a.lens ? 1 + (nb + a.win) / a.hop : a.T;
    const long ylen = (long)a.hop * (Tb - 1);
    const long src = (long)i + 2L * a.win;
    float v = 0.f;
    if (i < nb && src < ylen) {
        const long p = src + a.win / 2;
        const int t1 = (int)(p / a.hop);
        const int t0 = t1 - 1;
        float y = 0.f, wss = 0.f;
        if (t0 >= 0) {
            int k = (int)(p - (long)t0 * a.hop);
            if (k < a.win) { y += a.frames[((size_t)b * a.T + t0) * a.win + k]; wss += a.window[k] * a.window[k]; }
        }
        if (t1 < Tb) {
            int k = (int)(p - (long)t1 * a.hop);
            y += a.frames[((size_t)b * a.T + t1) * a.win + k]; wss += a.window[k] * a.window[k];
        }
        v = wss > 1.17549435e-38f ? y / wss : y;
    }
    a.out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// flat reference state <-> time-halo tensors (CyclicBuffer semantics, reference
// onnx_model/layers.py:68-107: each FIFO holds the last `cap` frames, oldest first).
// import: halo frame (-cap+1+j) <- buf[1+j];  export: buf[j] <- frame (Tc-cap+j).
struct StateIoArgs {
    float* state; long S;
    float* feat_erb; float* feat_spec; float* c0; float* xs; float* coefs; float* xm;
    int off_erb_buf, off_df_buf, off_convp, off_mask, off_coefs, off_spec;
    int B, Tc, E, D, F;
    int do_export;
    int seg_lo, seg_hi;    // this launch handles segments [seg_lo, seg_hi): 0 erb_conv0, 1 df_conv0, 2 mask spec, 3 df_convp | 4 coefs, 5 masked spec
    // import only, a streaming call's prologue (streams_enqueue): one more grid.y row behind the snapshot rows stages [analysis tail | new samples]
    // of every stream for the STFT and hands the tails over (stream_stage_in_kernel's work: one dependent launch less in front of the STFT)
    const float* si_pcm; float* si_tail; float* si_xbuf; float* si_snap; int si_hops, si_hop;
    float* snap; int snap_y;   // import only: grid.y rows behind the segments copy the whole state to `snap` (the streaming calls' pre-call copy,
                               // taken by the first launch that touches the state instead of by a copy on another stream: a cross-stream wait
                               // on the hop's chain costs ~10 us)
};
__device__ __forceinline__ void fifo_io(float* st, float* tensor_frame0 /* frame t=0 of clip */, long frame_sz,
                                        int cap, int Tc, int do_export, int tid, int nthreads, int j) {
    // plain layout: state frame j <-> tensor frame (export: Tc-cap+j, import: j-cap, j>=1); one FIFO frame per blockIdx.z
    if (j >= (do_export ? 0 : 1) && j < cap) {
        long tf = do_export ? (long)Tc - cap + j : (long)j - cap;
        float* tp = tensor_frame0 + tf * frame_sz;
        float* sp = st + (long)j * frame_sz;
        for (long i = tid; i < frame_sz; i += nthreads) {
            if (do_export) sp[i] = tp[i]; else tp[i] = sp[i];
        }
    }
}
__global__ __launch_bounds__(256) void state_io_kernel(StateIoArgs a) {
    const int b = blockIdx.x, seg = a.seg_lo + blockIdx.y, tid = threadIdx.x;
    const int jz = blockIdx.z;         // FIFO frame handled by this workgroup (grid.z = 5 = deepest FIFO)
    float* st = a.state + (long)b * a.S;
    if (seg >= a.seg_hi + a.snap_y) {
        if (a.si_xbuf && seg == a.seg_hi + a.snap_y && jz == 0) {
            const int hop = a.si_hop, n = a.si_hops * hop;
            float* xb = a.si_xbuf + (size_t)b * (n + hop);
            for (int i = tid; i < hop; i += 256) {
                const float v = a.si_tail[(size_t)b * hop + i];
                xb[i] = v;
                if (a.si_snap) a.si_snap[(size_t)b * hop + i] = v;
            }
            for (int i = tid; i < n; i += 256) xb[hop + i] = a.si_pcm[(size_t)b * n + i];
            __syncthreads();
            for (int i = tid; i < hop; i += 256) a.si_tail[(size_t)b * hop + i] = xb[n + i];
        }
        return;
    }
    if (seg >= a.seg_hi) {
        if (a.snap) {
            const int bid = (seg - a.seg_hi) * gridDim.z + jz, nb = a.snap_y * gridDim.z;
            float* dp = a.snap + (long)b * a.S;
            if ((a.S & 3) == 0) for (long i = (long)bid * 256 + tid; i < a.S / 4; i += (long)nb * 256) ((float4*)dp)[i] = ((const float4*)st)[i];
            else for (long i = (long)bid * 256 + tid; i < a.S; i += (long)nb * 256) dp[i] = st[i];
        }
        return;
    }
    const int Tc = a.Tc;
    if (seg == 0) {
        fifo_io(st + a.off_erb_buf, a.feat_erb + ((size_t)b * (Tc + 2) + 2) * a.E, a.E, 3, Tc, a.do_export, tid, 256, jz);
    } else if (seg == 1) {
        fifo_io(st + a.off_df_buf, a.feat_spec + ((size_t)b * (Tc + 2) + 2) * 2 * a.D, 2 * a.D, 3, Tc, a.do_export, tid, 256, jz);
    } else if (seg == 2) {
        fifo_io(st + a.off_mask, a.xs + ((size_t)b * (Tc + 2) + 2) * a.F * 2, a.F * 2, 3, Tc, a.do_export, tid, 256, jz);
    } else if (seg == 5) {
        fifo_io(st + a.off_spec, a.xm + ((size_t)b * (Tc + 4) + 4) * a.F * 2, a.F * 2, 5, Tc, a.do_export, tid, 256, jz);
    } else if (seg == 3) {
        // df_convp_buf [5][64][D] (channel-first) <-> c0 [B][4+Tc][D][64]
        const long fsz = 64L * a.D;
        float* t0 = a.c0 + ((size_t)b * (Tc + 4) + 4) * fsz;
        const int j = jz;
        if (j >= (a.do_export ? 0 : 1) && j < 5) {
            // a [64][D] <-> [D][64] transpose per frame: through LDS, so that BOTH sides move whole cache lines (the
            // direct form touched 64 lines per wave on the tensor side: 29 us per launch of the headline workload)
            __shared__ float Tt[64][97];                                    // D = 96 (+1: conflict-free columns)
            long tf = a.do_export ? (long)Tc - 5 + j : (long)j - 5;
            float* tp = t0 + tf * fsz;
            float* sp = st + a.off_convp + (long)j * fsz;
            if (a.do_export) {
                for (int i = tid; i < (int)fsz; i += 256) Tt[i & 63][i >> 6] = tp[i];          // tensor index = f*64 + c
                __syncthreads();
                for (int i = tid; i < (int)fsz; i += 256) { int c = i / a.D; sp[i] = Tt[c][i - c * a.D]; }   // state index = c*D + f
            } else {
                for (int i = tid; i < (int)fsz; i += 256) { int c = i / a.D; Tt[c][i - c * a.D] = sp[i]; }
                __syncthreads();
                for (int i = tid; i < (int)fsz; i += 256) tp[i] = Tt[i & 63][i >> 6];
            }
        }
    } else if (seg == 4) {
        // coefs_buf [3][5][D][2] <-> coefs [B][2+Tc][D][10]
        const long fsz = 10L * a.D;
        float* t0 = a.coefs + ((size_t)b * (Tc + 2) + 2) * fsz;
        const int j = jz;
        if (j >= (a.do_export ? 0 : 1) && j < 3) {
            long tf = a.do_export ? (long)Tc - 3 + j : (long)j - 3;
            float* tp = t0 + tf * fsz;
            float* sp = st + a.off_coefs + (long)j * fsz;
            for (long i = tid; i < fsz; i += 256) {
                int n = (int)(i / (2 * a.D)); int rem = (int)(i - (long)n * 2 * a.D);
                int f = rem >> 1, p = rem & 1;                               // state index = (n*D + f)*2 + p
                if (a.do_export) sp[i] = tp[(long)f * 10 + 2 * n + p]; else tp[(long)f * 10 + 2 * n + p] = sp[i];
            }
        }
    }
}

__global__ void fill_state_kernel(float* state, const float* init, long S, int B) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)S * B) return;
    state[idx] = init[idx % S];
}

// Streams that do not take part in a call keep their state: the active ones are packed into a dense batch before the call
// and unpacked after it (dpdf_streams_process_masked).  k-th active stream = idx[k]; one workgroup row per (k, slice).
//   pack:   state / in_tail / ola_tail of stream idx[k] -> slot k of the c* buffers, PCM row idx[k] of src -> row k of pcm_c
//   unpack: the reverse for the state and the tails, row k of pcm_c -> row idx[k] of dst; the device error flag -> host_err
struct StreamPackArgs {
    float* state; float* in_tail; float* ola_tail;          // [S][.]
    float* cstate; float* cin; float* cola;                 // [n_act][.]
    const float* pcm_src; float* pcm_c;                     // pack: src [S][npcm] -> pcm_c [n_act][npcm]
    const float* pcm_c_out; float* pcm_dst;                 // unpack: [n_act][npcm] -> dst [S][npcm]
    const int* idx; long S_state; int hop, npcm;
    const int* dev_err; int* host_err;
};
template <bool UNPACK>
__global__ void stream_pack_kernel(StreamPackArgs a) {
    const int k = blockIdx.x, st = a.idx[k];
    const int t0 = blockIdx.y * blockDim.x + threadIdx.x, tn = gridDim.y * blockDim.x;
    float* full = a.state + (size_t)st * a.S_state; float* comp = a.cstate + (size_t)k * a.S_state;
    if (UNPACK) { for (long i = t0; i < a.S_state; i += tn) full[i] = comp[i]; }
    else { for (long i = t0; i < a.S_state; i += tn) comp[i] = full[i]; }
    for (int i = t0; i < a.hop; i += tn) {
        if (UNPACK) { a.in_tail[(size_t)st * a.hop + i] = a.cin[(size_t)k * a.hop + i]; a.ola_tail[(size_t)st * a.hop + i] = a.cola[(size_t)k * a.hop + i]; }
        else { a.cin[(size_t)k * a.hop + i] = a.in_tail[(size_t)st * a.hop + i]; a.cola[(size_t)k * a.hop + i] = a.ola_tail[(size_t)st * a.hop + i]; }
    }
    for (int i = t0; i < a.npcm; i += tn) {
        if (UNPACK) a.pcm_dst[(size_t)st * a.npcm + i] = a.pcm_c_out[(size_t)k * a.npcm + i];
        else a.pcm_c[(size_t)k * a.npcm + i] = a.pcm_src[(size_t)st * a.npcm + i];
    }
    if (UNPACK && a.host_err && k == 0 && t0 == 0) *a.host_err = *a.dev_err;
}
// progress of an offline call, written straight into pinned host memory at the end of every chunk's stage 2
__global__ void progress_kernel(int* host_word, int frames_done) { *host_word = frames_done; }

// dst <- src for n floats (snapshot / restore of the streaming state around a call: dpdf_streams_process)
__global__ void copy_f4_kernel(float* dst, const float* src, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) *(float4*)(dst + i) = *(const float4*)(src + i);
    else for (size_t j = i; j < n; ++j) dst[j] = src[j];
}

// y += x (n multiple of 4)
__global__ void axpy_kernel(float* y, const float* x, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = *(float4*)(y + i), b = *(const float4*)(x + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *(float4*)(y + i) = a;
}

// streaming glue (reference package/src/dpdfnet/stream.py:116-156): causal analysis buffer and
// overlap-add with carried tails, hop = win/2.
//   xbuf[s] = in_tail[s] (hop samples) ++ pcm_in[s] (n_hops*hop samples); new in_tail = last hop samples
// snap_in (may be null): pre-call copy of the analysis tails, taken here because this kernel is the one that overwrites them
__global__ void stream_stage_in_kernel(const float* pcm_in, float* in_tail, float* xbuf, int S, int n_hops, int hop, float* snap_in) {
    const int s = blockIdx.x;
    const int n = n_hops * hop;
    float* xb = xbuf + (size_t)s * (n + hop);
    for (int i = threadIdx.x; i < hop; i += blockDim.x) {
        const float v = in_tail[(size_t)s * hop + i];
        xb[i] = v;
        if (snap_in) snap_in[(size_t)s * hop + i] = v;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) xb[hop + i] = pcm_in[(size_t)s * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < hop; i += blockDim.x) in_tail[(size_t)s * hop + i] = xb[n + i];
}
//   out hop j = (j ? frames[j-1][hop:] : ola_tail) + frames[j][:hop];  new ola_tail = frames[T-1][hop:]
// The same on K-split partial frames: frame value (row, c) = window[c] * sum over z (in order) of part[row][z][c]
// (the streaming iSTFT of a few frames runs split over K, gemm_rows.h; this kernel is its summing half).
// dev_err / host_err: the last kernel of a streaming call mirrors the device error flag into pinned host memory (null: no mirror)
__global__ void stream_ola_ksplit_kernel(const float* part, int ks, int W, const float* window, float* ola_tail, float* pcm_out, int S, int n_hops, int hop,
                                         const int* dev_err, int* host_err, float* snap_ola) {
    const int s = blockIdx.x;
    if (host_err && s == 0 && threadIdx.x == 0) *host_err = *dev_err;
    if (snap_ola) { for (int i = threadIdx.x; i < hop; i += blockDim.x) snap_ola[(size_t)s * hop + i] = ola_tail[(size_t)s * hop + i]; }
    const int win = 2 * hop;
    auto fr = [&](int j, int c) {
        const float* p = part + ((size_t)(s * n_hops + j) * ks) * W + c;
        float v = p[0];
        for (int z = 1; z < ks; ++z) v += p[(size_t)z * W];
        return v * window[c];
    };
    for (int idx = threadIdx.x; idx < n_hops * hop; idx += blockDim.x) {
        int j = idx / hop, i = idx - j * hop;
        float prev = j ? fr(j - 1, hop + i) : ola_tail[(size_t)s * hop + i];
        pcm_out[(size_t)s * n_hops * hop + idx] = prev + fr(j, i);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hop; i += blockDim.x) ola_tail[(size_t)s * hop + i] = fr(n_hops - 1, hop + i);
}

__global__ void stream_ola_kernel(const float* frames, float* ola_tail, float* pcm_out, int S, int n_hops, int hop, const int* dev_err, int* host_err, float* snap_ola) {
    const int s = blockIdx.x;
    if (host_err && s == 0 && threadIdx.x == 0) *host_err = *dev_err;
    if (snap_ola) { for (int i = threadIdx.x; i < hop; i += blockDim.x) snap_ola[(size_t)s * hop + i] = ola_tail[(size_t)s * hop + i]; }
    const int win = 2 * hop;
    const float* fr = frames + (size_t)s * n_hops * win;
    for (int idx = threadIdx.x; idx < n_hops * hop; idx += blockDim.x) {
        int j = idx / hop, i = idx - j * hop;
        float prev = j ? fr[(size_t)(j - 1) * win + hop + i] : ola_tail[(size_t)s * hop + i];
        pcm_out[(size_t)s * n_hops * hop + idx] = prev + fr[(size_t)j * win + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hop; i += blockDim.x) ola_tail[(size_t)s * hop + i] = fr[(size_t)(n_hops - 1) * win + hop + i];
}
