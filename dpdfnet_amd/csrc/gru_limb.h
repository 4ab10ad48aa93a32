// gru_limb.h -- the DEFAULT GRU-64 throughput kernels (dpdf_set_option "gru64_limbs" = 3; 0 selects the fp32-MFMA kernels of gru_scan.h, which the
// bench line times beside them): the GRU(64) scans with every fp32 product formed from THREE bf16 limbs per operand on the bf16 matrix pipe.
//
// The fp32 matrix rate of CDNA4 is 256 FLOP/cycle/CU (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD), the bf16 rate 4096
// (v_mfma_f32_16x16x32_bf16: 16 cycles for eight times the MACs).  The GRU-64 kernels of gru_scan.h sit at 77-81 % matrix-pipe busy
// -- the fp32 rate IS their bound.  But
//   * an fp32 value is EXACTLY hi + mid + lo with three bf16 values (8 + 8 + 8 significand bits; hi = rne(x), mid = rne(x - hi),
//     lo = x - hi - mid: the residues are exact fp32 subtractions and the last one fits bf16 exactly),
//   * the product of two bf16 values is exact in fp32, and the MFMA accumulates in fp32,
// so a . w = sum of nine limb products, of which lo.lo, lo.mid, mid.lo are < 2^-24 of the term and are dropped.  Six bf16 MFMAs per
// fp32 product term: 2.67 x the fp32 matrix rate at the accuracy of the fp32 kernels (tools/gru64_limb_bench.hip measures both against a
// float64 recurrence: the limb form is the CLOSER one, its products being exact where the fp32 MFMA rounds every partial sum).
//
// Layout (one 256-thread workgroup = 16 rows, wave w = hidden units [16w, 16w + 16), as in gru_scan.h), TRANSPOSED against the fp32
// kernels: D[unit][row] = W[unit][k] . act[row][k], i.e.
//   * A operand = WEIGHTS: lane (q, m) of k-chunk c holds W[unit 16w + m][32c + 8q .. + 7] -- 6 matrices (ih / hh x r, z, n) x 2 chunks x
//     3 limbs x 4 VGPRs = 144 VGPRs, resident for the whole scan;
//   * B operand = ACTIVATIONS: lane (q, n) reads 16 bytes act_limb[row n][32c + 8q .. + 7] from an LDS limb plane ([limb][row][72 bf16]:
//     144-byte rows, consecutive rows 4 banks apart -- conflict-free ds_read_b128);
//   * D: lane (q, n), register i = (unit 16w + 4q + i, row n): a lane ends up with FOUR CONSECUTIVE units of one row -- its h' goes to
//     HBM as one float4 and into the limb planes as three 8-byte stores (the fp32 kernels' C layout has four rows of one unit per lane:
//     twelve 2-byte stores).
// The split of h' and of the next x tile into limbs is ~22 VALU instructions per lane and side (v_cvt_pk_bf16_f32 is RNE).
// Reference: onnx_model/layers.py:159-196, 1235-1259.
#pragma once
#include "common.h"
#include "gru_scan.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int GRU64L_FRAG_PER_WAVE = 6 * 2 * 3;      // [mat = side * 3 + gate][k-chunk][limb] uint4 fragments per lane

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // low half = bf16(a), high half = bf16(b), round to nearest even
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// four fp32 values -> three limbs of four bf16 each (exact: v = hi + mid + lo)
__device__ __forceinline__ void split3(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk_bf16(v.x, v.y); hi.y = pk_bf16(v.z, v.w);
    float r0 = v.x - bf_lo(hi.x), r1 = v.y - bf_hi(hi.x), r2 = v.z - bf_lo(hi.y), r3 = v.w - bf_hi(hi.y);
    mid.x = pk_bf16(r0, r1); mid.y = pk_bf16(r2, r3);
    r0 -= bf_lo(mid.x); r1 -= bf_hi(mid.x); r2 -= bf_lo(mid.y); r3 -= bf_hi(mid.y);
    lo.x = pk_bf16(r0, r1); lo.y = pk_bf16(r2, r3);
}

#ifndef GRU64L_VAR
#define GRU64L_VAR 0      // tools/gru64_limb_bench.hip A/B switches
#endif
// the six limb pairs of a product, smallest first: (weight limb, activation limb)
#define GRU64L_TERMS(F) F(2, 0) F(0, 2) F(1, 1) F(1, 0) F(0, 1) F(0, 0)

// ---------------------------------------------------------------------------------------------------------------------------------
// gru64_l3_kernel<MODE>: the three throughput kernels of a DPRNN block (gru_scan.h: gru64_scan_kernel, gru64_epi_kernel<1>, <2>) on limbs.
//   MODE 0  intra-band FORWARD scan.  Its h' sequence is needed by nothing but fc_intra (the Linear over [hf | hb], reference
//           onnx_model/layers.py:178-181), so it leaves as pf(p) = W_fc[:, 0:64] hf(p) -- 64 floats per position like hf itself, and the
//           backward kernel's fc shrinks to its own half (one fc matrix per kernel: the 24 VGPRs that still fit beside the GRU's 144);
//   MODE 2  intra-band BACKWARD scan: y(p) = x(p) + LN(pf(p) + W_fc[:, 64:128] hb(p) + b);
//   MODE 1  inter-band scan (state carried): y(s) = x(s) + LN(W_fc h'(s) + b).
// The fc product of step s - 1 rides on the h operand of step s (the same limb registers) and passes through an LDS tile to the
// row-contiguous lanes, which store it (MODE 0; laid out like x) or, holding the residual x, normalise it first (MODES 1 / 2: 16 lanes x
// 4 values = one 64-channel row, statistics are two DPP butterflies) -- one 16-byte store per lane and step, whole 256-byte rows per 16
// lanes; output lags by two steps.
struct Gru64LArgs {
    Gru64Args g;            // MODE 0: g.out = pf; MODES 1, 2: g.out unused
    const uint4* wl;        // GRU limb fragments [dir][wave 4][mat 6][chunk 2][limb 3][lane 64]
    const uint4* fcl;       // fc limb fragments [wave 4][chunk 2][limb 3][lane 64] (the half / matrix this kernel multiplies by its own h')
    const float* fc_bias; const float* ln_g; const float* ln_b;   // [64] (MODES 1, 2)
    const float* extra;     // MODE 2: pf, addressed like x
    float* y;               // MODES 1, 2: output, addressed like x
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gru64_l3_kernel(Gru64LArgs ea) {
    const Gru64Args& a = ea.g;
    constexpr bool EPI = MODE != 0;
    __shared__ __attribute__((aligned(16))) unsigned short Hp[2][3][16][72];
    __shared__ __attribute__((aligned(16))) unsigned short Xp[2][3][16][72];
    __shared__ __attribute__((aligned(16))) float Xs[EPI ? 4 : 1][EPI ? 16 : 1][EPI ? 68 : 4];     // residual x, fp32: x(s - 2) must outlive x(s + 1)'s staging
    __shared__ __attribute__((aligned(16))) float Ys[2][16][68];
    __shared__ __attribute__((aligned(16))) float Lp[7][64];                                       // fc bias | ln gamma | ln beta | GRU biases r, z, in, hn
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = MODE == 2 ? 1 : 0;
    const int row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;

    uint4 wk[6][2][3], wf[2][3];
    {
        const uint4* wp = ea.wl + ((size_t)(dir * 4 + w) * GRU64L_FRAG_PER_WAVE) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int l = 0; l < 3; ++l) wk[mt][c][l] = wp[(size_t)((mt * 2 + c) * 3 + l) * 64];
        const uint4* fp = ea.fcl + ((size_t)w * 6) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int l = 0; l < 3; ++l) wf[c][l] = fp[(size_t)(c * 3 + l) * 64];
    }
    const int u0 = 16 * w + 4 * q;                       // this lane's four hidden units / fc output channels
    if (tid < 64) {
        if (EPI) { Lp[0][tid] = ea.fc_bias[tid]; Lp[1][tid] = ea.ln_g[tid]; Lp[2][tid] = ea.ln_b[tid]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) Lp[3 + k][tid] = a.bias[(size_t)dir * 256 + 64 * k + tid];
    }

    const int hi0 = row0 / a.rdiv, lo0 = row0 - hi0 * a.rdiv;
    const float* xbase = a.x + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo;
    const float* ebase = MODE == 2 ? ea.extra + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo : nullptr;
    float* ybase = EPI ? ea.y + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo : nullptr;
    float* obase = MODE == 0 ? a.out + (long)hi0 * a.x_hi + (long)lo0 * a.x_lo : nullptr;            // pf is laid out (and addressed) like x
    const int srow = 4 * w + q, scol = 4 * cl;
    unsigned sx_off; bool s_ok, o_ok;
    {
        int rs = row0 + srow; s_ok = rs < a.nrows; if (!s_ok) rs = a.nrows - 1;
        sx_off = (unsigned)((long)(rs / a.rdiv - hi0) * a.x_hi + (long)(rs % a.rdiv - lo0) * a.x_lo) + scol;
        int rc = row0 + cl; o_ok = rc < a.nrows; if (!o_ok) rc = a.nrows - 1;
    }
    float* hp = nullptr;
    float4 h_own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.hstate) {
        int rc = row0 + cl; if (rc >= a.nrows) rc = a.nrows - 1;
        hp = a.hstate + (long)(rc / a.rdiv) * a.h_hi + (long)(rc % a.rdiv) * a.h_lo + u0;
        h_own = make_float4(ld_agent(hp), ld_agent(hp + 1), ld_agent(hp + 2), ld_agent(hp + 3));      // (the state stride need not be a multiple of 16 bytes: no vector access)
    }
    const int n = a.nsteps;
    auto pos_of = [&](int s) { s = s < 0 ? 0 : (s < n ? s : n - 1); return dir ? n - 1 - s : s; };    // clamped: head and tail re-read a valid tile
    {
        uint2 l0, l1, l2;
        split3(h_own, l0, l1, l2);
        *(uint2*)&Hp[1][0][cl][u0] = l0; *(uint2*)&Hp[1][1][cl][u0] = l1; *(uint2*)&Hp[1][2][cl][u0] = l2;
        const float4 x0 = *(const float4*)((xbase + (long)pos_of(0) * a.x_step) + sx_off);
        split3(x0, l0, l1, l2);
        *(uint2*)&Xp[0][0][srow][scol] = l0; *(uint2*)&Xp[0][1][srow][scol] = l1; *(uint2*)&Xp[0][2][srow][scol] = l2;
        if (EPI) *(float4*)&Xs[0][srow][scol] = x0;
    }
    float4 e_prev = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    constexpr int LAG = 2;
    for (int s = 0; s < n + LAG; ++s) {
        const int buf = s & 1;
        // ---- finalize step s - 2: LayerNorm + residual on the row-contiguous pieces, one 16-byte store per lane
        if (EPI && s >= 2) {
            float4 yv = *(const float4*)&Ys[buf ^ 1][srow][scol];                 // fc(h'(s - 2)), written during step s - 1
            const float4 rv = *(const float4*)&Xs[(s - 2) & 3][srow][scol];       // residual x(s - 2)
            const float4 fb = *(const float4*)&Lp[0][scol];
            yv.x += fb.x; yv.y += fb.y; yv.z += fb.z; yv.w += fb.w;
            if (MODE == 2) { yv.x += e_prev.x; yv.y += e_prev.y; yv.z += e_prev.z; yv.w += e_prev.w; }
            const float mean = row16_allreduce_sum(yv.x + yv.y + yv.z + yv.w) * (1.0f / 64.0f);
            const float d0 = yv.x - mean, d1 = yv.y - mean, d2 = yv.z - mean, d3 = yv.w - mean;
            const float s2 = row16_allreduce_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
            const float inv = rsqrtf(s2 * (1.0f / 64.0f) + 1e-5f);
            const float4 gg = *(const float4*)&Lp[1][scol], bb = *(const float4*)&Lp[2][scol];
            float4 o;
            o.x = rv.x + d0 * inv * gg.x + bb.x; o.y = rv.y + d1 * inv * gg.y + bb.y;
            o.z = rv.z + d2 * inv * gg.z + bb.z; o.w = rv.w + d3 * inv * gg.w + bb.w;
            if (s_ok) *(float4*)((ybase + (long)pos_of(s - 2) * a.x_step) + sx_off) = o;
        }
        if (MODE == 0 && s >= 2) {        // pf(s - 2) out of the exchange tile: whole 256-byte rows per 16 lanes
            const float4 yv = *(const float4*)&Ys[buf ^ 1][srow][scol];
            if (s_ok) *(float4*)((obase + (long)pos_of(s - 2) * a.x_step) + sx_off) = yv;
        }
        // ---- loads for step s + 1 (and the pf tile of step s - 1: consumed by the finalize of the next iteration)
        const float4 xnext = *(const float4*)((xbase + (long)pos_of(s + 1) * a.x_step) + sx_off);
        float4 e_ld = xnext;
        if (MODE == 2) e_ld = *(const float4*)((ebase + (long)pos_of(s - 1) * a.x_step) + sx_off);
#if !(GRU64L_VAR & 1)
        __builtin_amdgcn_sched_barrier(0);
#endif
        uint4 xb[2][3], hb[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                hb[c][l] = *(const uint4*)&Hp[buf ^ 1][l][cl][32 * c + 8 * q];
                xb[c][l] = *(const uint4*)&Xp[buf][l][cl][32 * c + 8 * q];
            }
        const float4 b_r = *(const float4*)&Lp[3][u0], b_z = *(const float4*)&Lp[4][u0], b_in = *(const float4*)&Lp[5][u0], b_hn = *(const float4*)&Lp[6][u0];
        f32x4 ar = {b_r.x, b_r.y, b_r.z, b_r.w}, az = {b_z.x, b_z.y, b_z.z, b_z.w};
        f32x4 axn = {b_in.x, b_in.y, b_in.z, b_in.w}, ahn = {b_hn.x, b_hn.y, b_hn.z, b_hn.w};
        f32x4 ay = {0.f, 0.f, 0.f, 0.f};
#define GRU64L_STEP(WL, AL) \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) { \
            ar = mfma_bf16(wk[3][c][WL], hb[c][AL], ar); az = mfma_bf16(wk[4][c][WL], hb[c][AL], az); ahn = mfma_bf16(wk[5][c][WL], hb[c][AL], ahn); \
            ay = mfma_bf16(wf[c][WL], hb[c][AL], ay); \
            ar = mfma_bf16(wk[0][c][WL], xb[c][AL], ar); az = mfma_bf16(wk[1][c][WL], xb[c][AL], az); axn = mfma_bf16(wk[2][c][WL], xb[c][AL], axn); \
        }
        GRU64L_TERMS(GRU64L_STEP)
#undef GRU64L_STEP
        float4 hn;
        hn.x = gru64_cell(ar[0], az[0], axn[0], ahn[0], h_own.x); hn.y = gru64_cell(ar[1], az[1], axn[1], ahn[1], h_own.y);
        hn.z = gru64_cell(ar[2], az[2], axn[2], ahn[2], h_own.z); hn.w = gru64_cell(ar[3], az[3], axn[3], ahn[3], h_own.w);
        if (s < n) h_own = hn;                            // (iterations past the last step only drain the epilogue)
        uint2 l0, l1, l2;
        split3(hn, l0, l1, l2);
        *(uint2*)&Hp[buf][0][cl][u0] = l0; *(uint2*)&Hp[buf][1][cl][u0] = l1; *(uint2*)&Hp[buf][2][cl][u0] = l2;
        *(float4*)&Ys[buf][cl][u0] = make_float4(ay[0], ay[1], ay[2], ay[3]);         // fc(h'(s - 1)): read by the finalize / store of iteration s + 1
#if !(GRU64L_VAR & 2)
        __builtin_amdgcn_sched_barrier(0);
#endif
        split3(xnext, l0, l1, l2);
        *(uint2*)&Xp[buf ^ 1][0][srow][scol] = l0; *(uint2*)&Xp[buf ^ 1][1][srow][scol] = l1; *(uint2*)&Xp[buf ^ 1][2][srow][scol] = l2;
        if (EPI) *(float4*)&Xs[(s + 1) & 3][srow][scol] = xnext;
        e_prev = e_ld;
        __syncthreads();
    }
    if (hp && o_ok) {
        __hip_atomic_store(hp, h_own.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(hp + 1, h_own.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 2, h_own.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(hp + 3, h_own.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
