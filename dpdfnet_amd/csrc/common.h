// common.h -- shared device helpers for the gfx950 DPDFNet kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 FMA chain.
// lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15];
// D reg i holds D[row = (l>>4)*4 + i][col = l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // tanh(x) = sign(x) * (1 - e) / (1 + e),  e = exp(-2|x|)  (no overflow, abs err ~1e-7)
    float e = fast_exp(-2.0f * fabsf(x));
    float t = (1.0f - e) * fast_rcp(1.0f + e);
    return copysignf(t, x);
}

// GRU cell tail (reference onnx_model/layers.py:1235-1259: n = tanh(x_n + r * h_n), h' = (1-z) n + z h) with the FMA
// contraction pinned, so that every row a lane holds -- and every kernel variant -- rounds identically (identical
// clips in different batch slots must come out bit-identical; left to the compiler, unrolled copies may differ).
__device__ __forceinline__ float gru_candidate(float r, float hn, float xn) { return tanh_f(__builtin_fmaf(r, hn, xn)); }
__device__ __forceinline__ float gru_blend(float z, float n, float h) { return __builtin_fmaf(z, h - n, n); }

// GRU(64) cell tail on PRE-SCALED pre-activations (build_gru64 folds -log2 e into the r/z rows and -2 log2 e into the
// candidate rows of W_ih, W_hh and the biases): r, z = 1/(1 + 2^a); n = tanh(.) = 2/(1 + 2^t) - 1 with t = xn + r hn;
// h' = n + z (h - n).  6 transcendentals + 7 plain VALU per hidden unit; exp2 -> inf / 0 saturates to -1 / +1 cleanly.
__device__ __forceinline__ float sigmoid_pre(float a) { return fast_rcp(1.0f + __builtin_amdgcn_exp2f(a)); }
__device__ __forceinline__ float gru64_cell(float ar, float az, float axn, float ahn, float h) {
    const float r = sigmoid_pre(ar), z = sigmoid_pre(az);
    const float n = __builtin_fmaf(2.0f, sigmoid_pre(__builtin_fmaf(r, ahn, axn)), -1.0f);
    return __builtin_fmaf(z, h - n, n);
}

// An agent-scope, single-dword load: for values another workgroup of a RUNNING launch (or a kernel of another stream that is still
// running) has written.  (Round 5 read the deep-filter taps of df_apply / mask_df through it as a guard against a corruption whose cause
// was not known; round 6 found the cause -- packed FP32 arithmetic, not memory: DESIGN.md section 6 -- and those kernels read plain again.)
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// K index supplied by lane-quad q (= lane>>4) for MFMA kb (0..3) of 16-wide K chunk c.
// All A fragments (4 contiguous floats per lane per chunk) and all packed B fragments use it.
__host__ __device__ inline int kperm(int c, int q, int kb) { return 16 * c + 4 * q + kb; }

// ReLU that keeps a NaN (torch.relu(NaN) = NaN: the reference's frame function carries a NaN that has entered a clip through every layer and
// state of that clip -- tests/golden/nonfinite_*.npz); fmaxf / v_max_f32 return the other operand.  v_cmp + v_cndmask instead of one v_max.
__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return relu_f(v);
    if (act == ACT_TANH) return tanh_f(v);
    if (act == ACT_SIGMOID) return sigmoid_f(v);
    return v;
}

// all-reduce (sum) inside each 16-lane DPP row: four row-rotate adds on the VALU, no LDS crossbar.
// (A 64-channel row of the row-contiguous tile layout is exactly one DPP row x 4 values per lane.)
__device__ __forceinline__ float row16_allreduce_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

#ifdef DPDF_PHASE_TRACE
// timing builds only (tools/glue_trace.py): s_memtime stamps of one workgroup per role of the streaming hop's DPRNN launches
__device__ unsigned long long dpdf_trace_buf[32];
#endif
