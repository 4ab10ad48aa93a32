// Research probe for a later round (not part of the product): does the VALU overlap with BF16 MFMAs on gfx950, and how
// fast is v_mfma_f32_16x16x32_bf16 against the fp32 one?  (fp32 MFMA and VALU do NOT overlap: mfma_valu_overlap.hip.)
// If they do overlap, an fp32 GRU step emulated with 3 bf16 limbs per operand (6 limb products, fp32 accumulate)
// would be bounded by max(6/16 * S, V + split cost) instead of S + V.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>   // bit 0: bf16 MFMA waves (0-3), bit 1: FMA waves (4-7), bit 2: transcendental waves (4-7)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int w = threadIdx.x >> 6;
    float x = threadIdx.x * 1e-3f + 1.0f, y = blockIdx.x * 1e-6f + 0.5f;
    if (w < 4) {
        if (!(MODE & 1)) return;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(x + i); b[i] = (__bf16)(y - i); }
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(MODE & 6)) return;
        float v0 = x, v1 = y, v2 = x + y, v3 = x - y;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE & 2) {
                    v0 = __builtin_fmaf(v0, 0.999f, 0.001f); v1 = __builtin_fmaf(v1, 0.999f, 0.001f);
                    v2 = __builtin_fmaf(v2, 0.999f, 0.001f); v3 = __builtin_fmaf(v3, 0.999f, 0.001f);
                } else {
                    v0 = __builtin_amdgcn_rcpf(v0 + 1.0f); v1 = __builtin_amdgcn_exp2f(v1 - 1.0f);
                    v2 = __builtin_amdgcn_rcpf(v2 + 1.0f); v3 = __builtin_amdgcn_exp2f(v3 - 1.0f);
                }
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3;
    }
}
template <int MODE> float run(float* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    float m = run<1>(out, iters);
    double flop = 256.0 * 4 * iters * 32.0 * (2.0 * 16 * 16 * 32);
    printf("bf16 MFMA only       %.3f ms  (%.0f TFLOP/s; one wave per SIMD)\n", m, flop / m / 1e9);
    printf("FMA only             %.3f ms\n", run<2>(out, iters));
    printf("trans only           %.3f ms\n", run<4>(out, iters));
    printf("bf16 MFMA + FMA      %.3f ms\n", run<3>(out, iters));
    printf("bf16 MFMA + trans    %.3f ms\n", run<5>(out, iters));
    return 0;
}
