// Do fp32 MFMA and VALU instructions of two waves on the same SIMD overlap on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (the second wave of each SIMD)
// issue plain FMAs / transcendentals.  Times: MFMA only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // bit 0: MFMA waves active, bit 1: FMA waves, bit 2: transcendental waves
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int w = threadIdx.x >> 6;
    float x = threadIdx.x * 1e-3f + 1.0f, y = blockIdx.x * 1e-6f + 0.5f;
    if (w < 4) {
        if (!(MODE & 1)) return;
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(MODE & 6)) return;
        float v0 = x, v1 = y, v2 = x + y, v3 = x - y;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE & 2) {     // 32 x 4 = 128 FMAs... same instruction count as the MFMA waves: 32 per iteration
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
    printf("per iteration: 32 MFMA 16x16x4 f32 per MFMA wave; 32 v_fma (or 16 v_rcp+16 v_exp + 32 v_add) per VALU wave\n");
    printf("MFMA only          %.3f ms\n", run<1>(out, iters));
    printf("FMA only           %.3f ms\n", run<2>(out, iters));
    printf("trans only         %.3f ms\n", run<4>(out, iters));
    printf("MFMA + FMA waves   %.3f ms\n", run<3>(out, iters));
    printf("MFMA + trans waves %.3f ms\n", run<5>(out, iters));
    return 0;
}
