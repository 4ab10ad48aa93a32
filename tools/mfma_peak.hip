// sustained fp32 MFMA rate of the whole chip (register-only loop, no memory): what "100 %" can be under the power cap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 768, 1024}) {
        for (int iters : {2000, 20000}) {
            hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)wgs * 4 * iters * 32.0 * 2048.0;
            printf("wgs %4d iters %5d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", wgs, iters, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
        }
    }
    return 0;
}
