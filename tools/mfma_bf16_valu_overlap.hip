// How much VALU work hides under v_mfma_f32_16x16x32_bf16 on gfx950?  (tools only)
//  A: 512-thread workgroups, waves 0-3 issue bf16 MFMAs, waves 4-7 (the second wave of each SIMD) issue FMAs / transcendentals;
//  B: ONE wave per SIMD interleaving each MFMA with K independent VALU instructions in program order (K = 0..6, fma or exp/rcp).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mf(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <int MODE>   // bit 0: MFMA waves active, bit 1: FMA waves, bit 2: transcendental waves
__global__ __launch_bounds__(512) void kA(float* out, int iters) {
    const int w = threadIdx.x >> 6;
    float x = threadIdx.x * 1e-3f + 1.0f, y = blockIdx.x * 1e-6f + 0.5f;
    if (w < 4) {
        if (!(MODE & 1)) return;
        uint4 a = {threadIdx.x, 0x3f803f80u, 0x3f803f80u, blockIdx.x}, b = {0x3f803f80u, threadIdx.x, 1u, 0x3f803f80u};
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0 = mf(a, b, a0); a1 = mf(b, a, a1); a2 = mf(a, a, a2); a3 = mf(b, b, a3); }
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
template <int K, bool TRANS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kB(float* out, int iters) {
    float x = threadIdx.x * 1e-3f + 1.0f;
    uint4 a = {threadIdx.x, 0x3f803f80u, 0x3f803f80u, blockIdx.x}, b = {0x3f803f80u, threadIdx.x, 1u, 0x3f803f80u};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            acc[j & 3] = mf(a, b, acc[j & 3]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float& t = v[(j * K + k) & 7];
                if (TRANS) t = (k & 1) ? __builtin_amdgcn_exp2f(t) : __builtin_amdgcn_rcpf(t);
                else t = __builtin_fmaf(t, 0.999f, 0.001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int MODE> float runA(float* out, int iters) { return timeit([&] { hipLaunchKernelGGL(kA<MODE>, dim3(256), dim3(512), 0, 0, out, iters); }); }
template <int K, bool T, int W> float runB(float* out, int iters) { return timeit([&] { hipLaunchKernelGGL((kB<K, T, W>), dim3(256), dim3(64 * W), 0, 0, out, iters); }); }
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    printf("A: per iteration 32 bf16 MFMA 16x16x32 per MFMA wave; 32 v_fma (or 16 v_rcp + 16 v_exp + 32 v_add) per VALU wave\n");
    printf("MFMA only          %.3f ms\n", runA<1>(out, iters));
    printf("FMA only           %.3f ms\n", runA<2>(out, iters));
    printf("trans only         %.3f ms\n", runA<4>(out, iters));
    printf("MFMA + FMA waves   %.3f ms\n", runA<3>(out, iters));
    printf("MFMA + trans waves %.3f ms\n", runA<5>(out, iters));
    printf("B: one wave per SIMD (4 per CU), 32 MFMA per iteration each followed by K VALU in program order: ms (cycles per MFMA slot at 2.4 GHz)\n");
#define ROW(K) { float f = runB<K, false, 4>(out, iters), t = runB<K, true, 4>(out, iters), f2 = runB<K, false, 8>(out, iters), t2 = runB<K, true, 8>(out, iters); \
    printf("K=%d  fma %.3f (%.1f)  trans %.3f (%.1f)   | two waves per SIMD: fma %.3f (%.1f per wave-MFMA)  trans %.3f (%.1f)\n", K, f, f * 2.4e6 / (iters * 32.0), t, t * 2.4e6 / (iters * 32.0), \
           f2, f2 * 2.4e6 / (iters * 32.0), t2, t2 * 2.4e6 / (iters * 32.0)); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6)
    return 0;
}
