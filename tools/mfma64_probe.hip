// sustained float64 MFMA rate (v_mfma_f64_16x16x4_f64, register-only loop) and its dependent-issue latency: the yardstick of dft64.h
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma64_loop(double* out, int iters, unsigned long long* cyc) {
    f64x4 a[NACC];
    for (int j = 0; j < NACC; ++j) a[j] = f64x4{0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int j = 0; j < NACC; ++j) s += a[j][j & 3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(double* out, unsigned long long* cyc, int wgs, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma64_loop<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma64_loop<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC, flop = (double)wgs * 4 * n * 2048.0;
    printf("acc %d wgs %4d: %.3f ms  %.1f TFLOP/s f64;  wave 0: %.1f clock ticks per MFMA (s_memtime/100MHz-class counter: compare the two rows)\n", NACC, wgs, ms, flop / ms / 1e9, (double)c / n);
}
int main() {
    double* out; hipMalloc(&out, 4096 * 256 * 8);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    for (int wgs : {256, 512, 1024}) { run<1>(out, cyc, wgs, 4000); run<2>(out, cyc, wgs, 4000); run<4>(out, cyc, wgs, 4000); }
    return 0;
}
