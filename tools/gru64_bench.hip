// micro-benchmark of gru64_scan_kernel variants (tools only; not part of the product build)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef GRU64_VARIANT
#define GRU64_VARIANT 0
#endif
#include "../dpdfnet_amd/csrc/gru_scan.h"
int main() {
    const int rows = 32768, Fp = 48;
    float *x, *out, *wf, *bias;
    hipMalloc(&x, (size_t)rows * Fp * 64 * 4); hipMalloc(&out, (size_t)rows * Fp * 128 * 4);
    hipMalloc(&wf, 2 * 4 * 2 * 3 * 16 * 64 * 4); hipMalloc(&bias, 2 * 256 * 4);
    std::vector<float> h((size_t)rows * Fp * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 199) - 1.0f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> w(2 * 4 * 2 * 3 * 16 * 64);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.002f * (float)((i * 40503u) % 101) - 0.1f;
    hipMemcpy(wf, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 2 * 256 * 4);
    Gru64Args a{}; a.x = x; a.out = out; a.wfrag = wf; a.bias = bias; a.hstate = nullptr;
    a.nrows = rows; a.nsteps = Fp; a.ndirs = 2; a.rdiv = 1; a.x_hi = Fp * 64; a.x_lo = 0; a.x_step = 64;
    a.o_hi = Fp * 128; a.o_lo = 0; a.o_step = 128; a.o_dir_off = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    hipEventRecord(e0);
    const int N = 10;
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    double flops = (double)rows * Fp * 2 * 49152.0;
    printf("variant %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", GRU64_VARIANT, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    return 0;
}
