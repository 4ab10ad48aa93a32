// micro-benchmark of the GRU-64 scan kernels and their ablation variants (tools only; not part of the product build)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#ifndef GRU64_VARIANT
#define GRU64_VARIANT 0
#endif
#ifndef BENCH_ROWS
#define BENCH_ROWS 36864   // x 2 directions = 4608 workgroups = 6 full rounds of 768 slots (3 per CU)
#endif
#include "../dpdfnet_amd/csrc/gru_scan.h"
int main() {
    const int rows = BENCH_ROWS, Fp = 48;
    float *x, *out, *out2, *wf, *bias;
    const size_t no = (size_t)rows * Fp * 128;
    (void)hipMalloc(&x, (size_t)rows * Fp * 64 * 4); (void)hipMalloc(&out, no * 4); (void)hipMalloc(&out2, no * 4);
    (void)hipMalloc(&wf, 2 * 4 * 2 * 3 * 16 * 64 * 4); (void)hipMalloc(&bias, 2 * 256 * 4);
    std::vector<float> h((size_t)rows * Fp * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 199) - 1.0f;
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> w(2 * 4 * 2 * 3 * 16 * 64);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.002f * (float)((i * 40503u) % 101) - 0.1f;
    (void)hipMemcpy(wf, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(bias, 0, 2 * 256 * 4);
    Gru64Args a{}; a.x = x; a.out = out; a.wfrag = wf; a.bias = bias; a.hstate = nullptr;
    a.nrows = rows; a.nsteps = Fp; a.ndirs = 2; a.rdiv = 1; a.x_hi = Fp * 64; a.x_lo = 0; a.x_step = 64;
    a.o_hi = Fp * 128; a.o_lo = 0; a.o_step = 128; a.o_dir_off = 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 10;
    const double flops = (double)rows * Fp * 2 * 49152.0;
    float ms;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e0);
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("scan    variant %d rows %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", GRU64_VARIANT, rows, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    return 0;
}
