// dec_seg_bench.hip -- the 48 kHz decoder stage kernels alone: dec_seg_kernel (dec_last.h) against the tile-pipelined form
// (dpdfnet_amd/csrc/dec_seg2.h), outputs compared bit for bit.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dec_seg_bench.hip -o tools/dec_seg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../dpdfnet_amd/csrc/dec_seg2.h"

static unsigned long long sd = 88172645463325252ull;
static float rnd() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 11) & 0xfffff) / 1048576.f * 2.f - 1.f; }
static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (auto& v : h) v = scale * rnd();
    float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) / 32768.f - 1.f;
    }
}
template <class F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

template <int S, int R, bool LAST>
static void run(int BT, int FO) {
    const int FI = FO / S;
    const size_t nin = (size_t)BT * FI * 64, nout = LAST ? (size_t)BT * FO * 4 : (size_t)BT * FO * 64, ne0 = LAST ? (size_t)BT * FO * 64 : 64;
    float *e, *prev, *out0, *out1, *e0;
    (void)hipMalloc(&e, nin * 4); (void)hipMalloc(&prev, nin * 4); (void)hipMalloc(&out0, nout * 4); (void)hipMalloc(&out1, nout * 4); (void)hipMalloc(&e0, ne0 * 4);
    fill_kernel<<<2048, 256>>>(e, nin, 1u); fill_kernel<<<2048, 256>>>(prev, nin, 2u); fill_kernel<<<2048, 256>>>(e0, ne0, 3u);
    DecSegArgs a{e, prev, LAST ? nullptr : out0, dev_rand(64, 1.f), dev_rand(64, 0.3f), dev_rand(S * 64 * 3, 0.5f), dev_rand(64 * 64, 0.15f), dev_rand(64, 0.2f),
                 e0, LAST ? out0 : nullptr, dev_rand(64, 1.f), dev_rand(64, 0.3f), dev_rand(64 * 3, 0.3f), BT, FO};
    DecSegArgs b = a; if (LAST) b.ssum = out1; else b.out = out1;
    const long ntiles = (long)BT * (FO / R);
    const double gb = (double)ntiles * ((R / S + 2) * 2 * 256 + (LAST ? R * 256 + R * 16 : R * 256)) / 1e9;
    (void)hipMemset(out0, 0, nout * 4); (void)hipMemset(out1, 0xff, nout * 4);
    const float t0 = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<S, R, LAST>), dim3((unsigned)std::min<long>(ntiles, 2048)), dim3(256), 0, 0, a); }, 5);
    printf("S %d R %d FO %3d  %ld tiles  %.2f GB   dec_seg_kernel       %7.3f ms  %6.0f GB/s\n", S, R, FO, ntiles, gb, t0, gb / t0 * 1e3);
    for (int grid : {256, 1024, 2048}) {
        const float t1 = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<S, R, LAST>), dim3((unsigned)std::min<long>(ntiles, grid)), dim3(512), 0, 0, b); }, 5);
        printf("                                 dec_seg2 grid %4d   %7.3f ms  %6.0f GB/s  (var %d)\n", grid, t1, gb / t1 * 1e3, DS2_VAR);
    }
    std::vector<float> h0(nout), h1(nout);
    (void)hipMemcpy(h0.data(), out0, nout * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), out1, nout * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double ss = 0, dd = 0, dmax = 0;
    for (size_t i = 0; i < nout; ++i) {
        bad += memcmp(&h0[i], &h1[i], 4) != 0; ss += (double)h0[i] * h0[i];
        const double d = (double)h0[i] - h1[i]; dd += d * d; if (fabs(d) > dmax) dmax = fabs(d);
    }
    printf("        outputs differ in %zu of %zu words (rms %.4f; difference rms %.3g max %.3g)\n", bad, nout, sqrt(ss / nout), sqrt(dd / nout), dmax);
    (void)hipFree(e); (void)hipFree(prev); (void)hipFree(out0); (void)hipFree(out1); (void)hipFree(e0);
}

int main(int argc, char** argv) {
    const int BT = argc > 1 ? atoi(argv[1]) : 49152;          // 256 clips x a 192-frame chunk
    const bool quick = argc > 2;                              // timing builds: the two big shapes only
    if (!quick) run<2, 80, false>(BT, 80);
    run<2, 80, false>(BT, 160);
    run<3, 96, true>(BT, 480);
    if (quick) return 0;
    run<2, 80, false>(1000, 160);                             // one clip
    run<3, 96, true>(1000, 480);
    run<3, 96, true>(37, 480);                                // ragged tail
    return 0;
}
