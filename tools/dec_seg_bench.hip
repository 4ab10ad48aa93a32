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
        const float t1 = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<S, R, LAST>), dim3((unsigned)std::min<long>(BT, grid)), dim3(512), 0, 0, b); }, 5);
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

// the three stages chained as in the engine (d3 -> d2 -> tap sums): three dec_seg_kernel launches, three dec_seg2_kernel launches, one dec_seg2_all_kernel
static void run_chain(int BT) {
    const size_t n3 = (size_t)BT * 40 * 64, n2 = (size_t)BT * 80 * 64, n1 = (size_t)BT * 160 * 64, n0 = (size_t)BT * 480 * 64, ns = (size_t)BT * 480 * 4;
    float *e3, *emb, *e2, *e1, *e0, *d3[3], *d2[3], *ss[3];
    (void)hipMalloc(&e3, n3 * 4); (void)hipMalloc(&emb, n3 * 4); (void)hipMalloc(&e2, n2 * 4); (void)hipMalloc(&e1, n1 * 4); (void)hipMalloc(&e0, n0 * 4);
    for (int k = 0; k < 3; ++k) { (void)hipMalloc(&d3[k], n2 * 4); (void)hipMalloc(&d2[k], n1 * 4); (void)hipMalloc(&ss[k], ns * 4); (void)hipMemset(ss[k], 0xff, ns * 4); }
    fill_kernel<<<2048, 256>>>(e3, n3, 1u); fill_kernel<<<2048, 256>>>(emb, n3, 2u); fill_kernel<<<2048, 256>>>(e2, n2, 3u);
    fill_kernel<<<2048, 256>>>(e1, n1, 4u); fill_kernel<<<2048, 256>>>(e0, n0, 5u);
    const float *ps = dev_rand(64, 1.f), *pb = dev_rand(64, 0.3f), *dw2 = dev_rand(2 * 64 * 3, 0.5f), *dw3 = dev_rand(3 * 64 * 3, 0.5f), *pw = dev_rand(64 * 64, 0.15f), *bi = dev_rand(64, 0.2f), *w0 = dev_rand(64 * 3, 0.3f);
    auto A3 = [&](int k) { return DecSegArgs{e3, emb, d3[k], ps, pb, dw2, pw, bi, nullptr, nullptr, nullptr, nullptr, nullptr, BT, 80}; };
    auto A2 = [&](int k) { return DecSegArgs{e2, d3[k], d2[k], ps, pb, dw2, pw, bi, nullptr, nullptr, nullptr, nullptr, nullptr, BT, 160}; };
    auto A1 = [&](int k) { return DecSegArgs{e1, d2[k], nullptr, ps, pb, dw3, pw, bi, e0, ss[k], ps, pb, w0, BT, 480}; };
    const float t0 = time_ms([&] {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<2, 80, false>), dim3((unsigned)std::min<long>(BT, 2048)), dim3(256), 0, 0, A3(0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<2, 80, false>), dim3((unsigned)std::min<long>(2L * BT, 2048)), dim3(256), 0, 0, A2(0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<3, 96, true>), dim3((unsigned)std::min<long>(5L * BT, 2048)), dim3(256), 0, 0, A1(0)); }, 5);
    const float t1 = time_ms([&] {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), dim3((unsigned)std::min(BT, 256)), dim3(512), 0, 0, A3(1));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), dim3((unsigned)std::min(BT, 256)), dim3(512), 0, 0, A2(1));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<3, 96, true>), dim3((unsigned)std::min(BT, 256)), dim3(512), 0, 0, A1(1)); }, 5);
    const float t2 = time_ms([&] { hipLaunchKernelGGL(dec_seg2_all_kernel, dim3((unsigned)std::min(BT, 256)), dim3(512), 0, 0, A3(2), A2(2), A1(2)); }, 5);
    printf("chain, %d frames: dec_seg_kernel x 3 %.3f ms | dec_seg2_kernel x 3 %.3f ms | dec_seg2_all_kernel %.3f ms\n", BT, t0, t1, t2);
    std::vector<float> h0(ns), h1(ns), h2(ns), g1(n1), g2(n1);
    (void)hipMemcpy(h0.data(), ss[0], ns * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), ss[1], ns * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h2.data(), ss[2], ns * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(g1.data(), d2[1], n1 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(g2.data(), d2[2], n1 * 4, hipMemcpyDeviceToHost);
    size_t bad12 = 0, badd = 0; double dmax = 0;
    for (size_t i = 0; i < ns; ++i) { bad12 += memcmp(&h1[i], &h2[i], 4) != 0; const double d = fabs((double)h0[i] - h2[i]); if (d > dmax) dmax = d; }
    for (size_t i = 0; i < n1; ++i) badd += memcmp(&g1[i], &g2[i], 4) != 0;
    printf("        one launch vs three dec_seg2 launches: %zu tap-sum words and %zu d2 words differ; max |tap sum - dec_seg_kernel's| %.3g\n", bad12, badd, dmax);
    (void)hipFree(e3); (void)hipFree(emb); (void)hipFree(e2); (void)hipFree(e1); (void)hipFree(e0);
    for (int k = 0; k < 3; ++k) { (void)hipFree(d3[k]); (void)hipFree(d2[k]); (void)hipFree(ss[k]); }
}

int main(int argc, char** argv) {
    const int BT = argc > 1 ? atoi(argv[1]) : 49152;          // 256 clips x a 192-frame chunk
    run_chain(BT); run_chain(1067);
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
