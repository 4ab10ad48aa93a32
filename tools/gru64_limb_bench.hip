// gru64_limb_bench.hip -- micro-benchmark + accuracy check of the GRU-64 scan with fp32 products formed from THREE bf16 limbs per operand
// (tools only; `hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gru64_limb_bench.hip -o tools/gru64_limb_bench`).
//
// Why: the fp32 matrix rate of CDNA4 is 256 FLOP/cycle/CU (v_mfma_f32_16x16x4_f32, 32 cycles per SIMD), the bf16 rate 4096 (v_mfma_f32_
// 16x16x32_bf16, 16 cycles).  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits, round-to-nearest residues
// are exact), a product of two bf16 values is exact in fp32, and the MFMA accumulates in fp32: a . b = sum of the nine limb products, of which
// lo x lo, lo x mid, mid x lo are below 2^-24 of the result.  Six bf16 MFMAs therefore reproduce an fp32 product term to fp32 rounding -- 2.67 x
// the fp32 matrix rate -- and the recurrence's error against float64 is that of the fp32 kernels (measured below against a double reference).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#ifndef BENCH_ROWS
#define BENCH_ROWS 36864
#endif
#include "../dpdfnet_amd/csrc/gru_scan.h"
#include "../dpdfnet_amd/csrc/gru_limb.h"

static unsigned short bf16_rne(float x) {
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int rows = BENCH_ROWS, Fp = 48;
    float *x, *out, *out2, *wf, *bias; uint4* wl;
    const size_t no = (size_t)rows * Fp * 128;
    (void)hipMalloc(&x, (size_t)rows * Fp * 64 * 4); (void)hipMalloc(&out, no * 4); (void)hipMalloc(&out2, no * 4);
    (void)hipMalloc(&wf, 2 * 4 * 2 * 3 * 16 * 64 * 4); (void)hipMalloc(&bias, 2 * 256 * 4);
    (void)hipMalloc(&wl, (size_t)2 * 4 * GRU64L_FRAG_PER_WAVE * 64 * sizeof(uint4));
    std::vector<float> h((size_t)rows * Fp * 64);
    unsigned long long sd = 88172645463325252ull;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (float)((sd >> 11) & 0xfffff) / 1048576.f * 2.f - 1.f; };
    for (auto& v : h) v = rnd();
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // canonical weights W[dir][side ih/hh][gate][unit][k], pre-scaled as build_gru64 does (the kernels' gate math takes exp2 arguments)
    std::vector<float> W((size_t)2 * 2 * 3 * 64 * 64), B(2 * 256);
    for (auto& v : W) v = 0.25f * rnd();
    for (auto& v : B) v = 0.2f * rnd();
    auto Wc = [&](int dir, int side, int g, int unit, int k) -> float& { return W[((((size_t)dir * 2 + side) * 3 + g) * 64 + unit) * 64 + k]; };
    // fp32 fragments: [dir][wave][part][gate][chunk][kb][lane], value W[gate][16w + cl][16c + 4q + kb]
    std::vector<float> w((size_t)2 * 4 * 2 * 3 * 16 * 64);
    for (int dir = 0; dir < 2; ++dir) for (int wv = 0; wv < 4; ++wv) for (int pt = 0; pt < 2; ++pt) for (int g = 0; g < 3; ++g)
        for (int c = 0; c < 4; ++c) for (int kb = 0; kb < 4; ++kb) for (int lane = 0; lane < 64; ++lane) {
            const int cl = lane & 15, q = lane >> 4;
            w[(((((size_t)(dir * 4 + wv) * 2 + pt) * 3 + g) * 16 + c * 4 + kb)) * 64 + lane] = Wc(dir, pt, g, 16 * wv + cl, 16 * c + 4 * q + kb);
        }
    (void)hipMemcpy(wf, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    // limb fragments: [dir][wave][mat = side * 3 + gate][chunk 2][limb 3][lane] of 8 bf16: A operand row m = lane & 15 = unit 16w + m, k = 32c + 8(lane >> 4) + j
    std::vector<unsigned short> wlh((size_t)2 * 4 * GRU64L_FRAG_PER_WAVE * 64 * 8);
    for (int dir = 0; dir < 2; ++dir) for (int wv = 0; wv < 4; ++wv) for (int side = 0; side < 2; ++side) for (int g = 0; g < 3; ++g)
        for (int c = 0; c < 2; ++c) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const float v = Wc(dir, side, g, 16 * wv + (lane & 15), 32 * c + 8 * (lane >> 4) + j);
            const unsigned short hi = bf16_rne(v); const float r1 = v - bf16_f(hi);
            const unsigned short mi = bf16_rne(r1); const float r2 = r1 - bf16_f(mi);
            const unsigned short lo = bf16_rne(r2);
            const unsigned short limbs[3] = {hi, mi, lo};
            for (int l = 0; l < 3; ++l)
                wlh[((((size_t)(dir * 4 + wv) * 6 + side * 3 + g) * 2 + c) * 3 + l) * 64 * 8 + (size_t)lane * 8 + j] = limbs[l];
        }
    (void)hipMemcpy(wl, wlh.data(), wlh.size() * 2, hipMemcpyHostToDevice);

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
    printf("fp32 MFMA scan      rows %d: %.3f ms  %.1f TFLOP/s useful (%.1f%% of the fp32 peak 157.3)\n", rows, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    Gru64Args b = a; b.out = out2;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gru64_scan_l3_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, b, (const uint4*)wl);
    (void)hipEventRecord(e0);
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(gru64_scan_l3_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, b, (const uint4*)wl);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms2; (void)hipEventElapsedTime(&ms2, e0, e1); ms2 /= N;
    printf("3-limb bf16 scan    rows %d: %.3f ms  %.1f TFLOP/s useful = %.1f TFLOP/s of bf16 MFMA issued (%.1f%% of 2500)   speed-up %.2f x\n", rows, ms2,
           flops / ms2 / 1e9, 6 * flops / ms2 / 1e9, 6 * flops / ms2 / 1e9 / 2500 * 100, ms / ms2);
    // accuracy: both against a double-precision recurrence on the first 32 rows (two tiles), both directions
    const int R = 32;
    std::vector<float> o1((size_t)R * Fp * 128), o2((size_t)R * Fp * 128);
    (void)hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2.data(), out2, o2.size() * 4, hipMemcpyDeviceToHost);
    double e1s = 0, e2s = 0, d12 = 0, sig = 0, m1 = 0, m2 = 0; size_t cnt = 0;
    for (int r = 0; r < R; ++r) for (int dir = 0; dir < 2; ++dir) {
        double hh[64] = {0};
        for (int s = 0; s < Fp; ++s) {
            const int p = dir ? Fp - 1 - s : s;
            const float* xr = &h[((size_t)r * Fp + p) * 64];
            double pre[2][3][64];
            for (int side = 0; side < 2; ++side) for (int g = 0; g < 3; ++g) for (int u = 0; u < 64; ++u) {
                double acc = 0;
                for (int k = 0; k < 64; ++k) acc += (double)Wc(dir, side, g, u, k) * (side ? hh[k] : (double)xr[k]);
                pre[side][g][u] = acc;
            }
            double hn[64];
            for (int u = 0; u < 64; ++u) {
                const double ar = pre[0][0][u] + pre[1][0][u] + B[dir * 256 + u], az = pre[0][1][u] + pre[1][1][u] + B[dir * 256 + 64 + u];
                const double rr = 1.0 / (1.0 + exp2(ar)), zz = 1.0 / (1.0 + exp2(az));
                const double t = pre[0][2][u] + B[dir * 256 + 128 + u] + rr * (pre[1][2][u] + B[dir * 256 + 192 + u]);
                const double n = 2.0 / (1.0 + exp2(t)) - 1.0;
                hn[u] = n + zz * (hh[u] - n);
            }
            for (int u = 0; u < 64; ++u) {
                hh[u] = hn[u];
                const size_t o = ((size_t)r * Fp + p) * 128 + dir * 64 + u;
                const double a1 = o1[o] - hn[u], a2 = o2[o] - hn[u];
                e1s += a1 * a1; e2s += a2 * a2; sig += hn[u] * hn[u]; d12 += ((double)o1[o] - o2[o]) * ((double)o1[o] - o2[o]); ++cnt;
                if (fabs(a1) > m1) m1 = fabs(a1);
                if (fabs(a2) > m2) m2 = fabs(a2);
            }
        }
    }
    printf("against the float64 recurrence (%d rows x %d steps x 2 directions, signal RMS %.3f):\n  fp32 MFMA kernel   RMS %.3e  max %.3e\n  3-limb bf16 kernel RMS %.3e  max %.3e\n  kernel vs kernel   RMS %.3e\n",
           R, Fp, sqrt(sig / cnt), sqrt(e1s / cnt), m1, sqrt(e2s / cnt), m2, sqrt(d12 / cnt));
    return 0;
}
