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

__global__ __launch_bounds__(256, 1) void cotenant_kernel(float* junk, int iters) {
    __shared__ float big[33000];                        // 132 KB: one workgroup per CU, nothing beside it
    for (int i = threadIdx.x; i < 33000; i += 256) big[i] = i;
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc += big[(threadIdx.x * 17 + it * 31) % 33000] + __hip_atomic_load(junk + (blockIdx.x * 256 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    junk[blockIdx.x * 256 + threadIdx.x] = acc * 1e-30f;
}
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

    // fc_intra [64 out][128 in] (in 0..63 = hf, 64..127 = hb), fc_inter [64][64], biases, LayerNorm parameters
    std::vector<float> Fi(64 * 128), Fe(64 * 64), fb(64), lg(64), lb(64);
    for (auto& v : Fi) v = 0.15f * rnd();
    for (auto& v : Fe) v = 0.15f * rnd();
    for (int i = 0; i < 64; ++i) { fb[i] = 0.1f * rnd(); lg[i] = 1.0f + 0.2f * rnd(); lb[i] = 0.1f * rnd(); }
    auto pack_epi = [&](const float* Wm, int ld, int koff) {
        std::vector<float> f((size_t)4 * 16 * 64);
        for (int wv = 0; wv < 4; ++wv) for (int c = 0; c < 4; ++c) for (int kb = 0; kb < 4; ++kb) for (int lane = 0; lane < 64; ++lane)
            f[((size_t)wv * 16 + c * 4 + kb) * 64 + lane] = Wm[(size_t)(16 * wv + (lane & 15)) * ld + koff + 16 * c + 4 * (lane >> 4) + kb];
        return f;
    };
    auto pack_fcl = [&](const float* Wm, int ld, int koff) {       // [wave][chunk][limb][lane] x 8 bf16
        std::vector<unsigned short> f((size_t)4 * 6 * 64 * 8);
        for (int wv = 0; wv < 4; ++wv) for (int c = 0; c < 2; ++c) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const float v = Wm[(size_t)(16 * wv + (lane & 15)) * ld + koff + 32 * c + 8 * (lane >> 4) + j];
            const unsigned short hi = bf16_rne(v); const float r1 = v - bf16_f(hi);
            const unsigned short mi = bf16_rne(r1); const float r2 = r1 - bf16_f(mi);
            const unsigned short limbs[3] = {hi, mi, bf16_rne(r2)};
            for (int l = 0; l < 3; ++l) f[(((size_t)wv * 2 + c) * 3 + l) * 64 * 8 + (size_t)lane * 8 + j] = limbs[l];
        }
        return f;
    };
    auto up = [&](const void* src, size_t bytes) { void* d; (void)hipMalloc(&d, bytes); (void)hipMemcpy(d, src, bytes, hipMemcpyHostToDevice); return d; };
    std::vector<float> fi_epi = pack_epi(Fi.data(), 128, 64), fi1 = pack_epi(Fi.data(), 128, 0);
    fi_epi.insert(fi_epi.end(), fi1.begin(), fi1.end());
    std::vector<float> fe_epi = pack_epi(Fe.data(), 64, 0);
    const float* d_fi_epi = (const float*)up(fi_epi.data(), fi_epi.size() * 4); const float* d_fe_epi = (const float*)up(fe_epi.data(), fe_epi.size() * 4);
    auto fl_b = pack_fcl(Fi.data(), 128, 64), fl_f = pack_fcl(Fi.data(), 128, 0), fl_e = pack_fcl(Fe.data(), 64, 0);
    const uint4* d_fl_b = (const uint4*)up(fl_b.data(), fl_b.size() * 2); const uint4* d_fl_f = (const uint4*)up(fl_f.data(), fl_f.size() * 2);
    const uint4* d_fl_e = (const uint4*)up(fl_e.data(), fl_e.size() * 2);
    const float* d_fb = (const float*)up(fb.data(), 256); const float* d_lg = (const float*)up(lg.data(), 256); const float* d_lb = (const float*)up(lb.data(), 256);
    float *hf, *pf, *y1, *y2, *hs1, *hs2;
    const size_t nx = (size_t)rows * Fp * 64;
    (void)hipMalloc(&hf, nx * 4); (void)hipMalloc(&pf, nx * 4); (void)hipMalloc(&y1, nx * 4); (void)hipMalloc(&y2, nx * 4);
    (void)hipMalloc(&hs1, (size_t)rows * 64 * 4); (void)hipMalloc(&hs2, (size_t)rows * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 10;
    auto timeit = [&](auto f) { for (int i = 0; i < 2; ++i) f(); (void)hipEventRecord(e0); for (int i = 0; i < N; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                                float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / N; };
    auto cmp = [&](const char* what, const float* da, const float* db, size_t cnt) {
        std::vector<float> A(cnt), Bv(cnt);
        (void)hipMemcpy(A.data(), da, cnt * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(Bv.data(), db, cnt * 4, hipMemcpyDeviceToHost);
        double d = 0, sg = 0, mx = 0; bool fin = true;
        for (size_t i = 0; i < cnt; ++i) { const double e = (double)A[i] - Bv[i]; d += e * e; sg += (double)A[i] * A[i]; if (fabs(e) > mx) mx = fabs(e); fin = fin && std::isfinite(Bv[i]); }
        printf("  %s: limb vs fp32 kernels RMS %.3e  max %.3e  (signal RMS %.3f, finite %d)\n", what, sqrt(d / cnt), mx, sqrt(sg / cnt), (int)fin);
    };
    // ---- intra-band pair: rows = frames, steps = band positions
    {
        Gru64Args a{}; a.x = x; a.wfrag = wf; a.bias = bias; a.hstate = nullptr;
        a.nrows = rows; a.nsteps = Fp; a.rdiv = 1; a.x_hi = Fp * 64; a.x_lo = 0; a.x_step = 64;
        a.out = hf; a.ndirs = 1; a.o_hi = Fp * 64; a.o_lo = 0; a.o_step = 64; a.o_dir_off = 0;
        Gru64EpiArgs e2{a, d_fi_epi, d_fb, d_lg, d_lb, hf, y1};
        const float t_f = timeit([&] { hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 1), dim3(256), 0, 0, a); });
        const float t_b = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_epi_kernel<2>), dim3(rows / 16), dim3(256), 0, 0, e2); });
        Gru64Args b = a; b.out = pf;
        Gru64LArgs l0{b, (const uint4*)wl, d_fl_f, nullptr, nullptr, nullptr, nullptr, nullptr};
        Gru64LArgs l2{b, (const uint4*)wl, d_fl_b, d_fb, d_lg, d_lb, pf, y2};
        const float u_f = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<0>), dim3(rows / 16), dim3(256), 0, 0, l0); });
        const float u_b = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<2>), dim3(rows / 16), dim3(256), 0, 0, l2); });
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        printf("intra forward : fp32 %.3f ms   limbs %.3f ms   %.2f x\nintra backward: fp32 %.3f ms   limbs %.3f ms   %.2f x\n", t_f, u_f, t_f / u_f, t_b, u_b, t_b / u_b);
        cmp("intra block output y", y1, y2, (size_t)4096 * Fp * 64);
        // the same pair beside a co-tenant on another stream (64 workgroups that take a whole CU each: 132 KB of LDS, spinning on memory):
        // results must be bit-identical to the run alone
        std::vector<float> ref(nx), got(nx);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(ref.data(), y2, nx * 4, hipMemcpyDeviceToHost);
        hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
        float* junk; (void)hipMalloc(&junk, 64 * 256 * 4);
        int nbad_total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            (void)hipMemsetAsync(y2, 0, nx * 4, s1); (void)hipMemsetAsync(pf, 0, nx * 4, s1);
            hipLaunchKernelGGL(cotenant_kernel, dim3(64), dim3(256), 0, s2, junk, 3000 + 500 * rep);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<0>), dim3(rows / 16), dim3(256), 0, s1, l0);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<2>), dim3(rows / 16), dim3(256), 0, s1, l2);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(got.data(), y2, nx * 4, hipMemcpyDeviceToHost);
            size_t nbad = 0, first = 0;
            for (size_t i = 0; i < nx; ++i) if (got[i] != ref[i]) { if (!nbad) first = i; ++nbad; }
            if (nbad) printf("  rep %d beside a co-tenant: %zu values differ from the run alone (first at row %zu pos %zu ch %zu)\n", rep, nbad, first / (Fp * 64), first / 64 % Fp, first % 64);
            nbad_total += nbad != 0;
        }
        printf("  intra pair beside a co-tenant: %d of 20 runs differ from the run alone\n", nbad_total);
    }
    // ---- inter-band: rows = (clip, band position), steps = frames, carried state
    {
        const int Tc = 192, Bc = rows * Fp / (Tc * Fp) , nr = Bc * Fp;          // same number of x elements
        Gru64Args a{}; a.x = x; a.wfrag = wf; a.bias = bias;
        a.nrows = nr; a.nsteps = Tc; a.ndirs = 1; a.rdiv = Fp;
        a.x_hi = (long)Tc * Fp * 64; a.x_lo = 64; a.x_step = (long)Fp * 64; a.o_hi = a.x_hi; a.o_lo = 64; a.o_step = a.x_step; a.o_dir_off = 0;
        a.h_hi = (long)Fp * 64; a.h_lo = 64; a.out = nullptr;
        std::vector<float> h0((size_t)rows * 64);
        for (auto& v : h0) v = 0.5f * rnd();
        Gru64Args a1 = a; a1.hstate = hs1; Gru64Args a2 = a; a2.hstate = hs2;
        Gru64EpiArgs e1a{a1, d_fe_epi, d_fb, d_lg, d_lb, nullptr, y1};
        Gru64LArgs l1{a2, (const uint4*)wl, d_fl_e, d_fb, d_lg, d_lb, nullptr, y2};
        const float t_e = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_epi_kernel<1>), dim3((nr + 15) / 16), dim3(256), 0, 0, e1a); });
        const float u_e = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<1>), dim3((nr + 15) / 16), dim3(256), 0, 0, l1); });
        printf("inter (%d rows x %d frames): fp32 %.3f ms   limbs %.3f ms   %.2f x\n", nr, Tc, t_e, u_e, t_e / u_e);
        (void)hipMemcpy(hs1, h0.data(), h0.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(hs2, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_epi_kernel<1>), dim3((nr + 15) / 16), dim3(256), 0, 0, e1a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<1>), dim3((nr + 15) / 16), dim3(256), 0, 0, l1);
        (void)hipDeviceSynchronize();
        cmp("inter block output y", y1, y2, (size_t)16 * Tc * Fp * 64);
        cmp("inter carried state ", hs1, hs2, (size_t)nr * 64);
        // ---- both kernel families against a FLOAT64 recurrence: the carried state after Tc steps of the first 64 rows (same pre-scaled weights:
        // r, z = 1 / (1 + 2^a), n = 2 / (1 + 2^(xn + r hn)) - 1, h' = n + z (h - n): common.h gru64_cell)
        {
            const int R = 64 < nr ? 64 : nr;
            std::vector<double> hd((size_t)R * 64);
            for (int r = 0; r < R; ++r) for (int u = 0; u < 64; ++u) hd[(size_t)r * 64 + u] = h0[(size_t)r * 64 + u];
            std::vector<double> g(6 * 64);
            for (int r = 0; r < R; ++r) {
                const int hi = r / Fp, lo = r % Fp;
                for (int t = 0; t < Tc; ++t) {
                    const float* xr = h.data() + (size_t)hi * a.x_hi + (size_t)lo * a.x_lo + (size_t)t * a.x_step;
                    double* hr = &hd[(size_t)r * 64];
                    for (int gt = 0; gt < 3; ++gt) for (int u = 0; u < 64; ++u) {
                        double sx = 0, sh = 0;
                        for (int k = 0; k < 64; ++k) { sx += (double)Wc(0, 0, gt, u, k) * xr[k]; sh += (double)Wc(0, 1, gt, u, k) * hr[k]; }
                        g[gt * 64 + u] = sx; g[(3 + gt) * 64 + u] = sh;
                    }
                    double hn_[64];
                    for (int u = 0; u < 64; ++u) {
                        const double ar = g[u] + g[192 + u] + B[u], az = g[64 + u] + g[256 + u] + B[64 + u];
                        const double rr_ = 1.0 / (1.0 + exp2(ar)), zz = 1.0 / (1.0 + exp2(az));
                        const double nn = 2.0 / (1.0 + exp2((g[128 + u] + B[128 + u]) + rr_ * (g[320 + u] + B[192 + u]))) - 1.0;
                        hn_[u] = nn + zz * (hr[u] - nn);
                    }
                    for (int u = 0; u < 64; ++u) hr[u] = hn_[u];
                }
            }
            std::vector<float> s1((size_t)R * 64), s2((size_t)R * 64);
            (void)hipMemcpy(s1.data(), hs1, s1.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(s2.data(), hs2, s2.size() * 4, hipMemcpyDeviceToHost);
            double d1 = 0, d2 = 0, m1 = 0, m2 = 0, sg = 0;
            for (size_t i = 0; i < hd.size(); ++i) {
                const double e1 = s1[i] - hd[i], e2 = s2[i] - hd[i];
                d1 += e1 * e1; d2 += e2 * e2; sg += hd[i] * hd[i]; if (fabs(e1) > m1) m1 = fabs(e1); if (fabs(e2) > m2) m2 = fabs(e2);
            }
            printf("float64 recurrence (%d rows x %d steps, state RMS %.3f): fp32-MFMA kernel RMS %.3e max %.3e | limb kernel RMS %.3e max %.3e\n",
                   R, Tc, sqrt(sg / hd.size()), sqrt(d1 / hd.size()), m1, sqrt(d2 / hd.size()), m2);
        }
    }
    return 0;
}
