// Research probe for a later round (NOT part of the product): the GRU-64 scan step with every fp32 operand split into
// three bf16 limbs (a = a1 + a2 + a3 to 24 bits) and the six leading limb products issued as v_mfma_f32_16x16x32_bf16
// with fp32 accumulation -- fp32-accurate to ~2^-22, on the matrix pipe that is 14x faster than the fp32 MFMA and that
// overlaps with the VALU (tools/bf16_overlap.hip).  Prints time, fp32-equivalent TFLOP/s and the deviation from the
// shipped fp32 kernel on the same data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "../dpdfnet_amd/csrc/gru_scan.h"
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float a, __bf16& l1, __bf16& l2, __bf16& l3) {
    l1 = (__bf16)a; float r = a - (float)l1;
    l2 = (__bf16)r; r -= (float)l2;
    l3 = (__bf16)r;
}

// wlimb: [dir][wave 4][gate 3][kblock 4 (x0,x1,h0,h1)][limb 3][lane 64][8] bf16, element j = W[k = 32 kb' + 8 (lane>>4) + j][col]
__global__ __launch_bounds__(256, 2) void gru64_scan_bf16x3_kernel(Gru64Args a, const __bf16* wlimb) {
    __shared__ __attribute__((aligned(16))) __bf16 Hl[2][3][16][72];    // [buf][limb][row][k] (72: 16-byte aligned rows, no 2-way conflicts)
    __shared__ __attribute__((aligned(16))) __bf16 Xl[2][3][16][72];
    __shared__ __attribute__((aligned(16))) float Ho[16][68];           // fp32 h' for the row-contiguous global store
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int dir = blockIdx.y, row0 = blockIdx.x * 16;
    const int cl = lane & 15, q = lane >> 4;
    bf16x8 wb[3][4][3];
    {
        const bf16x8* wp = (const bf16x8*)wlimb + ((size_t)(dir * 4 + w) * 3 * 4 * 3) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int l = 0; l < 3; ++l) wb[g][kb][l] = wp[(size_t)((g * 4 + kb) * 3 + l) * 64];
    }
    const float* bp = a.bias + (size_t)dir * 256 + 16 * w + cl;
    const float b_r = bp[0], b_z = bp[64], b_in = bp[128], b_hn = bp[192];
    const float* xbase = a.x + (long)row0 * a.x_hi;
    float* obase = a.out + (long)row0 * a.o_hi + dir * a.o_dir_off;
    const int srow = 4 * w + q, scol = 4 * cl;
    const unsigned sx_off = (unsigned)((long)srow * a.x_hi) + scol, so_off = (unsigned)((long)srow * a.o_hi) + scol;
    float h_own[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 3 * 16 * 72; i += 256) (&Hl[1][0][0][0])[i] = (__bf16)0.f;
    auto stage_x = [&](int buf, float4 v) {
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 l1, l2, l3; split3(xv[j], l1, l2, l3);
            Xl[buf][0][srow][scol + j] = l1; Xl[buf][1][srow][scol + j] = l2; Xl[buf][2][srow][scol + j] = l3;
        }
    };
    {
        const int p0 = dir ? a.nsteps - 1 : 0;
        stage_x(0, *(const float4*)((xbase + (long)p0 * a.x_step) + sx_off));
    }
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < a.nsteps; ++s) {
        if (s > 0) {
            const int pp = dir ? a.nsteps - s : s - 1;
            *(float4*)((obase + (long)pp * a.o_step) + so_off) = *(const float4*)&Ho[srow][scol];
        }
        const int sn = s + 1 < a.nsteps ? s + 1 : s;
        const float4 xnext = *(const float4*)((xbase + (long)(dir ? a.nsteps - 1 - sn : sn) * a.x_step) + sx_off);
        f32x4 ar = {b_r, b_r, b_r, b_r}, az = {b_z, b_z, b_z, b_z};
        f32x4 axn = {b_in, b_in, b_in, b_in}, ahn = {b_hn, b_hn, b_hn, b_hn};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 al[3];
#pragma unroll
            for (int l = 0; l < 3; ++l)
                al[l] = kb < 2 ? *(const bf16x8*)&Xl[buf][l][cl][32 * kb + 8 * q]
                               : *(const bf16x8*)&Hl[buf ^ 1][l][cl][32 * (kb - 2) + 8 * q];
            // six leading limb products, small ones first
#define LIMB6(acc, g)                                                                               \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[2], wb[g][kb][0], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], wb[g][kb][1], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][2], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1], wb[g][kb][0], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][1], acc, 0, 0, 0);        \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wb[g][kb][0], acc, 0, 0, 0);
            LIMB6(ar, 0) LIMB6(az, 1)
            if (kb < 2) { LIMB6(axn, 2) } else { LIMB6(ahn, 2) }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float h = gru64_cell(ar[i], az[i], axn[i], ahn[i], h_own[i]);
            h_own[i] = h;
            Ho[q * 4 + i][16 * w + cl] = h;
            __bf16 l1, l2, l3; split3(h, l1, l2, l3);
            Hl[buf][0][q * 4 + i][16 * w + cl] = l1; Hl[buf][1][q * 4 + i][16 * w + cl] = l2; Hl[buf][2][q * 4 + i][16 * w + cl] = l3;
        }
        stage_x(buf ^ 1, xnext);
        __syncthreads();
        buf ^= 1;
    }
    if (a.nsteps > 0) {
        const int pp = dir ? 0 : a.nsteps - 1;
        *(float4*)((obase + (long)pp * a.o_step) + so_off) = *(const float4*)&Ho[srow][scol];
    }
}

int main() {
    const int rows = 36864, Fp = 48;
    float *x, *out, *out2, *wf, *bias; __bf16* wl;
    const size_t no = (size_t)rows * Fp * 128;
    (void)hipMalloc(&x, (size_t)rows * Fp * 64 * 4); (void)hipMalloc(&out, no * 4); (void)hipMalloc(&out2, no * 4);
    (void)hipMalloc(&wf, 2 * 4 * 2 * 3 * 16 * 64 * 4); (void)hipMalloc(&bias, 2 * 256 * 4);
    std::vector<float> h((size_t)rows * Fp * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 199) - 1.0f;
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // dense weights W[dir][part ih/hh][gate][col 64][k 64], the same values in both packings
    std::vector<float> W((size_t)2 * 2 * 3 * 64 * 64);
    for (size_t i = 0; i < W.size(); ++i) W[i] = (0.002f * (float)((i * 40503u) % 101) - 0.1f) * (i % 3 == 0 ? -1.4f : 1.0f);
    auto Wat = [&](int d, int part, int g, int col, int k) { return W[((((size_t)d * 2 + part) * 3 + g) * 64 + col) * 64 + k]; };
    std::vector<float> frag((size_t)2 * 4 * 2 * 3 * 16 * 64);      // fp32 kernel: [dir][wave][part][gate][c*4+kb][lane]
    for (int d = 0; d < 2; ++d) for (int w = 0; w < 4; ++w) for (int part = 0; part < 2; ++part) for (int g = 0; g < 3; ++g)
        for (int c = 0; c < 4; ++c) for (int kb = 0; kb < 4; ++kb) for (int lane = 0; lane < 64; ++lane)
            frag[(((((size_t)(d * 4 + w) * 2 + part) * 3 + g) * 16) + c * 4 + kb) * 64 + lane] = Wat(d, part, g, 16 * w + (lane & 15), kperm(c, lane >> 4, kb));
    (void)hipMemcpy(wf, frag.data(), frag.size() * 4, hipMemcpyHostToDevice);
    std::vector<__bf16> limbs((size_t)2 * 4 * 3 * 4 * 3 * 64 * 8);
    for (int d = 0; d < 2; ++d) for (int w = 0; w < 4; ++w) for (int g = 0; g < 3; ++g) for (int kb = 0; kb < 4; ++kb)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int part = kb >> 1, k = 32 * (kb & 1) + 8 * (lane >> 4) + j;
            float a = Wat(d, part, g, 16 * w + (lane & 15), k);
            float r = a; __bf16 l[3];
            for (int t = 0; t < 3; ++t) { l[t] = (__bf16)r; r -= (float)l[t]; }
            for (int t = 0; t < 3; ++t)
                limbs[((((((size_t)(d * 4 + w) * 3 + g) * 4 + kb) * 3 + t) * 64) + lane) * 8 + j] = l[t];
        }
    (void)hipMalloc(&wl, limbs.size() * 2);
    (void)hipMemcpy(wl, limbs.data(), limbs.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(bias, 0, 2 * 256 * 4);
    Gru64Args a{}; a.x = x; a.out = out; a.wfrag = wf; a.bias = bias; a.hstate = nullptr;
    a.nrows = rows; a.nsteps = Fp; a.ndirs = 2; a.rdiv = 1; a.x_hi = Fp * 64; a.x_lo = 0; a.x_step = 64;
    a.o_hi = Fp * 128; a.o_lo = 0; a.o_step = 128; a.o_dir_off = 64;
    Gru64Args b = a; b.out = out2;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 10; float ms; const double flops = (double)rows * Fp * 2 * 49152.0;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e0);
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(gru64_scan_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("fp32 MFMA scan      : %.3f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(gru64_scan_bf16x3_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, b, (const __bf16*)wl);
    (void)hipEventRecord(e0);
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(gru64_scan_bf16x3_kernel, dim3(rows / 16, 2), dim3(256), 0, 0, b, (const __bf16*)wl);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("bf16x3 split scan   : %.3f ms  %.1f fp32-equivalent TFLOP/s (hipGetLastError: %s)\n", ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    std::vector<float> r1(no), r2(no);
    (void)hipMemcpy(r1.data(), out, no * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(r2.data(), out2, no * 4, hipMemcpyDeviceToHost);
    double maxd = 0, sum2 = 0, ref2 = 0;
    for (size_t i = 0; i < no; ++i) { double d = (double)r1[i] - r2[i]; if (fabs(d) > maxd) maxd = fabs(d); sum2 += d * d; ref2 += (double)r1[i] * r1[i]; }
    printf("deviation from the fp32 kernel: max abs %.3e, rms %.3e (signal rms %.3e)\n", maxd, sqrt(sum2 / no), sqrt(ref2 / no));
    return 0;
}
