// mfma4_probe.hip -- layout and rate of v_mfma_f32_4x4x1_16b_f32 with the A-broadcast modifiers (CBSZ / ABID) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4_probe tools/mfma4_probe.hip && /tmp/mfma4_probe
// Question behind it: a latency-bound GRU step on few rows pays for 16 rows with the 16x16x4 shape; the 4x4x1 x 16-block
// shape with A broadcast from one block computes 4 rows x 64 columns x 1 k per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void layout_kernel(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

__global__ void rate_kernel(float* out, int iters, long long* cyc) {
    const int l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 3, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 5, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 4, 7, 0);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2];
}
__global__ void rate_dep_kernel(float* out, int iters, long long* cyc) {
    const int l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 48; ++k) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 9, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
}

template <int CBSZ, int ABID>
static void run_layout() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = (float)(l + 1); b[l] = (float)(100 * (l + 1)); }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(layout_kernel<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    printf("CBSZ=%d ABID=%d: D[lane][i] = a_lane_src * b_lane_src/100 -> (a_src, b_src)\n", CBSZ, ABID);
    for (int l = 0; l < 64; ++l) {
        printf(" lane %2d:", l);
        for (int i = 0; i < 4; ++i) {
            // d = a[x] * b[y] with a = x+1, b = 100 (y+1): find (x, y)
            int fx = -1, fy = -1;
            for (int x = 0; x < 64 && fx < 0; ++x) for (int y = 0; y < 64; ++y) if (d[l * 4 + i] == a[x] * b[y]) { fx = x; fy = y; break; }
            printf(" i%d=(a%2d,b%2d)", i, fx, fy);
        }
        printf("\n");
        if (l == 7) { l = 55; printf(" ...\n"); }
    }
    hipFree(da); hipFree(db); hipFree(dd);
}

int main() {
    run_layout<0, 0>();
    run_layout<4, 0>();
    run_layout<4, 5>();
    run_layout<2, 1>();
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    long long h;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64 * waves), 0, 0, out, 1000, cyc);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("rate: %d waves/WG, 3 accumulators: %.2f clock64 ticks per MFMA per wave (48000 MFMAs)\n", waves, (double)h / 48000.0);
    }
    hipLaunchKernelGGL(rate_dep_kernel, dim3(1), dim3(64), 0, 0, out, 1000, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("rate: 1 wave, ONE accumulator (dependent chain): %.2f ticks per MFMA\n", (double)h / 48000.0);
    // clock64 tick rate: compare against wall clock
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, out, 100000, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("clock64: %.1f ticks per us; 1 wave 3 acc: %.2f ns per MFMA\n", (double)h / (ms * 1e3), ms * 1e6 / 4.8e6);
    return 0;
}
