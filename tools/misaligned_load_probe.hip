// Do under-aligned global_load_dwordx4 (8-byte aligned, 40-byte stride: the deep-filter taps as df_apply_kernel read them) always return
// the bytes that are in memory?  The buffer is written once, never again; readers compare what they load with the known pattern while a
// co-tenant streams through the caches on another stream.  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void fill(float* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (float)(i & 0xfffff);
}
template <int MODE>   // 0: two misaligned dwordx4 + one dwordx2 (what the compiler made of ten scalar loads), 1: ten dword loads
__global__ void reader(const float* buf, size_t nrec, unsigned* bad) {
    unsigned n = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        const float* c = buf + r * 10;                 // 40-byte records: 8-byte aligned
        float v[10];
        if (MODE == 0) {
            f4 a, b; float2 d;
            asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx2 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(d) : "v"(c) : "memory");
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3]; v[8] = d.x; v[9] = d.y;
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) v[k] = *(volatile const float*)(c + k);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) n += v[k] != (float)((r * 10 + k) & 0xfffff);
    }
    if (n) atomicAdd(bad, n);
}
__global__ void thrash(float* junk, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) junk[i] = junk[i] * 1.0001f + 1.f;
}
// a co-tenant shaped like the GRU-256 cluster exchange: 64 workgroups in clusters of four, each step every workgroup publishes 8-byte
// {epoch, value} granules with agent-scope stores and sweeps its three peers' granules with agent-scope loads until they carry the epoch
__global__ __launch_bounds__(256, 1) void exchange(unsigned long long* xb, int steps) {
    __shared__ float pad[30000];
    pad[threadIdx.x] = 0.f;
    const int tile = blockIdx.x >> 2, j = blockIdx.x & 3;
    unsigned long long* base = xb + (size_t)tile * 2 * 4 * 1024;
    for (int t = 0; t < steps; ++t) {
        unsigned long long* slot = base + (size_t)(t & 1) * 4 * 1024;
        const unsigned epoch = (unsigned)t + 1u;
        for (int i = threadIdx.x; i < 1024; i += 256)
            __hip_atomic_store(slot + j * 1024 + i, ((unsigned long long)epoch << 32) | (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int s = 1; s < 4; ++s) {
            const int peer = (j + s) & 3;
            for (int i = threadIdx.x; i < 1024; i += 256) {
                unsigned spins = 0;
                while ((unsigned)(__hip_atomic_load(slot + peer * 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != epoch && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (pad[threadIdx.x] == 1.f) xb[0] = 0;
}
int main() {
    const size_t nrec = (size_t)256 * 66 * 96, n = nrec * 10;          // the taps of 256 clips x 66 ring rows x 96 bins
    float* buf; unsigned* bad; float* junk; const size_t nj = (size_t)256 << 20;
    (void)hipMalloc(&buf, n * 4 + 64); (void)hipMalloc(&bad, 4); (void)hipMalloc(&junk, nj * 4); (void)hipMemset(junk, 0, nj * 4);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, buf, n);
    (void)hipDeviceSynchronize();
    hipStream_t s, s2; (void)hipStreamCreate(&s); (void)hipStreamCreate(&s2);
    unsigned long long* xb; (void)hipMalloc(&xb, (size_t)16 * 2 * 4 * 1024 * 8); (void)hipMemset(xb, 0, (size_t)16 * 2 * 4 * 1024 * 8);
    hipStream_t s3; (void)hipStreamCreate(&s3);
    for (int mode = 0; mode < 2; ++mode)
        for (int busy = 0; busy < 4; ++busy) {
            unsigned total = 0; int bad_runs = 0;
            for (int e = 0; e < 400; ++e) {
                (void)hipMemsetAsync(bad, 0, 4, s);
                if (busy & 1) hipLaunchKernelGGL(thrash, dim3(4096), dim3(256), 0, s2, junk, nj);
                if (busy & 2) { (void)hipMemsetAsync(xb, 0, (size_t)16 * 2 * 4 * 1024 * 8, s3); hipLaunchKernelGGL(exchange, dim3(64), dim3(256), 0, s3, xb, 60); }
                if (mode == 0) hipLaunchKernelGGL(reader<0>, dim3(6336), dim3(256), 0, s, buf, nrec, bad);
                else hipLaunchKernelGGL(reader<1>, dim3(6336), dim3(256), 0, s, buf, nrec, bad);
                unsigned h; (void)hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
                total += h; bad_runs += h != 0;
            }
            (void)hipDeviceSynchronize();
            printf("%s, co-tenant %d (1 = streaming, 2 = cluster-style exchange, 3 = both): %u wrong values in %d of 400 runs\n", mode == 0 ? "misaligned dwordx4 + dwordx4 + dwordx2" : "ten dword loads                       ", busy, total, bad_runs);
        }
    return 0;
}
