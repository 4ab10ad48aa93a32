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
int main() {
    const size_t nrec = (size_t)256 * 66 * 96, n = nrec * 10;          // the taps of 256 clips x 66 ring rows x 96 bins
    float* buf; unsigned* bad; float* junk; const size_t nj = (size_t)256 << 20;
    (void)hipMalloc(&buf, n * 4 + 64); (void)hipMalloc(&bad, 4); (void)hipMalloc(&junk, nj * 4); (void)hipMemset(junk, 0, nj * 4);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, buf, n);
    (void)hipDeviceSynchronize();
    hipStream_t s, s2; (void)hipStreamCreate(&s); (void)hipStreamCreate(&s2);
    for (int mode = 0; mode < 2; ++mode)
        for (int busy = 0; busy < 2; ++busy) {
            unsigned total = 0; int bad_runs = 0;
            for (int e = 0; e < 400; ++e) {
                (void)hipMemsetAsync(bad, 0, 4, s);
                if (busy) hipLaunchKernelGGL(thrash, dim3(4096), dim3(256), 0, s2, junk, nj);
                if (mode == 0) hipLaunchKernelGGL(reader<0>, dim3(6336), dim3(256), 0, s, buf, nrec, bad);
                else hipLaunchKernelGGL(reader<1>, dim3(6336), dim3(256), 0, s, buf, nrec, bad);
                unsigned h; (void)hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
                total += h; bad_runs += h != 0;
            }
            (void)hipDeviceSynchronize();
            printf("%s, co-tenant %d: %u wrong values in %d of 400 runs\n", mode == 0 ? "misaligned dwordx4 + dwordx4 + dwordx2" : "ten dword loads                       ", busy, total, bad_runs);
        }
    return 0;
}
