// Is a 128-byte line that two workgroups on DIFFERENT XCDs write piecewise (64 bytes each) seen whole by the next kernel of the stream?
// (tools only)  writer: workgroups 2p and 2p + 1 (consecutive ids: different XCDs) write the two halves of every line of pair p with the
// launch's epoch; reader (next launch, same stream): workgroup r reads whole lines with plain loads and counts values != epoch.
// MODE of the reader's line -> workgroup map: 0 = same as writer 2p (same XCD as one writer), 1 = shifted by 3 workgroups (another XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void writer(float* buf, int nlines, int npairs, float epoch, int piece) {
    const int p = blockIdx.x >> 1, half = blockIdx.x & 1;
    for (int l = p; l < nlines; l += npairs)
        for (int i = threadIdx.x; i < piece; i += blockDim.x) {
            // `piece` floats per workgroup and line: 16 = two 64-byte halves; 60 = 240-byte pieces straddling lines like df_out's groups
            buf[(size_t)l * 2 * piece + half * piece + i] = epoch;
        }
}
__global__ void reader(const float* buf, int nlines, int nwg, float epoch, int piece, int shift, unsigned* bad, int agent) {
    const int r = (blockIdx.x + shift) % nwg;
    unsigned n = 0;
    for (int l = r; l < nlines; l += nwg)
        for (int i = threadIdx.x; i < 2 * piece; i += blockDim.x) {
            const float* p = buf + (size_t)l * 2 * piece + i;
            const float v = agent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
            n += v != epoch;
        }
    if (n) atomicAdd(bad, n);
}
// piece = 60 only: records of 10 floats read as dwordx4 + dwordx4 + dwordx2 at 8-byte alignment (what the compiler made of df_apply's taps)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void reader_vec(const float* buf, int nlines, int nwg, float epoch, int shift, unsigned* bad) {
    const int r = (blockIdx.x + shift) % nwg;
    unsigned n = 0;
    for (int l = r; l < nlines; l += nwg)
        for (int i = threadIdx.x; i < 12; i += blockDim.x) {            // 120 floats per "line" = 12 records
            const float* c = buf + (size_t)l * 120 + i * 10;
            f4v a, b; float2 d;
            asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx2 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(d) : "v"(c) : "memory");
            n += (a[0] != epoch) + (a[1] != epoch) + (a[2] != epoch) + (a[3] != epoch) + (b[0] != epoch) + (b[1] != epoch) + (b[2] != epoch) + (b[3] != epoch) + (d.x != epoch) + (d.y != epoch);
        }
    if (n) atomicAdd(bad, n);
}
__global__ void thrash(float* junk, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) junk[i] = junk[i] * 1.0001f + 1.f;
}
int main() {
    const int nlines = 1 << 16, npairs = 512;
    float* buf; unsigned* bad; float* junk;
    const size_t nj = (size_t)64 << 20;
    (void)hipMalloc(&buf, (size_t)nlines * 2 * 64 * 4); (void)hipMalloc(&bad, 4); (void)hipMalloc(&junk, nj * 4);
    (void)hipMemset(junk, 0, nj * 4);
    hipStream_t s, s2; (void)hipStreamCreate(&s); (void)hipStreamCreate(&s2);
    for (int piece : {16, 60})
        for (int shift : {0, 3})
            for (int agent : {0, 1})
                for (int busy : {0, 1}) {
                    unsigned total = 0; int bad_epochs = 0;
                    (void)hipMemset(buf, 0, (size_t)nlines * 2 * 64 * 4);
                    for (int e = 1; e <= 300; ++e) {
                        (void)hipMemsetAsync(bad, 0, 4, s);
                        if (busy) hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, s2, junk, nj);      // a co-tenant streaming through every L2
                        hipLaunchKernelGGL(writer, dim3(2 * npairs), dim3(64), 0, s, buf, nlines, npairs, (float)e, piece);
                        hipLaunchKernelGGL(reader, dim3(2 * npairs), dim3(64), 0, s, buf, nlines, 2 * npairs, (float)e, piece, shift, bad, agent);
                        unsigned h; (void)hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
                        total += h; bad_epochs += h != 0;
                    }
                    (void)hipDeviceSynchronize();
                    printf("piece %2d floats, reader shift %d, %s loads, co-tenant %d: %u stale values in %d of 300 epochs\n", piece, shift, agent ? "agent-scope" : "plain      ", busy, total, bad_epochs);
                }
    for (int shift : {0, 3})
        for (int busy : {0, 1}) {
            unsigned total = 0; int bad_epochs = 0;
            for (int e = 1; e <= 300; ++e) {
                (void)hipMemsetAsync(bad, 0, 4, s);
                if (busy) hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, s2, junk, nj);
                hipLaunchKernelGGL(writer, dim3(2 * npairs), dim3(64), 0, s, buf, nlines, npairs, (float)(1000 + e), 60);
                hipLaunchKernelGGL(reader_vec, dim3(2 * npairs), dim3(64), 0, s, buf, nlines, 2 * npairs, (float)(1000 + e), shift, bad);
                unsigned h; (void)hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
                total += h; bad_epochs += h != 0;
            }
            (void)hipDeviceSynchronize();
            printf("piece 60 floats, reader shift %d, under-aligned dwordx4 loads, co-tenant %d: %u stale values in %d of 300 epochs\n", shift, busy, total, bad_epochs);
        }
    return 0;
}
