// handoff_probe.hip -- what does a dependent step cost on this chip, as a kernel boundary and inside one launch?
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe tools/handoff_probe.hip && ./handoff_probe
// (a) chain of N dependent tiny kernels on one stream (each reads the previous one's output): us per boundary, for 1 / 32 / 192 workgroups
// (b) ONE launch of G co-resident workgroups doing N phases separated by a grid barrier (arrival counter in global memory, relaxed
//     agent-scope atomics, the payload written with plain stores + __threadfence() = release / acquire at agent scope): us per phase
// (c) the same with the payload published as {epoch, value} granules (sc1 write-through stores, relaxed agent-scope polling loads: the
//     GRU-256 cluster scans' protocol) from every workgroup to its ring neighbour -- no fences, no counter: us per phase
// (d) see flagged_kernel.
// The decision they inform (docs/HISTORY.md section 7): a persistent DPRNN-branch kernel replaces 2 nb kernel boundaries + entry phases
// by 2 nb in-kernel hand-offs of (b) or (c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny_kernel(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}

// (b) grid barrier: counter += 1 per workgroup and phase; wait until counter >= (phase + 1) * G
__global__ void barrier_kernel(float* buf, unsigned* counter, int phases, int n_per_wg, unsigned base) {
    const int G = gridDim.x, g = blockIdx.x;
    float* mine = buf + (size_t)g * n_per_wg;
    const float* peer = buf + (size_t)((g + 1) % G) * n_per_wg;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) mine[i] = acc + (float)p;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = base + (unsigned)(p + 1) * (unsigned)G;
            while ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();
        acc = peer[threadIdx.x % n_per_wg] * 0.5f;      // consume the neighbour's payload of this phase
    }
    if (threadIdx.x == 0) mine[0] = acc;
}

// (c) granule ring: every workgroup publishes n granules per phase, polls its neighbour's
__global__ void granule_kernel(unsigned long long* gbuf, float* out, int phases, int n_per_wg, unsigned base) {
    const int G = gridDim.x, g = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        unsigned long long* mine = gbuf + ((size_t)(p & 1) * G + g) * n_per_wg;
        const unsigned long long* peer = gbuf + ((size_t)(p & 1) * G + (g + 1) % G) * n_per_wg;
        const unsigned epoch = base + (unsigned)p + 1u;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x)
            __hip_atomic_store(mine + i, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(acc + (float)i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float got = 0.f;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) {
            unsigned long long v;
            do { v = __hip_atomic_load(peer + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((int)((unsigned)(v >> 32) - epoch) < 0);   // (a ring neighbour may be a phase ahead: accept newer)
            got += __uint_as_float((unsigned)v);
        }
        acc = got * 1e-3f;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[g] = acc;
}

// (d) payload as plain agent-scope (write-through) stores, ONE {epoch} flag granule per workgroup behind them (stores drained: the
//     workgroup barrier's release waits for vmcnt(0)); the consumer polls the flag, then reads the payload with agent-scope loads
__global__ void flagged_kernel(float* buf, unsigned long long* flags, float* out, int phases, int n_per_wg, unsigned base) {
    const int G = gridDim.x, g = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* mine = buf + ((size_t)(p & 1) * G + g) * n_per_wg;
        const float* peer = buf + ((size_t)(p & 1) * G + (g + 1) % G) * n_per_wg;
        const unsigned epoch = base + (unsigned)p + 1u;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x)
            __hip_atomic_store(mine + i, acc + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                 // every store of this wave has been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + g, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            while ((int)((unsigned)__hip_atomic_load(flags + (g + 1) % G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) ;
        }
        __syncthreads();
        float got = 0.f;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) got += __hip_atomic_load(peer + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc = got * 1e-3f;
    }
    if (threadIdx.x == 0) out[g] = acc;
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int N = 400;
    float *a, *b; CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22)); CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int wgs : {1, 32, 192}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(st));
            double t0 = now();
            for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny_kernel, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, wgs * 256); }
            CK(hipStreamSynchronize(st));
            if (rep) printf("(a) kernel boundary, %3d workgroups of 256 threads: %.2f us per dependent launch\n", wgs, (now() - t0) / N);
        }
    }
    unsigned* counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    unsigned long long* gbuf; CK(hipMalloc(&gbuf, (size_t)2 * 256 * 4096 * 8)); CK(hipMemset(gbuf, 0, (size_t)2 * 256 * 4096 * 8));
    unsigned base = 0, ebase = 0;
    for (int G : {2, 32, 192, 224}) {
        for (int npw : {256, 4096}) {           // payload floats per workgroup and phase (1 KB / 16 KB)
            const int P = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(counter, 0, 4)); base = 0;
                CK(hipDeviceSynchronize());
                double t0 = now();
                hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, st, a, counter, P, npw, base);
                CK(hipStreamSynchronize(st));
                double dt = now() - t0;
                if (rep) printf("(b) grid barrier (counter + fences), %3d workgroups, %5d floats each: %.2f us per phase\n", G, npw, dt / P);
            }
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                hipLaunchKernelGGL(granule_kernel, dim3(G), dim3(256), 0, st, gbuf, b, P, npw, ebase);
                CK(hipStreamSynchronize(st));
                double dt = now() - t0;
                ebase += P;
                if (rep) printf("(c) granule ring (sc1 data = flag),   %3d workgroups, %5d granules each: %.2f us per phase\n", G, npw, dt / P);
            }
        }
    }
    unsigned long long* flags; CK(hipMalloc(&flags, 256 * 8)); CK(hipMemset(flags, 0, 256 * 8));
    float* pay; CK(hipMalloc(&pay, (size_t)2 * 256 * 32768 * 4));
    unsigned fbase = 0;
    for (int G : {2, 32, 192, 224}) {
        for (int npw : {256, 4096, 16384}) {    // 1 KB / 16 KB / 64 KB per workgroup and phase
            const int P = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                hipLaunchKernelGGL(flagged_kernel, dim3(G), dim3(256), 0, st, pay, flags, b, P, npw, fbase);
                CK(hipStreamSynchronize(st));
                double dt = now() - t0;
                fbase += P;
                if (rep) printf("(d) write-through payload + one flag,  %3d workgroups, %5d floats each: %.2f us per phase\n", G, npw, dt / P);
            }
        }
    }
    return 0;
}
