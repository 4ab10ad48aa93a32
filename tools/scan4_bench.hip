// scan4_bench.hip -- gru64_scan4_gi_kernel alone: time per dependent step (long scan) and per launch (48 steps), random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dpdfnet_amd/csrc -o /tmp/scan4_bench tools/scan4_bench.hip && /tmp/scan4_bench
// -I picks the header under test (an older gru_scan4.h for an A/B); -DDPDF_SCAN4_... variants pass through.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gru_scan4.h"
__global__ __launch_bounds__(256) void scan4_handoff_kernel(Gru64Args a, const float* wfrag4, const float* gi, int gw) {
    gru64_scan4_body<true>(a, wfrag4, gi, gw, blockIdx.x, blockIdx.y);      // h' through agent-scope (write-through) stores, as in dprnn_hop_block.h
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64;
    for (int nsteps : {48, 480}) {
        const size_t M = (size_t)rows * nsteps;
        std::vector<float> w(2 * 4 * 64 * 64), b(2 * 256), gi(M * 384);
        srand(1);
        for (auto& v : w) v = 0.05f * (rand() / (float)RAND_MAX - 0.5f);
        for (auto& v : b) v = 0.1f * (rand() / (float)RAND_MAX - 0.5f);
        for (auto& v : gi) v = (rand() / (float)RAND_MAX - 0.5f);
        float *dw, *db, *dgi, *dout;
        CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dgi, gi.size() * 4)); CK(hipMalloc(&dout, M * 128 * 4));
        CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dgi, gi.data(), gi.size() * 4, hipMemcpyHostToDevice));
        Gru64Args a{};
        a.x = nullptr; a.out = dout; a.wfrag = nullptr; a.bias = db; a.hstate = nullptr;
        a.nrows = rows; a.nsteps = nsteps; a.ndirs = 2; a.rdiv = 1;
        a.x_hi = (long)nsteps * 64; a.x_lo = 0; a.x_step = 64;
        a.o_hi = (long)nsteps * 128; a.o_lo = 0; a.o_step = 128; a.o_dir_off = 64;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 200;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gru64_scan4_gi_kernel, dim3((rows + 3) / 4, 2), dim3(256), 0, 0, a, dw, (const float*)dgi, 384);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gru64_scan4_gi_kernel, dim3((rows + 3) / 4, 2), dim3(256), 0, 0, a, dw, (const float*)dgi, 384);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> out(M * 128);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double cs = 0; for (float v : out) cs += v;
        printf("rows %d nsteps %d: %.2f us per launch, %.3f us per step incl. launch; checksum %.6f\n", rows, nsteps, 1e3 * ms / reps, 1e3 * ms / reps / nsteps, cs);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(scan4_handoff_kernel, dim3((rows + 3) / 4, 2), dim3(256), 0, 0, a, dw, (const float*)dgi, 384);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scan4_handoff_kernel, dim3((rows + 3) / 4, 2), dim3(256), 0, 0, a, dw, (const float*)dgi, 384);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   with write-through stores: %.2f us per launch, %.3f us per step incl. launch\n", 1e3 * ms / reps, 1e3 * ms / reps / nsteps);
        (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dgi); (void)hipFree(dout);
    }
    return 0;
}
