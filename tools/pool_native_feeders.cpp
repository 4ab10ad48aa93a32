// pool_native_feeders.cpp -- measurement helper (not part of the product library): NATIVE host threads feeding the library's stream
// pool (dpdf_streams_submit_block, include/dpdfnet_hip.h), i.e. the reference's pattern of N independent StreamEnhancer callers
// (package/src/dpdfnet/stream.py:13-72) without an interpreter lock between them.  bench.py calls it beside the same pattern driven
// by Python threads, so that the line shows what the pool itself costs and what CPython's GIL hand-offs add.
// build: g++ -O2 -std=c++17 -shared -fPIC -pthread tools/pool_native_feeders.cpp -o tools/libpool_native_feeders.so
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

typedef int (*submit_block_fn)(void* s, int n, const int* slots, const float* in_block, int k_hops, float* out_block, int flags);

extern "C" int pool_native_feeders(void* streams, void* submit_block, int S, int nthreads, int rounds, int hop, const float* pcm /* [S][hop] */,
                                   float* last_out /* [S][hop] or NULL */, double* us_per_round) {
    submit_block_fn submit = (submit_block_fn)submit_block;
    std::atomic<int> ready{0}, go{0}, failed{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            std::vector<int> slots;
            for (int i = t; i < S; i += nthreads) slots.push_back(i);
            const int n = (int)slots.size();
            std::vector<float> in((size_t)n * hop), out((size_t)n * hop);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            for (int r = 0; r < rounds; ++r) {
                for (int i = 0; i < n; ++i) memcpy(&in[(size_t)i * hop], pcm + (size_t)slots[i] * hop, hop * sizeof(float));   // "the chunks arrive"
                if (submit(streams, n, slots.data(), in.data(), 1, out.data(), 0)) { failed.fetch_add(1); break; }
            }
            if (last_out) for (int i = 0; i < n; ++i) memcpy(last_out + (size_t)slots[i] * hop, &out[(size_t)i * hop], hop * sizeof(float));
        });
    while (ready.load() < nthreads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1);
    for (auto& x : th) x.join();
    *us_per_round = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / rounds;
    return failed.load();
}

// The lock-step hop from a NATIVE loop: `calls` consecutive dpdf_streams_process(streams, pcm, 1, out, flags) of ONE host thread, timed here
// -- what a C / C++ host pays per hop (bench.py's `us_per_call` drives the same entry point from a Python loop: + the interpreter's ~20 us).
typedef int (*process_fn)(void* s, const float* pcm_in, int n_hops, float* pcm_out, int flags);
extern "C" int native_hop_loop(void* streams, void* process, int calls, int warm, const float* pcm, float* out, int flags, double* us_per_call) {
    process_fn f = (process_fn)process;
    for (int i = 0; i < warm; ++i) if (f(streams, pcm, 1, out, flags)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; ++i) if (f(streams, pcm, 1, out, flags)) return 1;
    *us_per_call = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / calls;
    return 0;
}
