// dpdf_model.hip -- C ABI + host orchestration of the MI355X DPDFNet engine.
//
// Schedule: "layer at a time over every frame of every clip" instead of the reference's
// "every layer for one frame" (reference package/src/dpdfnet/api.py:96-104): each layer is ONE
// kernel over B*Tc frames; recurrences over time (EMA norms, inter-band GRUs, the 256-wide GRU
// stacks) are scans inside persistent workgroups; FIFO buffers (CyclicBuffer,
// onnx_model/layers.py:68-107) become time-halo index shifts.  SURVEY.md appendix A.3 gives the
// time indexing; the reference's offline twin (model/dpdfnet.py) proves the equivalence.
// The device state between chunks/calls is kept in the reference's own flat layout, so
// `dpdf_run_frames` is a drop-in for T consecutive session.run calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <sys/mman.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23      // Linux >= 5.14: prefault writable, contents untouched
#endif

#include "../../include/dpdfnet_hip.h"
#include "../../include/dpdf_norm_init.h"
#include "common.h"
#include "gemm_rows.h"
#include "gru_scan.h"
#include "gru_limb.h"
#include "misc_kernels.h"
#include "resample.h"
#include "df_ring.h"
#include "dec_last.h"
#include "dec_seg2.h"
#include "gru_stack.h"
#include "fcln_gi.h"
#include "gru_scan4.h"
#include "dprnn_hop_block.h"
#include "dprnn_hop_stack.h"
#include "small_fused_mfma.h"
#include "enc_seg.h"
#include "dec_pyr.h"
#include "dft2stage.h"
#include "gru_clusterx.h"
#include "dft64.h"

// ------------------------------------------------------------------------------------------------
// HIP multiplexes every stream of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4); the engine runs four
// streams per handle concurrently (stage 1, ERB branch, stage 2, DF decoder), so with the default a second handle --
// even an idle one -- or any other stream user makes two of them share a queue and serialise: one 10 s clip 11.4 ms
// alone, 16.0 ms beside a second handle; with 8 queues 11.4 ms in both cases (tools/clock_probe.py).  The variable is
// read when the HIP runtime initialises, which a library cannot influence once its host has touched the GPU: the
// LIBRARY does not set it.  The Python package (dpdfnet_amd/__init__.py) and bench.py set a default of 8 before HIP
// comes up and say so when they are too late; C hosts set GPU_MAX_HW_QUEUES=8 themselves.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return set_err(DPDF_E_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

extern "C" const char* dpdf_last_error(void) { return g_err; }
extern "C" int dpdf_abi_version(void) { return 1; }
extern "C" int dpdf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------
// N3: device resampler (model-independent; one tap table per (device, up, down), one stream per device)
// ------------------------------------------------------------------------------------------------
namespace {
struct ResampleDev { hipStream_t stream = nullptr; std::map<std::pair<int, int>, std::pair<float*, ResampleDesign>> taps;
                     float* in = nullptr; float* out = nullptr; size_t in_cap = 0, out_cap = 0; };
std::mutex g_rs_mu;
std::map<int, ResampleDev> g_rs;
}
extern "C" long dpdf_resample_len(long n_in, int sr_in, int sr_out) {
    if (n_in < 0 || sr_in <= 0 || sr_out <= 0) return -1;
    long g = resample_gcd(sr_in, sr_out);
    return resample_out_len(n_in, (int)(sr_out / g), (int)(sr_in / g));
}
extern "C" int dpdf_resample(int device, const float* in, int B, long n_in, int sr_in, int sr_out, float* out, int flags) {
    if (!in || !out) return set_err(DPDF_E_INVALID, "null argument");
    if (B <= 0 || n_in < 0 || sr_in <= 0 || sr_out <= 0) return set_err(DPDF_E_INVALID, "bad resample geometry B=%d n=%ld %d->%d", B, n_in, sr_in, sr_out);
    if (n_in == 0) return DPDF_OK;
    std::lock_guard<std::mutex> lk(g_rs_mu);
    HIP_TRY(hipSetDevice(device));
    ResampleDev& R = g_rs[device];
    if (!R.stream) HIP_TRY(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
    long g = resample_gcd(sr_in, sr_out);
    const std::pair<int, int> key((int)(sr_out / g), (int)(sr_in / g));
    auto it = R.taps.find(key);
    if (it == R.taps.end()) {
        ResampleDesign d = design_resampler(sr_in, sr_out);
        std::vector<float> hf(d.hp.begin(), d.hp.end());
        float* dp = nullptr;
        HIP_TRY(hipMalloc((void**)&dp, hf.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(dp, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice));
        it = R.taps.emplace(key, std::make_pair(dp, std::move(d))).first;
    }
    const ResampleDesign& d = it->second.second;
    const long n_out = resample_out_len(n_in, d.up, d.down);
    const bool host = !(flags & DPDF_DEVICE_PTRS);
    const float* d_in = in; float* d_out = out;
    if (host) {
        const size_t ni = (size_t)B * n_in, no = (size_t)B * n_out;
        if (ni > R.in_cap) { if (R.in) (void)hipFree(R.in); R.in = nullptr; R.in_cap = 0; HIP_TRY(hipMalloc((void**)&R.in, ni * sizeof(float))); R.in_cap = ni; }
        if (no > R.out_cap) { if (R.out) (void)hipFree(R.out); R.out = nullptr; R.out_cap = 0; HIP_TRY(hipMalloc((void**)&R.out, no * sizeof(float))); R.out_cap = no; }
        HIP_TRY(hipMemcpyAsync(R.in, in, ni * sizeof(float), hipMemcpyHostToDevice, R.stream));
        d_in = R.in; d_out = R.out;
    }
    ResampleArgs a{d_in, d_out, it->second.first, n_in, n_out, d.pre, d.up, d.down, (int)d.hp.size()};
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)B), dim3(256), 0, R.stream, a);
    HIP_TRY(hipGetLastError());
    if (host) HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)B * n_out * sizeof(float), hipMemcpyDeviceToHost, R.stream));
    HIP_TRY(hipStreamSynchronize(R.stream));
    return DPDF_OK;
}

extern "C" size_t dpdf_weight_count(const dpdf_cfg* cfg) { return dpdf_manifest(cfg, nullptr, nullptr); }
extern "C" int dpdf_query_dims(const dpdf_cfg* cfg, dpdf_dims* out) {
    if (!out || dpdf_get_dims(cfg, out) != 0) return set_err(DPDF_E_INVALID, "unsupported model config");
    return DPDF_OK;
}
namespace {
struct TxtCtx { char* buf; size_t cap, len; };
void text_cb(void* ud, const char* name, const int* shape, int ndim, size_t off, size_t cnt) {
    TxtCtx* t = (TxtCtx*)ud;
    char line[256];
    int k = snprintf(line, sizeof(line), "%s %zu %zu ", name, off, cnt);
    for (int i = 0; i < ndim; ++i) k += snprintf(line + k, sizeof(line) - k, i ? ",%d" : "%d", shape[i]);
    k += snprintf(line + k, sizeof(line) - k, "\n");
    if (t->buf && t->len + k < t->cap) memcpy(t->buf + t->len, line, (size_t)k);
    t->len += (size_t)k;
}
}  // namespace
extern "C" size_t dpdf_manifest_text(const dpdf_cfg* cfg, char* buf, size_t cap) {
    TxtCtx t{buf, cap, 0};
    dpdf_manifest(cfg, text_cb, &t);
    if (buf && t.len < cap) buf[t.len] = 0;
    return t.len;
}

// ------------------------------------------------------------------------------------------------
// host-side weight preparation
// ------------------------------------------------------------------------------------------------
namespace {

struct Blob {
    const float* base;
    std::map<std::string, size_t> off;
    const float* get(const std::string& n) const {
        auto it = off.find(n);
        if (it == off.end()) { fprintf(stderr, "dpdfnet_hip: missing tensor %s\n", n.c_str()); abort(); }
        return base + it->second;
    }
};
void blob_cb(void* ud, const char* name, const int*, int, size_t off, size_t) { ((Blob*)ud)->off[name] = off; }

struct Arena {                     // one device allocation for every prepared constant
    std::vector<float> h;
    size_t add(const std::vector<float>& v) {
        size_t o = (h.size() + 63) & ~size_t(63);
        h.resize(o + v.size());
        std::copy(v.begin(), v.end(), h.begin() + o);
        return o;
    }
};

// pack W (math orientation out = A . W, W[k][n] given by accessor) into MFMA B-fragment order
// [chunk][tile][kb][lane] (K padded to 16, N to NT*16)
template <class Fn>
std::vector<float> pack_frag(int K, int N, int NT, Fn w) {
    const int nch = (K + 15) / 16;
    std::vector<float> out((size_t)nch * NT * 256, 0.f);
    for (int c = 0; c < nch; ++c)
        for (int nt = 0; nt < NT; ++nt)
            for (int kb = 0; kb < 4; ++kb)
                for (int lane = 0; lane < 64; ++lane) {
                    int k = kperm(c, lane >> 4, kb), n = nt * 16 + (lane & 15);
                    if (k < K && n < N) out[(((size_t)c * NT + nt) * 4 + kb) * 64 + lane] = w(k, n);
                }
    return out;
}

struct BnFold { std::vector<float> scale, shift; };
BnFold fold_bn(const Blob& B, const std::string& p, int ch) {
    BnFold f; f.scale.resize(ch); f.shift.resize(ch);
    const float *w = B.get(p + ".weight"), *b = B.get(p + ".bias"), *m = B.get(p + ".running_mean"), *v = B.get(p + ".running_var");
    for (int c = 0; c < ch; ++c) {
        f.scale[c] = w[c] / std::sqrt(v[c] + 1e-5f);
        f.shift[c] = b[c] - m[c] * f.scale[c];
    }
    return f;
}

struct SepConvW { size_t dw, pwfrag, bias; int nsub; };        // arena offsets
struct PathW { size_t ps, pb; };
struct GruW64 { size_t wfrag, bias; int ndirs;
                size_t hh4;                   // W_hh for the 4-row scan (gru_scan4.h): [dir][wave 4][instruction 64][lane 4b+i]: gate i of unit 16 wave + b (i = 3: zero)
                size_t wl;                    // the same weights as bf16 limb fragments (gru_limb.h): [dir][wave 4][mat 6][chunk 2][limb 3][lane 64] x 8 bf16
                size_t ih_frag, ih_bias; };   // W_ih as a gemm_rows operand: [dir*3+gate][chunk][nt][kb][lane] + bias [dir*3+gate][64] (small-batch scan)
struct GlW { size_t frag, bias; int G, Og, Ig, NT; };
struct Gru256W { size_t ih_frag_s, ih_bias, hh_frag, b_hn, ih_as_hh; };   // ih_as_hh: W_ih packed like hh_frag (second cell of a stacked pair, gru_stack.h)   // ih_frag_s: the same W_ih in 24 column blocks of 32 (few-row launches)   // hh_frag: [unit-group 16][gate 3][chunk 16][kb 4][lane 64]
struct DprnnW { GruW64 intra, inter; size_t fci_frag, fci_b, lni_g, lni_b, fce_frag, fce_b, lne_g, lne_b;
                size_t fci_epi, fce_epi;      // fc fragments for the fused-epilogue scans: [part][wave][16][lane]
                size_t fci_lb, fci_lf, fce_l; };   // ... as bf16 limb fragments (gru_limb.h): fc_intra's hb half, its hf half, fc_inter: [wave 4][chunk 2][limb 3][lane 64] x 8 bf16

}  // namespace

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    float* p = nullptr; size_t n = 0;
    int ensure(size_t need) {
        if (need <= n) return DPDF_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
#ifdef DPDF_HAZARD_PROBE
        hipError_t e = uncached ? hipExtMallocWithFlags((void**)&p, need * sizeof(float), hipDeviceMallocUncached) : hipMalloc((void**)&p, need * sizeof(float));
#else
        hipError_t e = hipMalloc((void**)&p, need * sizeof(float));
#endif
        if (e != hipSuccess) return set_err(DPDF_E_RUNTIME, "hipMalloc(%zu floats) failed: %s", need, hipGetErrorString(e));
        n = need;
        return DPDF_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
#ifdef DPDF_HAZARD_PROBE
    bool uncached = false;
#endif
};

// Tensors that cross from stage 1 (features, encoder convs, DPRNNs) to stage 2 (embedding GRUs,
// decoders, mask, deep filter).  Double-buffered by chunk parity so that stage 2 of chunk i runs on
// its own HIP stream underneath stage 1 of chunk i+1.
struct XSet {
    DevBuf xs, e0, e1, e2, e3, xe_a, xe_b, c0, c1, xd_a, xd_b, pconv;
    const float* e3d = nullptr; const float* c1d = nullptr;
    bool have_pconv = false;          // stage 1 produced the DF pathway conv (df_ring_kernel) for this chunk
    void release() {
        DevBuf* all[] = {&xs, &e0, &e1, &e2, &e3, &xe_a, &xe_b, &c0, &c1, &xd_a, &xd_b, &pconv};
        for (DevBuf* b : all) b->release();
    }
};
constexpr int NRING = 2;            // stage-crossing tensors are double-buffered by chunk parity
struct Workspace {
    int Bcap = 0, Tcap = 0;
    XSet x[NRING];
    // stage-1 temporaries (DF branch on the main stream, ERB branch on its own stream)
    DevBuf feat_erb, feat_spec, hcat, hin, hcat_e, hin_e;
    DevBuf gi64, gi64_e;               // input-side GRU-64 pre-activations of the small-batch scans (grown on first use)
    // stage-2 temporaries
    DevBuf embin, g256a, g256b, g256c, gi, emb, demb, demb2, d3, d2, d1, m, dfo, coefs, xm;
    DevBuf g256d, g256e, g256f, gi2;   // DF-decoder chain's own scratch (runs beside the ERB decoder)
    DevBuf skipb;                      // df_skip(emb) of the fused small-launch form (emb_out_mfma_kernel)
    void release() {
        for (int k = 0; k < NRING; ++k) x[k].release();
        DevBuf* all[] = {&feat_erb, &feat_spec, &hcat, &hin, &hcat_e, &hin_e, &gi64, &gi64_e,
                         &embin, &g256a, &g256b, &g256c, &gi, &g256d, &g256e, &g256f, &gi2, &skipb, &emb, &demb, &demb2, &d3, &d2, &d1, &m, &dfo, &coefs, &xm};
        for (DevBuf* b : all) b->release();
        Bcap = Tcap = 0;
    }
};

// One independent execution lane: its own streams, events, workspace and GRU-256 exchange buffer.
// Clips are independent, so a batch is split over two lanes whose kernels the GPU interleaves:
// HBM-bound phases of one lane run under MFMA-bound scans of the other.
// Engine handles alive in this process.  HIP multiplexes the streams of ALL handles onto a few hardware queues, and a queue runs its
// kernels in order: a kernel that WAITS for a kernel of another stream (the counter join of a streaming hop, run_stage1) is only
// safe while no other handle's kernels can sit between the two in a shared queue -- with several handles alive it is not used.
static std::atomic<int> g_live_models{0};

struct Lane {
    hipStream_t sA = nullptr, sB = nullptr, sC = nullptr, sD = nullptr;   // sD: DF-decoder half of stage 2
    void sync_all() const {
        hipStream_t all[] = {sA, sB, sC, sD};
        for (hipStream_t st : all) if (st) (void)hipStreamSynchronize(st);
    }
    hipEvent_t ev_s1[NRING] = {}, ev_s2[NRING] = {}, ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr;
    hipEvent_t ev_fk[NRING] = {}, ev_jn[NRING] = {};   // ERB-branch fork/join, per chunk-ring slot
    hipEvent_t ev_dfk[NRING] = {}, ev_djn[NRING] = {}; // decoder fork/join inside stage 2
    bool s2_pending[NRING] = {};
    bool single_chunk = false;          // this call is one chunk: stage 2 on the main stream (run_stage2)
    bool s1_imported = false;           // stage 1's FIFO import of the coming one-frame chunk was done by the caller's prologue launch
    bool defer_export = false, export_pending = false; StateIoArgs pending_sio{}; int pending_B = 0;   // a one-chunk call whose caller launches the export later (join_export)
    hipEvent_t ev_x2 = nullptr; bool x2_pending = false;   // behind the stage-2 FIFO export of the latest chunk (joined at the END of a call: run_chunks / join_export)
    bool mask_from_sums = false;                       // this chunk's mask is still three tap sums per band in ws.d1 (run_dec_convs -> run_mask_df)
    Workspace ws;
    // GRU-256 cluster exchange granules: [0] embedding + ERB-decoder cells, [1] DF-decoder cells (they may run concurrently)
    unsigned long long* gru_xbuf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; int gru_xbuf_tiles[5] = {0, 0, 0, 0, 0}; unsigned gru_epoch[5] = {0, 0, 0, 0, 0};
    // stacked decoder pairs (gru256_stack16_kernel): per pair [tiles][Tcap + 2][16][256] granules = cell A's per-frame ring + cell B's two slots
    unsigned* join_ctr = nullptr; unsigned join_total = 0; bool join_want = false, join_armed = false;   // streaming hop: stage 2's first kernel waits for the ERB stack's last block by counter, not by event
    // dprnn_hop_stack_kernel (a whole stack as one persistent launch): per branch the scans' granules [2][M][128], the later blocks' input projections
    // [2][M][384], the glue tiles' flags [S][4] and the epoch of the next launch's first block
    unsigned long long* hs_hcat[2] = {nullptr, nullptr}; float* hs_gi[2] = {nullptr, nullptr}; unsigned* hs_flags[2] = {nullptr, nullptr};
    int hs_M[2] = {0, 0}, hs_S[2] = {0, 0}; unsigned hs_epoch[2] = {1, 1};
    unsigned* hop_flags[2] = {nullptr, nullptr}; int hop_flags_n[2] = {0, 0}; unsigned hop_epoch[2] = {0, 0};   // dprnn_hop_block_kernel: [0] DF stack, [1] ERB stack (they run side by side)
    unsigned* arrive[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; int arrive_tiles[5] = {0, 0, 0, 0, 0}; unsigned arrive_count[5] = {0, 0, 0, 0, 0};   // gru256_step_kernel
    unsigned long long* gru_sbuf[2] = {nullptr, nullptr}; int gru_sbuf_tiles[2] = {0, 0}, gru_sbuf_T[2] = {0, 0}; unsigned gru_sepoch[2] = {0, 0};
    const float* dbg_e3d = nullptr; const float* dbg_c1d = nullptr; const float* dbg_emb = nullptr; int dbg_B = 0, dbg_Tc = 0, dbg_parity = 0;
};

struct ProfEntry { double ms = 0; long calls = 0; };

// ---- host I/O pipeline of the batch entry points (enhance_host_pipelined) --------------------------------------------------
// A few worker threads that copy rows between the caller's (pageable) memory and pinned staging: a blocking parallel-for
// in which the calling thread takes part.  One 256-clip time slice is 31 MB each way; four threads move it in ~1 ms.
struct HostCopyPool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv, cv_done;
    const std::function<void(int)>* job = nullptr;
    int n_items = 0, busy = 0; std::atomic<int> next{0}; unsigned long gen = 0; bool stop = false;
    void worker() {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn; int n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; fn = job; n = n_items; ++busy;
            }
            // (a worker that wakes up only after its generation's run() has returned finds job == nullptr and must not touch the
            // counter, which may already belong to the next generation)
            if (fn) for (int i; (i = next.fetch_add(1)) < n;) (*fn)(i);
            { std::lock_guard<std::mutex> lk(mu); if (--busy == 0) cv_done.notify_all(); }
        }
    }
    void ensure(int n_threads) {
        while ((int)th.size() < n_threads - 1) th.emplace_back([this] { worker(); });
    }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        { std::lock_guard<std::mutex> lk(mu); job = &fn; n_items = n; next.store(0); ++gen; }
        cv.notify_all();
        for (int i; (i = next.fetch_add(1)) < n;) fn(i);
        std::unique_lock<std::mutex> lk(mu);
        // every worker that woke up for this generation has drained the counter; workers that have not woken up yet will find it drained
        cv_done.wait(lk, [&] { return busy == 0; });
        job = nullptr; n_items = 0;
    }
    ~HostCopyPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};
struct HostPipe {
    static constexpr int R = 3;        // ring depth: slice k+2 is staged and slice k-2 drained while chunk k is enqueued
    hipStream_t s_up = nullptr, s_down = nullptr;
    float* pin_in[R] = {}; float* pin_out[R] = {}; size_t cap_in = 0, cap_out = 0;     // floats per slot
    hipEvent_t ev_up[R] = {}, ev_down[R] = {}, ev_s2[R] = {};
    HostCopyPool pool;
};

struct dpdf_model {
    dpdf_cfg cfg; dpdf_dims d; dpdf_state_layout L;
    int device = 0;
    hipStream_t stream = nullptr;      // main stream (= lanes[0].sA): I/O, STFT/iSTFT, stage 1 of lane 0
    hipStream_t cur = nullptr;         // stream the helper launchers enqueue on
    Lane lanes[1]; Lane* ln = nullptr; // the execution lane: streams, events, workspace (one per handle)
    // bit 0: stage 2 on its own stream; bit 1: ERB encoder branch on its own stream; bit 3: DF decoder beside the ERB decoder
    // inside stage 2 (needs bit 0); bit 4: GRU-256 scans on 8 / 16 workgroups per tile for launches of few tiles.
    // 0 = everything serial on the main stream (A/B timing).  (Bits 2 and 5 -- two lanes, five-stream sub-stage pipeline --
    // were measured slower and removed: docs/HISTORY.md section 7; they are ignored.)
#ifdef DPDF_HAZARD_PROBE
    int probe_taps = -1, probe_wait = 0, probe_late = 0; bool probe_dump_on = false; unsigned* probe_dump = nullptr; int probe_dump_T = 0; size_t probe_dump_n = 0;
#endif
    int overlap = 27;
    int inter_fuse_rows = 1024;        // inter-band scan: fused form (fc + LN inside the scan) from this many (stream, band) rows on, hoisted-input form below
    int scan4_max_wgs = 512;           // hoisted-input GRU-64 scans on 4-row tiles (gru_scan4.h) while the launch has at most this many workgroups (0 = never)
    int hoist_gi = 1;                  // small chunks: input-side GRU-64 GEMM hoisted out of the scans
    int gru256_stack = 1;              // one or two tiles: the two cells of each decoder stack as one wavefront launch (gru_stack.h)
    int tail_frames = 32;              // throughput regime: frames of the short chunk split off a long last chunk (pipeline drain; 0 = off)
    int gru256_step = 1;               // single-hop streaming: input projection + GRUCell(256) step as one launch per cell
    bool counted = false;              // this handle is in g_live_models
    int hop_spin_join = 1;             // ... and stage 2's emb_in waits for the ERB stack's last block by a counter instead of a cross-stream event (~10 us)
    int n_cus = 256;                   // compute units of the device (co-residency checks of the persistent launches)
    int hop_stack = 0;                 // OPT-IN: a whole DPRNN stack of a hop as ONE persistent launch (dprnn_hop_stack.h; measured equal to the per-block launches)
    int hop_fused = 1;                 // ... and that glue in the SAME launch as the scan in front of it (dprnn_hop_block.h): one launch per block
    int hop_glue = 1;                  // single-hop streaming: one glue launch per DPRNN block between the intra-band scans (fcln_gi.h)
    int fcln_gi = 1;                   // small batches: fc + LayerNorm GEMMs of the DPRNN also produce the next recurrence's input projection (fcln_gi.h)
    int gru256_c8_tiles = 4;           // launches of <= this many tiles use eight workgroups per tile (gru256_cluster8_kernel)
    int gru256_c16_tiles = 2;          // launches of <= this many 16-row tiles use sixteen workgroups per tile (gru256_cluster16_kernel)
    int df_ring = 2;                   // big batches: 1 = df_conv1 + DF pathway conv as one time-walking pass over c0 (df_ring.h), 2 = df_conv0 in it too
    int dec_seg = 2;                   // 48 kHz decoder stages as band-segment tiles with inputs read once: 2 = tile-pipelined (dec_seg2.h), three launches; 3 = the same in ONE launch
                                       // (dec_seg2_all_kernel; A/B: holding every CU for the whole decoder costs stage 1 more than the two re-acquisitions cost stage 2: 131.0 -> 134.1 ms/step);
                                       // 1 = dec_last.h: dec_seg_kernel, 0 = gemm_rows producers
    int dec_seg_all_frames = 8192;     // ... one launch from this many frames per chunk on
    int dec_seg_grid = 256;            // dec_seg2 workgroups (512 threads, 110 / 149 KB of LDS: one per CU)
    int fuse_mask = 1;                 // mask head's 64->1 contraction in the convt1 epilogue (0: stand-alone mask_out_kernel, A/B)
    int fuse_dprnn = 1;                // fc + LayerNorm + residual fused into the GRU-64 scans: 0 never (separate GEMM kernels),
                                       // 1 auto (only when B*Tc fills the chip; measured crossover ~3k frame rows), 2 always
    std::mutex mu;
    float* consts = nullptr;           // device arena
    int* iconsts = nullptr;            // band_start[33] | band_of[F]
    std::vector<float> erb_norm_init, spec_norm_init;
    float* d_init_state = nullptr;     // [S]
    int chunk_frames = 0;
    int* d_err = nullptr;
    int* d_lens = nullptr; size_t d_lens_cap = 0; std::vector<int> h_lens;   // per-clip lengths of a ragged batch
    int use_gru256_cluster = 1;
    // single-hop streaming: chores of the call's front end that the fused feature kernel of the hop picks up (feat_hop_kernel):
    // the sum over K-split STFT partials and the hand-over of the analysis buffers.  Set by streams_enqueue, consumed by stage 1.
    float* snap_dst = nullptr;         // pending pre-call state copy of a streaming call (consumed by the first stage-1 import)
    struct HopExtras { const float* part = nullptr; int ks = 0, W = 0; const float* pcm_new = nullptr; float* in_tail = nullptr; float* snap_in = nullptr; bool armed = false; } hx;
    int fuse_gl = 1;                   // small launches: grouped linears around the GRU-256 cells chained per 16-row tile in one launch each (0: A/B)
    int glue8 = 1;                     // single-hop DPRNN glue on eight waves per tile (0: four; A/B)
    int fuse_small = 1;                // launches of <= 512 rows: small dependent kernels merged (mask + deep filter, the embedding fan-in / fan-out linears; 0: A/B)
    int fuse_enc = 1;                  // ... and the ERB encoder's four convolutions (erb_enc_seg_kernel; 0: A/B)
    int seg10 = 1;                     // 48 kHz pyramid kernels: 10-position segments when 8-position ones would exceed one workgroup per CU (0: A/B)
    int dfout_in_decin = 1;            // decoders in series: df_out shares the ERB decoder's dec_in launch (0: A/B)
    int hop_pconv = 1;                 // streaming hops: the DF decoder's pathway conv inside df_enc_seg_kernel (0: its own launch in stage 2; A/B)
    int dual_step = 1;                 // streaming hops with the decoders in series: the two decoders' GRU-256 steps pairwise in one launch (0: A/B)
    int hop_dec_fork = 0;              // one-chunk calls: 1 = the DF decoder forks onto its own stream beside the ERB decoder (measured 4-14 us slower per hop than in series: two handoffs)
    int hop_prologue = 1;              // single hops of > 4 streams: staging + stage-1 FIFO import + state copy as one launch in front of the STFT (0: A/B)
    int enc_seg_rows = 512, dec_pyr_rows = 512;      // frame rows up to which the pyramid kernels (enc_seg.h, dec_pyr.h) replace the per-layer launches.  They are latency forms
                                                      // (weights re-read per workgroup): at 256 clips x 10 s they are bit-identical but not faster (tools/offline_ab.py: 48 kHz 152.9 -> 153.5 /
                                                      // 162.8 ms per step, 16 kHz 107.0 -> 109.2 / 109.6)
    int late_export = 1;               // streaming hops: the FIFO export behind the overlap-add, the host waits for the output only (0: A/B)
    int snapshot = 1;                  // streaming calls keep a pre-call copy of state and tails for the re-run after a device-side timeout (0: A/B only)
    int single_chunk_inline = 1;       // one-chunk calls: stage 2 on the main stream instead of the stage-2 stream (0: A/B)
    int fuse_dec = 1;                  // ... and the ERB decoder's three stages + mask head (dec_pyr_kernel; 0: A/B)
    int interleave = 1;                // the two encoder branches' blocks enqueued alternately (0: one branch after the other; A/B)
    int hop_feat = 1;                  // single-hop calls: features A + B (+ those chores) as one launch (0: separate kernels, A/B)
    int* pin_progress = nullptr;       // pinned host word: frames of the running offline call whose stage 2 is complete (dpdf_progress)
    bool progress_on = false;          // set by the offline entry points only (a streaming hop does not pay for the extra launch)
    long recoveries = 0;               // calls re-run on the non-spinning GRU-256 kernels after a cluster exchange timed out (dpdf_recovery_count)
    // prepared weights (arena offsets)
    size_t conv0_w, conv0_b;
    SepConvW erb_conv1, erb_conv2, erb_conv3, df_conv1, convt3, convt2, convt1;
    size_t dfc0_pwfrag, dfc0_bias;      // df_conv0 folded to one im2col operand [32][64] + BN shift
    std::vector<DprnnW> dprnn_erb, dprnn_df;
    GlW enc_erb_fc, df_fc_emb, enc_lin_in, enc_lin_out, ed_lin_in, ed_lin_out, ed_erb_fc, df_lin_in, df_skip, df_out;
    Gru256W enc_gru, ed_gru0, ed_gru1, df_gru0, df_gru1;
    PathW conv3p, conv2p, conv1p, conv0p;
    size_t c0out_w; float c0out_bias;
    size_t convp_frag, convp_bias;
    size_t window, stft_frag_s, istft_frag;
    int stft_groups_s, istft_groups, istft_K;
    DevBuf io_spec, io_spec_e, io_state, io_wav, io_out, frames, raw_spec, enh_spec, batch_state, stft_part;
    size_t dft_f1 = 0, dft_f2 = 0, dft_iA = 0, dft_iB = 0;    // operands of the two-stage DFT (dft2stage.h)
    DevBuf dft_mid_f, dft_mid_i;       // its intermediates [frames][30][64] (analysis / synthesis: they may run on different streams)
    long dbg_nspec = 0, dbg_nframes = 0;
    int dft2 = 1;                      // big launches: STFT / iSTFT as two small matrix stages (0: one [win x 2F] GEMM; A/B)
    size_t dft64_tw1 = 0, dft64_twm = 0, dft64_tw2 = 0;   // operands of the float64 analysis DFT (dft64.h; doubles stored in the float arena)
    int gru64_limbs = 0;               // OPT-IN (default off: the headline arithmetic is fp32 MFMA, reviews of rounds 1 and 2); bit 0: intra-band pair, bit 1: inter-band scan: the GRU-64 throughput kernels on bf16 limbs (gru_limb.h: every fp32 product from three bf16 limbs per operand, six
                                       // bf16 MFMAs per term, fp32 accumulation -- fp32-exact products at 2.67 x the fp32 matrix rate); 0 = the fp32-MFMA kernels of gru_scan.h
    int dft64 = 2;                     // analysis STFT in float64 on every call path (dft64.h): 1 = 48 kHz models only (per-bin log-magnitude features), 2 = 16 kHz too (default: one analysis everywhere), 0 = the fp32 forms (A/B)
    bool use_dft64() const { return (d.win == 960 && dft64 >= 1) || (d.win == 320 && dft64 >= 2); }
    HostPipe hp;                       // pinned staging ring + copy streams of the host-pointer batch calls
    int host_pipe = 1;                 // host-pointer batch calls pipelined over time slices (0: one upload, compute, one download; A/B)
    int gru256_fused_x_tiles = 6;      // ... from this many 16-row tiles on
    int gru256_fused_x = 1;            // big batches: GRU-256 input projection inside the four-workgroup cluster scan (gru_clusterx.h; 0: hoisted GEMM + scan, A/B)
    int chunk_io = 0;                  // device-pointer batch calls of several chunks: STFT / iSTFT + overlap-add per chunk beside the frame function (0: two whole-batch launches; A/B)
    int host_prefault = 1;             // pipelined host calls: a helper thread populates the caller's output rows while the first chunk computes (0: A/B)
    int host_copy_threads = 4;         // threads (incl. the caller's) that move rows between the caller's memory and pinned staging
    int stft_ksplit = 7;               // few frames: bit 0 STFT split five ways over K (stft_small), bit 1 streaming iSTFT split seven ways (summed by the overlap-add kernel)
    // profiling
    bool prof_on = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> prof_events; int prof_used = 0;
    std::vector<std::pair<const char*, int>> prof_pending;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const float* C(size_t off) const { return consts + off; }
};

struct dpdf_streams {
    dpdf_model* m; int S;
    DevBuf state, in_tail, ola_tail, spec, spec_e, pcm_in, pcm_out;
    std::vector<int> primed;
    // host-pointer calls: PCM staged through pinned, GPU-visible host memory -- the first kernel of a hop reads the noisy PCM
    // straight out of it and the last one writes the enhanced PCM (and the device error flag) straight into it: no copy
    // commands, no second blocking read-back of the flag
    float* pin_in = nullptr; float* pin_out = nullptr; int* pin_err = nullptr; size_t pin_cap = 0;
    // masked calls (dpdf_streams_process_masked): the active streams packed into a dense batch
    DevBuf cstate, cin, cola, cpcm_in, cpcm_out; int* pin_idx = nullptr;
    // pre-call copy of the state and the tails, taken at the start of every host-pointer call: a GRU-256 exchange that timed out
    // leaves the in-place state half advanced -- the call is then restored from here and re-run on the kernels that
    // do not spin (recover_and_rerun)
    DevBuf snap_state, snap_in, snap_ola;
    hipEvent_t ev_snap = nullptr;
    hipEvent_t ev_out = nullptr;       // behind the overlap-add of the latest call: the output is in place (the state export follows it)
    struct StreamPoolC* pool = nullptr; // native coalescing of independent submitters (dpdf_streams_submit*), created on first use
    // (Measured and dropped: replaying a captured hipGraph of the hop -- ~110 launches over four streams -- instead of
    // enqueueing them: 781 / 319 / 446 us per hop against 748 / 307 / 433 us with plain launches for 64 x 48 kHz dpdfnet8,
    // one 16 kHz dpdfnet2 and eight dpdfnet4 streams: the hop is bound by the dependent kernels on the GPU, not by the
    // host's 190-270 us of enqueueing (tools/hop_probe.py); and the HIP runtime bundled with torch recurses without end in
    // hipStreamEndCapture on this four-stream fork/join pattern.)
};

// The GRU-256 cluster scans exchange h' between workgroups by spinning on granules (gru_scan.h).  A spin that times out
// (peer workgroups never became co-resident: GPU shared with other processes, oversubscribed queues) raises the device
// flag d_err and the scan carries on with stale data -- so every point where results become visible to the caller
// reads the flag back and turns it into DPDF_E_RUNTIME instead of returning corrupted audio with DPDF_OK.
static int check_device_err(dpdf_model* m) {
    if (!m->d_err) return DPDF_OK;
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, m->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (!flag) return DPDF_OK;
    HIP_TRY(hipMemset(m->d_err, 0, sizeof(int)));
    return set_err(DPDF_E_RUNTIME, "GRU-256 cluster exchange timed out (peer workgroups were not co-resident) in an asynchronous "
                                   "(device-pointer) call: its results are invalid and a state updated in place is half advanced -- "
                                   "streams must be reset (dpdf_streams_reset) or restored (dpdf_streams_set_state), batch calls "
                                   "re-issued; host-pointer calls recover by themselves (dpdf_recovery_count)");
}

// Host-pointer calls of the batch entry points recover from a timed-out exchange by themselves: their state starts from the
// host's copy (or the initial state), so the whole call simply runs again with every GRU-256 recurrence on the single-
// workgroup scan, which has no cross-workgroup waits.  DPDF_RETRY is the internal "flag was set" code of the call bodies.
constexpr int DPDF_RETRY = -1000;
static int device_err_or_retry(dpdf_model* m) {
    if (!m->d_err) return DPDF_OK;
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, m->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (!flag) return DPDF_OK;
    HIP_TRY(hipMemset(m->d_err, 0, sizeof(int)));
    return DPDF_RETRY;
}
template <class Body>
static int with_recovery(dpdf_model* m, Body body) {
    int rc = body();
    if (rc != DPDF_RETRY) return rc;
    const int saved = m->use_gru256_cluster;
    m->use_gru256_cluster = 0;
    rc = body();
    m->use_gru256_cluster = saved;
    ++m->recoveries;
    if (rc == DPDF_RETRY) return set_err(DPDF_E_RUNTIME, "device error flag raised again on the non-spinning path");
    return rc;
}

// streams + events of the lane
static int init_lane(Lane& L) {
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (!L.sA) {
        HIP_TRY(hipStreamCreateWithFlags(&L.sA, hipStreamNonBlocking));
        // stage 2 is latency-bound (GRU-256 cluster scans): give its workgroups dispatch priority
        HIP_TRY(hipStreamCreateWithPriority(&L.sB, hipStreamNonBlocking, hi));
        HIP_TRY(hipStreamCreateWithFlags(&L.sC, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithPriority(&L.sD, hipStreamNonBlocking, hi));
        for (int p = 0; p < NRING; ++p) {
            HIP_TRY(hipEventCreateWithFlags(&L.ev_s1[p], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&L.ev_s2[p], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&L.ev_fk[p], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&L.ev_jn[p], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&L.ev_dfk[p], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&L.ev_djn[p], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.ev_x2, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
    }
    return DPDF_OK;
}


namespace {

struct ProfScope {
    // Records a HIP event pair around a launch class on the model's stream; NO host sync here --
    // elapsed times are resolved in dpdf_profile_report after the stream has drained, so profiling
    // can stay on inside a timed region.
    dpdf_model* m; const char* name; int idx = -1;
    ProfScope(dpdf_model* m_, const char* n) : m(m_), name(n) {
        if (!m->prof_on) return;
        if (m->prof_used + 2 > (int)m->prof_events.size()) {
            size_t old = m->prof_events.size();
            m->prof_events.resize(old + 512, nullptr);
            for (size_t i = old; i < m->prof_events.size(); ++i) (void)hipEventCreate(&m->prof_events[i]);
        }
        idx = m->prof_used; m->prof_used += 2;
        (void)hipEventRecord(m->prof_events[idx], m->cur);
    }
    ~ProfScope() {
        if (idx < 0) return;
        (void)hipEventRecord(m->prof_events[idx + 1], m->cur);
        m->prof_pending.push_back({name, idx});
    }
};

// ----- weight builders ------------------------------------------------------------------------
SepConvW build_sepconv(Arena& A, const Blob& B, const std::string& p, int nsub) {
    SepConvW s; s.nsub = nsub < 1 ? 1 : nsub;
    std::vector<float> dw((size_t)s.nsub * 64 * 3);
    for (int k = 0; k < s.nsub; ++k) {
        const float* w = nsub <= 1 ? B.get(p + ".0.weight") : B.get(p + ".0.convs." + std::to_string(k) + ".weight");
        std::copy(w, w + 192, dw.begin() + (size_t)k * 192);
    }
    s.dw = A.add(dw);
    BnFold f = fold_bn(B, p + ".2", 64);
    const float* pw = B.get(p + ".1.weight");           // [out][in]
    s.pwfrag = A.add(pack_frag(64, 64, 4, [&](int k, int n) { return pw[n * 64 + k] * f.scale[n]; }));
    s.bias = A.add(f.shift);
    return s;
}
PathW build_path(Arena& A, const Blob& B, const std::string& p) {
    BnFold f = fold_bn(B, p + ".1", 64);
    const float* sc = B.get(p + ".0.weight");
    std::vector<float> ps(64), pb(64);
    for (int c = 0; c < 64; ++c) { ps[c] = sc[c] * f.scale[c]; pb[c] = f.shift[c]; }
    PathW w; w.ps = A.add(ps); w.pb = A.add(pb);
    return w;
}

// fp32 -> three bf16 limbs, v = hi + mid + lo exactly (round to nearest even; the residues are exact fp32 subtractions): gru_limb.h
static inline unsigned short bf16_rne_bits(float x) {
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_bits_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void limbs3(float v, unsigned short out[3]) {
    out[0] = bf16_rne_bits(v); const float r1 = v - bf16_bits_f(out[0]);
    out[1] = bf16_rne_bits(r1); const float r2 = r1 - bf16_bits_f(out[1]);
    out[2] = bf16_rne_bits(r2);
}
// limb fragments of `nmat` matrices [16-row wave block 4][k-chunk 2][limb 3][lane 64] x 8 bf16 as an arena blob (bit patterns in floats):
// A operand of v_mfma_f32_16x16x32_bf16 -- lane (q, m) holds W(mat, row 16 w + m, k = 32 c + 8 q + j), j = 0..7
template <class Fn>
static std::vector<float> pack_limb_frags(int nouter, int nmat, Fn w) {      // layout [outer][wave][mat][chunk][limb][lane][8]
    std::vector<unsigned short> f((size_t)nouter * 4 * nmat * 2 * 3 * 64 * 8);
    for (int o = 0; o < nouter; ++o) for (int wv = 0; wv < 4; ++wv) for (int mt = 0; mt < nmat; ++mt) for (int c = 0; c < 2; ++c)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            unsigned short l3[3];
            limbs3(w(o, mt, 16 * wv + (lane & 15), 32 * c + 8 * (lane >> 4) + j), l3);
            for (int l = 0; l < 3; ++l)
                f[((((((size_t)o * 4 + wv) * nmat + mt) * 2 + c) * 3 + l) * 64 + lane) * 8 + j] = l3[l];
        }
    std::vector<float> out(f.size() / 2);
    memcpy(out.data(), f.data(), f.size() * 2);
    return out;
}

// [dir][wave][part][gate][chunk*4+kb][lane] + bias [dir][4][64]
GruW64 build_gru64(Arena& A, const Blob& B, const std::vector<std::pair<std::string, std::string>>& dirs) {
    GruW64 g; g.ndirs = (int)dirs.size();
    // The exponent scales of the gate non-linearities are folded into the packed weights, so that the accumulators
    // come out of the MFMAs ready for v_exp_f32 (2^x): sigma(a) = 1/(1 + 2^(-a log2 e)) for r and z, and
    // tanh(t) = 2/(1 + 2^(-2 t log2 e)) - 1 for the candidate (common.h gru64_cell).  VALU instructions and fp32 MFMAs
    // share the SIMD datapath (docs/HISTORY.md section 3), so the four multiplies per hidden unit this removes are MFMA time.
    const float gate_scale[3] = {-1.4426950408889634f, -1.4426950408889634f, -2.8853900817779268f};
    std::vector<float> frag((size_t)g.ndirs * 4 * 2 * 3 * 16 * 64), bias((size_t)g.ndirs * 256);
    for (int d = 0; d < g.ndirs; ++d) {
        const std::string &p = dirs[d].first, &sfx = dirs[d].second;
        const float* wih = B.get(p + ".weight_ih" + sfx); const float* whh = B.get(p + ".weight_hh" + sfx);
        const float* bih = B.get(p + ".bias_ih" + sfx);   const float* bhh = B.get(p + ".bias_hh" + sfx);
        for (int w = 0; w < 4; ++w)
            for (int part = 0; part < 2; ++part)
                for (int gate = 0; gate < 3; ++gate)
                    for (int c = 0; c < 4; ++c)
                        for (int kb = 0; kb < 4; ++kb)
                            for (int lane = 0; lane < 64; ++lane) {
                                int k = kperm(c, lane >> 4, kb);
                                int j = gate * 64 + 16 * w + (lane & 15);
                                const float* W = part == 0 ? wih : whh;
                                frag[(((((size_t)(d * 4 + w) * 2 + part) * 3 + gate) * 16) + c * 4 + kb) * 64 + lane] = W[j * 64 + k] * gate_scale[gate];
                            }
        for (int j = 0; j < 64; ++j) {
            bias[d * 256 + j] = (bih[j] + bhh[j]) * gate_scale[0];
            bias[d * 256 + 64 + j] = (bih[64 + j] + bhh[64 + j]) * gate_scale[1];
            bias[d * 256 + 128 + j] = bih[128 + j] * gate_scale[2];
            bias[d * 256 + 192 + j] = bhh[128 + j] * gate_scale[2];
        }
    }
    g.wfrag = A.add(frag); g.bias = A.add(bias);
    {   // bf16 limb fragments (gru_limb.h) of the SAME scaled fp32 values: mat = side * 3 + gate
        std::vector<const float*> wi(g.ndirs), wh(g.ndirs);
        for (int d = 0; d < g.ndirs; ++d) { wi[d] = B.get(dirs[d].first + ".weight_ih" + dirs[d].second); wh[d] = B.get(dirs[d].first + ".weight_hh" + dirs[d].second); }
        g.wl = A.add(pack_limb_frags(g.ndirs, 6, [&](int d, int mt, int unit, int k) {
            const int gate = mt % 3;
            return (mt < 3 ? wi[d] : wh[d])[(size_t)(gate * 64 + unit) * 64 + k] * gate_scale[gate];
        }));
    }
    {   // 4-row scan (gru_scan4.h): A operand of instruction t = 4m + g of wave w, lane 4b + i = scaled W_hh[gate i][unit 16w + b][k],
        // k = 4m + ((g + b) & 3) -- the k that block b meets under the B operand's lane-group broadcast g (i = 3: zero)
        std::vector<float> f4((size_t)g.ndirs * 4 * 64 * 64, 0.f);
        for (int d = 0; d < g.ndirs; ++d) {
            const float* whh = B.get(dirs[d].first + ".weight_hh" + dirs[d].second);
            for (int w = 0; w < 4; ++w) for (int t = 0; t < 64; ++t) for (int b = 0; b < 16; ++b) for (int i = 0; i < 3; ++i) {
                const int k = 4 * (t >> 2) + (((t & 3) + b) & 3);
                f4[(((size_t)(d * 4 + w) * 64) + t) * 64 + 4 * b + i] = whh[(size_t)(i * 64 + 16 * w + b) * 64 + k] * gate_scale[i];
            }
        }
        g.hh4 = A.add(f4);
    }
    {   // the same W_ih (and input-side biases), scaled alike, as an ordinary GEMM operand for gru64_scan_gi_kernel
        std::vector<float> gfrag, gbias;
        for (int d = 0; d < g.ndirs; ++d) {
            const std::string &p = dirs[d].first, &sfx = dirs[d].second;
            const float* wih = B.get(p + ".weight_ih" + sfx);
            const float* bih = B.get(p + ".bias_ih" + sfx); const float* bhh = B.get(p + ".bias_hh" + sfx);
            for (int gate = 0; gate < 3; ++gate) {
                auto f = pack_frag(64, 64, 4, [&](int k, int n) { return wih[(gate * 64 + n) * 64 + k] * gate_scale[gate]; });
                gfrag.insert(gfrag.end(), f.begin(), f.end());
                for (int j = 0; j < 64; ++j)
                    gbias.push_back((gate < 2 ? bih[gate * 64 + j] + bhh[gate * 64 + j] : bih[128 + j]) * gate_scale[gate]);
            }
        }
        g.ih_frag = A.add(gfrag); g.ih_bias = A.add(gbias);
    }
    return g;
}
GlW build_gl(Arena& A, const Blob& B, const std::string& p, int G, int Og, int Ig) {
    GlW g; g.G = G; g.Og = Og; g.Ig = Ig; g.NT = (Og + 15) / 16;
    const float* w = B.get(p + ".weight"); const float* b = B.get(p + ".bias");
    std::vector<float> frag;
    for (int gi = 0; gi < G; ++gi) {
        auto f = pack_frag(Ig, Og, g.NT, [&](int k, int n) { return w[((size_t)gi * Og + n) * Ig + k]; });
        frag.insert(frag.end(), f.begin(), f.end());
    }
    g.frag = A.add(frag);
    g.bias = A.add(std::vector<float>(b, b + (size_t)G * Og));
    return g;
}
Gru256W build_gru256(Arena& A, const Blob& B, const std::string& p) {
    Gru256W g;
    const float* wih = B.get(p + ".weight_ih"); const float* whh = B.get(p + ".weight_hh");
    const float* bih = B.get(p + ".bias_ih");   const float* bhh = B.get(p + ".bias_hh");
    {   // input projection in 24 blocks of 32 columns: few rows -> one workgroup per (row tile, block) (gemm_rows,
        // latency-bound); many rows -> four blocks per workgroup, one per wave (gemm_rows_wn)
        std::vector<float> fs;
        for (int g24 = 0; g24 < 24; ++g24) {
            auto f = pack_frag(256, 32, 2, [&](int k, int n) { return wih[(size_t)(g24 * 32 + n) * 256 + k]; });
            fs.insert(fs.end(), f.begin(), f.end());
        }
        g.ih_frag_s = A.add(fs);
    }
    std::vector<float> bias(768), bhn(256);
    for (int j = 0; j < 256; ++j) {
        bias[j] = bih[j] + bhh[j]; bias[256 + j] = bih[256 + j] + bhh[256 + j]; bias[512 + j] = bih[512 + j];
        bhn[j] = bhh[512 + j];
    }
    g.ih_bias = A.add(bias); g.b_hn = A.add(bhn);
    // recurrent: [wave 16][gate 3][chunk 16][kb 4][lane 64]
    std::vector<float> hh((size_t)16 * 3 * 64 * 64);
    for (int w = 0; w < 16; ++w)
        for (int gate = 0; gate < 3; ++gate)
            for (int c = 0; c < 16; ++c)
                for (int kb = 0; kb < 4; ++kb)
                    for (int lane = 0; lane < 64; ++lane) {
                        int k = kperm(c, lane >> 4, kb), j = gate * 256 + 16 * w + (lane & 15);
                        hh[((((size_t)w * 3 + gate) * 16 + c) * 4 + kb) * 64 + lane] = whh[(size_t)j * 256 + k];
                    }
    g.hh_frag = A.add(hh);
    for (int w = 0; w < 16; ++w)
        for (int gate = 0; gate < 3; ++gate)
            for (int c = 0; c < 16; ++c)
                for (int kb = 0; kb < 4; ++kb)
                    for (int lane = 0; lane < 64; ++lane) {
                        int k = kperm(c, lane >> 4, kb), j = gate * 256 + 16 * w + (lane & 15);
                        hh[((((size_t)w * 3 + gate) * 16 + c) * 4 + kb) * 64 + lane] = wih[(size_t)j * 256 + k];
                    }
    g.ih_as_hh = A.add(hh);
    return g;
}
std::vector<DprnnW> build_dprnn(Arena& A, const Blob& B, const std::string& p, int nb) {
    std::vector<DprnnW> v;
    for (int i = 0; i < nb; ++i) {
        std::string q = p + ".blocks." + std::to_string(i);
        DprnnW w;
        w.intra = build_gru64(A, B, {{q + ".intra_gru", "_l0"}, {q + ".intra_gru", "_l0_reverse"}});
        w.inter = build_gru64(A, B, {{q + ".inter_gru.grucell", ""}});
        const float* fi = B.get(q + ".fc_intra.weight");   // [64][128]
        w.fci_frag = A.add(pack_frag(128, 64, 4, [&](int k, int n) { return fi[n * 128 + k]; }));
        w.fci_b = A.add(std::vector<float>(B.get(q + ".fc_intra.bias"), B.get(q + ".fc_intra.bias") + 64));
        w.lni_g = A.add(std::vector<float>(B.get(q + ".ln_intra.weight"), B.get(q + ".ln_intra.weight") + 64));
        w.lni_b = A.add(std::vector<float>(B.get(q + ".ln_intra.bias"), B.get(q + ".ln_intra.bias") + 64));
        const float* fe = B.get(q + ".fc_inter.weight");   // [64][64]
        w.fce_frag = A.add(pack_frag(64, 64, 4, [&](int k, int n) { return fe[n * 64 + k]; }));
        w.fce_b = A.add(std::vector<float>(B.get(q + ".fc_inter.bias"), B.get(q + ".fc_inter.bias") + 64));
        w.lne_g = A.add(std::vector<float>(B.get(q + ".ln_inter.weight"), B.get(q + ".ln_inter.weight") + 64));
        w.lne_b = A.add(std::vector<float>(B.get(q + ".ln_inter.bias"), B.get(q + ".ln_inter.bias") + 64));
        {   // epilogue-fused forms: wave w owns output columns [16w,16w+16)
            auto pack_epi = [&](const float* W, int ld, int koff) {
                std::vector<float> f((size_t)4 * 16 * 64);
                for (int wv = 0; wv < 4; ++wv)
                    for (int c = 0; c < 4; ++c)
                        for (int kb = 0; kb < 4; ++kb)
                            for (int lane = 0; lane < 64; ++lane)
                                f[((size_t)wv * 16 + c * 4 + kb) * 64 + lane] = W[(size_t)(16 * wv + (lane & 15)) * ld + koff + kperm(c, lane >> 4, kb)];
                return f;
            };
            std::vector<float> fi = pack_epi(B.get(q + ".fc_intra.weight"), 128, 64);      // part 0: fed by hb (this scan's h')
            std::vector<float> fi1 = pack_epi(B.get(q + ".fc_intra.weight"), 128, 0);      // part 1: fed by hf
            fi.insert(fi.end(), fi1.begin(), fi1.end());
            w.fci_epi = A.add(fi);
            w.fce_epi = A.add(pack_epi(B.get(q + ".fc_inter.weight"), 64, 0));
            const float* fiw = B.get(q + ".fc_intra.weight"); const float* few = B.get(q + ".fc_inter.weight");
            w.fci_lb = A.add(pack_limb_frags(1, 1, [&](int, int, int n, int k) { return fiw[(size_t)n * 128 + 64 + k]; }));
            w.fci_lf = A.add(pack_limb_frags(1, 1, [&](int, int, int n, int k) { return fiw[(size_t)n * 128 + k]; }));
            w.fce_l = A.add(pack_limb_frags(1, 1, [&](int, int, int n, int k) { return few[(size_t)n * 64 + k]; }));
        }
        v.push_back(w);
    }
    return v;
}

// vorbis window (reference package/src/dpdfnet/audio.py:84-88)
std::vector<float> vorbis(int n) {
    std::vector<float> w(n);
    const double h = n / 2.0;
    for (int i = 0; i < n; ++i) { double s = std::sin(0.5 * M_PI * (i + 0.5) / h); w[i] = (float)std::sin(0.5 * M_PI * s * s); }
    return w;
}
// ERB band edges (reference model/utils.py:265-324), 16 kHz: 32 bands over 161 bins, min width 1
void erb_bands(int nfft, int fs, std::vector<int>& start, std::vector<int>& band_of) {
    const int nf = 32, F = nfft / 2 + 1;
    const double fw = (double)fs / nfft;
    const double lo = 9.265 * std::log1p(0.0), hi = 9.265 * std::log1p((fs / 2.0) / (24.7 * 9.265));
    const double step = (hi - lo) / nf;
    std::vector<int> bins(nf + 1);
    for (int i = 0; i <= nf; ++i) bins[i] = (int)std::nearbyint(24.7 * 9.265 * (std::exp((lo + i * step) / 9.265) - 1.0) / fw);
    bins[nf] = F;
    start.assign(nf + 1, 0); band_of.assign(F, 0);
    int over = 0;
    for (int j = 0; j < nf; ++j) {
        int a = bins[j] + over, b = bins[j + 1];
        if (b - a < 1) { over = 1 - (b - a); b = std::min(b + over, F); } else over = 0;
        start[j] = a; start[j + 1] = b;
        for (int f = a; f < b; ++f) band_of[f] = j;
    }
}

// Row count below which the wide-N GEMMs switch to their narrow-column packing: with <= 8 row tiles the launch is a
// handful of workgroups walking all K panels one after the other; narrower column blocks multiply the workgroups.
// Also the limit of the fused small-launch forms (small_fused_mfma.h, mask_df_kernel).
constexpr int SMALL_M_ROWS = 512;

int ensure_xset(dpdf_model* m, XSet& x, int B, int Tc) {
    const dpdf_dims& d = m->d;
    const size_t BT = (size_t)B * Tc;
    int rc = DPDF_OK;
#define ENSX(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
    ENSX(x.xs, (size_t)B * (Tc + 2) * d.F * 2);
    ENSX(x.e0, BT * d.Ec * 64); ENSX(x.e1, BT * d.F1 * 64); ENSX(x.e2, BT * d.F2 * 64); ENSX(x.e3, BT * d.F3 * 64);
    ENSX(x.xe_a, BT * d.F3 * 64); ENSX(x.xe_b, BT * d.F3 * 64);
    ENSX(x.c0, (size_t)B * (Tc + 4) * d.D * 64); ENSX(x.c1, BT * d.Fd * 64);
    ENSX(x.xd_a, BT * d.Fd * 64); ENSX(x.xd_b, BT * d.Fd * 64);
    ENSX(x.pconv, BT * d.D * 10);
#undef ENSX
    return DPDF_OK;
}

int ensure_ws(dpdf_model* m, int B, int Tc) {
    Workspace& w = m->ln->ws;
    {   // hoisted input-side GRU-64 pre-activations (run_dprnn): intra form (2 dirs x 192 per band row) only below 3072
        // frame rows, inter form (192 per band row) whenever B*F' is too small to fill the chip.  Not monotone in B,
        // so checked on every call; growing waits for the streams like the rest of the workspace.
        const dpdf_dims& d = m->d;
        const size_t BT = (size_t)B * Tc, bt_small = std::min(BT, (size_t)3071);
        const size_t need_d = std::max(bt_small * d.Fd * 384, (long)B * d.Fd < m->inter_fuse_rows ? BT * d.Fd * 192 : (size_t)0);
        const size_t need_e = std::max(bt_small * d.F3 * 384, (long)B * d.F3 < m->inter_fuse_rows ? BT * d.F3 * 192 : (size_t)0);
        if (need_d > w.gi64.n || need_e > w.gi64_e.n) {
            m->ln->sync_all();
            int rc = w.gi64.ensure(need_d); if (rc) return rc;
            rc = w.gi64_e.ensure(need_e); if (rc) return rc;
        }
    }
    if (B <= w.Bcap && Tc <= w.Tcap) return DPDF_OK;    // every size below is monotone in B and Tc
    // growing: make sure nothing in flight still uses the old buffers
    m->ln->sync_all();
    B = std::max(B, w.Bcap); Tc = std::max(Tc, w.Tcap);
    const dpdf_dims& d = m->d;
    const size_t BT = (size_t)B * Tc;
    int rc = DPDF_OK;
#define ENS(buf, n) do { rc = (buf).ensure(n); if (rc) return rc; } while (0)
    for (int k = 0; k < 2; ++k) { rc = ensure_xset(m, w.x[k], B, Tc); if (rc) return rc; }
    ENS(w.feat_erb, (size_t)B * (Tc + 2) * d.E);
    ENS(w.feat_spec, (size_t)B * (Tc + 2) * 2 * d.D);
    ENS(w.hcat, BT * d.Fd * 128); ENS(w.hin, BT * d.Fd * 64);
    ENS(w.hcat_e, BT * d.F3 * 128); ENS(w.hin_e, BT * d.F3 * 64);
    // GRU-256 scan inputs / outputs: rows rounded up to whole 16-clip tiles (gru256_ring_kernel addresses rows
    // unclamped; the padding rows are read, never written or used)
    const size_t BTp = (size_t)((B + 15) & ~15) * Tc;
    ENS(w.embin, BT * 1024); ENS(w.g256a, BTp * 256); ENS(w.g256b, BTp * 256); ENS(w.g256c, BTp * 256);
    ENS(w.gi, BTp * 768); ENS(w.emb, BT * 512); ENS(w.demb, BT * 512);
    ENS(w.g256d, BTp * 256); ENS(w.g256e, BTp * 256); ENS(w.g256f, BTp * 256); ENS(w.gi2, BTp * 768);
    ENS(w.skipb, std::min(BTp, (size_t)SMALL_M_ROWS + 16) * 256);
    ENS(w.demb2, BT * (size_t)d.F3 * 64);
    ENS(w.d3, BT * d.F2 * 64); ENS(w.d2, BT * d.F1 * 64); ENS(w.d1, BT * d.Ec * 64);
    ENS(w.m, BT * d.E); ENS(w.dfo, BT * d.D * 10);
    ENS(w.coefs, (size_t)B * (Tc + 2) * d.D * 10); ENS(w.xm, (size_t)B * (Tc + 4) * d.F * 2);
#undef ENS
    w.Bcap = B; w.Tcap = Tc;
    return DPDF_OK;
}

int ensure_gru_xbuf(dpdf_model* m, int ntiles, int which) {
    Lane& L = *m->ln;
    if (ntiles <= L.gru_xbuf_tiles[which] && L.gru_xbuf[which] && m->d_err) return DPDF_OK;
    if (L.gru_xbuf[which]) {
        L.sync_all();
        (void)hipFree(L.gru_xbuf[which]); L.gru_xbuf[which] = nullptr;
    }
    const size_t bytes = (size_t)ntiles * 2 * 16 * 256 * 8;
    if (hipMalloc((void**)&L.gru_xbuf[which], bytes) != hipSuccess) { L.gru_xbuf_tiles[which] = 0; return DPDF_E_RUNTIME; }
    (void)hipMemsetAsync(L.gru_xbuf[which], 0, bytes, m->cur);
    L.gru_epoch[which] = 0;
    L.gru_xbuf_tiles[which] = ntiles;
    if (!m->d_err) {
        if (hipMalloc((void**)&m->d_err, sizeof(int)) != hipSuccess) return DPDF_E_RUNTIME;
        (void)hipMemsetAsync(m->d_err, 0, sizeof(int), m->cur);
    }
    return DPDF_OK;
}
template <int NT, int KP>
void run_gl(dpdf_model* m, const GlW& g, const float* in, size_t lda, float* out, size_t ldo, int M, int act) {
    PlainA<KP> ap{in, lda, g.Ig, g.Ig};
    BiasActStore<NT> ep{out, ldo, g.Og, m->C(g.bias), g.Og, g.Og, act};
    launch_gemm_rows<NT, KP, false>(m->cur, ap, m->C(g.frag), ep, M, g.Ig, g.G);
}
void run_gl_auto(dpdf_model* m, const GlW& g, const float* in, size_t lda, float* out, size_t ldo, int M, int act) {
    if (g.NT == 1 && g.Ig % 32 == 0 && g.Ig != 64) run_gl<1, 32>(m, g, in, lda, out, ldo, M, act);
    else if (g.NT == 1 && g.Ig == 64) run_gl<1, 64>(m, g, in, lda, out, ldo, M, act);
    else if (g.NT == 1) run_gl<1, 16>(m, g, in, lda, out, ldo, M, act);
    else if (g.NT == 2 && g.Ig == 64) run_gl<2, 64>(m, g, in, lda, out, ldo, M, act);
    else if (g.NT == 2) run_gl<2, 16>(m, g, in, lda, out, ldo, M, act);
    else if (g.NT == 4) run_gl<4, 16>(m, g, in, lda, out, ldo, M, act);
    else run_gl<5, 16>(m, g, in, lda, out, ldo, M, act);
}

// SqueezedGRU_S cell: gi = W_ih x + b (all frames, one GEMM) then the recurrent scan
// which: 0 = embedding / ERB-decoder cells (scratch ws.gi, granules [0]); 1 = DF-decoder cells (ws.gi2, granules [1])
void run_gru256_proj(dpdf_model* m, const Gru256W& g, const float* x, float* gi, int M) {
    ProfScope ps(m, "gru256_proj");
    PlainA<64> ap{x, 256, 0, 256};
    if (M <= SMALL_M_ROWS) {
        BiasActStore<2> ep{gi, 768, 32, m->C(g.ih_bias), 32, 32, ACT_NONE};
        launch_gemm_rows<2, 64, false>(m->cur, ap, m->C(g.ih_frag_s), ep, M, 256, 24);
    } else {
        // many rows: waves split over columns (same 32-column packing, 6 quadruples of column groups): a quarter
        // of the B-fragment loads of the row-split form, 7.3 -> 5.6 ms per step of the headline workload
        BiasActStore<2> ep{gi, 768, 32, m->C(g.ih_bias), 32, 32, ACT_NONE};
        launch_gemm_rows_wn<2, 64>(m->cur, ap, m->C(g.ih_frag_s), ep, M, 256, 6);
    }
}

// Two stacked cells (g0 -> g1, the second one's input is the first one's hidden state) as one wavefront launch
// (gru_stack.h) when the launch is one or two tiles; false = not eligible, run the cells one after the other.
bool run_gru256_stack(dpdf_model* m, const Gru256W& g0, const Gru256W& g1, const float* x, float* out0, float* out1, float* state, long S,
                      int hoff, int B, int Tc, int which) {
    const int ntiles = (B + 15) / 16;
    if (Tc == 1 && m->gru256_step) return false;       // one frame: a step kernel per cell (run_gru256)
    if (!m->gru256_stack || !(m->overlap & 16) || !m->use_gru256_cluster || ntiles > m->gru256_c16_tiles || which < 0 || which > 1) return false;
    Lane& L = *m->ln;
    if (ntiles > L.gru_sbuf_tiles[which] || Tc > L.gru_sbuf_T[which] || !L.gru_sbuf[which] || !m->d_err) {
        if (L.gru_sbuf[which]) { L.sync_all(); (void)hipFree(L.gru_sbuf[which]); L.gru_sbuf[which] = nullptr; }
        const int nt = std::max(ntiles, L.gru_sbuf_tiles[which]), T = std::max(Tc, L.gru_sbuf_T[which]);
        const size_t bytes = (size_t)nt * (T + 2) * 16 * 256 * 8;
        if (hipMalloc((void**)&L.gru_sbuf[which], bytes) != hipSuccess) { L.gru_sbuf_tiles[which] = L.gru_sbuf_T[which] = 0; return false; }
        (void)hipMemsetAsync(L.gru_sbuf[which], 0, bytes, m->cur);
        L.gru_sepoch[which] = 0; L.gru_sbuf_tiles[which] = nt; L.gru_sbuf_T[which] = T;
        if (!m->d_err) {
            if (hipMalloc((void**)&m->d_err, sizeof(int)) != hipSuccess) return false;
            (void)hipMemsetAsync(m->d_err, 0, sizeof(int), m->cur);
        }
    }
    if (L.gru_sepoch[which] > 0xF0000000u) {     // epoch wrap: re-zero the granules
        (void)hipMemsetAsync(L.gru_sbuf[which], 0, (size_t)L.gru_sbuf_tiles[which] * (L.gru_sbuf_T[which] + 2) * 16 * 256 * 8, m->cur);
        L.gru_sepoch[which] = 0;
    }
    float* gi = which ? L.ws.gi2.p : L.ws.gi.p;
    run_gru256_proj(m, g0, x, gi, B * Tc);
    ProfScope ps(m, "gru256_scan");
    unsigned long long* ring = L.gru_sbuf[which];
    unsigned long long* xb = ring + (size_t)ntiles * Tc * 16 * 256;      // cell B's two slots behind this launch's rings
    Gru256SArgs a{gi, out0, out1, m->C(g0.hh_frag), m->C(g0.b_hn), m->C(g1.ih_as_hh), m->C(g1.hh_frag), m->C(g1.ih_bias), m->C(g1.b_hn),
                  state + hoff, state + hoff + 256, S, B, Tc, ring, xb, L.gru_sepoch[which], m->d_err};
    L.gru_sepoch[which] += (unsigned)Tc;
    hipLaunchKernelGGL(gru256_stack16_kernel, dim3(ntiles * 32), dim3(256), 0, m->cur, a);
    return true;
}

// One frame per stream: input projection + cell step as ONE launch (gru_stack.h: gru256_step_kernel)
bool prep_gru256_step(dpdf_model* m, const Gru256W& g, const float* x, float* out, float* state, long S, int hoff, int B, int which, Gru256StepArgs& a) {
    if (!m->gru256_step || !m->use_gru256_cluster || which < 0 || which > 4) return false;
    Lane& L = *m->ln;
    const int ntiles = (B + 15) / 16;
    if (ntiles > L.arrive_tiles[which] || !L.arrive[which] || !m->d_err) {
        if (L.arrive[which]) { L.sync_all(); (void)hipFree(L.arrive[which]); L.arrive[which] = nullptr; }
        if (hipMalloc((void**)&L.arrive[which], (size_t)ntiles * sizeof(unsigned)) != hipSuccess) { L.arrive_tiles[which] = 0; return false; }
        (void)hipMemsetAsync(L.arrive[which], 0, (size_t)ntiles * sizeof(unsigned), m->cur);
        L.arrive_tiles[which] = ntiles; L.arrive_count[which] = 0;      // (counter wrap: 2^32 / 16 launches -- decades of hops)
        if (!m->d_err) {
            if (hipMalloc((void**)&m->d_err, sizeof(int)) != hipSuccess) return false;
            (void)hipMemsetAsync(m->d_err, 0, sizeof(int), m->cur);
        }
    }
    if (L.arrive_count[which] > 0xF0000000u) {       // launches since the counters were zeroed: re-zero long before a 32-bit wrap
        (void)hipMemsetAsync(L.arrive[which], 0, (size_t)L.arrive_tiles[which] * sizeof(unsigned), m->cur);
        L.arrive_count[which] = 0;
    }
    L.arrive_count[which] += 16;
    a = Gru256StepArgs{x, out, m->C(g.ih_as_hh), m->C(g.hh_frag), m->C(g.ih_bias), m->C(g.b_hn), state + hoff, S, B, L.arrive[which], m->d_err};
    return true;
}
bool run_gru256_step(dpdf_model* m, const Gru256W& g, const float* x, float* out, float* state, long S, int hoff, int B, int which) {
    Gru256StepArgs a{};
    if (!prep_gru256_step(m, g, x, out, state, S, hoff, B, which, a)) return false;
    ProfScope ps(m, "gru256_scan");
    hipLaunchKernelGGL(gru256_step_kernel, dim3(((B + 15) / 16) * 16), dim3(256), 0, m->cur, a);
    return true;
}
// two independent cells' steps as one launch (arrival counter sets wa / wb); false: not eligible, nothing launched
bool run_gru256_step_dual(dpdf_model* m, const Gru256W& ga, const float* xa, float* oa, int ha, int wa,
                          const Gru256W& gb, const float* xb, float* ob, int hb, int wb, float* state, long S, int B) {
    if (!m->gru256_step || !m->use_gru256_cluster) return false;
    Gru256StepArgs a{}, b{};
    if (!prep_gru256_step(m, ga, xa, oa, state, S, ha, B, wa, a)) return false;
    if (!prep_gru256_step(m, gb, xb, ob, state, S, hb, B, wb, b)) { m->ln->arrive_count[wa] -= 16; return false; }
    ProfScope ps(m, "gru256_scan");
    const int n0 = ((B + 15) / 16) * 16;
    hipLaunchKernelGGL(gru256_step_dual_kernel, dim3(2 * n0), dim3(256), 0, m->cur, a, b, n0);
    return true;
}

void run_gru256(dpdf_model* m, const Gru256W& g, const float* x, float* out, float* state, long S, int hoff, int B, int Tc, int which = 0,
                float* gi_buf = nullptr) {
    if (Tc == 1 && !gi_buf && run_gru256_step(m, g, x, out, state, S, hoff, B, which)) return;
    const int M = B * Tc;
    float* gi = gi_buf ? gi_buf : (which ? m->ln->ws.gi2.p : m->ln->ws.gi.p);
    // Big batches (the four-workgroup cluster form): the input projection runs INSIDE the scan, in the time a wave would otherwise
    // spend waiting for its peers' granules (gru_clusterx.h) -- no chip-wide GEMM in front, no 3 KB per row through HBM.
    {
        const int ntiles = (B + 15) / 16;
        const bool four = !((m->overlap & 16) && ntiles <= std::max(m->gru256_c16_tiles, m->gru256_c8_tiles));
        // (from six tiles = the throughput regime of the chunk schedule on: 80 clips 44.0 -> 44.4 ms -- there the stage-2 chain's latency
        // counts and a step is 6.0 instead of 5.1 us --, 128 clips 59.1 -> 58.6, 256 clips 106.5 -> 105.5, 512 clips 208.0 -> 203.3)
        if (m->gru256_fused_x && four && ntiles >= m->gru256_fused_x_tiles && !gi_buf && Tc > 1 && m->use_gru256_cluster && ensure_gru_xbuf(m, ntiles, which) == DPDF_OK) {
            ProfScope ps(m, "gru256_scan");
            Lane& L = *m->ln;
            if (L.gru_epoch[which] > 0xF0000000u) {
                (void)hipMemsetAsync(L.gru_xbuf[which], 0, (size_t)L.gru_xbuf_tiles[which] * 2 * 16 * 256 * 8, m->cur);
                L.gru_epoch[which] = 0;
            }
            Gru256XArgs a{x, out, m->C(g.hh_frag), m->C(g.ih_as_hh), m->C(g.ih_bias), m->C(g.b_hn), state + hoff, S, B, Tc,
                          L.gru_xbuf[which], L.gru_epoch[which], m->d_err};
            L.gru_epoch[which] += (unsigned)Tc;
            hipLaunchKernelGGL(gru256_clusterx_kernel, dim3(ntiles * 4), dim3(256), 0, m->cur, a);
            return;
        }
    }
    run_gru256_proj(m, g, x, gi, M);
    {
        ProfScope ps(m, "gru256_scan");
        const int ntiles = (B + 15) / 16;
        if (m->use_gru256_cluster && ensure_gru_xbuf(m, ntiles, which) == DPDF_OK) {
            Lane& L = *m->ln;
            if (L.gru_epoch[which] > 0xF0000000u) {     // epoch wrap: re-zero the granules (once per ~4e9 steps)
                (void)hipMemsetAsync(L.gru_xbuf[which], 0, (size_t)L.gru_xbuf_tiles[which] * 2 * 16 * 256 * 8, m->cur);
                L.gru_epoch[which] = 0;
            }
            Gru256CArgs a{gi, out, m->C(g.hh_frag), m->C(g.b_hn), state + hoff, S, B, Tc, L.gru_xbuf[which], L.gru_epoch[which], m->d_err};
            L.gru_epoch[which] += (unsigned)Tc;
            // Forward progress of the cluster scans (peers spin on each other's granules under an ordinary, non-cooperative
            // launch) rests on ONE assumption: workgroups are dispatched in blockIdx order.  The block -> (tile, slice)
            // maps of both kernels put all workgroups of a tile inside one aligned run of 32 (resp. 64) consecutive
            // blocks, so the resident set always contains whole clusters, these finish, and later blocks get their CUs
            // (2048 clips = 512 workgroups on 256 CUs is covered by tests/test_gpu_fullsize.py).  If the assumption ever
            // fails the spin times out, d_err is raised and the call returns DPDF_E_RUNTIME (check_device_err).
            // sixteen / eight workgroups per tile for launches of few tiles (step latency), four from there on (tools/sweep2.sh)
            if ((m->overlap & 16) && ntiles <= m->gru256_c16_tiles) hipLaunchKernelGGL(gru256_cluster16_kernel, dim3(ntiles * 16), dim3(256), 0, m->cur, a);
            else if ((m->overlap & 16) && ntiles <= m->gru256_c8_tiles) hipLaunchKernelGGL(gru256_cluster8_kernel, dim3(ntiles * 8), dim3(256), 0, m->cur, a);
            else hipLaunchKernelGGL(gru256_cluster_kernel, dim3(ntiles * 4), dim3(256), 0, m->cur, a);
        } else {
            Gru256Args a{gi, out, m->C(g.hh_frag), m->C(g.b_hn), state + hoff, S, B, Tc};
            hipLaunchKernelGGL(gru256_scan_kernel, dim3(ntiles), dim3(1024), 0, m->cur, a);
        }
    }
}

// DPRNN (reference onnx_model/layers.py:159-196, 278-302): x [B*Tc][Fp][64] -> same, in xa (uses xb as scratch)
//
// Each of the two recurrences of a block picks its form from the parallelism it actually has (tiles of 16 rows):
//   intra-band: B*Tc/16 tiles x 2 directions, Fp steps;   inter-band: B*Fp/16 tiles, Tc steps.
//   * enough tiles to oversubscribe the 256 CUs  -> fused scans (fc + LayerNorm + residual inside the scan step):
//     the MFMA count is what matters and the fc rides along;
//   * fewer                                       -> a scan step is pure latency: W_ih x is hoisted into one GEMM over
//     all (row, step) pairs (gru64_scan_gi_kernel keeps the 48 h-part MFMAs), fc + LN run as a wide GEMM afterwards.
// Measured (tools/sweep2.sh, tools/latency_bench.py): intra crossover at 192 tiles (3072 frame rows).  Inter: the
// hoisted form wins below ~100 tiles when run alone (8 clips x 10 s: 8.7 -> 6.6 ms) but costs throughput inside the
// stream pipeline of a big batch (256 clips, ERB branch, 128 tiles: 125.6 -> 128.7 ms/step), so it is used below 64.
// xin is read only (it stays valid for its other consumers: e3 is the decoder's skip input); the blocks ping-pong
// between xa and xb, so no staging copy of the input is needed.
// geometry of the grouped linears around the GRU-256 cells that the chained small-launch kernels (small_fused_mfma.h) are written for
static bool small_gl_dims(const dpdf_model* m) {
    const dpdf_dims& d = m->d;
    return m->enc_lin_in.Ig == 64 && m->enc_lin_in.Og == 16 && m->df_fc_emb.Og == 16 && m->df_fc_emb.Ig == 96 &&
                         (!d.is48 || (m->enc_erb_fc.Og == 16 && m->enc_erb_fc.Ig == 80)) &&
                         m->enc_lin_out.Ig == 16 && m->enc_lin_out.Og == 32 && m->ed_lin_in.Ig == 32 && m->ed_lin_in.Og == 16 &&
                         m->df_skip.Ig == 32 && m->df_skip.Og == 16 && m->df_lin_in.Ig == 64 && m->df_lin_in.Og == 32 && m->df_lin_in.G == 8 &&
                         m->ed_lin_out.Ig == 16 && m->ed_lin_out.Og == 32 && (!d.is48 || (m->ed_erb_fc.Ig == 16 && m->ed_erb_fc.Og == 80)) &&
                         (d.is48 || d.F3 * 64 == 512);
}
// One DPRNN stack as a walk over its blocks: block(bi) enqueues block bi on m->cur.  The two encoder branches are walked
// alternately by run_stage1 (the DF stack on the main stream, the ERB stack on its own), so that in the latency regime -- where
// the host is only just ahead of the GPU -- neither chain waits for the other one's ~20 launches to be enqueued.
struct DprnnWalk {
    dpdf_model* m; const std::vector<DprnnW>& blocks; float* xin; float* xa; float* xb; float* hcat; float* hin; DevBuf& gibuf; int Fp;
    float* state; long S; int soff, B, Tc;
    int M; float* x; float* y;
    bool can_fuse, fuse_intra, fuse_inter, gi_intra, gi_inter, df, chain_gi, hop_glue, intra_gi_ready = false;
    DprnnWalk(dpdf_model* m_, const std::vector<DprnnW>& blocks_, float* xin_, float* xa_, float* xb_, float* hcat_, float* hin_, DevBuf& gibuf_, int Fp_,
              float* state_, long S_, int soff_, int B_, int Tc_)
        : m(m_), blocks(blocks_), xin(xin_), xa(xa_), xb(xb_), hcat(hcat_), hin(hin_), gibuf(gibuf_), Fp(Fp_), state(state_), S(S_), soff(soff_), B(B_), Tc(Tc_) {
        M = B * Tc * Fp;
        x = xin; y = xa;
        can_fuse = (Fp % 4 == 0) && m->fuse_dprnn != 0;
        fuse_intra = can_fuse && (m->fuse_dprnn == 2 || (long)B * Tc >= 3072);
        fuse_inter = can_fuse && (m->fuse_dprnn == 2 || (long)B * Fp >= m->inter_fuse_rows);
        gi_intra = !fuse_intra && m->hoist_gi && (size_t)M * 384 <= gibuf.n;
        gi_inter = !fuse_inter && m->hoist_gi && Tc >= 4 && (size_t)M * 192 <= gibuf.n;
        df = Fp >= 48;
        // small batches: each fc + LayerNorm GEMM also computes the input projection of the recurrence that follows it
        // (fcln_gi.h) -- two dependent launches per block fewer
        chain_gi = m->fcln_gi != 0;
        hop_glue = m->hop_glue && Tc == 1 && gi_intra;      // one frame per stream: everything between two intra scans in one launch
    }
    size_t size() const { return blocks.size(); }
    float* result() const { return x; }
    HopGlueArgs glue_args(size_t bi) const {
        const DprnnW& w = blocks[bi];
        const bool next = bi + 1 < blocks.size();
        return HopGlueArgs{hcat, x, y, m->C(w.fci_frag), m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b),
                           m->C(w.inter.wfrag), m->C(w.inter.bias), state + soff + (long)bi * Fp * 64, S, 64, Fp,
                           m->C(w.fce_frag), m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b),
                           gibuf.p, next ? m->C(blocks[bi + 1].intra.ih_frag) : nullptr, next ? m->C(blocks[bi + 1].intra.ih_bias) : nullptr, M};
    }
    void after_glue(size_t bi) {
        intra_gi_ready = bi + 1 < blocks.size();
        float* freed = x == xin ? xb : x;
        x = y; y = freed;
    }
    // single-hop streaming: the intra-band scan and the glue behind it as ONE launch (dprnn_hop_block.h); false: not available
    bool hop_block(size_t bi, const Gru64Args& ai) {
        Lane& L = *m->ln;
        const int br = df ? 0 : 1, nx = (ai.nrows + 3) / 4;
        if (!m->d_err) {
            if (hipMalloc((void**)&m->d_err, sizeof(int)) != hipSuccess) return false;
            (void)hipMemsetAsync(m->d_err, 0, sizeof(int), m->cur);
        }
        if (L.hop_flags_n[br] < 2 * nx || L.hop_epoch[br] > 0xF0000000u) {
            if (L.hop_flags_n[br] < 2 * nx) {
                if (L.hop_flags[br]) { L.sync_all(); (void)hipFree(L.hop_flags[br]); L.hop_flags[br] = nullptr; L.hop_flags_n[br] = 0; }
                const int n = std::max(2 * nx, 64);
                if (hipMalloc((void**)&L.hop_flags[br], (size_t)n * sizeof(unsigned)) != hipSuccess) return false;
                L.hop_flags_n[br] = n;
            }
            (void)hipMemsetAsync(L.hop_flags[br], 0, (size_t)L.hop_flags_n[br] * sizeof(unsigned), m->cur);
            L.hop_epoch[br] = 0;
        }
        const DprnnW& w = blocks[bi];
        ProfScope ps(m, "dprnn_hop_block");
        const bool next = bi + 1 < blocks.size();
        unsigned* done = nullptr;
        if (!next && !df && L.join_want) {
            if (!L.join_ctr) {
                if (hipMalloc((void**)&L.join_ctr, sizeof(unsigned)) != hipSuccess) return false;
                // zeroed and VISIBLE before anything can poll or bump it (this stack's launch and stage 2's emb_in sit on two streams)
                if (hipMemset(L.join_ctr, 0, sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
                L.join_total = 0;
            }
            done = L.join_ctr; L.join_total += (unsigned)((M + 15) / 16); L.join_armed = true;
        }
        HopBlockArgs ha{ai, m->C(w.intra.hh4), (const float*)gibuf.p, 384, glue_args(bi), L.hop_flags[br], ++L.hop_epoch[br], nx, Fp, m->d_err, done};
        const unsigned grid = (unsigned)(2 * nx + (M + 15) / 16);
        if (next) hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_block_kernel<true>), dim3(grid), dim3(512), 0, m->cur, ha);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_block_kernel<false>), dim3(grid), dim3(512), 0, m->cur, ha);
        after_glue(bi);
        return true;
    }
    // single-hop streaming: ALL blocks of this stack as one persistent launch (dprnn_hop_stack.h); false: not available here
    bool stack_ok() const {
        if (!(m->hop_stack && hop_glue && m->glue8 && m->hop_fused && m->use_gru256_cluster)) return false;
        if (blocks.empty() || blocks.size() > (size_t)HOP_STACK_MAX_BLOCKS || Fp > 48 || Fp % 8 || Tc != 1) return false;
        // every workgroup of BOTH stacks of the hop must be resident at once (one 512-thread workgroup per CU)
        const int wgs = 2 * ((B + 3) / 4) + B;
        return 2 * wgs <= m->n_cus && (size_t)M * 384 < (1u << 30);
    }
    bool stack() {
        Lane& L = *m->ln;
        const int br = df ? 0 : 1, S_ = B, nbk = (int)blocks.size();
        if (!m->d_err) {
            if (hipMalloc((void**)&m->d_err, sizeof(int)) != hipSuccess) return false;
            (void)hipMemsetAsync(m->d_err, 0, sizeof(int), m->cur);
        }
        if (L.hs_M[br] < M || L.hs_S[br] < S_ || L.hs_epoch[br] > 0xF0000000u) {
            L.sync_all();
            if (L.hs_hcat[br]) (void)hipFree(L.hs_hcat[br]);
            if (L.hs_gi[br]) (void)hipFree(L.hs_gi[br]);
            if (L.hs_flags[br]) (void)hipFree(L.hs_flags[br]);
            L.hs_hcat[br] = nullptr; L.hs_gi[br] = nullptr; L.hs_flags[br] = nullptr; L.hs_M[br] = L.hs_S[br] = 0;
            const int Mc = std::max(M, L.hs_M[br]), Sc = std::max(S_, L.hs_S[br]);
            if (hipMalloc((void**)&L.hs_hcat[br], (size_t)2 * Mc * 128 * sizeof(unsigned long long)) != hipSuccess ||
                hipMalloc((void**)&L.hs_gi[br], (size_t)2 * Mc * 384 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&L.hs_flags[br], (size_t)Sc * 4 * sizeof(unsigned)) != hipSuccess) return false;
            // epochs start at 1: zeroed granules and flags are "never written"
            if (hipMemset(L.hs_hcat[br], 0, (size_t)2 * Mc * 128 * sizeof(unsigned long long)) != hipSuccess ||
                hipMemset(L.hs_flags[br], 0, (size_t)Sc * 4 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
            L.hs_M[br] = Mc; L.hs_S[br] = Sc; L.hs_epoch[br] = 1;
        }
        if (!intra_gi_ready) {        // block 0's input projection (the encoder launch did not bring it along)
            const DprnnW& w0 = blocks[0];
            PlainA<64> ap{x, 64, 0, 64};
            BiasActStore<4> ep{gibuf.p, 384, 64, m->C(w0.intra.ih_bias), 64, 64, ACT_NONE};
            launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w0.intra.ih_frag), ep, M, 64, 6);
        }
        ProfScope ps(m, "dprnn_hop_stack");
        HopStackArgs ha{};
        for (int bi = 0; bi < nbk; ++bi) {
            const DprnnW& w = blocks[bi];
            const bool next = bi + 1 < nbk;
            ha.blk[bi] = HopStackBlock{m->C(w.intra.hh4), m->C(w.intra.bias), m->C(w.fci_frag), m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b),
                                       m->C(w.inter.wfrag), m->C(w.inter.bias), state + soff + (long)bi * Fp * 64,
                                       m->C(w.fce_frag), m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b),
                                       next ? m->C(blocks[bi + 1].intra.ih_frag) : nullptr, next ? m->C(blocks[bi + 1].intra.ih_bias) : nullptr};
        }
        ha.nb = nbk; ha.S = S_; ha.Fp = Fp; ha.h_hi = S;
        ha.x0 = x; ha.gi0 = gibuf.p; ha.gi = L.hs_gi[br]; ha.hcat = L.hs_hcat[br]; ha.y_out = xa == x ? xb : xa;
        ha.gi_flags = L.hs_flags[br]; ha.epoch0 = L.hs_epoch[br]; L.hs_epoch[br] += (unsigned)nbk;
        ha.err = m->d_err; ha.done = nullptr;
        if (!df && L.join_want) {
            if (!L.join_ctr) {
                if (hipMalloc((void**)&L.join_ctr, sizeof(unsigned)) != hipSuccess) return false;
                if (hipMemset(L.join_ctr, 0, sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
                L.join_total = 0;
            }
            ha.done = L.join_ctr; L.join_total += (unsigned)S_; L.join_armed = true;
        }
        const unsigned grid = (unsigned)(2 * ((S_ + 3) / 4) + S_);
        hipLaunchKernelGGL(dprnn_hop_stack_kernel, dim3(grid), dim3(512), 0, m->cur, ha);
        x = ha.y_out; intra_gi_ready = false;
        return true;
    }
    void block(size_t bi) {
        const DprnnW& w = blocks[bi];
        bool inter_gi_ready = false;
        Gru64Args ai{};     // intra-band bi-GRU over frequency, h0 = 0: rows = frames, steps = band positions
        ai.x = x; ai.wfrag = m->C(w.intra.wfrag); ai.bias = m->C(w.intra.bias); ai.hstate = nullptr;
        ai.nrows = B * Tc; ai.nsteps = Fp; ai.rdiv = 1;
        ai.x_hi = (long)Fp * 64; ai.x_lo = 0; ai.x_step = 64;
        if (fuse_intra && (m->gru64_limbs & 1)) {
            // bf16-limb kernels (gru_limb.h): the forward scan leaves pf = W_fc[:, 0:64] hf in `hin`, the backward scan adds its own half
            ai.out = hin; ai.ndirs = 1; ai.o_hi = (long)Fp * 64; ai.o_lo = 0; ai.o_step = 64; ai.o_dir_off = 0;
            {
                ProfScope ps(m, df ? "gru64_l3_kernel<0>/intra_fwd_df" : "gru64_l3_kernel<0>/intra_fwd_erb");
                Gru64LArgs la{ai, (const uint4*)m->C(w.intra.wl), (const uint4*)m->C(w.fci_lf), nullptr, nullptr, nullptr, nullptr, nullptr};
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<0>), dim3((ai.nrows + 15) / 16), dim3(256), 0, m->cur, la);
            }
            {
                ProfScope ps(m, df ? "gru64_l3_kernel<2>/intra_bwd_df" : "gru64_l3_kernel<2>/intra_bwd_erb");
                Gru64LArgs la{ai, (const uint4*)m->C(w.intra.wl), (const uint4*)m->C(w.fci_lb), m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b), hin, y};
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<2>), dim3((ai.nrows + 15) / 16), dim3(256), 0, m->cur, la);
            }
        } else if (fuse_intra) {
            {   // forward direction: plain scan, hf -> `hin` scratch [rows][Fp][64]
                ProfScope ps(m, df ? "gru64_scan_kernel/intra_fwd_df" : "gru64_scan_kernel/intra_fwd_erb");
                ai.out = hin; ai.ndirs = 1; ai.o_hi = (long)Fp * 64; ai.o_lo = 0; ai.o_step = 64; ai.o_dir_off = 0;
                hipLaunchKernelGGL(gru64_scan_kernel, dim3((ai.nrows + 15) / 16, 1), dim3(256), 0, m->cur, ai);
            }
            {   // backward direction + fc_intra + ln_intra + residual
                ProfScope ps(m, df ? "gru64_epi_kernel<2>/intra_bwd_df" : "gru64_epi_kernel<2>/intra_bwd_erb");
                Gru64EpiArgs ea{ai, m->C(w.fci_epi), m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b), hin, y};
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_epi_kernel<2>), dim3((ai.nrows + 15) / 16), dim3(256), 0, m->cur, ea);
            }
        } else {
            ai.out = hcat; ai.ndirs = 2;
            ai.o_hi = (long)Fp * 128; ai.o_lo = 0; ai.o_step = 128; ai.o_dir_off = 64;
            if (gi_intra) {     // W_ih x for every (frame, band) in one GEMM, then the h-only scan
                ProfScope ps(m, df ? "gru64_scan_gi_kernel/intra_df" : "gru64_scan_gi_kernel/intra_erb");
                if (!intra_gi_ready) {
                    PlainA<64> ap{x, 64, 0, 64};
                    BiasActStore<4> ep{gibuf.p, 384, 64, m->C(w.intra.ih_bias), 64, 64, ACT_NONE};
                    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.intra.ih_frag), ep, M, 64, 6);
                }
                const bool scan4 = ((ai.nrows + 3) / 4) * 2 <= m->scan4_max_wgs && (size_t)M * 384 < (1u << 30);     // (32-bit lane offsets in the kernel)
                // (use_gru256_cluster = 0 is the recovery re-run: no kernel that waits for another workgroup)
                if (scan4 && hop_glue && m->glue8 && m->hop_fused && m->use_gru256_cluster && hop_block(bi, ai)) return;
                if (scan4)
                    hipLaunchKernelGGL(gru64_scan4_gi_kernel, dim3((ai.nrows + 3) / 4, 2), dim3(256), 0, m->cur, ai, m->C(w.intra.hh4), (const float*)gibuf.p, 384);
                else
                    hipLaunchKernelGGL(gru64_scan_gi_kernel, dim3((ai.nrows + 15) / 16, 2), dim3(256), 0, m->cur, ai, (const float*)gibuf.p, 384);
            } else {
                ProfScope ps(m, df ? "gru64_scan_kernel/intra_df" : "gru64_scan_kernel/intra_erb");
                hipLaunchKernelGGL(gru64_scan_kernel, dim3((ai.nrows + 15) / 16, 2), dim3(256), 0, m->cur, ai);
            }
            if (hop_glue) {
                // single-hop streaming: fc_intra + LN, the inter-band GRUCell step, fc_inter + LN and the next block's
                // intra input projection as ONE launch (fcln_gi.h); the block output goes to y, x0 becomes the free buffer
                ProfScope ps(m, "dprnn_hop_glue");
                const bool next = bi + 1 < blocks.size();
                const HopGlueArgs ha = glue_args(bi);
                if (m->glue8) {      // eight waves per tile: half the dependent MFMAs and operand loads per wave (fcln_gi.h)
                    if (next) hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue8_kernel<true>), dim3((M + 15) / 16), dim3(512), 0, m->cur, ha);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue8_kernel<false>), dim3((M + 15) / 16), dim3(512), 0, m->cur, ha);
                }
                else if (next) hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue_kernel<true>), dim3((M + 15) / 16), dim3(256), 0, m->cur, ha);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue_kernel<false>), dim3((M + 15) / 16), dim3(256), 0, m->cur, ha);
                after_glue(bi);
                return;
            }
            {   // fc_intra + ln_intra + residual (+ the inter-band cell's input projection)
                ProfScope ps(m, "dprnn_fc_ln");
                if (chain_gi && gi_inter) {
                    FclnGiArgs fa{hcat, 128, x, y, m->C(w.fci_frag), m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b), gibuf.p, 192,
                                  m->C(w.inter.ih_frag), m->C(w.inter.ih_bias), M};
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(fcln_gi_kernel<128, 3>), dim3((M + 15) / 16), dim3(256), 0, m->cur, fa);
                    inter_gi_ready = true;
                } else {
                    PlainA<128> ap{hcat, 128, 0, 128};
                    LnResStore ep{y, x, m->C(w.fci_b), m->C(w.lni_g), m->C(w.lni_b)};
                    launch_gemm_rows<4, 128, true>(m->cur, ap, m->C(w.fci_frag), ep, M, 128, 1);
                }
            }
        }
        intra_gi_ready = false;
        std::swap(x, y);
        if (y == xin) y = xb;
        Gru64Args ae{};     // inter-band GRUCell over time, one hidden state per band position
        ae.x = x; ae.wfrag = m->C(w.inter.wfrag); ae.bias = m->C(w.inter.bias);
        ae.hstate = state + soff + (long)bi * Fp * 64;
        ae.nrows = B * Fp; ae.nsteps = Tc; ae.ndirs = 1; ae.rdiv = Fp;
        ae.x_hi = (long)Tc * Fp * 64; ae.x_lo = 64; ae.x_step = (long)Fp * 64;
        ae.o_hi = ae.x_hi; ae.o_lo = 64; ae.o_step = ae.x_step; ae.o_dir_off = 0;
        ae.h_hi = S; ae.h_lo = 64;
        if (fuse_inter && (m->gru64_limbs & 2)) {
            ProfScope ps(m, df ? "gru64_l3_kernel<1>/inter_df" : "gru64_l3_kernel<1>/inter_erb");
            ae.out = nullptr;
            Gru64LArgs la{ae, (const uint4*)m->C(w.inter.wl), (const uint4*)m->C(w.fce_l), m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b), nullptr, y};
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_l3_kernel<1>), dim3((ae.nrows + 15) / 16), dim3(256), 0, m->cur, la);
        } else if (fuse_inter) {   // inter scan + fc_inter + ln_inter + residual
            ProfScope ps(m, df ? "gru64_epi_kernel<1>/inter_df" : "gru64_epi_kernel<1>/inter_erb");
            ae.out = nullptr;
            Gru64EpiArgs ea{ae, m->C(w.fce_epi), m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b), nullptr, y};
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gru64_epi_kernel<1>), dim3((ae.nrows + 15) / 16), dim3(256), 0, m->cur, ea);
        } else {
            ae.out = hin;
            if (gi_inter) {
                ProfScope ps(m, df ? "gru64_scan_gi_kernel/inter_df" : "gru64_scan_gi_kernel/inter_erb");
                if (!inter_gi_ready) {
                    PlainA<64> ap{x, 64, 0, 64};
                    BiasActStore<4> ep{gibuf.p, 192, 64, m->C(w.inter.ih_bias), 64, 64, ACT_NONE};
                    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.inter.ih_frag), ep, M, 64, 3);
                }
                if ((ae.nrows + 3) / 4 <= m->scan4_max_wgs && (size_t)M * 384 < (1u << 30))
                    hipLaunchKernelGGL(gru64_scan4_gi_kernel, dim3((ae.nrows + 3) / 4, 1), dim3(256), 0, m->cur, ae, m->C(w.inter.hh4), (const float*)gibuf.p, 192);
                else
                    hipLaunchKernelGGL(gru64_scan_gi_kernel, dim3((ae.nrows + 15) / 16, 1), dim3(256), 0, m->cur, ae, (const float*)gibuf.p, 192);
            } else {
                ProfScope ps(m, df ? "gru64_scan_kernel/inter_df" : "gru64_scan_kernel/inter_erb");
                hipLaunchKernelGGL(gru64_scan_kernel, dim3((ae.nrows + 15) / 16, 1), dim3(256), 0, m->cur, ae);
            }
            {   // fc_inter + ln_inter + residual (+ the next block's intra-band input projection)
                ProfScope ps(m, "dprnn_fc_ln");
                if (chain_gi && gi_intra && bi + 1 < blocks.size()) {
                    const DprnnW& wn = blocks[bi + 1];
                    FclnGiArgs fa{hin, 64, x, y, m->C(w.fce_frag), m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b), gibuf.p, 384,
                                  m->C(wn.intra.ih_frag), m->C(wn.intra.ih_bias), M};
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(fcln_gi_kernel<64, 6>), dim3((M + 15) / 16), dim3(256), 0, m->cur, fa);
                    intra_gi_ready = true;
                } else {
                    PlainA<64> ap{hin, 64, 0, 64};
                    LnResStore ep{y, x, m->C(w.fce_b), m->C(w.lne_g), m->C(w.lne_b)};
                    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.fce_frag), ep, M, 64, 1);
                }
            }
        }
        std::swap(x, y);
        if (y == xin) y = xb;
    }
};
float* run_dprnn(dpdf_model* m, const std::vector<DprnnW>& blocks, float* xin, float* xa, float* xb, float* hcat, float* hin, DevBuf& gibuf, int Fp,
                 float* state, long S, int soff, int B, int Tc) {
    DprnnWalk wk(m, blocks, xin, xa, xb, hcat, hin, gibuf, Fp, state, S, soff, B, Tc);
    for (size_t bi = 0; bi < wk.size(); ++bi) wk.block(bi);
    return wk.result();
}

template <int S>
void run_dwconv(dpdf_model* m, const SepConvW& w, TView in, TView out, int B, int Tc) {
    RowMap rm = RowMap::make(Tc, out.Fp);
    DwConvA<S> ap{in, rm, m->C(w.dw)};
    BiasReluToView ep{out, rm, m->C(w.bias)};
    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.pwfrag), ep, B * Tc * out.Fp, 64, 1);
}
void run_dwconv_s(dpdf_model* m, const SepConvW& w, TView in, TView out, int B, int Tc, int stride) {
    if (stride == 1) run_dwconv<1>(m, w, in, out, B, Tc);
    else if (stride == 2) run_dwconv<2>(m, w, in, out, B, Tc);
    else run_dwconv<3>(m, w, in, out, B, Tc);
}
template <int S>
void run_subpix(dpdf_model* m, const SepConvW& w, const PathW& p, TView e, TView prev, TView out, int B, int Tc) {
    RowMap rm = RowMap::make(Tc, out.Fp);
    SubpixA<S> ap{e, prev, rm, m->C(p.ps), m->C(p.pb), m->C(w.dw)};
    BiasReluToView ep{out, rm, m->C(w.bias)};
    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.pwfrag), ep, B * Tc * out.Fp, 64, 1);
}
// convt1 with the mask head's 64 -> 1 contraction in its epilogue (MaskSumEpi): d1 never reaches HBM
template <int S>
void run_subpix_mask(dpdf_model* m, const SepConvW& w, const PathW& p, TView e, TView prev, const float* e0, float* ssum, int Fo, int B, int Tc) {
    RowMap rm = RowMap::make(Tc, Fo);
    SubpixA<S> ap{e, prev, rm, m->C(p.ps), m->C(p.pb), m->C(w.dw)};
    MaskSumEpi ep{e0, ssum, m->C(w.bias), m->C(m->conv0p.ps), m->C(m->conv0p.pb), m->C(m->c0out_w)};
    launch_gemm_rows<4, 64, true>(m->cur, ap, m->C(w.pwfrag), ep, B * Tc * Fo, 64, 1);
}
void run_subpix_s(dpdf_model* m, const SepConvW& w, const PathW& p, TView e, TView prev, TView out, int B, int Tc, int s) {
    if (s == 1) run_subpix<1>(m, w, p, e, prev, out, B, Tc);
    else if (s == 2) run_subpix<2>(m, w, p, e, prev, out, B, Tc);
    else run_subpix<3>(m, w, p, e, prev, out, B, Tc);
}

// ------------------------------------------------------------------------------------------------
// One chunk of the frame function for B streams x Tc frames, as a two-stage pipeline.
//   raw:   unnormalised spec, frame t of clip b at raw + b*raw_clip_stride + t*F*2
//   state: device [B][S] reference flat layout, updated in place
//   out:   enhanced spec, frame (out_t0 + t) of clip b at out + b*out_clip_stride + ...
// stage 1 (main stream; ERB branch forked onto stream_c): features, encoder convs, both DPRNNs.
// stage 2 (stream_b): embedding GRU, both decoders, mask, deep filter -- mostly latency-bound
//   256-wide GRU scans that occupy 64 of the 256 CUs, so stage 2 of chunk i runs UNDER stage 1 of
//   chunk i+1.  The two stages touch disjoint segments of the flat state and disjoint temporaries;
//   the tensors that cross (XSet) are double-buffered by chunk parity.
// ------------------------------------------------------------------------------------------------
struct ChunkArgs {
    const float* raw; size_t raw_clip_stride; int B, Tc; float* state;
    float* out; size_t out_clip_stride; int out_t0; const float* attn_raw; float alpha;
    int parity;
};

StateIoArgs make_sio(dpdf_model* m, const ChunkArgs& c, XSet& x) {
    const dpdf_dims& d = m->d; const dpdf_state_layout& L = m->L; Workspace& w = m->ln->ws;
    return StateIoArgs{c.state, (long)d.state_size, w.feat_erb.p, w.feat_spec.p, x.c0.p, x.xs.p, w.coefs.p, w.xm.p,
                       L.erb_conv0_buf, L.df_conv0_buf, L.df_convp_buf, L.mask_buf, L.df_coefs_buf, L.df_spec_buf,
                       c.B, c.Tc, d.E, d.D, d.F, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0};
}

int run_stage1(dpdf_model* m, const ChunkArgs& c) {
    const dpdf_dims& d = m->d; const dpdf_state_layout& L = m->L;
    Workspace& w = m->ln->ws; XSet& x = w.x[c.parity];
    const int B = c.B, Tc = c.Tc, BT = B * Tc;
    const long S = d.state_size;
    float* state = c.state;
    hipStream_t sA = m->ln->sA, sC = (m->overlap & 2) ? m->ln->sC : m->ln->sA;
    m->cur = sA;
    if (m->ln->s2_pending[c.parity]) {      // stage 2 of chunk i-2 must be done with this XSet
        HIP_TRY(hipStreamWaitEvent(sA, m->ln->ev_s2[c.parity], 0));
        m->ln->s2_pending[c.parity] = false;
    }
    StateIoArgs sio = make_sio(m, c, x);
    sio.seg_lo = 0; sio.seg_hi = m->ln->single_chunk ? 6 : 4;      // erb_conv0 / df_conv0 / mask(spec) / df_convp FIFOs (+ stage 2's two in a one-chunk call: one launch less on its chain)
    if (m->ln->s1_imported) {       // a streaming hop's prologue launch did it (streams_enqueue)
        m->ln->s1_imported = false;
    } else {
        ProfScope ps(m, "state_io");
        // (a streaming call's pre-call copy of the state rides along in the first import of the call: StateIoArgs.snap)
        if (m->snap_dst) { sio.snap = m->snap_dst; sio.snap_y = 4; m->snap_dst = nullptr; }
        hipLaunchKernelGGL(state_io_kernel, dim3(B, sio.seg_hi + sio.snap_y, 5), dim3(256), 0, sA, sio);
        sio.snap = nullptr; sio.snap_y = 0;
    }
    sio.seg_hi = 4;
    {
        ProfScope ps(m, "features");
        FeatAArgs fa{c.raw, c.raw_clip_stride, x.xs.p, w.feat_erb.p, d.is48 ? nullptr : m->iconsts, B, Tc, d.F, d.E, d.is48, d.wnorm};
        FeatBArgs fb{w.feat_erb.p, x.xs.p, w.feat_spec.p, state, S, L.erb_norm, L.spec_norm, B, Tc, d.F, d.E, d.D};
        if (Tc == 1 && m->hx.armed) {      // a streaming hop: one launch, with the front end's chores folded in
            FeatHopArgs fh{fa, fb, m->hx.part, m->hx.ks, m->hx.W, m->hx.pcm_new, m->hx.in_tail, m->hx.snap_in, d.hop};
            hipLaunchKernelGGL(feat_hop_kernel, dim3(B), dim3(256), 0, sA, fh);
            m->hx = dpdf_model::HopExtras{};
        } else {
            hipLaunchKernelGGL(feat_a_kernel, dim3(BT), dim3(256), 0, sA, fa);
            int nth = ((d.E + d.D + 63) / 64) * 64;
            hipLaunchKernelGGL(feat_b_kernel, dim3(B), dim3(nth), 0, sA, fb);
        }
    }
    // Two independent chains from here: the DF branch on the main stream, the ERB branch on its own.  They are ENQUEUED
    // alternately, block by block (DprnnWalk): in the latency regime the host is only just ahead of the GPU (~3 us per launch),
    // and a branch whose ~20 launches are enqueued behind the other one's starts that much later -- with 48 band positions
    // against 40 (48 kHz) both chains are critical.  The fork point is the same either way.
    DprnnWalk wdf(m, m->dprnn_df, x.c1.p, x.xd_a.p, x.xd_b.p, w.hcat.p, w.hin.p, w.gi64, d.Fd, state, S, L.dprnn_df, B, Tc);
    DprnnWalk werb(m, m->dprnn_erb, x.e3.p, x.xe_a.p, x.xe_b.p, w.hcat_e.p, w.hin_e.p, w.gi64_e, d.F3, state, S, L.dprnn_erb, B, Tc);
    if (sC != sA) { HIP_TRY(hipEventRecord(m->ln->ev_fk[c.parity], sA)); HIP_TRY(hipStreamWaitEvent(sC, m->ln->ev_fk[c.parity], 0)); }
    // A streaming hop: the first kernel of stage 2 (emb_in, on this stream) waits for the ERB stack's last block by a counter that
    // block's tiles bump (DprnnWalk::hop_block) -- a kernel that waits for an EVENT of another stream starts ~10 us after it
    m->ln->join_want = m->hop_spin_join && g_live_models.load() == 1 && sC != sA && m->ln->single_chunk && Tc == 1 && d.nb > 0 && m->fuse_small && m->fuse_gl &&
                       BT <= SMALL_M_ROWS && small_gl_dims(m);
    m->ln->join_armed = false;
    // the small-launch forms of the two front ends (enc_seg.h), with the first DPRNN block's input projection riding along
    const bool small_enc = m->fuse_small && m->fuse_enc && BT <= m->enc_seg_rows;
    const bool df_seg_ok = small_enc && d.D == 2 * d.Fd && d.Fd % 16 == 0;
    x.have_pconv = m->df_ring && B * 3 >= 192;
    const bool df_seg = df_seg_ok && !x.have_pconv;
    const bool erb_exact = d.F2 == d.F3 * d.s3 && d.F1 == d.F2 * d.s2 && d.Ec == d.F1 * d.s1 && d.F3 % 8 == 0;
    const int erb_geo = !(small_enc && erb_exact) ? 0 : (d.s1 == 2 && d.s2 == 2 && d.s3 == 1) ? 16 : (d.s1 == 3 && d.s2 == 2 && d.s3 == 2) ? 48 : 0;
    const bool df_gi = d.nb > 0 && wdf.gi_intra, erb_gi = d.nb > 0 && werb.gi_intra;
    DfEncArgs dfa{w.feat_spec.p, x.c0.p, x.c1.p, df_gi ? w.gi64.p : nullptr, m->C(m->dfc0_pwfrag), m->C(m->dfc0_bias),
                  m->C(m->df_conv1.dw), m->C(m->df_conv1.pwfrag), m->C(m->df_conv1.bias),
                  df_gi ? m->C(m->dprnn_df[0].intra.ih_frag) : nullptr, df_gi ? m->C(m->dprnn_df[0].intra.ih_bias) : nullptr, B, Tc, d.D, d.Fd,
                  nullptr, m->C(m->convp_frag), m->C(m->convp_bias)};
    if (df_seg && Tc == 1 && m->hop_pconv) {      // a streaming hop: the DF decoder's pathway conv rides along (stage 2's df_out epilogue adds it)
        dfa.p = x.pconv.p; x.have_pconv = true;
    }
    ErbEncArgs era{w.feat_erb.p, x.e0.p, x.e1.p, x.e2.p, x.e3.p, m->C(m->conv0_w), m->C(m->conv0_b),
                   m->C(m->erb_conv1.dw), m->C(m->erb_conv1.pwfrag), m->C(m->erb_conv1.bias),
                   m->C(m->erb_conv2.dw), m->C(m->erb_conv2.pwfrag), m->C(m->erb_conv2.bias),
                   m->C(m->erb_conv3.dw), m->C(m->erb_conv3.pwfrag), m->C(m->erb_conv3.bias), B, Tc, d.E, d.Ec, d.F1, d.F2, d.F3,
                   erb_gi ? w.gi64_e.p : nullptr, erb_gi ? m->C(m->dprnn_erb[0].intra.ih_frag) : nullptr, erb_gi ? m->C(m->dprnn_erb[0].intra.ih_bias) : nullptr};
    // ---- encoder, DF branch (dpdfnet.py:221-234) on the main stream ----
    m->cur = sA;
    TView c0v{x.c0.p, Tc + 4, 4, d.D, 64}, c1v{x.c1.p, Tc, 0, d.Fd, 64};
    {
        ProfScope ps(m, "enc_convs_df");
        // df_conv1 (+ the DF decoder's pathway conv, + df_conv0 itself): one time-walking pass when clips x 3 workgroups
        // fill the chip, else the time-parallel gemm_rows forms (df_ring.h)
        if (df_seg) {       // latency regime: df_conv0 + df_conv1 + the first block's input projection (+ pathway conv) as one launch (enc_seg.h)
            if (dfa.p) hipLaunchKernelGGL(HIP_KERNEL_NAME(df_enc_seg_kernel<true>), dim3(d.Fd / 16, BT), dim3(256), 0, sA, dfa);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(df_enc_seg_kernel<false>), dim3(d.Fd / 16, BT), dim3(256), 0, sA, dfa);
            wdf.intra_gi_ready = df_gi;
        } else {
            const bool conv0_in_ring = x.have_pconv && m->df_ring >= 2;
            if (!conv0_in_ring) {
                RowMap rm = RowMap::make(Tc, d.D);
                Conv0DfA ap{w.feat_spec.p, Tc + 2, d.D, rm};
                BiasReluToView ep{c0v, rm, m->C(m->dfc0_bias)};
                launch_gemm_rows<4, 32, true>(sA, ap, m->C(m->dfc0_pwfrag), ep, BT * d.D, 32, 1);
            }
            if (x.have_pconv) {
                DfRingArgs ra{x.c0.p, x.c1.p, x.pconv.p, m->C(m->df_conv1.dw), m->C(m->df_conv1.pwfrag), m->C(m->df_conv1.bias),
                              m->C(m->convp_frag), m->C(m->convp_bias), B, Tc, w.feat_spec.p, m->C(m->dfc0_pwfrag), m->C(m->dfc0_bias)};
                if (conv0_in_ring) hipLaunchKernelGGL(HIP_KERNEL_NAME(df_ring_kernel<true>), dim3(B * 3), dim3(256), 0, sA, ra);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(df_ring_kernel<false>), dim3(B * 3), dim3(256), 0, sA, ra);
            } else {
                run_dwconv_s(m, m->df_conv1, c0v, c1v, B, Tc, 2);
            }
        }
    }
    // ---- encoder, ERB branch (reference onnx_model/dpdfnet.py:206-219) on its own stream ----
    m->cur = sC;
    TView e0v{x.e0.p, Tc, 0, d.Ec, 64}, e1v{x.e1.p, Tc, 0, d.F1, 64}, e2v{x.e2.p, Tc, 0, d.F2, 64}, e3v{x.e3.p, Tc, 0, d.F3, 64};
    {
        ProfScope ps(m, "enc_convs_erb");
        if (erb_geo) {      // latency regime: four dependent launches -> one (enc_seg.h)
            if (erb_geo == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(erb_enc_seg_kernel<2, 2, 1, 8>), dim3(d.F3 / 8, BT), dim3(256), 0, sC, era);
            // 48 kHz: segments of 10 positions once segments of 8 would be more workgroups than CUs (64 streams: 256 instead of 320 --
            // two workgroups sharing a CU's matrix pipes take twice as long, the launch ends with the slowest)
            else if (m->seg10 && d.F3 % 10 == 0 && (long)BT * (d.F3 / 8) > 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(erb_enc_seg_kernel<3, 2, 2, 10>), dim3(d.F3 / 10, BT), dim3(256), 0, sC, era);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(erb_enc_seg_kernel<3, 2, 2, 8>), dim3(d.F3 / 8, BT), dim3(256), 0, sC, era);
            werb.intra_gi_ready = erb_gi;
        } else {
            Conv0ErbArgs ca{w.feat_erb.p, x.e0.p, m->C(m->conv0_w), m->C(m->conv0_b), B, Tc, d.E, d.Ec};
            size_t rows16 = ((size_t)BT * d.Ec + 15) / 16;
            hipLaunchKernelGGL(conv0_erb_kernel, dim3((unsigned)std::min<size_t>(rows16, 8192)), dim3(256), 0, sC, ca);
            run_dwconv_s(m, m->erb_conv1, e0v, e1v, B, Tc, d.s1);
            run_dwconv_s(m, m->erb_conv2, e1v, e2v, B, Tc, d.s2);
            run_dwconv_s(m, m->erb_conv3, e2v, e3v, B, Tc, d.s3);
        }
    }
    x.c1d = x.c1.p; x.e3d = x.e3.p;
    if (d.nb > 0) {
        // a streaming hop whose stacks fit the chip: one persistent launch per stack (both must take this form: they share the CUs)
        bool stacked = false;
        if (wdf.stack_ok() && werb.stack_ok()) {
            m->cur = sA; const bool a_ = wdf.stack();
            m->cur = sC; const bool b_ = a_ && werb.stack();
            stacked = a_ && b_;
            if (a_ && !b_) return set_err(DPDF_E_RUNTIME, "dprnn_hop_stack: allocation failed");
        }
        if (stacked) {
        } else
        if (m->interleave) {
            for (size_t bi = 0; bi < wdf.size(); ++bi) {
                m->cur = sA; wdf.block(bi);
                m->cur = sC; werb.block(bi);
            }
        } else {
            m->cur = sA; for (size_t bi = 0; bi < wdf.size(); ++bi) wdf.block(bi);
            m->cur = sC; for (size_t bi = 0; bi < werb.size(); ++bi) werb.block(bi);
        }
        x.c1d = wdf.result(); x.e3d = werb.result();
    }
    m->cur = sA;
    if (sC != sA) { HIP_TRY(hipEventRecord(m->ln->ev_jn[c.parity], sC)); if (!m->ln->join_armed) HIP_TRY(hipStreamWaitEvent(sA, m->ln->ev_jn[c.parity], 0)); }
    // stage 2 may start here: the FIFO export below only reads stage-1 tensors that stage 2 does not write, and writes state
    // segments stage 2 does not touch -- it runs beside the first kernels of stage 2 instead of in front of them
    if ((m->overlap & 1) && !m->ln->single_chunk) HIP_TRY(hipEventRecord(m->ln->ev_s1[c.parity], sA));      // (its only waiter: stage 2 on the stage-2 stream)
    if (!m->ln->single_chunk) {
        ProfScope ps(m, "state_io");
        sio.do_export = 1;
        hipLaunchKernelGGL(state_io_kernel, dim3(B, 4, 5), dim3(256), 0, sA, sio);
    }       // (a one-chunk call exports all six FIFOs in one launch at the end of stage 2)
    m->ln->dbg_e3d = x.e3d; m->ln->dbg_c1d = x.c1d; m->ln->dbg_B = B; m->ln->dbg_Tc = Tc; m->ln->dbg_parity = c.parity;
    HIP_TRY(hipGetLastError());
    return DPDF_OK;
}

// ERB decoder convs + mask head (reference onnx_model/dpdfnet.py:361-366) on stream st: dembp [B*Tc][F3][64] -> w.m
void run_dec_convs(dpdf_model* m, XSet& x, float* dembp, int B, int Tc, hipStream_t st) {
    const dpdf_dims& d = m->d; Workspace& w = m->ln->ws;
    const int BT = B * Tc;
    m->cur = st;
    TView e1v{x.e1.p, Tc, 0, d.F1, 64}, e2v{x.e2.p, Tc, 0, d.F2, 64}, e3v{x.e3.p, Tc, 0, d.F3, 64};
    ProfScope ps(m, "dec_convs");
    m->ln->mask_from_sums = false;
    TView dembv{dembp, Tc, 0, d.F3, 64};
    TView d3v{w.d3.p, Tc, 0, d.F2, 64}, d2v{w.d2.p, Tc, 0, d.F1, 64}, d1v{w.d1.p, Tc, 0, d.Ec, 64};
    const bool geo16 = m->fuse_mask && !d.is48 && d.s1 == 2 && d.s2 == 2 && d.s3 == 1 && d.Ec == 32 && d.F1 == 16 && d.F2 == 8 && d.F3 == 8;
    const bool geo48 = m->dec_seg && BT >= 1024 && d.is48 &&   // (few frames: the gemm_rows forms spread over more workgroups: 64 x 48 kHz streams, one hop 767 -> 753 us)
                        m->fuse_mask && d.s3 == 2 && d.s2 == 2 && d.s1 == 3 && d.F2 % 80 == 0 && d.F1 % 80 == 0 && d.Ec % 96 == 0;
    const bool exact = d.F2 == d.F3 * d.s3 && d.F1 == d.F2 * d.s2 && d.Ec == d.F1 * d.s1;
    const int pyr = !(m->fuse_small && m->fuse_dec && m->fuse_mask && exact && BT <= m->dec_pyr_rows) ? 0
                    : geo16 ? 16 : (d.is48 && d.s1 == 3 && d.s2 == 2 && d.s3 == 2 && d.F3 % 8 == 0) ? 48 : 0;
    if (pyr) {      // latency regime: the three stages + the mask head's tap sums as one launch (dec_pyr.h)
        DecPyrArgs pa{x.e3.p, dembp, x.e2.p, x.e1.p, x.e0.p, pyr == 48 ? w.d1.p : nullptr, pyr == 16 ? w.m.p : nullptr,
                      m->C(m->conv3p.ps), m->C(m->conv3p.pb), m->C(m->convt3.dw), m->C(m->convt3.pwfrag), m->C(m->convt3.bias),
                      m->C(m->conv2p.ps), m->C(m->conv2p.pb), m->C(m->convt2.dw), m->C(m->convt2.pwfrag), m->C(m->convt2.bias),
                      m->C(m->conv1p.ps), m->C(m->conv1p.pb), m->C(m->convt1.dw), m->C(m->convt1.pwfrag), m->C(m->convt1.bias),
                      m->C(m->conv0p.ps), m->C(m->conv0p.pb), m->C(m->c0out_w), m->c0out_bias, BT, d.F3, d.F2, d.F1, d.Ec, d.E};
        if (pyr == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_pyr_kernel<2, 2, 1, 8, true>), dim3(1, BT), dim3(256), 0, st, pa);
        else if (m->seg10 && d.F3 % 10 == 0 && (long)BT * (d.F3 / 8) > 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_pyr_kernel<3, 2, 2, 10, false>), dim3(d.F3 / 10, BT), dim3(256), 0, st, pa);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_pyr_kernel<3, 2, 2, 8, false>), dim3(d.F3 / 8, BT), dim3(256), 0, st, pa);
        m->ln->mask_from_sums = pyr == 48 && BT <= SMALL_M_ROWS;      // (the tap sums are finished inside mask_df_kernel)
        if (pyr == 48 && !m->ln->mask_from_sums) {
            MaskFinArgs mf{w.d1.p, w.m.p, m->c0out_bias, BT * d.Ec, d.Ec, d.E, d.is48};
            hipLaunchKernelGGL(mask_fin_kernel, dim3((BT * d.Ec + 255) / 256), dim3(256), 0, st, mf);
        }
        return;
    }
    if (geo48) {    // 48 kHz geometry: tiles of 80 / 80 / 96 output bands of one frame, inputs loaded once (dec_last.h: dec_seg_kernel)
        const long cap = 256 * 2 * 4;
        const bool pipe = m->dec_seg >= 2;
        const bool one = m->dec_seg >= 3 && BT >= m->dec_seg_all_frames;   // (few frames per workgroup: three launches fill and drain faster)
        auto grid = [&](long tiles) { return dim3((unsigned)std::min<long>(pipe ? BT : tiles, pipe ? m->dec_seg_grid : cap)); };
        DecSegArgs a3{x.e3.p, dembp, w.d3.p, m->C(m->conv3p.ps), m->C(m->conv3p.pb), m->C(m->convt3.dw), m->C(m->convt3.pwfrag), m->C(m->convt3.bias),
                      nullptr, nullptr, nullptr, nullptr, nullptr, BT, d.F2};
        DecSegArgs a2{x.e2.p, w.d3.p, w.d2.p, m->C(m->conv2p.ps), m->C(m->conv2p.pb), m->C(m->convt2.dw), m->C(m->convt2.pwfrag), m->C(m->convt2.bias),
                      nullptr, nullptr, nullptr, nullptr, nullptr, BT, d.F1};
        // convt1 + mask head: w.d1 holds the three tap sums per row ([rows][4]) instead of the 64-channel d1 rows
        DecSegArgs a1{x.e1.p, w.d2.p, nullptr, m->C(m->conv1p.ps), m->C(m->conv1p.pb), m->C(m->convt1.dw), m->C(m->convt1.pwfrag), m->C(m->convt1.bias),
                      x.e0.p, w.d1.p, m->C(m->conv0p.ps), m->C(m->conv0p.pb), m->C(m->c0out_w), BT, d.Ec};
        if (one && d.F2 == 80 && d.F1 == 160 && d.Ec == 480) {
            hipLaunchKernelGGL(dec_seg2_all_kernel, grid(BT), dim3(512), 0, st, a3, a2, a1);
        } else {
            if (pipe) hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), grid((long)BT * (d.F2 / 80)), dim3(512), 0, st, a3);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<2, 80, false>), grid((long)BT * (d.F2 / 80)), dim3(256), 0, st, a3);
            if (pipe) hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), grid((long)BT * (d.F1 / 80)), dim3(512), 0, st, a2);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<2, 80, false>), grid((long)BT * (d.F1 / 80)), dim3(256), 0, st, a2);
            if (pipe) hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<3, 96, true>), grid((long)BT * (d.Ec / 96)), dim3(512), 0, st, a1);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg_kernel<3, 96, true>), grid((long)BT * (d.Ec / 96)), dim3(256), 0, st, a1);
        }
        MaskFinArgs mf{w.d1.p, w.m.p, m->c0out_bias, BT * d.Ec, d.Ec, d.E, d.is48};
        hipLaunchKernelGGL(mask_fin_kernel, dim3((BT * d.Ec + 255) / 256), dim3(256), 0, st, mf);
        return;
    }
    if (geo16) {    // 16 kHz geometry: whole frames per 64-row tile, inputs loaded once (dec_last.h)
        const int cap = 256 * 3 * 4;
        DecStageArgs a3{x.e3.p, dembp, w.d3.p, m->C(m->conv3p.ps), m->C(m->conv3p.pb), m->C(m->convt3.dw), m->C(m->convt3.pwfrag), m->C(m->convt3.bias), BT};
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_stage_kernel<1, 8>), dim3(std::min((BT + 7) / 8, cap)), dim3(256), 0, st, a3);
        DecStageArgs a2{x.e2.p, w.d3.p, w.d2.p, m->C(m->conv2p.ps), m->C(m->conv2p.pb), m->C(m->convt2.dw), m->C(m->convt2.pwfrag), m->C(m->convt2.bias), BT};
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_stage_kernel<2, 16>), dim3(std::min((BT + 3) / 4, cap)), dim3(256), 0, st, a2);
    } else {
        run_subpix_s(m, m->convt3, m->conv3p, e3v, dembv, d3v, B, Tc, d.s3);
        run_subpix_s(m, m->convt2, m->conv2p, e2v, d3v, d2v, B, Tc, d.s2);
    }
    if (m->fuse_mask && !d.is48 && d.s1 == 2 && d.Ec == 32 && d.F1 == 16) {
        // 16 kHz geometry: last decoder stage + mask head in one kernel, m written directly (dec_last.h)
        DecLastArgs da{x.e1.p, w.d2.p, x.e0.p, w.m.p, m->C(m->conv1p.ps), m->C(m->conv1p.pb), m->C(m->convt1.dw),
                       m->C(m->convt1.pwfrag), m->C(m->convt1.bias), m->C(m->conv0p.ps), m->C(m->conv0p.pb),
                       m->C(m->c0out_w), m->c0out_bias, BT};
        const int ntiles = (BT + 1) / 2;
        hipLaunchKernelGGL(dec_last_kernel, dim3(std::min(ntiles, 256 * 3 * 4)), dim3(256), 0, st, da);
    } else if (m->fuse_mask) {
        // w.d1 holds the three tap sums per row ([rows][4]) instead of the 64-channel d1 rows
        if (d.s1 == 2) run_subpix_mask<2>(m, m->convt1, m->conv1p, e1v, d2v, x.e0.p, w.d1.p, d.Ec, B, Tc);
        else run_subpix_mask<3>(m, m->convt1, m->conv1p, e1v, d2v, x.e0.p, w.d1.p, d.Ec, B, Tc);
        // small launches at 48 kHz: the tap sums are finished inside mask_df_kernel (one launch fewer on the hop's chain)
        m->ln->mask_from_sums = m->fuse_small && d.is48 && BT <= SMALL_M_ROWS;
        if (!m->ln->mask_from_sums) {
            MaskFinArgs mf{w.d1.p, w.m.p, m->c0out_bias, BT * d.Ec, d.Ec, d.E, d.is48};
            hipLaunchKernelGGL(mask_fin_kernel, dim3((BT * d.Ec + 255) / 256), dim3(256), 0, st, mf);
        }
    } else {
        run_subpix_s(m, m->convt1, m->conv1p, e1v, d2v, d1v, B, Tc, d.s1);
        MaskOutArgs ma{x.e0.p, w.d1.p, w.m.p, m->C(m->conv0p.ps), m->C(m->conv0p.pb), m->C(m->c0out_w), m->c0out_bias,
                       BT * d.Ec, d.Ec, d.E, d.is48};
        hipLaunchKernelGGL(mask_out_kernel, dim3((BT * d.Ec + 3) / 4), dim3(256), 0, st, ma);
    }
}
// mask + deep filter (layers.py:414-445, multiframe.py:200-232) on stream st
void run_mask_df(dpdf_model* m, const ChunkArgs& c, XSet& x, hipStream_t st) {
    const dpdf_dims& d = m->d; Workspace& w = m->ln->ws;
    const int B = c.B, Tc = c.Tc, BT = B * Tc;
    m->cur = st;
    ProfScope ps(m, "mask_df");
    MaskApplyArgs mk{x.xs.p, w.m.p, w.xm.p, d.is48 ? nullptr : m->iconsts + 33, B, Tc, d.F, d.E};
    size_t total = (size_t)BT * d.F;
    DfApplyArgs da{w.xm.p, w.coefs.p, c.out, c.out_clip_stride, c.out_t0, c.attn_raw, c.alpha, (float)(1.0 - (double)c.alpha),
                   B, Tc, d.F, d.D, (float)(1.0 / (double)d.wnorm)};
    if (m->fuse_small && BT <= SMALL_M_ROWS) {     // latency regime: one launch (mask_df_kernel)
        MaskDfArgs md{mk, da, m->ln->mask_from_sums ? w.d1.p : nullptr, m->c0out_bias, d.Ec};
        hipLaunchKernelGGL(mask_df_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, md);
        return;
    }
    hipLaunchKernelGGL(mask_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, mk);
#ifdef DPDF_HAZARD_PROBE
    da.dump = m->probe_dump_on ? m->probe_dump : nullptr; da.dump_T = m->probe_dump_T;
    if (m->probe_taps >= 0) {
        const dim3 g((unsigned)((total + 255) / 256));
#define DPDF_PROBE_CASE(T, W, LT) case (LT) * 1000 + (T) * 20 + (W): hipLaunchKernelGGL(HIP_KERNEL_NAME(df_apply_probe_kernel<T, W, LT>), g, dim3(256), 0, st, da); break;
        switch (m->probe_late * 1000 + m->probe_taps * 20 + m->probe_wait) {
        DPDF_PROBE_CASE(0, 0, 0) DPDF_PROBE_CASE(1, 0, 0) DPDF_PROBE_CASE(2, 0, 0) DPDF_PROBE_CASE(3, 0, 0) DPDF_PROBE_CASE(4, 0, 0) DPDF_PROBE_CASE(5, 0, 0) DPDF_PROBE_CASE(6, 0, 0)
        DPDF_PROBE_CASE(1, 1, 0) DPDF_PROBE_CASE(2, 1, 0) DPDF_PROBE_CASE(4, 1, 0) DPDF_PROBE_CASE(2, 2, 0) DPDF_PROBE_CASE(4, 2, 0)
        DPDF_PROBE_CASE(2, 10, 0) DPDF_PROBE_CASE(2, 11, 0)
        DPDF_PROBE_CASE(2, 6, 0) DPDF_PROBE_CASE(2, 7, 0) DPDF_PROBE_CASE(2, 8, 0) DPDF_PROBE_CASE(2, 9, 0) DPDF_PROBE_CASE(2, 6, 1) DPDF_PROBE_CASE(2, 8, 1)
        DPDF_PROBE_CASE(2, 4, 0) DPDF_PROBE_CASE(4, 4, 0) DPDF_PROBE_CASE(2, 5, 0) DPDF_PROBE_CASE(4, 5, 0) DPDF_PROBE_CASE(3, 4, 0) DPDF_PROBE_CASE(0, 4, 0)
        DPDF_PROBE_CASE(2, 3, 0) DPDF_PROBE_CASE(4, 3, 0) DPDF_PROBE_CASE(1, 3, 0) DPDF_PROBE_CASE(0, 3, 0)
        DPDF_PROBE_CASE(0, 0, 1) DPDF_PROBE_CASE(2, 0, 1) DPDF_PROBE_CASE(4, 0, 1) DPDF_PROBE_CASE(2, 1, 1) DPDF_PROBE_CASE(4, 1, 1)
        default: fprintf(stderr, "probe: no kernel for taps %d wait %d late %d\n", m->probe_taps, m->probe_wait, m->probe_late); abort();
        }
#undef DPDF_PROBE_CASE
        return;
    }
#endif
    hipLaunchKernelGGL(df_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, da);
}

int run_stage2(dpdf_model* m, const ChunkArgs& c) {
    const dpdf_dims& d = m->d; const dpdf_state_layout& L = m->L;
    Workspace& w = m->ln->ws; XSet& x = w.x[c.parity];
    const int B = c.B, Tc = c.Tc, BT = B * Tc;
    const long S = d.state_size;
    float* state = c.state;
    // A call of ONE chunk (a streaming hop, a short clip) has no next chunk whose stage 1 stage 2 could run under: stage 2 stays on
    // the main stream -- the two cross-stream handoffs (into the stage-2 stream, back for the iSTFT) cost ~10 us each against
    // ~4 us of a same-stream dependent launch (one 16 kHz stream 248 -> 221 us/hop, 64 x 48 kHz streams 638 -> 612).
    hipStream_t st = ((m->overlap & 1) && !m->ln->single_chunk) ? m->ln->sB : m->ln->sA;
    m->cur = st;
    StateIoArgs sio = make_sio(m, c, x);
    sio.seg_lo = 4; sio.seg_hi = 6;                    // DF coefs delay / masked-spec FIFOs
    if (!m->ln->single_chunk) {   // the FIFO import touches stage-2 tensors and stage-2 state only: it runs BEFORE the wait for stage 1 (a one-chunk call: done by stage 1's import launch)
        ProfScope ps(m, "state_io");
        hipLaunchKernelGGL(state_io_kernel, dim3(B, 2, 5), dim3(256), 0, st, sio);
    }
    if (st != m->ln->sA) HIP_TRY(hipStreamWaitEvent(st, m->ln->ev_s1[c.parity], 0));
    const float* e3d = x.e3d; const float* c1d = x.c1d;
    TView e1v{x.e1.p, Tc, 0, d.F1, 64}, e2v{x.e2.p, Tc, 0, d.F2, 64}, e3v{x.e3.p, Tc, 0, d.F3, 64};
    TView c0v{x.c0.p, Tc + 4, 4, d.D, 64};
    // ---- embedding (dpdfnet.py:233-241; 48k hr.py:285-293).  channels-last [f][c] IS the (f,c) flatten ----
    // Small launches (<= 512 rows): the grouped linears chained in one launch each, per 16-row tile on the matrix cores
    // (small_fused_mfma.h).  (A per-row VALU form was measured too: equal for one row, worse from a few dozen rows on -- every
    // row's workgroup re-reads all weights: 64 x 48 kHz streams 313 -> 340 us -- and is gone.)
    const bool gl_dims = small_gl_dims(m);
    const bool smallm = m->fuse_small && m->fuse_gl && BT <= SMALL_M_ROWS && gl_dims;
    if (m->ln->join_armed && !smallm) { HIP_TRY(hipStreamWaitEvent(st, m->ln->ev_jn[c.parity], 0)); m->ln->join_armed = false; }     // (not reached: join_want asks for the same conditions)
    auto glfrag = [&](const GlW& g) { return GlFrag{m->C(g.frag), m->C(g.bias), g.G, g.Og, g.Ig, g.NT}; };
    const GlFrag nofrag{nullptr, nullptr, 0, 0, 0, 0};
    if (smallm) {
        ProfScope ps(m, "grouped_linear");
        EmbInMArgs ea{c1d, d.Fd * 64, e3d, d.F3 * 64, glfrag(m->df_fc_emb), d.is48 ? glfrag(m->enc_erb_fc) : nofrag, glfrag(m->enc_lin_in), w.g256a.p, BT,
                      nullptr, 0u, m->d_err};
        if (m->ln->join_armed) { ea.wait_ctr = m->ln->join_ctr; ea.wait_target = m->ln->join_total; m->ln->join_armed = false; }
        hipLaunchKernelGGL(emb_in_mfma_kernel, dim3((BT + 63) / 64, 16), dim3(256), 0, st, ea);
    } else {
        ProfScope ps(m, "grouped_linear");
        run_gl_auto(m, m->df_fc_emb, c1d, (size_t)d.Fd * 64, w.embin.p + 512, 1024, BT, ACT_RELU);
        if (d.is48) run_gl_auto(m, m->enc_erb_fc, e3d, (size_t)d.F3 * 64, w.embin.p, 1024, BT, ACT_RELU);
        else HIP_TRY(hipMemcpy2DAsync(w.embin.p, 1024 * sizeof(float), e3d, 512 * sizeof(float), 512 * sizeof(float), BT,
                                      hipMemcpyDeviceToDevice, st));
        run_gl_auto(m, m->enc_lin_in, w.embin.p, 1024, w.g256a.p, 256, BT, ACT_RELU);
    }
    run_gru256(m, m->enc_gru, w.g256a.p, w.g256b.p, state, S, L.emb_gru, B, Tc);
    const bool fork = (m->overlap & 8) && (st != m->ln->sA || (m->ln->single_chunk && m->hop_dec_fork));
    // fanned: the four linears behind the embedding GRU in one launch (emb_out_mfma_kernel) and the DF decoder's sum in df_out's A
    // producer -- with the decoders side by side, and in a one-chunk call also when they run one after the other on the main stream
    // (four dependent launches less); the DF decoder then works in its own granule buffers either way
    const bool fanned = smallm && (fork || m->ln->single_chunk);
    const bool sep = fork || fanned;
    float* df_ga = sep ? w.g256d.p : w.g256a.p;
    if (fanned) {
        ProfScope ps(m, "grouped_linear");
        EmbOutMArgs ea{w.g256b.p, glfrag(m->enc_lin_out), glfrag(m->df_lin_in), glfrag(m->ed_lin_in), glfrag(m->df_skip),
                       w.emb.p, df_ga, w.g256a.p, w.skipb.p, BT};
        hipLaunchKernelGGL(emb_out_mfma_kernel, dim3((BT + 63) / 64, 8), dim3(256), 0, st, ea);
    } else {
        ProfScope ps(m, "grouped_linear");
        run_gl_auto(m, m->enc_lin_out, w.g256b.p, 256, w.emb.p, 512, BT, ACT_RELU);
    }
    // The two decoders only share `emb`: the DF decoder (2 GRU-256 cells, df_out, pathway conv) runs on its own stream
    // beside the ERB decoder (2 cells, transposed convs, mask) -- the five latency-bound cell scans become three deep.
    hipStream_t sd = fork ? m->ln->sD : st;
    bool dec_steps_done = false, dfout_in_decin = false;
    if (fork) { HIP_TRY(hipEventRecord(m->ln->ev_dfk[c.parity], st)); HIP_TRY(hipStreamWaitEvent(sd, m->ln->ev_dfk[c.parity], 0)); }
    // ---- DF decoder (dpdfnet.py:486-519) ----
    {
        m->cur = sd;
        float* ga = sep ? w.g256d.p : w.g256a.p; float* gb = sep ? w.g256e.p : w.g256b.p; float* gc = sep ? w.g256f.p : w.g256c.p;
        const int which = fork ? 1 : 0;
        if (!fanned) {
            ProfScope ps(m, "grouped_linear");
            run_gl_auto(m, m->df_lin_in, w.emb.p, 512, ga, 256, BT, ACT_RELU);
        }
        // one frame per stream, the decoders one after the other on this stream: their first cells step in one launch, then their
        // second cells (they only share `emb`) -- four dependent step launches become two
        if (fanned && !fork && Tc == 1 && m->dual_step &&
            run_gru256_step_dual(m, m->df_gru0, ga, gb, L.df_dec_gru, 1, m->ed_gru0, w.g256a.p, w.g256b.p, L.erb_dec_gru, 0, state, S, B) &&
            run_gru256_step_dual(m, m->df_gru1, gb, gc, L.df_dec_gru + 256, 1, m->ed_gru1, w.g256b.p, w.g256c.p, L.erb_dec_gru + 256, 0, state, S, B)) {
            dec_steps_done = true;
        } else
        if (!run_gru256_stack(m, m->df_gru0, m->df_gru1, ga, gb, gc, state, S, L.df_dec_gru, B, Tc, which)) {
            run_gru256(m, m->df_gru0, ga, gb, state, S, L.df_dec_gru, B, Tc, which);
            run_gru256(m, m->df_gru1, gb, gc, state, S, L.df_dec_gru + 256, B, Tc, which);
        }
        if (!fanned) {
            ProfScope ps(m, "grouped_linear");
            run_gl_auto(m, m->df_skip, w.emb.p, 512, ga, 256, BT, ACT_NONE);   // c = df_gru(emb) + df_skip(emb)
        }
        {
            ProfScope ps(m, "df_coefs");
            size_t n = (size_t)BT * 256;
            const GlW& go = m->df_out;
            dfout_in_decin = fanned && !fork && x.have_pconv && m->dfout_in_decin && go.G == 16 && go.Ig == 16 && go.Og == 60 && go.NT == 4 && d.D * 10 == 960;
            if (dfout_in_decin) {                    // rides in the ERB decoder's dec_in launch below (decoders in series)
            } else if (fanned && x.have_pconv) {     // the sum rides in df_out's A producer
                const GlW& g = m->df_out;
                SumA<16> ap{gc, w.skipb.p, 256, g.Ig, g.Ig};
                DfOutEpi ep{w.coefs.p, Tc, FastDiv::make(Tc), x.pconv.p, m->C(g.bias), g.Og};
                launch_gemm_rows<4, 16, false>(sd, ap, m->C(g.frag), ep, BT, g.Ig, g.G);
            } else if (fanned) {
                const GlW& g = m->df_out;
                SumA<16> ap{gc, w.skipb.p, 256, g.Ig, g.Ig};
                BiasActStore<4> ep{w.dfo.p, (size_t)d.D * 10, g.Og, m->C(g.bias), g.Og, g.Og, ACT_TANH};
                launch_gemm_rows<4, 16, false>(sd, ap, m->C(g.frag), ep, BT, g.Ig, g.G);
                RowMap rm = RowMap::make(Tc, d.D);
                ConvpA ap2{c0v, rm};
                ConvpEpi ep2{w.coefs.p, Tc + 2, rm, w.dfo.p, m->C(m->convp_bias)};
                launch_gemm_rows<1, 64, false>(sd, ap2, m->C(m->convp_frag), ep2, BT * d.D, 320, 1);
            } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(axpy_kernel), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, sd, gc, ga, n);
            if (x.have_pconv) {      // pathway conv already done in stage 1: df_out's epilogue adds it and writes the taps
                const GlW& g = m->df_out;
                PlainA<16> ap{gc, 256, g.Ig, g.Ig};
                DfOutEpi ep{w.coefs.p, Tc, FastDiv::make(Tc), x.pconv.p, m->C(g.bias), g.Og};
                launch_gemm_rows<4, 16, false>(sd, ap, m->C(g.frag), ep, BT, g.Ig, g.G);
            } else {
                run_gl_auto(m, m->df_out, gc, 256, w.dfo.p, (size_t)d.D * 10, BT, ACT_TANH);
                RowMap rm = RowMap::make(Tc, d.D);
                ConvpA ap{c0v, rm};
                ConvpEpi ep{w.coefs.p, Tc + 2, rm, w.dfo.p, m->C(m->convp_bias)};
                launch_gemm_rows<1, 64, false>(sd, ap, m->C(m->convp_frag), ep, BT * d.D, 320, 1);
            }
            }
        }
        m->cur = st;
    }
    // ---- ERB decoder (dpdfnet.py:343-368; 48k hr.py:405-432) ----
    if (!fanned) {
        ProfScope ps(m, "grouped_linear");
        run_gl_auto(m, m->ed_lin_in, w.emb.p, 512, w.g256a.p, 256, BT, ACT_RELU);
    }
    if (dec_steps_done) {
    } else if (!run_gru256_stack(m, m->ed_gru0, m->ed_gru1, w.g256a.p, w.g256b.p, w.g256c.p, state, S, L.erb_dec_gru, B, Tc, 0)) {
        run_gru256(m, m->ed_gru0, w.g256a.p, w.g256b.p, state, S, L.erb_dec_gru, B, Tc);
        run_gru256(m, m->ed_gru1, w.g256b.p, w.g256c.p, state, S, L.erb_dec_gru + 256, B, Tc);
    }
    float* dembp = d.is48 ? w.demb2.p : w.demb.p;
    if (smallm) {
        ProfScope ps(m, "grouped_linear");
        DecInMArgs da{w.g256c.p, glfrag(m->ed_lin_out), d.is48 ? glfrag(m->ed_erb_fc) : nofrag, w.demb.p, w.demb2.p, d.F3 * 64, BT,
                      glfrag(m->df_out), w.g256f.p, w.skipb.p, x.pconv.p, w.coefs.p, Tc};
        hipLaunchKernelGGL(dec_in_mfma_kernel, dim3((BT + 63) / 64, dfout_in_decin ? 32 : 16), dim3(256), 0, st, da);
    } else {
        ProfScope ps(m, "grouped_linear");
        run_gl_auto(m, m->ed_lin_out, w.g256c.p, 256, w.demb.p, 512, BT, ACT_RELU);
        if (d.is48) run_gl_auto(m, m->ed_erb_fc, w.demb.p, 512, w.demb2.p, (size_t)d.F3 * 64, BT, ACT_RELU);
    }
    run_dec_convs(m, x, dembp, B, Tc, st);
    if (fork) { HIP_TRY(hipEventRecord(m->ln->ev_djn[c.parity], sd)); HIP_TRY(hipStreamWaitEvent(st, m->ln->ev_djn[c.parity], 0)); }
    run_mask_df(m, c, x, st);
    if (m->progress_on && m->pin_progress) hipLaunchKernelGGL(progress_kernel, dim3(1), dim3(1), 0, st, m->pin_progress, c.out_t0 + Tc);
    // the chunk's output is complete HERE: whoever waits for stage 2 (the next-but-one chunk's stage 1 for this XSet, the caller's
    // iSTFT) does not wait for the FIFO export behind it, which only moves stage-2 tensors into stage-2 state segments
    if (st != m->ln->sA) { HIP_TRY(hipEventRecord(m->ln->ev_s2[c.parity], st)); m->ln->s2_pending[c.parity] = true; }
    if (m->ln->single_chunk) {
        // One chunk: stage 1's export waited until here (it reads stage-1 tensors that stage 2 does not write) and both go out as one
        // launch, on the main stream.  (Measured: the same launch on the stage-2 stream, beside the caller's iSTFT instead of in front
        // of it -- one 16 kHz stream 197 -> 225 us per hop: a cross-stream handoff costs more than the launch it hides.)
        ProfScope ps(m, "state_io");
        sio.seg_lo = 0; sio.seg_hi = 6; sio.do_export = 1;
        if (m->ln->defer_export) { m->ln->pending_sio = sio; m->ln->pending_B = B; m->ln->export_pending = true; }     // (join_export)
        else hipLaunchKernelGGL(state_io_kernel, dim3(B, 6, 5), dim3(256), 0, st, sio);
    } else {
        {
            ProfScope ps(m, "state_io");
            sio.do_export = 1;
            hipLaunchKernelGGL(state_io_kernel, dim3(B, 2, 5), dim3(256), 0, st, sio);
        }
        if (st != m->ln->sA) { HIP_TRY(hipEventRecord(m->ln->ev_x2, st)); m->ln->x2_pending = true; }
    }
    m->ln->dbg_emb = w.emb.p;
    HIP_TRY(hipGetLastError());
    m->cur = m->ln->sA;
    return DPDF_OK;
}

// all chunks of a [B][T] problem; on return every stream's work is ordered before the main stream
// The main stream behind the last chunk's stage-2 FIFO export: the state is complete.  run_chunks does this itself unless the
// caller asks to do it later (a streaming hop: after the iSTFT and the overlap-add, which do not need the state).
int join_export(dpdf_model* m) {
    Lane& L = m->lanes[0];
    if (L.export_pending) {     // a streaming hop: the FIFO export goes BEHIND the iSTFT and the overlap-add -- the caller's wait for the output does not include it
        hipLaunchKernelGGL(state_io_kernel, dim3(L.pending_B, 6, 5), dim3(256), 0, m->stream, L.pending_sio);
        L.export_pending = false;
    }
    if (L.x2_pending) { HIP_TRY(hipStreamWaitEvent(m->stream, L.ev_x2, 0)); L.x2_pending = false; }
    return DPDF_OK;
}
// The time chunks of a [B][T] problem.
std::vector<int> chunk_schedule(const dpdf_model* m, int B, int T) {
    // chunk_frames: >0 explicit, <0 whole sequence, 0 auto (below); small batches: 256 frames per
    // chunk -- a small batch is latency-bound and wants several chunks so that stage 2 of one runs under stage 1 of the
    // next (tools/latency_bench.py --chunks: 1 clip x 10 s 22.0 -> 18.2 ms, 32 clips 39.3 -> 29.7 ms)
    int chunk = T;
    if (m->chunk_frames > 0) chunk = std::min(m->chunk_frames, T);
    // Throughput regime (>= 96 streams): 192 frames.  The intra-band launches have streams x frames / 16 workgroups and the
    // GRU-64 kernels are resident three (scan, <1>) or two (<2>) to a CU: at 256 clips 192 frames = 3072 workgroups =
    // 4 x 768 = 6 x 512 fills whole rounds of both (107.8 ms/step; 128 frames = 2.67 rounds, 108.7; 168 = 3.5 rounds, 109.6);
    // 128 clips 58.9 ms at 192 vs 60.1 at 256, 512 clips 207.3 vs 211.3 at 96 -- as long as a chunk stays below 128k frame
    // rows (~33 GB of workspace).  Fewer streams: 256 frames (latency regime, above).
    else if (m->chunk_frames == 0) {
        if (B < 96) chunk = std::min(T, 256);
        else chunk = std::min(T, std::min(192, std::max(64, 131072 / B)));
    }
    // (Measured and dropped: ending on a quarter-size chunk to shorten the pipeline drain -- 115.0 vs 114.6 ms/step; splitting
    // the batch over two independent lanes -- 180 vs 120 ms/step; stage 2 as a five-stream pipeline of sub-stages across
    // chunks for small batches -- 13.3 vs 10.3 ms for one clip: docs/HISTORY.md section 7.)
    std::vector<int> sizes;
    for (int rem = T; rem > 0; rem -= std::min(chunk, rem)) sizes.push_back(std::min(chunk, rem));
    // Pipeline drain: stage 2 of the LAST chunk has no stage 1 to run under -- its latency-bound GRU-256 scans (three deep,
    // 64 CUs) cost ~15 us per frame of that chunk with the rest of the chip idle, while an extra chunk costs ~0.6 ms.  So
    // in the throughput regime a last chunk of 96 frames or more gives up a short tail chunk (1003 frames = 5 x 192 + 43
    // stays as it is; 703 = 3 x 192 + 127 becomes ... + 95 + 32).
    if (m->tail_frames > 0 && m->chunk_frames == 0 && B >= 96 && sizes.size() > 1 && sizes.back() >= 96) {
        const int last = sizes.back();
        sizes.back() = last - m->tail_frames;
        sizes.push_back(m->tail_frames);
    }
    return sizes;
}
// Callbacks around the chunks of run_chunks (the pipelined host path: per-chunk STFT in front of stage 1, per-chunk iSTFT /
// overlap-add / download behind stage 2).  stage2_stream = the stream stage 2 of that chunk was enqueued on.
struct ChunkHooks {
    std::function<int(int ci, int t0, int Tc)> pre;
    std::function<int(int ci, int t0, int Tc, hipStream_t stage2_stream)> post;
};
int run_chunks(dpdf_model* m, const float* raw, size_t clip_stride, int B, int T, float* state,
               float* out, const float* attn_raw, float alpha, bool defer_export_join = false, const ChunkHooks* hooks = nullptr) {
    const dpdf_dims& d = m->d;
    int rc;
    m->ln = &m->lanes[0];
    m->lanes[0].export_pending = false;      // (never armed across calls: join_export launches it, error exits disarm it)
    if ((rc = init_lane(m->lanes[0]))) return rc;
    const std::vector<int> sizes = chunk_schedule(m, B, T);
    if ((rc = ensure_ws(m, B, *std::max_element(sizes.begin(), sizes.end())))) return rc;
    // Stage 2 imports its FIFOs BEFORE it waits for stage 1 of the chunk (run_stage2), so nothing else orders the stage-2 stream
    // behind what the caller queued on the main stream in front of this call -- the upload or the initialisation of the very
    // state that import reads.  One event at the head of the call does.
    m->ln->single_chunk = sizes.size() == 1 && m->single_chunk_inline;
    m->ln->defer_export = defer_export_join && m->late_export;
    if ((m->overlap & 1) && !m->ln->single_chunk) {
        HIP_TRY(hipEventRecord(m->lanes[0].ev_fork, m->stream));
        HIP_TRY(hipStreamWaitEvent(m->lanes[0].sB, m->lanes[0].ev_fork, 0));
    }
    int i = 0, t0 = 0;
    for (size_t ci = 0; ci < sizes.size(); t0 += sizes[ci], ++ci, ++i) {
        ChunkArgs c{raw + (size_t)t0 * d.F * 2, clip_stride, B, sizes[ci], state, out, clip_stride, t0, attn_raw, alpha, i & 1};
        if (hooks && hooks->pre && (rc = hooks->pre((int)ci, t0, sizes[ci]))) return rc;
        if ((rc = run_stage1(m, c)) || (rc = run_stage2(m, c))) return rc;
        if (hooks && hooks->post && (rc = hooks->post((int)ci, t0, sizes[ci], ((m->overlap & 1) && !m->ln->single_chunk) ? m->ln->sB : m->ln->sA))) return rc;
    }
    Lane& L = m->lanes[0];
    for (int p = 0; p < NRING; ++p)
        if (L.s2_pending[p]) { HIP_TRY(hipStreamWaitEvent(m->stream, L.ev_s2[p], 0)); L.s2_pending[p] = false; }
    m->cur = m->stream;
    if (!defer_export_join) return join_export(m);
    return DPDF_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
extern "C" int dpdf_create(const dpdf_cfg* cfg, const float* weights, size_t n_floats, int device, dpdf_model** out) {
    if (!cfg || !weights || !out) return set_err(DPDF_E_INVALID, "null argument");
    dpdf_dims d;
    if (dpdf_get_dims(cfg, &d) != 0) return set_err(DPDF_E_INVALID, "unsupported model config (sample_rate=%d nb=%d)", cfg->sample_rate, cfg->nb);
    const size_t need = dpdf_manifest(cfg, nullptr, nullptr);
    if (need != n_floats) return set_err(DPDF_E_INVALID, "weight blob has %zu floats, model needs %zu", n_floats, need);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(DPDF_E_RUNTIME, "no HIP device available: the MI355X engine has no CPU fallback");
    if (device < 0 || device >= ndev) return set_err(DPDF_E_INVALID, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    dpdf_model* m = new dpdf_model();
    m->cfg = *cfg; m->d = d; m->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->n_cus = cus; }
    dpdf_get_state_layout(&d, &m->L);
    Blob B; B.base = weights;
    dpdf_manifest(cfg, blob_cb, &B);
    Arena A;

    // ---- encoder convs ----
    {
        BnFold f = fold_bn(B, "enc.erb_conv0.2", 64);
        const float* w = B.get("enc.erb_conv0.1.weight");
        std::vector<float> wf(64 * 9);
        for (int c = 0; c < 64; ++c) for (int k = 0; k < 9; ++k) wf[c * 9 + k] = w[c * 9 + k] * f.scale[c];
        m->conv0_w = A.add(wf); m->conv0_b = A.add(f.shift);
    }
    m->erb_conv1 = build_sepconv(A, B, "enc.erb_conv1", 1);
    m->erb_conv2 = build_sepconv(A, B, "enc.erb_conv2", 1);
    m->erb_conv3 = build_sepconv(A, B, "enc.erb_conv3", 1);
    {
        // grouped 3x3 conv -> pointwise -> BN is linear end to end: fold it into one [K = 32][64] im2col operand,
        // row k = kt*8 + g*4 + kf (Conv0DfA), products accumulated in double
        const float *w0 = B.get("enc.df_conv0.1.convs.0.weight"), *w1 = B.get("enc.df_conv0.1.convs.1.weight");
        BnFold f = fold_bn(B, "enc.df_conv0.3", 64);
        const float* pw = B.get("enc.df_conv0.2.weight");           // [out 64][in 64]
        m->dfc0_pwfrag = A.add(pack_frag(32, 64, 4, [&](int k, int n) -> float {
            const int kt = k >> 3, g = (k >> 2) & 1, kf = k & 3;
            if (kt > 2 || kf > 2) return 0.f;
            const float* wg = g ? w1 : w0;                          // [32][1][3][3]
            double acc = 0.0;
            for (int cl = 0; cl < 32; ++cl) acc += (double)pw[n * 64 + g * 32 + cl] * (double)wg[cl * 9 + kt * 3 + kf];
            return (float)(acc * (double)f.scale[n]);
        }));
        m->dfc0_bias = A.add(f.shift);
    }
    m->df_conv1 = build_sepconv(A, B, "enc.df_conv1", 1);
    m->dprnn_erb = build_dprnn(A, B, "enc.dprnn_erb", d.nb);
    m->dprnn_df = build_dprnn(A, B, "enc.dprnn_df", d.nb);
    if (d.is48) m->enc_erb_fc = build_gl(A, B, "enc.erb_fc_emb.0", 32, d.emb / 32, d.C * d.F3 / 32);
    m->df_fc_emb = build_gl(A, B, "enc.df_fc_emb.0", 32, d.emb / 32, d.C * d.Fd / 32);
    m->enc_lin_in = build_gl(A, B, "enc.emb_gru.linear_in.0", 16, d.H / 16, 2 * d.emb / 16);
    m->enc_gru = build_gru256(A, B, "enc.emb_gru.gru.0.grucell");
    m->enc_lin_out = build_gl(A, B, "enc.emb_gru.linear_out.0", 16, d.emb / 16, d.H / 16);
    m->ed_lin_in = build_gl(A, B, "erb_dec.emb_gru.linear_in.0", 16, d.H / 16, d.emb / 16);
    m->ed_gru0 = build_gru256(A, B, "erb_dec.emb_gru.gru.0.grucell");
    m->ed_gru1 = build_gru256(A, B, "erb_dec.emb_gru.gru.1.grucell");
    m->ed_lin_out = build_gl(A, B, "erb_dec.emb_gru.linear_out.0", 16, d.emb / 16, d.H / 16);
    if (d.is48) m->ed_erb_fc = build_gl(A, B, "erb_dec.erb_fc_emb.0", 32, d.C * d.F3 / 32, d.emb / 32);
    m->conv3p = build_path(A, B, "erb_dec.conv3p"); m->convt3 = build_sepconv(A, B, "erb_dec.convt3", d.s3 > 1 ? d.s3 : 1);
    m->conv2p = build_path(A, B, "erb_dec.conv2p"); m->convt2 = build_sepconv(A, B, "erb_dec.convt2", d.s2);
    m->conv1p = build_path(A, B, "erb_dec.conv1p"); m->convt1 = build_sepconv(A, B, "erb_dec.convt1", d.s1);
    m->conv0p = build_path(A, B, "erb_dec.conv0p");
    {
        BnFold f = fold_bn(B, "erb_dec.conv0_out.1", 1);
        const float* w = B.get("erb_dec.conv0_out.0.weight");   // [1][64][1][3]
        std::vector<float> wf(64 * 3);
        for (int i = 0; i < 192; ++i) wf[i] = w[i] * f.scale[0];
        m->c0out_w = A.add(wf); m->c0out_bias = f.shift[0];
    }
    {   // df_convp: grouped(2) 32->5 k(5,1) . pointwise 10->10 . BN  folded into one [320 x 10] matrix
        BnFold f = fold_bn(B, "df_dec.df_convp.3", 10);
        const float *w0 = B.get("df_dec.df_convp.1.convs.0.weight"), *w1 = B.get("df_dec.df_convp.1.convs.1.weight");
        const float* pw = B.get("df_dec.df_convp.2.weight");    // [10][10]
        auto wd = [&](int k, int oc) -> float {                 // dense grouped-conv weight, k = kt*64 + cin
            int kt = k / 64, cin = k % 64, g = cin / 32;
            if (oc / 5 != g) return 0.f;
            const float* w = g == 0 ? w0 : w1;
            return w[((oc % 5) * 32 + (cin % 32)) * 5 + kt];
        };
        m->convp_frag = A.add(pack_frag(320, 10, 1, [&](int k, int n) {
            float acc = 0.f;
            for (int oc = 0; oc < 10; ++oc) acc += pw[n * 10 + oc] * wd(k, oc);
            return acc * f.scale[n];
        }));
        m->convp_bias = A.add(f.shift);
    }
    m->df_lin_in = build_gl(A, B, "df_dec.df_gru.linear_in.0", 8, d.H / 8, d.emb / 8);
    m->df_gru0 = build_gru256(A, B, "df_dec.df_gru.gru.0.grucell");
    m->df_gru1 = build_gru256(A, B, "df_dec.df_gru.gru.1.grucell");
    m->df_skip = build_gl(A, B, "df_dec.df_skip", 16, d.H / 16, d.emb / 16);
    m->df_out = build_gl(A, B, "df_dec.df_out.0", 16, d.D * 2 * d.O / 16, d.H / 16);

    // ---- STFT / iSTFT as real-DFT GEMMs ----
    m->window = A.add(vorbis(d.win));
    {
        const int N2 = 2 * d.F;
        // 32-column blocks: few frames (streaming hops) -> one workgroup per (row tile, block), see SMALL_M_ROWS;
        // many frames -> four blocks per workgroup, one per wave (gemm_rows_wn)
        const int NTs = 2;
        m->stft_groups_s = ((((N2 + 15) / 16 + NTs - 1) / NTs + 3) / 4) * 4;      // whole quadruples for gemm_rows_wn (tail groups are zero)
        std::vector<float> fs;
        for (int g = 0; g < m->stft_groups_s; ++g) {
            auto f = pack_frag(d.win, NTs * 16, NTs, [&](int k, int n) -> float {
                int ng = g * NTs * 16 + n;
                if (ng >= N2) return 0.f;
                int fb = ng / 2; long idx = ((long)fb * k) % d.win;
                double ang = 2.0 * M_PI * (double)idx / d.win;
                return (ng & 1) ? (float)(-std::sin(ang)) : (float)std::cos(ang);
            });
            fs.insert(fs.end(), f.begin(), f.end());
        }
        m->stft_frag_s = A.add(fs);
    }
    {
        const int NT = 5;
        m->istft_K = ((2 * d.F + 47) / 48) * 48;
        m->istft_groups = d.win / (NT * 16);
        std::vector<float> frag;
        for (int g = 0; g < m->istft_groups; ++g) {
            auto f = pack_frag(m->istft_K, NT * 16, NT, [&](int k, int n) -> float {
                if (k >= 2 * d.F) return 0.f;
                int fb = k / 2, ng = g * NT * 16 + n;
                double cf = (fb == 0 || fb == d.F - 1) ? 1.0 : 2.0;
                long idx = ((long)fb * ng) % d.win;
                double ang = 2.0 * M_PI * (double)idx / d.win;
                double v = (k & 1) ? -cf * std::sin(ang) : cf * std::cos(ang);
                if ((k & 1) && (fb == 0 || fb == d.F - 1)) v = 0.0;   // irfft ignores Im of DC / Nyquist
                return (float)(v / d.win);
            });
            frag.insert(frag.end(), f.begin(), f.end());
        }
        m->istft_frag = A.add(frag);
    }

    if (d.win == 960 || d.win == 320) {
        // ---- two-stage DFT (dft2stage.h): N = 32 x N2, n = N2 n1 + n2, k = k1 + 32 k2 ----
        const int Nw = d.win, N2 = Nw / 32, NK2 = N2 / 2 + 1;
        const int KC = (2 * N2 + 15) / 16, NT2 = (2 * NK2 + 15) / 16, NTA = (2 * N2 + 15) / 16;
        auto ang = [](long num, int den) { return 2.0 * M_PI * (double)(num % den) / den; };
        m->dft_f1 = A.add(pack_frag(32, 64, 4, [&](int n1, int n) -> float {
            const int k1 = n >> 1; const double a = ang((long)n1 * k1, 32);
            return (n & 1) ? (float)(-std::sin(a)) : (float)std::cos(a);
        }));
        std::vector<float> f2, fa;
        for (int k1 = 0; k1 < 32; ++k1) {
            auto f = pack_frag(16 * KC, 16 * NT2, NT2, [&](int kk, int n) -> float {
                if (kk >= 2 * N2 || n >= 2 * NK2) return 0.f;
                const int n2 = kk >> 1, cc = kk & 1, k2 = n >> 1, cp = n & 1;
                const double th = ang((long)n2 * (k1 + 32 * k2), Nw);
                if (cc == cp) return (float)std::cos(th);
                return cc ? (float)std::sin(th) : (float)(-std::sin(th));
            });
            f2.insert(f2.end(), f.begin(), f.end());
            auto g = pack_frag(16 * KC, 16 * NTA, NTA, [&](int kk, int n) -> float {
                if (kk >= 2 * N2 || n >= 2 * N2) return 0.f;
                const int k2 = kk >> 1, cc = kk & 1, n2 = n >> 1, cp = n & 1;
                const int k = k1 + 32 * k2; const bool mir = k > Nw / 2; const int ks = mir ? Nw - k : k;
                if (cc == 1 && (ks == 0 || ks == Nw / 2)) return 0.f;       // irfft ignores Im of DC / Nyquist
                const double sg = mir ? -1.0 : 1.0, th = ang((long)n2 * k, Nw);
                if (cc == 0) return cp == 0 ? (float)std::cos(th) : (float)std::sin(th);
                return cp == 0 ? (float)(-sg * std::sin(th)) : (float)(sg * std::cos(th));
            });
            fa.insert(fa.end(), g.begin(), g.end());
        }
        m->dft_f2 = A.add(f2);
        m->dft_iA = A.add(fa);
        m->dft_iB = A.add(pack_frag(64, 32, 2, [&](int kk, int n1) -> float {
            const int k1 = kk >> 1; const double a = ang((long)n1 * k1, 32);
            return (float)(((kk & 1) ? -std::sin(a) : std::cos(a)) / (double)Nw);
        }));
    }

    if (d.win == 960 || d.win == 320) {
        // ---- float64 analysis DFT (dft64.h) ----
        std::vector<double> t1, tm, t2;
        if (d.win == 960) dft64_tables<30>(t1, tm, t2); else dft64_tables<10>(t1, tm, t2);
        auto as_floats = [](const std::vector<double>& v) { std::vector<float> f(v.size() * 2); memcpy(f.data(), v.data(), v.size() * sizeof(double)); return f; };
        m->dft64_tw1 = A.add(as_floats(t1));          // (arena slots start on 256-byte boundaries)
        m->dft64_twm = A.add(as_floats(tm));
        m->dft64_tw2 = A.add(as_floats(t2));
    }

    // ---- streams: lane 0 now; the second lane and the sub-stage pipeline's streams only when first used (init_lane):
    // HIP multiplexes streams onto a handful of hardware queues, and streams that merely exist still take part in that
    // mapping -- with 14 streams per handle the four active ones of a second handle ended up sharing queues (one
    // 10 s clip 11.5 -> 14.6 ms, one streaming hop 1.33 -> 2.1 ms when measured beside another live handle)
    HIP_TRY(hipSetDevice(device));
    { int rc_ = init_lane(m->lanes[0]); if (rc_) return rc_; }
    m->stream = m->lanes[0].sA;
    m->cur = m->stream;
    m->ln = &m->lanes[0];
    HIP_TRY(hipEventCreate(&m->ev0)); HIP_TRY(hipEventCreate(&m->ev1));
    HIP_TRY(hipMalloc((void**)&m->consts, A.h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(m->consts, A.h.data(), A.h.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        std::vector<int> ic(33 + d.F, 0);
        if (!d.is48) {
            std::vector<int> start, band_of;
            erb_bands(d.win, d.sr, start, band_of);
            std::copy(start.begin(), start.end(), ic.begin());
            std::copy(band_of.begin(), band_of.end(), ic.begin() + 33);
        }
        HIP_TRY(hipMalloc((void**)&m->iconsts, ic.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(m->iconsts, ic.data(), ic.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // ErbNorm / SpecNorm (16 kHz: linspace) and MagNorm48 / SpecNorm48 (48 kHz: empirical tables) initial states
    // (reference onnx_model/layers.py:455-463, 516-522, 575-730; onnx_model/init_norms.py:21-139)
    m->erb_norm_init.resize(d.E); m->spec_norm_init.resize(d.D);
    dpdf_default_norm_init(&d, m->erb_norm_init.data(), m->spec_norm_init.data());
    HIP_TRY(hipMalloc((void**)&m->d_init_state, (size_t)d.state_size * sizeof(float)));
    {
        std::vector<float> st(d.state_size, 0.f);
        std::copy(m->erb_norm_init.begin(), m->erb_norm_init.end(), st.begin() + m->L.erb_norm);
        std::copy(m->spec_norm_init.begin(), m->spec_norm_init.end(), st.begin() + m->L.spec_norm);
        HIP_TRY(hipMemcpy(m->d_init_state, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&m->d_err, sizeof(int)));
    HIP_TRY(hipMemset(m->d_err, 0, sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&m->pin_progress, sizeof(int), hipHostMallocDefault));
    *m->pin_progress = 0;
    m->counted = true; g_live_models.fetch_add(1);
    *out = m;
    return DPDF_OK;
}

extern "C" void dpdf_destroy(dpdf_model* m) {
    if (!m) return;
    if (m->counted) { g_live_models.fetch_sub(1); m->counted = false; }
    (void)hipSetDevice(m->device);
    for (int g = 0; g < 1; ++g) {
        Lane& L = m->lanes[g];
        L.sync_all();
        L.ws.release();
        for (int k = 0; k < 5; ++k) if (L.gru_xbuf[k]) (void)hipFree(L.gru_xbuf[k]);
        for (int k = 0; k < 2; ++k) if (L.hop_flags[k]) (void)hipFree(L.hop_flags[k]);
        if (L.join_ctr) (void)hipFree(L.join_ctr);
        for (int k = 0; k < 2; ++k) if (L.gru_sbuf[k]) (void)hipFree(L.gru_sbuf[k]);
        for (int k = 0; k < 5; ++k) if (L.arrive[k]) (void)hipFree(L.arrive[k]);
    }
    DevBuf* bufs[] = {&m->io_spec, &m->io_spec_e, &m->io_state, &m->io_wav, &m->io_out, &m->frames, &m->raw_spec, &m->enh_spec, &m->batch_state, &m->stft_part, &m->dft_mid_f, &m->dft_mid_i};
    for (DevBuf* b : bufs) b->release();
    if (m->consts) (void)hipFree(m->consts);
    if (m->iconsts) (void)hipFree(m->iconsts);
    if (m->d_err) (void)hipFree(m->d_err);
    if (m->pin_progress) (void)hipHostFree(m->pin_progress);
    for (int r = 0; r < HostPipe::R; ++r) {
        if (m->hp.pin_in[r]) (void)hipHostFree(m->hp.pin_in[r]);
        if (m->hp.pin_out[r]) (void)hipHostFree(m->hp.pin_out[r]);
        if (m->hp.ev_up[r]) (void)hipEventDestroy(m->hp.ev_up[r]);
        if (m->hp.ev_down[r]) (void)hipEventDestroy(m->hp.ev_down[r]);
        if (m->hp.ev_s2[r]) (void)hipEventDestroy(m->hp.ev_s2[r]);
    }
    if (m->hp.s_up) (void)hipStreamDestroy(m->hp.s_up);
    if (m->hp.s_down) (void)hipStreamDestroy(m->hp.s_down);
    if (m->d_lens) (void)hipFree(m->d_lens);
    if (m->d_init_state) (void)hipFree(m->d_init_state);
    for (hipEvent_t e : m->prof_events) if (e) (void)hipEventDestroy(e);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    for (int g = 0; g < 1; ++g) {
        Lane& L = m->lanes[g];
        for (int p = 0; p < NRING; ++p) { if (L.ev_s1[p]) (void)hipEventDestroy(L.ev_s1[p]); if (L.ev_s2[p]) (void)hipEventDestroy(L.ev_s2[p]); if (L.ev_fk[p]) (void)hipEventDestroy(L.ev_fk[p]); if (L.ev_jn[p]) (void)hipEventDestroy(L.ev_jn[p]); if (L.ev_dfk[p]) (void)hipEventDestroy(L.ev_dfk[p]); if (L.ev_djn[p]) (void)hipEventDestroy(L.ev_djn[p]); }
        if (L.ev_fork) (void)hipEventDestroy(L.ev_fork);
        if (L.ev_x2) (void)hipEventDestroy(L.ev_x2);
        if (L.ev_join) (void)hipEventDestroy(L.ev_join);
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
        if (L.sB) (void)hipStreamDestroy(L.sB);
        if (L.sC) (void)hipStreamDestroy(L.sC);
        if (L.sD) (void)hipStreamDestroy(L.sD);
        if (L.sA) (void)hipStreamDestroy(L.sA);
    }
    delete m;
}

extern "C" int dpdf_set_norm_init(dpdf_model* m, const float* e, int ne, const float* s, int ns) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    if (e) {
        if (ne != m->d.E) return set_err(DPDF_E_INVALID, "erb_norm_init has %d values, model needs %d", ne, m->d.E);
        m->erb_norm_init.assign(e, e + ne);
        HIP_TRY(hipMemcpy(m->d_init_state + m->L.erb_norm, e, ne * sizeof(float), hipMemcpyHostToDevice));
    }
    if (s) {
        if (ns != m->d.D) return set_err(DPDF_E_INVALID, "spec_norm_init has %d values, model needs %d", ns, m->d.D);
        m->spec_norm_init.assign(s, s + ns);
        HIP_TRY(hipMemcpy(m->d_init_state + m->L.spec_norm, s, ns * sizeof(float), hipMemcpyHostToDevice));
    }
    return DPDF_OK;
}
extern "C" int dpdf_state_size(const dpdf_model* m) { return m ? m->d.state_size : 0; }
extern "C" int dpdf_initial_state(const dpdf_model* m, float* state) {
    if (!m || !state) return set_err(DPDF_E_INVALID, "null argument");
    memset(state, 0, sizeof(float) * m->d.state_size);
    std::copy(m->erb_norm_init.begin(), m->erb_norm_init.end(), state + m->L.erb_norm);
    std::copy(m->spec_norm_init.begin(), m->spec_norm_init.end(), state + m->L.spec_norm);
    return DPDF_OK;
}
extern "C" int dpdf_win_len(const dpdf_model* m) { return m ? m->d.win : 0; }
extern "C" int dpdf_hop(const dpdf_model* m) { return m ? m->d.hop : 0; }
extern "C" int dpdf_freq_bins(const dpdf_model* m) { return m ? m->d.F : 0; }
extern "C" int dpdf_sample_rate(const dpdf_model* m) { return m ? m->d.sr : 0; }
extern "C" int dpdf_num_frames(const dpdf_model* m, int n) { return m ? 1 + (n + m->d.win) / m->d.hop : 0; }
extern "C" int dpdf_set_chunk_frames(dpdf_model* m, int frames) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    m->chunk_frames = frames;
    return DPDF_OK;
}
extern "C" int dpdf_set_overlap(dpdf_model* m, int on) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    (void)hipSetDevice(m->device);
    for (int g = 0; g < 1; ++g) m->lanes[g].sync_all();
    m->overlap = on;
    return DPDF_OK;
}
extern "C" int dpdf_set_fuse_dprnn(dpdf_model* m, int on) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    m->fuse_dprnn = on < 0 ? 0 : (on > 2 ? 2 : on);
    return DPDF_OK;
}
// A/B switches for measurements (never change results beyond rounding); unknown names are an error.
extern "C" int dpdf_set_option(dpdf_model* m, const char* name, int value) {
    if (!m || !name) return set_err(DPDF_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    (void)hipSetDevice(m->device);
    for (int g = 0; g < 1; ++g) m->lanes[g].sync_all();
    const std::string n(name);
    if (n == "fuse_mask") m->fuse_mask = value != 0;
    else if (n == "dec_seg") m->dec_seg = value;
    else if (n == "dec_seg_grid") m->dec_seg_grid = value > 0 ? value : 256;
    else if (n == "dec_seg_all_frames") m->dec_seg_all_frames = value;
    else if (n == "df_ring") m->df_ring = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "hoist_gi") m->hoist_gi = value != 0;
    else if (n == "scan4_max_wgs") m->scan4_max_wgs = value < 0 ? 0 : value;
    else if (n == "inter_fuse_rows") m->inter_fuse_rows = value < 16 ? 16 : value;
    else if (n == "gru256_stack") m->gru256_stack = value != 0;
    else if (n == "tail_frames") m->tail_frames = value < 0 ? 0 : std::min(value, 48);
    else if (n == "stft_ksplit") m->stft_ksplit = value & 7;
    else if (n == "dft64") m->dft64 = value;
    else if (n == "gru64_limbs") m->gru64_limbs = value;
    else if (n == "gru256_step") m->gru256_step = value != 0;
    else if (n == "hop_glue") m->hop_glue = value != 0;
    else if (n == "hop_fused") m->hop_fused = value != 0;
    else if (n == "hop_stack") m->hop_stack = value != 0;
    else if (n == "hop_spin_join") m->hop_spin_join = value != 0;
    else if (n == "hop_feat") m->hop_feat = value != 0;
    else if (n == "glue8") m->glue8 = value != 0;
    else if (n == "fuse_gl") m->fuse_gl = value != 0;
    else if (n == "fuse_small") m->fuse_small = value != 0;
    else if (n == "fuse_enc") m->fuse_enc = value != 0;
    else if (n == "seg10") m->seg10 = value != 0;
    else if (n == "dfout_in_decin") m->dfout_in_decin = value != 0;
    else if (n == "hop_pconv") m->hop_pconv = value != 0;
    else if (n == "dual_step") m->dual_step = value != 0;
    else if (n == "hop_dec_fork") m->hop_dec_fork = value != 0;
    else if (n == "hop_prologue") m->hop_prologue = value != 0;
    else if (n == "enc_seg_rows") m->enc_seg_rows = value;
    else if (n == "dec_pyr_rows") m->dec_pyr_rows = value;
    else if (n == "late_export") m->late_export = value != 0;
    else if (n == "snapshot") m->snapshot = value != 0;
    else if (n == "host_pipe") m->host_pipe = value != 0;
    else if (n == "host_prefault") m->host_prefault = value != 0;
    else if (n == "dft2") m->dft2 = value != 0;
    else if (n == "chunk_io") m->chunk_io = value != 0;
    else if (n == "gru256_fused_x") m->gru256_fused_x = value != 0;
    else if (n == "gru256_fused_x_tiles") m->gru256_fused_x_tiles = value < 1 ? 1 : value;
    else if (n == "host_copy_threads") m->host_copy_threads = value < 1 ? 1 : value;
    else if (n == "single_chunk_inline") m->single_chunk_inline = value != 0;
    else if (n == "fuse_dec") m->fuse_dec = value != 0;
    else if (n == "interleave") m->interleave = value != 0;
    else if (n == "fcln_gi") m->fcln_gi = value != 0;
    else if (n == "gru256_c8_tiles") m->gru256_c8_tiles = value < 0 ? 0 : value;
    else if (n == "gru256_c16_tiles") m->gru256_c16_tiles = value < 0 ? 0 : value;
    else if (n == "gru256_cluster") m->use_gru256_cluster = value != 0;
#ifdef DPDF_HAZARD_PROBE
    else if (n == "probe_taps") m->probe_taps = value;                      // -1: the shipped df_apply_kernel
    else if (n == "probe_wait") m->probe_wait = value;
    else if (n == "probe_late") m->probe_late = value;
    else if (n == "probe_dump") m->probe_dump_on = value != 0;
    else if (n == "probe_coefs_uncached") {
        m->lanes[0].ws.release();                                           // re-allocated by the next call
        m->lanes[0].ws.coefs.uncached = value != 0;
    }
#endif
    else return set_err(DPDF_E_INVALID, "unknown option '%s'", name);
    return DPDF_OK;
}
extern "C" int dpdf_sync(dpdf_model* m) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);     // the flag read-and-clear below must not interleave with another thread's call
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return check_device_err(m);      // device-pointer calls surface a failed GRU-256 exchange here
}
// Frames of the offline call in flight (or of the last one) whose enhanced spectra are complete: lock-free, callable from
// another host thread while dpdf_enhance_batch* runs (reference api.py:94-104 reports (t + 1, total) after every frame; here
// the figure advances once per time chunk).
extern "C" int dpdf_progress(const dpdf_model* m) { return (m && m->pin_progress) ? *(volatile int*)m->pin_progress : 0; }
extern "C" int dpdf_debug_raise_device_error(dpdf_model* m) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    if (!m->d_err) HIP_TRY(hipMalloc((void**)&m->d_err, sizeof(int)));
    const int one = 1;
    HIP_TRY(hipMemcpy(m->d_err, &one, sizeof(int), hipMemcpyHostToDevice));
    return DPDF_OK;
}
extern "C" int dpdf_profile_enable(dpdf_model* m, int on) {
    if (!m) return set_err(DPDF_E_INVALID, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    m->prof_on = on != 0;
    if (on) { m->prof.clear(); m->prof_pending.clear(); m->prof_used = 0; }
    return DPDF_OK;
}
extern "C" size_t dpdf_profile_report(dpdf_model* m, char* buf, size_t cap) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lk(m->mu);
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    for (auto& pe : m->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->prof_events[pe.second], m->prof_events[pe.second + 1]) == hipSuccess) {
            auto& e = m->prof[pe.first]; e.ms += ms; e.calls++;
        }
    }
    m->prof_pending.clear(); m->prof_used = 0;
    std::string s;
    for (auto& kv : m->prof) {
        char line[160];
        snprintf(line, sizeof(line), "%s %.4f %ld\n", kv.first.c_str(), kv.second.ms, kv.second.calls);
        s += line;
    }
    if (buf && cap) { size_t n = std::min(cap - 1, s.size()); memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}

// Analysis STFT of a FEW frames (streaming hops, tiny clips): one 64-row tile per column group, so the K loop -- win / 64
// panels, each a round of B-fragment and PCM loads -- is a chain of latencies (one hop of 64 x 48 kHz streams: 118 us of a
// 1.2 ms hop).  Split five ways over K: the partial sums of every split land side by side and are added in a fixed order.
// defer_sum: leave the partials of a K-split launch in m->stft_part and say so in m->hx (the hop's feature kernel adds them)
static int stft_small(dpdf_model* m, const StftA<64>& ap, float* spec, int M, bool defer_sum = false) {
    const dpdf_dims& d = m->d;
    // (16 kHz: 5 panels -- the extra summing launch costs what the split saves; 48 kHz: 15 panels, 1222 -> 1175 us per hop)
    // a hop whose feature kernel adds the partials anyway (defer_sum): one K panel per workgroup, the summing is free there
    const int npan = d.win / 64, W = m->stft_groups_s * 32;
    const int ks = !(m->stft_ksplit & 1) ? 1 : (defer_sum && (m->stft_ksplit & 4)) ? npan : (npan >= 10 && npan % 5 == 0) ? 5 : 1;
    if (ks == 1) {
        BiasActStore<2> ep{spec, (size_t)2 * d.F, 32, nullptr, 0, 32, ACT_NONE};
        ep.ncol_total = 2 * d.F;
        launch_gemm_rows<2, 64, false>(m->stream, ap, m->C(m->stft_frag_s), ep, M, d.win, m->stft_groups_s);
        return DPDF_OK;
    }
    int rc = m->stft_part.ensure((size_t)M * ks * W); if (rc) return rc;
    BiasActStore<2> ep{m->stft_part.p, (size_t)ks * W, 32, nullptr, 0, 32, ACT_NONE};
    launch_gemm_rows<2, 64, false>(m->stream, ap, m->C(m->stft_frag_s), ep, M, d.win, m->stft_groups_s, 2048, ks);
    if (defer_sum) { m->hx.part = m->stft_part.p; m->hx.ks = ks; m->hx.W = W; return DPDF_OK; }
    const size_t n = (size_t)M * 2 * d.F;
    hipLaunchKernelGGL(ksplit_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, (const float*)m->stft_part.p, spec, M, ks, W, 2 * d.F, (const float*)nullptr);
    return DPDF_OK;
}

// ------------------------------------------------------------------------------------------------
// frame function for B streams x T frames (drop-in for the session.run loop)
// ------------------------------------------------------------------------------------------------
extern "C" int dpdf_run_frames(dpdf_model* m, const float* spec, int B, int T, float* state, float* spec_e, int flags) {
    if (!m || !spec || !state || !spec_e) return set_err(DPDF_E_INVALID, "null argument");
    if (B <= 0 || T < 0) return set_err(DPDF_E_INVALID, "bad batch geometry B=%d T=%d", B, T);
    if (T == 0) return DPDF_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    return with_recovery(m, [&]() -> int {
    const size_t nspec = (size_t)B * T * m->d.F * 2, nstate = (size_t)B * m->d.state_size;
    const float* d_spec = spec; float* d_state = state; float* d_out = spec_e;
    const bool host = !(flags & DPDF_DEVICE_PTRS);
    if (host) {
        int rc;
        if ((rc = m->io_spec.ensure(nspec)) || (rc = m->io_spec_e.ensure(nspec)) || (rc = m->io_state.ensure(nstate))) return rc;
        HIP_TRY(hipMemcpyAsync(m->io_spec.p, spec, nspec * sizeof(float), hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(m->io_state.p, state, nstate * sizeof(float), hipMemcpyHostToDevice, m->stream));
        d_spec = m->io_spec.p; d_state = m->io_state.p; d_out = m->io_spec_e.p;
    }
    {
        int rc = run_chunks(m, d_spec, (size_t)T * m->d.F * 2, B, T, d_state, d_out, nullptr, 0.f);
        if (rc) return rc;
    }
    if (host) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        int er = device_err_or_retry(m);       // before the state is copied back: a retry starts from the caller's state again
        if (er) return er;
        HIP_TRY(hipMemcpyAsync(spec_e, d_out, nspec * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(state, d_state, nstate * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    return DPDF_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// offline batch path: enhance() for B clips
// ------------------------------------------------------------------------------------------------

// lengths: nullptr = every clip is N samples; else host array [B] of per-clip sample counts (<= N, the row stride)
// Rows of a host-pointer call: in[b] holds in_len[b] readable floats, out[b] takes out_len[b] (null: N each).
struct HostRows { const float* const* in; float* const* out; const int* in_len; const int* out_len; };

// Pinned staging ring, copy streams and the copy threads of the pipelined host path (grown on demand, kept by the handle).
static int ensure_host_pipe(dpdf_model* m, size_t slot_in, size_t slot_out) {
    HostPipe& hp = m->hp;
    if (!hp.s_up) {
        HIP_TRY(hipStreamCreateWithFlags(&hp.s_up, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&hp.s_down, hipStreamNonBlocking));
        for (int r = 0; r < HostPipe::R; ++r) {
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_up[r], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_down[r], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_s2[r], hipEventDisableTiming));
        }
    }
    hp.pool.ensure(std::max(1, std::min(m->host_copy_threads, 16)));
    if (slot_in > hp.cap_in) {
        HIP_TRY(hipStreamSynchronize(hp.s_up));
        for (int r = 0; r < HostPipe::R; ++r) {
            if (hp.pin_in[r]) (void)hipHostFree(hp.pin_in[r]);
            hp.pin_in[r] = nullptr;
        }
        hp.cap_in = 0;
        for (int r = 0; r < HostPipe::R; ++r) HIP_TRY(hipHostMalloc((void**)&hp.pin_in[r], slot_in * sizeof(float), hipHostMallocDefault));
        hp.cap_in = slot_in;
    }
    if (slot_out > hp.cap_out) {
        HIP_TRY(hipStreamSynchronize(hp.s_down));
        for (int r = 0; r < HostPipe::R; ++r) {
            if (hp.pin_out[r]) (void)hipHostFree(hp.pin_out[r]);
            hp.pin_out[r] = nullptr;
        }
        hp.cap_out = 0;
        for (int r = 0; r < HostPipe::R; ++r) HIP_TRY(hipHostMalloc((void**)&hp.pin_out[r], slot_out * sizeof(float), hipHostMallocDefault));
        hp.cap_out = slot_out;
    }
    return DPDF_OK;
}

static int enhance_impl(dpdf_model* m, const float* wav, int B, int N, const int* lengths, float attn_limit_db, float* out, int flags,
                        const HostRows* rows = nullptr) {
    if (!m || (!rows && (!wav || !out))) return set_err(DPDF_E_INVALID, "null argument");
    if (B <= 0 || N < 0) return set_err(DPDF_E_INVALID, "bad batch geometry B=%d N=%d", B, N);
    if (attn_limit_db < 0.f) return set_err(DPDF_E_INVALID, "attn_limit_db must be non-negative, infinity, or None.");
    if (lengths)
        for (int b = 0; b < B; ++b)
            if (lengths[b] < 0 || lengths[b] > N) return set_err(DPDF_E_INVALID, "lengths[%d] = %d outside [0, %d]", b, lengths[b], N);
    if (N == 0) return DPDF_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    // dpdf_progress counts THIS call's frames from here on (a poller must never see the previous call's final count while
    // this one is still allocating / uploading), and reads 0 again once the call has returned
    struct ProgressEpoch { dpdf_model* m; bool sync_call;
                           ProgressEpoch(dpdf_model* m_, bool s_) : m(m_), sync_call(s_) { if (m->pin_progress) *m->pin_progress = 0; }
                           ~ProgressEpoch() { m->progress_on = false; if (sync_call && m->pin_progress) *m->pin_progress = 0; }
    } progress_epoch(m, !(flags & DPDF_DEVICE_PTRS));
    const bool host = !(flags & DPDF_DEVICE_PTRS);
    if (rows && !host) return set_err(DPDF_E_INVALID, "the row-pointer form takes host pointers");
    // host-pointer calls see their rows through pointers (a [B][N] block is B rows N floats apart)
    std::vector<const float*> blk_in; std::vector<float*> blk_out;
    HostRows hr{nullptr, nullptr, nullptr, nullptr};
    if (host) {
        if (rows) hr = *rows;
        else {
            blk_in.resize(B); blk_out.resize(B);
            for (int b = 0; b < B; ++b) { blk_in[b] = wav + (size_t)b * N; blk_out[b] = out + (size_t)b * N; }
            hr = HostRows{blk_in.data(), blk_out.data(), nullptr, nullptr};
        }
    }
    return with_recovery(m, [&]() -> int {
    const dpdf_dims& d = m->d;
    const int T = 1 + (N + d.win) / d.hop;
    const int* d_lens = nullptr;
    if (lengths) {
        if ((size_t)B > m->d_lens_cap) {
            if (m->d_lens) { HIP_TRY(hipStreamSynchronize(m->stream)); (void)hipFree(m->d_lens); m->d_lens = nullptr; m->d_lens_cap = 0; }
            HIP_TRY(hipMalloc((void**)&m->d_lens, (size_t)B * sizeof(int)));
            m->d_lens_cap = (size_t)B;
        }
        m->h_lens.assign(lengths, lengths + B);      // staging copy that outlives the async upload
        HIP_TRY(hipMemcpyAsync(m->d_lens, m->h_lens.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, m->stream));
        d_lens = m->d_lens;
    }
    const size_t nw = (size_t)B * N, nspec = (size_t)B * T * d.F * 2;
    int rc;
    const float* d_wav = wav; float* d_out = out;
    // ---- host pointers: pipelined over TIME SLICES (the metric includes the H2D of the noisy and the D2H of the
    // enhanced PCM).  The frame function walks time chunks [t0, t1) anyway: chunk k needs the samples below (t1 - 1) hop + win / 2
    // and completes the output samples below (t1 - 5) hop (overlap-add of frames p / hop - 1 and p / hop at p = n + 2 win +
    // win / 2).  So slice k+2 is gathered into pinned staging by the copy threads and uploaded, and slice k-2 downloaded and
    // scattered to the caller's rows, while the GPU computes chunk k; the STFT and the iSTFT + overlap-add run per chunk
    // (the synthesis on the download stream).  Only the first slice's upload and the last one's download are exposed.
    const std::vector<int> sizes = chunk_schedule(m, B, T);
    // Device-pointer calls take the same per-chunk form without the copies ("chunk_io"): only the first chunk's STFT precedes the
    // frame function and only the last chunk's iSTFT + overlap-add follows it; the rest runs beside the chunks (the synthesis on
    // its own stream behind each chunk's stage 2) instead of as two whole-batch launches on the critical stream.
    bool piped = (host ? m->host_pipe : (m->chunk_io && sizes.size() > 1)) && !m->prof_on;
    for (int Tc : sizes) piped = piped && (long)B * Tc > SMALL_M_ROWS;
    struct Slice { int t0, Tc, u0, u1, v0, v1; };
    std::vector<Slice> sl;
    size_t slot_in = 0, slot_out = 0;
    if (piped) {
        int t0 = 0, u = 0, v = 0;
        for (size_t k = 0; k < sizes.size(); ++k) {
            const int t1 = t0 + sizes[k];
            const bool last = k + 1 == sizes.size();
            const int u1 = last ? N : std::max(u, std::min(N, (t1 - 1) * d.hop + d.win / 2));
            const int v1 = last ? N : std::max(v, std::min(N, (t1 - 5) * d.hop));
            sl.push_back(Slice{t0, sizes[k], u, u1, v, v1});
            slot_in = std::max(slot_in, (size_t)B * (u1 - u)); slot_out = std::max(slot_out, (size_t)B * (v1 - v));
            t0 = t1; u = u1; v = v1;
        }
    }
    std::vector<float> flat_in, flat_out;          // small host calls in the row-pointer form: one contiguous staging block each way
    if (host) {
        if ((rc = m->io_wav.ensure(nw)) || (rc = m->io_out.ensure(nw))) return rc;
        d_wav = m->io_wav.p; d_out = m->io_out.p;
        if (piped) { if ((rc = ensure_host_pipe(m, slot_in, slot_out))) return rc; }
        else {
            const float* src = wav;
            if (rows) {
                flat_in.assign(nw, 0.f);
                for (int b = 0; b < B; ++b) memcpy(flat_in.data() + (size_t)b * N, hr.in[b], (size_t)(hr.in_len ? hr.in_len[b] : N) * sizeof(float));
                src = flat_in.data();
            }
            HIP_TRY(hipMemcpyAsync(m->io_wav.p, src, nw * sizeof(float), hipMemcpyHostToDevice, m->stream));
            if (rows) HIP_TRY(hipStreamSynchronize(m->stream));       // flat_in is pageable: the copy has left it
        }
    }
    if (piped && !host && (rc = ensure_host_pipe(m, 0, 0))) return rc;        // (its download stream and events; no staging)
    if ((rc = m->raw_spec.ensure(nspec)) || (rc = m->enh_spec.ensure(nspec)) ||
        (rc = m->batch_state.ensure((size_t)B * d.state_size)) || (rc = m->frames.ensure((size_t)B * T * d.win))) return rc;
    // big launches: the analysis / synthesis DFT as two small matrix stages (dft2stage.h)
    m->dbg_nspec = (long)nspec; m->dbg_nframes = (long)B * T * d.win;
    const bool dft2 = (d.win == 960 || d.win == 320) && m->dft2 && (long)B * T > SMALL_M_ROWS;
    // (the synthesis side only at 960: at 320 its two launches measure 0.96 ms against 0.88 ms for the one GEMM + overlap-add)
    const bool dft2_inv = dft2 && d.win == 960;
    if (dft2) {
        const size_t rows = piped ? (size_t)B * *std::max_element(sizes.begin(), sizes.end()) : (size_t)B * T;
        const size_t per_frame = (size_t)(d.win / 32) * 64;
        if ((rc = m->dft_mid_f.ensure(rows * per_frame)) || (rc = m->dft_mid_i.ensure(rows * per_frame))) return rc;
    }
    HostPipe& hp = m->hp;
    constexpr int R = HostPipe::R;
    static const bool trace = getenv("DPDF_HOST_PIPE_TRACE") != nullptr;      // stderr: where the host thread of a pipelined call spends its time
    struct Tr { double t_stage = 0, t_upwait = 0, t_drainwait = 0, t_scatter = 0; } tr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_call0 = now(); double t_first_up = 0;
    // gather slice k of every row into its pinned slot and queue the upload
    auto stage_up = [&](int k) -> int {
        const Slice& q = sl[k];
        const int slot = k % R, w = q.u1 - q.u0;
        double t0_ = now();
        if (k >= R) HIP_TRY(hipEventSynchronize(hp.ev_up[slot]));           // the upload that last read this slot
        tr.t_upwait += now() - t0_; t0_ = now();
        if (w > 0) {
            float* pin = hp.pin_in[slot];
            const std::function<void(int)> fn = [&](int b) {
                const int have = hr.in_len ? hr.in_len[b] : N;
                const int n = std::max(0, std::min(have, q.u1) - q.u0);         // (samples beyond a short clip's end are never read)
                if (n > 0) memcpy(pin + (size_t)b * w, hr.in[b] + q.u0, (size_t)n * sizeof(float));
            };
            if (m->host_copy_threads > 1) hp.pool.run(B, fn); else for (int b = 0; b < B; ++b) fn(b);
            HIP_TRY(hipMemcpy2DAsync(m->io_wav.p + q.u0, (size_t)N * sizeof(float), pin, (size_t)w * sizeof(float), (size_t)w * sizeof(float), B,
                                     hipMemcpyHostToDevice, hp.s_up));
        }
        HIP_TRY(hipEventRecord(hp.ev_up[slot], hp.s_up));
        tr.t_stage += now() - t0_;
        return DPDF_OK;
    };
    // wait for slice k's download and scatter it to the caller's rows
    auto drain = [&](int k) -> int {
        const Slice& q = sl[k];
        const int slot = k % R, w = q.v1 - q.v0;
        double t0_ = now();
        HIP_TRY(hipEventSynchronize(hp.ev_down[slot]));
        tr.t_drainwait += now() - t0_; t0_ = now();
        if (trace) fprintf(stderr, "[host_pipe]   slice %d down at %.2f ms\n", k, now() - t_call0);
        if (w > 0) {
            const float* pin = hp.pin_out[slot];
            const std::function<void(int)> fn = [&](int b) {
                const int room = hr.out_len ? hr.out_len[b] : N;
                const int n = std::max(0, std::min(room, q.v1) - q.v0);
                if (n > 0) memcpy(hr.out[b] + q.v0, pin + (size_t)b * w, (size_t)n * sizeof(float));
            };
            if (m->host_copy_threads > 1) hp.pool.run(B, fn); else for (int b = 0; b < B; ++b) fn(b);
        }
        tr.t_scatter += now() - t0_;
        return DPDF_OK;
    };
    ChunkHooks hooks;
    // Output rows the caller has just allocated are not backed by pages yet: the first write to every 4 KB costs a fault and a
    // zeroed page (~40 000 of them for 256 x 10 s), and the writes of the LAST slices are the exposed tail of the call.  A helper
    // thread populates the rows (madvise MADV_POPULATE_WRITE: contents untouched, so no ordering against the scatter is needed)
    // while the GPU works on the first chunk.  Rows that are populated already cost a page-table walk.
    struct Prefault { std::thread th; ~Prefault() { if (th.joinable()) th.join(); } } prefault;
    if (piped && host && m->host_prefault && nw * sizeof(float) >= ((size_t)8 << 20)) {
        prefault.th = std::thread([&hr, B, N] {
            const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
            for (int b = 0; b < B; ++b) {
                const size_t n = (size_t)(hr.out_len ? hr.out_len[b] : N) * sizeof(float);
                size_t lo = ((size_t)hr.out[b] + pg - 1) / pg * pg, hi = ((size_t)hr.out[b] + n) / pg * pg;
                if (hi > lo && madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE) != 0) return;     // old kernel / odd mapping: leave it to the scatter
            }
        });
    }
    if (piped) {
        if (host) {
            if ((rc = stage_up(0))) return rc;
            t_first_up = now() - t_call0;
            if (sl.size() > 1 && (rc = stage_up(1))) return rc;
        } else {
            // the download stream starts behind whatever the caller queued in front of this call
            HIP_TRY(hipEventRecord(hp.ev_up[0], m->stream));
            HIP_TRY(hipStreamWaitEvent(hp.s_down, hp.ev_up[0], 0));
        }
        hooks.pre = [&](int k, int t0, int Tc) -> int {
            // A1 for the frames of this chunk, behind its slice's upload
            if (host) HIP_TRY(hipStreamWaitEvent(m->stream, hp.ev_up[k % R], 0));
            const RowSeg seg{Tc, T, t0};
            if (m->use_dft64()) {
                Dft64Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, seg, 0, nullptr, m->raw_spec.p, B * Tc,
                             (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
                launch_dft64_forward(m->stream, da, d.win);
                return DPDF_OK;
            }
            if (dft2) {
                Dft2Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, nullptr, m->raw_spec.p, m->dft_mid_f.p, m->C(m->dft_f1), m->C(m->dft_f2), seg, B * Tc};
                launch_dft2_forward(m->stream, da, d.win);
                return DPDF_OK;
            }
            StftSegA<64> ap{d_wav, N, T, d.win, d.hop, m->C(m->window), d_lens, seg};
            SegStore<2> ep{m->raw_spec.p, (size_t)2 * d.F, 32, 32, 2 * d.F, seg};
            launch_gemm_rows_wn<2, 64>(m->stream, ap, m->C(m->stft_frag_s), ep, B * Tc, d.win, m->stft_groups_s / 4);
            return DPDF_OK;
        };
        hooks.post = [&](int k, int t0, int Tc, hipStream_t s2) -> int {
            // A14 for the frames of this chunk + the output samples they complete, on the download stream behind stage 2
            const Slice& q = sl[k];
            const int slot = k % R, w = q.v1 - q.v0;
            HIP_TRY(hipEventRecord(hp.ev_s2[slot], s2));
            HIP_TRY(hipStreamWaitEvent(hp.s_down, hp.ev_s2[slot], 0));
            const RowSeg seg{Tc, T, t0};
            if (dft2_inv) {
                Dft2Args da{nullptr, N, T, d.hop, m->C(m->window), nullptr, m->frames.p, m->enh_spec.p, m->dft_mid_i.p, m->C(m->dft_iB), m->C(m->dft_iA), seg, B * Tc};
                launch_dft2_inverse(hp.s_down, da, d.win);
            } else {
            PlainSegA<48> ap{m->enh_spec.p, (size_t)2 * d.F, 2 * d.F, seg};
            WindowSegStore<5> ep{m->frames.p, d.win, m->C(m->window), seg};
            if (m->istft_groups % 4 == 0) launch_gemm_rows_wn<5, 48>(hp.s_down, ap, m->C(m->istft_frag), ep, B * Tc, m->istft_K, m->istft_groups / 4);
            else launch_gemm_rows<5, 48, false>(hp.s_down, ap, m->C(m->istft_frag), ep, B * Tc, m->istft_K, m->istft_groups);
            }
            if (w > 0) {
                OlaArgs oa{m->frames.p, m->C(m->window), d_out, B, T, N, d.win, d.hop, d_lens, q.v0, w};
                hipLaunchKernelGGL(ola_kernel, dim3((unsigned)(((size_t)B * w + 255) / 256)), dim3(256), 0, hp.s_down, oa);
                if (host) HIP_TRY(hipMemcpy2DAsync(hp.pin_out[slot], (size_t)w * sizeof(float), d_out + q.v0, (size_t)N * sizeof(float), (size_t)w * sizeof(float), B,
                                                   hipMemcpyDeviceToHost, hp.s_down));
            }
            HIP_TRY(hipEventRecord(hp.ev_down[slot], hp.s_down));
            if (!host) return DPDF_OK;
            if (k + 2 < (int)sl.size() && (rc = stage_up(k + 2))) return rc;
            if (k >= 2 && (rc = drain(k - 2))) return rc;
            return DPDF_OK;
        };
    } else {
        // A1: analysis STFT
        ProfScope ps(m, "stft");
        StftA<64> ap{d_wav, N, T, d.win, d.hop, m->C(m->window), 0, d_lens};
        if (m->use_dft64()) {
            Dft64Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, RowSeg{T, T, 0}, 0, nullptr, m->raw_spec.p, B * T,
                         (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
            launch_dft64_forward(m->stream, da, d.win);
        } else if (B * T <= SMALL_M_ROWS) {
            if ((rc = stft_small(m, ap, m->raw_spec.p, B * T))) return rc;
        } else if (dft2) {
            Dft2Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, nullptr, m->raw_spec.p, m->dft_mid_f.p, m->C(m->dft_f1), m->C(m->dft_f2), RowSeg{T, T, 0}, B * T};
            launch_dft2_forward(m->stream, da, d.win);
        } else {
            BiasActStore<2> ep{m->raw_spec.p, (size_t)2 * d.F, 32, nullptr, 0, 32, ACT_NONE};
            ep.ncol_total = 2 * d.F;
            launch_gemm_rows_wn<2, 64>(m->stream, ap, m->C(m->stft_frag_s), ep, B * T, d.win, m->stft_groups_s / 4);
        }
    }
    // A17/A20: initial state for every clip
    {
        size_t n = (size_t)B * d.state_size;
        hipLaunchKernelGGL(fill_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream,
                           m->batch_state.p, m->d_init_state, (long)d.state_size, B);
    }
    // A2..A13: frame function over time chunks (+ attenuation limit fused in the DF kernel)
    const bool attn = std::isfinite(attn_limit_db);
    const float alpha = attn ? (float)std::pow(10.0, -(double)attn_limit_db / 20.0) : 0.f;
    if (m->pin_progress) *m->pin_progress = 0;
    m->progress_on = true;
    rc = run_chunks(m, m->raw_spec.p, (size_t)T * d.F * 2, B, T, m->batch_state.p, m->enh_spec.p,
                    attn ? m->raw_spec.p : nullptr, alpha, false, piped ? &hooks : nullptr);
    m->progress_on = false;
    if (rc) {
        if (piped) { (void)hipStreamSynchronize(hp.s_up); (void)hipStreamSynchronize(hp.s_down); (void)hipStreamSynchronize(m->stream); }
        return rc;
    }
    if (piped && !host) {
        // the output is complete behind the synthesis of the last chunk: order the caller's stream behind it
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamWaitEvent(m->stream, hp.ev_down[((int)sl.size() - 1) % R], 0));
        return DPDF_OK;
    }
    if (piped) {
        HIP_TRY(hipGetLastError());
        const int n = (int)sl.size();
        for (int k = std::max(0, n - 2); k < n; ++k) if ((rc = drain(k))) return rc;
        HIP_TRY(hipStreamSynchronize(m->stream));        // the state exports of the last chunk; every stream is joined behind this
        if (trace) fprintf(stderr, "[host_pipe] call %.2f ms: first upload staged+queued by %.2f, gather+queue %.2f, wait(up slot) %.2f, wait(down) %.2f, scatter %.2f, enqueue+rest %.2f\n",
                           now() - t_call0, t_first_up, tr.t_stage, tr.t_upwait, tr.t_drainwait, tr.t_scatter,
                           now() - t_call0 - tr.t_stage - tr.t_upwait - tr.t_drainwait - tr.t_scatter);
        return device_err_or_retry(m);
    }
    // A14: synthesis
    {
        ProfScope ps(m, "istft");
        PlainA<48> ap{m->enh_spec.p, (size_t)2 * d.F, 0, 2 * d.F};
        WindowStore<5> ep{m->frames.p, d.win, m->C(m->window)};
        if (dft2_inv) {
            Dft2Args da{nullptr, N, T, d.hop, m->C(m->window), nullptr, m->frames.p, m->enh_spec.p, m->dft_mid_i.p, m->C(m->dft_iB), m->C(m->dft_iA), RowSeg{T, T, 0}, B * T};
            launch_dft2_inverse(m->stream, da, d.win);
        } else
        if (B * T > SMALL_M_ROWS && m->istft_groups % 4 == 0) launch_gemm_rows_wn<5, 48>(m->stream, ap, m->C(m->istft_frag), ep, B * T, m->istft_K, m->istft_groups / 4);
        else launch_gemm_rows<5, 48, false>(m->stream, ap, m->C(m->istft_frag), ep, B * T, m->istft_K, m->istft_groups);
        OlaArgs oa{m->frames.p, m->C(m->window), d_out, B, T, N, d.win, d.hop, d_lens};
        hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, m->stream, oa);
    }
    HIP_TRY(hipGetLastError());
    if (host) {
        float* dst = out;
        if (rows) { flat_out.resize(nw); dst = flat_out.data(); }
        HIP_TRY(hipMemcpyAsync(dst, d_out, nw * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (rows)
            for (int b = 0; b < B; ++b) memcpy(hr.out[b], flat_out.data() + (size_t)b * N, (size_t)(hr.out_len ? hr.out_len[b] : N) * sizeof(float));
        return device_err_or_retry(m);
    }
    return DPDF_OK;
    });
}

extern "C" int dpdf_enhance_batch(dpdf_model* m, const float* wav, int B, int N, float attn_limit_db, float* out, int flags) {
    return enhance_impl(m, wav, B, N, nullptr, attn_limit_db, out, flags);
}
extern "C" int dpdf_enhance_batch_ragged(dpdf_model* m, const float* wav, int B, int n_max, const int* lengths,
                                         float attn_limit_db, float* out, int flags) {
    if (!lengths) return set_err(DPDF_E_INVALID, "null lengths");
    return enhance_impl(m, wav, B, n_max, lengths, attn_limit_db, out, flags);
}
extern "C" int dpdf_enhance_batch_rows(dpdf_model* m, const float* const* in_rows, const int* lengths, int B, int n_max,
                                       float attn_limit_db, float* const* out_rows, int flags) {
    if (!in_rows || !out_rows) return set_err(DPDF_E_INVALID, "null argument");
    if (flags & DPDF_DEVICE_PTRS) return set_err(DPDF_E_INVALID, "dpdf_enhance_batch_rows takes host pointers");
    for (int b = 0; b < B; ++b)
        if (!in_rows[b] || !out_rows[b]) return set_err(DPDF_E_INVALID, "null row pointer %d", b);
    HostRows hr{in_rows, out_rows, lengths, lengths};
    return enhance_impl(m, nullptr, B, n_max, lengths, attn_limit_db, nullptr, flags, &hr);
}

// ------------------------------------------------------------------------------------------------
// device-resident streaming (StreamEnhancer hot loop for S concurrent streams)
// ------------------------------------------------------------------------------------------------
extern "C" int dpdf_streams_create(dpdf_model* m, int n_streams, dpdf_streams** out) {
    if (!m || !out || n_streams <= 0) return set_err(DPDF_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    dpdf_streams* s = new dpdf_streams();
    s->m = m; s->S = n_streams; s->primed.assign(n_streams, 0);
    int rc;
    if ((rc = s->state.ensure((size_t)n_streams * m->d.state_size)) || (rc = s->in_tail.ensure((size_t)n_streams * m->d.hop)) ||
        (rc = s->ola_tail.ensure((size_t)n_streams * m->d.hop))) { delete s; return rc; }
    *out = s;
    size_t n = (size_t)n_streams * m->d.state_size;
    hipLaunchKernelGGL(fill_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, s->state.p, m->d_init_state,
                       (long)m->d.state_size, n_streams);
    HIP_TRY(hipMemsetAsync(s->in_tail.p, 0, (size_t)n_streams * m->d.hop * sizeof(float), m->stream));
    HIP_TRY(hipMemsetAsync(s->ola_tail.p, 0, (size_t)n_streams * m->d.hop * sizeof(float), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return DPDF_OK;
}
static void pool_destroy(dpdf_streams* s);
extern "C" void dpdf_streams_destroy(dpdf_streams* s) {
    if (!s) return;
    (void)hipSetDevice(s->m->device);
    (void)hipStreamSynchronize(s->m->stream);
    DevBuf* bufs[] = {&s->state, &s->in_tail, &s->ola_tail, &s->spec, &s->spec_e, &s->pcm_in, &s->pcm_out,
                      &s->cstate, &s->cin, &s->cola, &s->cpcm_in, &s->cpcm_out, &s->snap_state, &s->snap_in, &s->snap_ola};
    for (DevBuf* b : bufs) b->release();
    if (s->pin_idx) (void)hipHostFree(s->pin_idx);
    if (s->ev_snap) (void)hipEventDestroy(s->ev_snap);
    if (s->ev_out) (void)hipEventDestroy(s->ev_out);
    if (s->pin_in) (void)hipHostFree(s->pin_in);
    if (s->pin_out) (void)hipHostFree(s->pin_out);
    if (s->pin_err) (void)hipHostFree(s->pin_err);
    pool_destroy(s);
    delete s;
}
extern "C" int dpdf_streams_reset(dpdf_streams* s, int stream) {
    if (!s) return set_err(DPDF_E_INVALID, "null streams");
    if (stream >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", stream, s->S);
    dpdf_model* m = s->m;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    const int lo = stream < 0 ? 0 : stream, cnt = stream < 0 ? s->S : 1;
    size_t n = (size_t)cnt * m->d.state_size;
    hipLaunchKernelGGL(fill_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream,
                       s->state.p + (size_t)lo * m->d.state_size, m->d_init_state, (long)m->d.state_size, cnt);
    HIP_TRY(hipMemsetAsync(s->in_tail.p + (size_t)lo * m->d.hop, 0, (size_t)cnt * m->d.hop * sizeof(float), m->stream));
    HIP_TRY(hipMemsetAsync(s->ola_tail.p + (size_t)lo * m->d.hop, 0, (size_t)cnt * m->d.hop * sizeof(float), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (int i = lo; i < lo + cnt; ++i) s->primed[i] = 0;
    return DPDF_OK;
}
extern "C" int dpdf_streams_prime(dpdf_streams* s, const float* pcm_in, int flags) {
    if (!s || !pcm_in) return set_err(DPDF_E_INVALID, "null argument");
    dpdf_model* m = s->m;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    const size_t n = (size_t)s->S * m->d.hop;
    HIP_TRY(hipMemcpyAsync(s->in_tail.p, pcm_in, n * sizeof(float),
                           (flags & DPDF_DEVICE_PTRS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (int i = 0; i < s->S; ++i) s->primed[i] = 1;
    return DPDF_OK;
}
// One streaming call on device buffers: src [S][T*hop] -> dst [S][T*hop], everything enqueued on the engine's streams and
// joined back into m->stream.  host_err (pinned host memory, may be null): the last kernel mirrors the device error flag into it.
// (S, state, in_tail, ola_tail): the stream set itself, or the packed active subset of a masked call.
// snap_in / snap_ola (null: none): where the staging and overlap-add kernels leave the pre-call tails; ev_state (null: none):
// event behind the pre-call copy of the state, awaited before the frame function starts to update the state in place
struct StreamView { int S; float* state; float* in_tail; float* ola_tail; float* snap_in = nullptr; float* snap_ola = nullptr; hipEvent_t ev_state = nullptr; };
static int streams_enqueue_body(dpdf_streams* s, const StreamView& v, const float* src, int T, float* dst, int* host_err);
// run_stage2 arms Lane.export_pending (+ pending_sio, which holds THIS stream set's state / workspace pointers) for the late FIFO
// export of a hop; an error exit between there and join_export must not leave it armed for the next call on the model
// (possibly an offline batch, or after this dpdf_streams is gone).
static int streams_enqueue(dpdf_streams* s, const StreamView& v, const float* src, int T, float* dst, int* host_err) {
    dpdf_model* m = s->m;
    const int rc = streams_enqueue_body(s, v, src, T, dst, host_err);
    if (rc) {
        Lane& L = m->lanes[0];
        L.export_pending = false;
        m->hx = dpdf_model::HopExtras{};
        m->snap_dst = nullptr;
    }
    return rc;
}
static int streams_enqueue_body(dpdf_streams* s, const StreamView& v, const float* src, int T, float* dst, int* host_err) {
    dpdf_model* m = s->m;
    const dpdf_dims& d = m->d;
    const int S = v.S;
    int rc;
    // causal analysis (stream.py:119-126): frame j = x[j*hop : j*hop+win] * window -> rfft with x = [analysis buffer | new samples],
    // both read in place.  The buffers are handed over (in_tail <- last new hop) only after the STFT has read them: by the hop's
    // fused feature kernel (single-hop calls), else by a small kernel of its own.
    // A handful of streams: the STFT reads [analysis buffer | new samples] in place -- also when the new samples sit in pinned
    // host memory (a few KB).  More streams: every column group of the STFT GEMM re-reads its A rows, which must not go over PCIe
    // 31 times (64 x 48 kHz streams: +70 us) -- the staging kernel copies them into HBM once and hands the buffers over itself.
    const bool hop_fused = T == 1 && m->hop_feat && S * T <= SMALL_M_ROWS;
    const bool in_place = S * T <= 4;
    m->hx = dpdf_model::HopExtras{};
    float* xbuf = s->pcm_in.p;                       // [S][(T+1)*hop]
    // Single hops with staging: ONE prologue launch in front of the STFT does the staging, stage 1's FIFO import (it depends on the
    // previous call only) and the pre-call state copy -- instead of a staging launch here and an import launch behind the STFT.
    m->lanes[0].s1_imported = false;
    const bool prologue = !in_place && T == 1 && m->hop_prologue && !m->prof_on && S <= SMALL_M_ROWS;
    if (prologue) {
        m->ln = &m->lanes[0];
        if ((rc = init_lane(m->lanes[0])) || (rc = ensure_ws(m, S, 1))) return rc;
        ChunkArgs c{s->spec.p, (size_t)T * d.F * 2, S, 1, v.state, s->spec_e.p, (size_t)T * d.F * 2, 0, nullptr, 0.f, 0};
        StateIoArgs sio = make_sio(m, c, m->ln->ws.x[0]);
        sio.seg_lo = 0; sio.seg_hi = m->single_chunk_inline ? 6 : 4;      // (as run_stage1's import of a one-chunk call)
        if (m->snap_dst) { sio.snap = m->snap_dst; sio.snap_y = 4; m->snap_dst = nullptr; }
        sio.si_pcm = src; sio.si_tail = v.in_tail; sio.si_xbuf = xbuf; sio.si_snap = v.snap_in; sio.si_hops = T; sio.si_hop = d.hop;
        hipLaunchKernelGGL(state_io_kernel, dim3(S, sio.seg_hi + sio.snap_y + 1, 5), dim3(256), 0, m->stream, sio);
        m->ln->s1_imported = true;
    } else
    if (!in_place) hipLaunchKernelGGL(stream_stage_in_kernel, dim3(S), dim3(256), 0, m->stream, src, v.in_tail, xbuf, S, T, d.hop, v.snap_in);
    {
        StftA<64> ap{in_place ? src : xbuf, (T + 1) * d.hop, T, d.win, d.hop, m->C(m->window), 1};
        if (in_place) ap.tail = v.in_tail;
        if (m->use_dft64()) {
            Dft64Args da{in_place ? src : xbuf, (T + 1) * d.hop, T, d.hop, m->C(m->window), nullptr, RowSeg{T, T, 0}, 1, in_place ? v.in_tail : nullptr,
                         s->spec.p, S * T, (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
            launch_dft64_forward(m->stream, da, d.win);
        } else if (S * T <= SMALL_M_ROWS) {
            if ((rc = stft_small(m, ap, s->spec.p, S * T, hop_fused))) return rc;
        } else {
            BiasActStore<2> ep{s->spec.p, (size_t)2 * d.F, 32, nullptr, 0, 32, ACT_NONE};
            ep.ncol_total = 2 * d.F;
            launch_gemm_rows_wn<2, 64>(m->stream, ap, m->C(m->stft_frag_s), ep, S * T, d.win, m->stft_groups_s / 4);
        }
    }
    if (hop_fused) {
        if (in_place) { m->hx.pcm_new = src; m->hx.in_tail = v.in_tail; m->hx.snap_in = v.snap_in; }
        m->hx.armed = true;
    } else if (in_place) hipLaunchKernelGGL(stream_tail_update_kernel, dim3(S), dim3(256), 0, m->stream, src, v.in_tail, v.snap_in, T, d.hop);
    if (v.ev_state) HIP_TRY(hipStreamWaitEvent(m->stream, v.ev_state, 0));
    rc = run_chunks(m, s->spec.p, (size_t)T * d.F * 2, S, T, v.state, s->spec_e.p, nullptr, 0.f, true);
    if (rc) return rc;
    {
        PlainA<48> ap{s->spec_e.p, (size_t)2 * d.F, 0, 2 * d.F};
        WindowStore<5> ep{m->frames.p, d.win, m->C(m->window)};
        // few frames: the K loop (istft_K / 48 panels on win / 80 workgroups) is a chain of load latencies -- split seven
        // ways over K, the overlap-add kernel sums the partial frames and applies the window (no extra launch)
        const int npan = m->istft_K / 48, ks = ((m->stft_ksplit & 2) && S * T <= SMALL_M_ROWS && npan % 7 == 0) ? 7 : 1;
        if (ks > 1) {
            const int W = m->istft_groups * 80;
            if ((rc = m->stft_part.ensure((size_t)S * T * ks * W))) return rc;
            BiasActStore<5> ep7{m->stft_part.p, (size_t)ks * W, 80, nullptr, 0, 80, ACT_NONE};
            launch_gemm_rows<5, 48, false>(m->stream, ap, m->C(m->istft_frag), ep7, S * T, m->istft_K, m->istft_groups, 2048, ks);
            hipLaunchKernelGGL(stream_ola_ksplit_kernel, dim3(S), dim3(256), 0, m->stream, (const float*)m->stft_part.p, ks, W, m->C(m->window), v.ola_tail, dst, S, T, d.hop,
                               (const int*)m->d_err, host_err, v.snap_ola);
        } else {
            if (S * T > SMALL_M_ROWS && m->istft_groups % 4 == 0) launch_gemm_rows_wn<5, 48>(m->stream, ap, m->C(m->istft_frag), ep, S * T, m->istft_K, m->istft_groups / 4);
            else launch_gemm_rows<5, 48, false>(m->stream, ap, m->C(m->istft_frag), ep, S * T, m->istft_K, m->istft_groups);
            hipLaunchKernelGGL(stream_ola_kernel, dim3(S), dim3(256), 0, m->stream, m->frames.p, v.ola_tail, dst, S, T, d.hop, (const int*)m->d_err, host_err, v.snap_ola);
        }
    }
    HIP_TRY(hipGetLastError());
    if (s->ev_out) HIP_TRY(hipEventRecord(s->ev_out, m->stream));
    return join_export(m);          // the state is complete behind this point of the main stream
}

// The body of a streaming call on device-visible buffers (src / dst: device memory or pinned host memory), all streams or
// the n_act packed ones listed in idx (device-visible).
// snap: take the pre-call copy (state on the stage-2 stream, tails inside the kernels that overwrite them).
static int streams_run(dpdf_streams* s, const float* src, int T, float* dst, int n_act, const int* idx, int* host_err, bool snap = false) {
    dpdf_model* m = s->m;
    const dpdf_dims& d = m->d;
    hipEvent_t ev_state = nullptr;
    m->snap_dst = nullptr;
    if (snap && n_act == s->S) {
        m->snap_dst = s->snap_state.p;       // all streams: the copy rides in the call's first state import (run_stage1)
    } else if (snap) {
        // the state copy runs on the stage-2 stream (idle until stage 1 of this call is through) beside the staging kernel and
        // the STFT; only the frame function waits for it
        const size_t ns = (size_t)s->S * d.state_size;
        hipStream_t sb = m->lanes[0].sB;
        hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((ns / 4 + 256) / 256)), dim3(256), 0, sb, s->snap_state.p, (const float*)s->state.p, ns);
        HIP_TRY(hipEventRecord(s->ev_snap, sb));
        ev_state = s->ev_snap;
    }
    if (n_act == s->S) {
        StreamView v{s->S, s->state.p, s->in_tail.p, s->ola_tail.p, snap ? s->snap_in.p : nullptr, snap ? s->snap_ola.p : nullptr, ev_state};
        const int rc = streams_enqueue(s, v, src, T, dst, host_err);
        m->snap_dst = nullptr;
        return rc;
    }
    if (snap) {     // masked call: the packed copies are what the kernels overwrite; the full-set tails are copied here (rare path)
        const size_t nt = (size_t)s->S * d.hop;
        hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((nt / 4 + 256) / 256)), dim3(256), 0, m->stream, s->snap_in.p, (const float*)s->in_tail.p, nt);
        hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((nt / 4 + 256) / 256)), dim3(256), 0, m->stream, s->snap_ola.p, (const float*)s->ola_tail.p, nt);
        HIP_TRY(hipStreamWaitEvent(m->stream, ev_state, 0));
    }
    const int npcm = T * d.hop;
    StreamPackArgs pa{s->state.p, s->in_tail.p, s->ola_tail.p, s->cstate.p, s->cin.p, s->cola.p, src, s->cpcm_in.p, s->cpcm_out.p, dst,
                      idx, (long)d.state_size, d.hop, npcm, (const int*)m->d_err, host_err};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stream_pack_kernel<false>), dim3(n_act, 16), dim3(256), 0, m->stream, pa);
    int rc = streams_enqueue(s, StreamView{n_act, s->cstate.p, s->cin.p, s->cola.p}, s->cpcm_in.p, T, s->cpcm_out.p, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stream_pack_kernel<true>), dim3(n_act, 16), dim3(256), 0, m->stream, pa);
    HIP_TRY(hipGetLastError());
    return DPDF_OK;
}

// A GRU-256 cluster / step kernel whose exchange timed out has left the in-place state half advanced.  Host-pointer calls
// recover by themselves: state and tails go back to the copy taken at the start of the call, the device flag is cleared and
// the call runs again with every GRU-256 recurrence on the single-workgroup scan (no cross-workgroup waits, so it cannot
// time out) -- same results to rounding.  Counted in dpdf_recovery_count.
static int streams_recover_and_rerun(dpdf_streams* s, const float* src, int T, float* dst, int n_act, const int* idx) {
    dpdf_model* m = s->m;
    const dpdf_dims& d = m->d;
    HIP_TRY(hipMemsetAsync(m->d_err, 0, sizeof(int), m->stream));
    const size_t ns = (size_t)s->S * d.state_size, nt = (size_t)s->S * d.hop;
    hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((ns / 4 + 256) / 256)), dim3(256), 0, m->stream, s->state.p, (const float*)s->snap_state.p, ns);
    hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((nt / 4 + 256) / 256)), dim3(256), 0, m->stream, s->in_tail.p, (const float*)s->snap_in.p, nt);
    hipLaunchKernelGGL(copy_f4_kernel, dim3((unsigned)((nt / 4 + 256) / 256)), dim3(256), 0, m->stream, s->ola_tail.p, (const float*)s->snap_ola.p, nt);
    const int saved = m->use_gru256_cluster;
    m->use_gru256_cluster = 0;
    *s->pin_err = 0;
    int rc = streams_run(s, src, T, dst, n_act, idx, s->pin_err);
    m->use_gru256_cluster = saved;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    ++m->recoveries;
    if (*s->pin_err) { *s->pin_err = 0; return check_device_err(m); }
    return DPDF_OK;
}

// ext_in / ext_out (both or neither): pinned, GPU-visible [S][n_hops * hop] blocks owned by the caller that already hold the input and take the
// output in place (the native pool's round buffers): the host-pointer path without its two staging copies.
static int streams_call(dpdf_streams* s, const float* pcm_in, int n_hops, float* pcm_out, const unsigned char* active, int flags,
                        float* ext_in = nullptr, float* ext_out = nullptr) {
    if (!s || ((!pcm_in || !pcm_out) && !ext_in)) return set_err(DPDF_E_INVALID, "null argument");
    if (n_hops <= 0) return set_err(DPDF_E_INVALID, "n_hops must be positive");
    dpdf_model* m = s->m;
    int n_act = 0;
    for (int i = 0; i < s->S; ++i) {
        if (active && !active[i]) continue;
        if (!s->primed[i]) return set_err(DPDF_E_STATE, "stream %d not primed: call dpdf_streams_prime with its first hop", i);
        ++n_act;
    }
    if (n_act == 0) return DPDF_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    const dpdf_dims& d = m->d;
    const int S = s->S, T = n_hops;
    const size_t npcm = (size_t)S * T * d.hop, nspec = (size_t)n_act * T * d.F * 2;
    const bool host = !(flags & DPDF_DEVICE_PTRS);
    int rc;
    if ((rc = s->spec.ensure(nspec)) || (rc = s->spec_e.ensure(nspec)) || (rc = m->frames.ensure((size_t)n_act * T * d.win)) ||
        (rc = s->pcm_in.ensure((size_t)n_act * (T + 1) * d.hop))) return rc;
    if (!s->pin_idx) HIP_TRY(hipHostMalloc((void**)&s->pin_idx, (size_t)S * sizeof(int), hipHostMallocDefault));
    if (n_act < S) {
        HIP_TRY(hipStreamSynchronize(m->stream));      // an earlier asynchronous masked call may still be reading pin_idx
        int k = 0;
        for (int i = 0; i < S; ++i) if (active[i]) s->pin_idx[k++] = i;
        if ((rc = s->cstate.ensure((size_t)n_act * d.state_size)) || (rc = s->cin.ensure((size_t)n_act * d.hop)) || (rc = s->cola.ensure((size_t)n_act * d.hop)) ||
            (rc = s->cpcm_in.ensure((size_t)n_act * T * d.hop)) || (rc = s->cpcm_out.ensure((size_t)n_act * T * d.hop))) return rc;
    }
    if (!host) return streams_run(s, pcm_in, T, pcm_out, n_act, s->pin_idx, nullptr);
    // ---- host pointers: pinned staging both ways, pre-call snapshot, self-recovery ----
    float* pin_in = ext_in; float* pin_out = ext_out;
    if (!ext_in) {
        if (npcm > s->pin_cap) {
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (s->pin_in) (void)hipHostFree(s->pin_in);
            if (s->pin_out) (void)hipHostFree(s->pin_out);
            s->pin_in = s->pin_out = nullptr; s->pin_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&s->pin_in, npcm * sizeof(float), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void**)&s->pin_out, npcm * sizeof(float), hipHostMallocDefault));
            s->pin_cap = npcm;
        }
        pin_in = s->pin_in; pin_out = s->pin_out;
    }
    if (!s->pin_err) { HIP_TRY(hipHostMalloc((void**)&s->pin_err, sizeof(int), hipHostMallocDefault)); *s->pin_err = 0; }
    if (!ext_in) {
        if (n_act == S) memcpy(pin_in, pcm_in, npcm * sizeof(float));
        else for (int k = 0; k < n_act; ++k) {
            const size_t o = (size_t)s->pin_idx[k] * T * d.hop;
            memcpy(pin_in + o, pcm_in + o, (size_t)T * d.hop * sizeof(float));
        }
    }
    {
        const size_t ns = (size_t)S * d.state_size, nt = (size_t)S * d.hop;
        if ((rc = s->snap_state.ensure(ns)) || (rc = s->snap_in.ensure(nt)) || (rc = s->snap_ola.ensure(nt))) return rc;
        if (!s->ev_snap) HIP_TRY(hipEventCreateWithFlags(&s->ev_snap, hipEventDisableTiming));
        if (!s->ev_out) HIP_TRY(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
    }
    if ((rc = streams_run(s, pin_in, T, pin_out, n_act, s->pin_idx, s->pin_err, m->snapshot != 0))) return rc;
    // all streams active: the output (and the error flag's mirror) is in place behind the overlap-add; the state export that follows it
    // on the stream is not waited for -- whatever touches the state next is ordered behind it, and the getters synchronise the stream
    if (n_act == S && m->late_export) HIP_TRY(hipEventSynchronize(s->ev_out));
    else HIP_TRY(hipStreamSynchronize(m->stream));
    if (*s->pin_err) {
        *s->pin_err = 0;
        // recovery restores the PRE-CALL copy of state and tails: with the copy switched off (option "snapshot" = 0) there is
        // nothing valid to go back to -- report the device error, the streams need reset / set_state (as for device pointers)
        if (!m->snapshot) return check_device_err(m);
        if ((rc = streams_recover_and_rerun(s, pin_in, T, pin_out, n_act, s->pin_idx))) return rc;
    }
    if (!ext_in) {
        if (n_act == S) memcpy(pcm_out, pin_out, npcm * sizeof(float));
        else for (int k = 0; k < n_act; ++k) {
            const size_t o = (size_t)s->pin_idx[k] * T * d.hop;
            memcpy(pcm_out + o, pin_out + o, (size_t)T * d.hop * sizeof(float));
        }
    }
    return DPDF_OK;
}
extern "C" int dpdf_streams_process_masked(dpdf_streams* s, const float* pcm_in, int n_hops, float* pcm_out, const unsigned char* active, int flags) {
    return streams_call(s, pcm_in, n_hops, pcm_out, active, flags);
}

// ------------------------------------------------------------------------------------------------
// Native coalescing of INDEPENDENT submitters (the reference's pattern: N StreamEnhancer objects, each fed by its own caller
// whenever it has a chunk, package/src/dpdfnet/stream.py:13-72, 74-165).  Host threads submit k hops for one slot each; the
// first submitter of a round leads it: it waits -- at most the window, and only until every slot in use has queued -- then
// issues ONE masked device call for everybody in the round.  Submitters write their samples straight into the round's pinned,
// GPU-visible input block and read their result straight out of its output block (both in parallel, outside the lock);
// two round buffers alternate, so the next round fills while this one is on the GPU.  A round is homogeneous in k: a request
// with another hop count waits for the open round to fire and opens / joins the next one.
// ------------------------------------------------------------------------------------------------
struct StreamPoolC {
    struct Round {
        enum { OPEN = 0, FIRING = 1, DONE = 2 };
        std::atomic<int> state{OPEN};                 // written under `mu`; pool_collect's bounded poll reads it without
        int k = 0, n_queued = 0, n_regular_queued = 0, copies_pending = 0, readers_left = 0, rc = 0;
        bool has_leader = false;
        std::atomic<long> id{0};
        std::vector<unsigned char> active;
        std::string err;
        float* pin_in = nullptr; float* pin_out = nullptr; size_t cap = 0;       // floats
    };
    std::mutex mu, exec_mu;
    std::condition_variable cv;                       // every state change (arrivals, copies finished, rounds done / recycled)
    Round rd[2];
    long open_id = 0;                                 // id of the round that takes submissions (buffer open_id & 1)
    std::vector<unsigned char> in_use; int n_in_use = 0;
    std::vector<long> slot_round;                     // round id of the slot's outstanding request (-1: none)
    double window_s = 2e-4;                           // wait for in-use slots that did NOT ride in the previous round
    double regular_window_s = 2e-3;                   // ... for the ones that did (they are feeding the pool hop after hop: worth waiting for)
    double spin_s = 0.0;                              // waiters poll (no futex sleep) for up to this long before they block on `cv`
    std::vector<unsigned char> prev_active; int n_prev_in_use = 0;     // who rode in the last round that fired (and is still in use)
    std::atomic<int> arrivals{0};                     // bumped by every join: what a polling leader watches
    int n_cv_waiters = 0;                             // threads asleep on `cv` (nobody: state changes skip the notify)
    int n_blocked = 0;                                // submitters waiting in pool_join (another hop count than the open round's, or its buffer not recycled yet): they cannot join the open round
    long device_calls = 0, rounds = 0;
    // where a round's time goes (dpdf_streams_pool_timing): leader waiting for the others | device call | end of a call -> next call
    double t_wait = 0, t_call = 0, t_gap = 0; std::chrono::steady_clock::time_point last_done{};
    // Is more than one host thread feeding the pool?  A leader that is the only recent submitter does not wait for others.
    std::thread::id last_tid{}; std::chrono::steady_clock::time_point other_seen{};
};
// The pool's critical sections are tens of instructions long; a std::mutex that is found locked puts the caller to sleep in the
// kernel (tens of microseconds, per feeder thread and round).  Try for a few microseconds first.
static inline void pool_lock(std::unique_lock<std::mutex>& lk) {
    for (int i = 0; i < 4000; ++i) {
        if (lk.try_lock()) return;
        __builtin_ia32_pause();
    }
    lk.lock();
}
static StreamPoolC* pool_of(dpdf_streams* s) {
    static std::mutex g_mu;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!s->pool) {
        s->pool = new StreamPoolC();
        s->pool->in_use.assign(s->S, 0);
        s->pool->slot_round.assign(s->S, -1);
        s->pool->prev_active.assign(s->S, 0);
        for (int b = 0; b < 2; ++b) { s->pool->rd[b].active.assign(s->S, 0); s->pool->rd[b].id = b; }
    }
    return s->pool;
}
static void pool_destroy(dpdf_streams* s) {
    if (!s->pool) return;
    for (int b = 0; b < 2; ++b) {
        if (s->pool->rd[b].pin_in) (void)hipHostFree(s->pool->rd[b].pin_in);
        if (s->pool->rd[b].pin_out) (void)hipHostFree(s->pool->rd[b].pin_out);
    }
    delete s->pool; s->pool = nullptr;
}
extern "C" int dpdf_streams_pool_config(dpdf_streams* s, double window_s) {
    if (!s || !(window_s >= 0.0)) return set_err(DPDF_E_INVALID, "bad argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    P->window_s = window_s;
    P->regular_window_s = window_s > 2e-3 ? window_s : 2e-3;          // (the default; dpdf_streams_pool_tune sets it on its own)
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_tune(dpdf_streams* s, double window_s, double regular_window_s, double spin_s) {
    if (!s || !(window_s >= 0.0) || !(regular_window_s >= 0.0) || !(spin_s >= 0.0)) return set_err(DPDF_E_INVALID, "bad argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    P->window_s = window_s; P->regular_window_s = regular_window_s > window_s ? regular_window_s : window_s; P->spin_s = spin_s;
    return DPDF_OK;
}
extern "C" int dpdf_streams_slot_use(dpdf_streams* s, int slot, int in_use) {
    if (!s || slot < 0 || slot >= s->S) return set_err(DPDF_E_INVALID, "bad slot");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    if (in_use && !P->in_use[slot]) { P->in_use[slot] = 1; ++P->n_in_use; }
    else if (!in_use && P->in_use[slot]) {
        P->in_use[slot] = 0; --P->n_in_use;
        if (P->prev_active[slot]) { P->prev_active[slot] = 0; --P->n_prev_in_use; }
        P->arrivals.fetch_add(1, std::memory_order_release); P->cv.notify_all();
    }
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_stats(dpdf_streams* s, long* device_calls, long* rounds) {
    if (!s) return set_err(DPDF_E_INVALID, "null streams");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    if (device_calls) *device_calls = P->device_calls;
    if (rounds) *rounds = P->rounds;
    return DPDF_OK;
}
extern "C" int dpdf_streams_pool_timing(dpdf_streams* s, double* out3) {
    if (!s || !out3) return set_err(DPDF_E_INVALID, "null argument");
    StreamPoolC* P = pool_of(s);
    std::lock_guard<std::mutex> lk(P->mu);
    out3[0] = P->t_wait; out3[1] = P->t_call; out3[2] = P->t_gap;
    return DPDF_OK;
}
// join the open round with k hops for each of `n` slots of ONE caller (idx[] picks the requests out of the caller's arrays), copying
// the samples in; *lead_id >= 0: this caller has become that round's leader.  The whole group joins ONE round under one acquisition of
// the pool's lock (a contended std::mutex puts the loser to sleep in the kernel: per-slot locking cost 20-40 us per feeder thread and
// round with four feeders).
static int pool_join(dpdf_streams* s, StreamPoolC* P, int n, const int* idx, const int* slots, const float* const* in_rows, int k, long* lead_id) {
    const dpdf_dims& d = s->m->d;
    const size_t row = (size_t)k * d.hop;
    std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
    for (int j = 0; j < n; ++j)
        if (P->slot_round[slots[idx[j]]] >= 0) return set_err(DPDF_E_STATE, "slot %d already has a request in flight", slots[idx[j]]);
    StreamPoolC::Round* R;
    bool blocked = false;
    for (;;) {
        R = &P->rd[P->open_id & 1];
        // the buffer of the open round is free once the readers of the round that used it before are through; and a round takes
        // one hop count only
        if (R->state == StreamPoolC::Round::OPEN && R->id == P->open_id && (R->n_queued == 0 || R->k == k)) break;
        if (!blocked) {        // said ONCE: the open round's leader counts us as "cannot come" (several blocked submitters that re-announced themselves on every wake-up kept waking each other until the round fired)
            blocked = true;
            P->n_blocked += n; P->arrivals.fetch_add(1, std::memory_order_release); P->cv.notify_all();
        }
        ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters;
    }
    if (blocked) P->n_blocked -= n;
    if (R->n_queued == 0) {
        R->k = k;
        const size_t need = (size_t)s->S * row;
        if (need > R->cap) {                           // (empty round, buffer idle: nobody reads or writes it)
            if (hipSetDevice(s->m->device) != hipSuccess) return set_err(DPDF_E_RUNTIME, "hipSetDevice failed");
            if (R->pin_in) (void)hipHostFree(R->pin_in);
            if (R->pin_out) (void)hipHostFree(R->pin_out);
            R->pin_in = R->pin_out = nullptr; R->cap = 0;
            if (hipHostMalloc((void**)&R->pin_in, need * sizeof(float), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void**)&R->pin_out, need * sizeof(float), hipHostMallocDefault) != hipSuccess)
                return set_err(DPDF_E_RUNTIME, "hipHostMalloc of the pool's round buffers failed");
            R->cap = need;
        }
    }
    const auto tid = std::this_thread::get_id();
    if (P->last_tid != std::thread::id{} && P->last_tid != tid) P->other_seen = std::chrono::steady_clock::now();
    P->last_tid = tid;
    for (int j = 0; j < n; ++j) {
        const int slot = slots[idx[j]];
        R->active[slot] = 1;
        if (P->prev_active[slot]) ++R->n_regular_queued;
        P->slot_round[slot] = R->id;
    }
    R->n_queued += n; ++R->copies_pending;
    if (!R->has_leader) { R->has_leader = true; *lead_id = R->id; }
    float* base = R->pin_in;
    lk.unlock();
    for (int j = 0; j < n; ++j) memcpy(base + (size_t)slots[idx[j]] * row, in_rows[idx[j]], row * sizeof(float));
    pool_lock(lk);
    --R->copies_pending;
    P->arrivals.fetch_add(1, std::memory_order_release);
    if (P->n_cv_waiters) P->cv.notify_all();
    return DPDF_OK;
}
static void pool_lead(dpdf_streams* s, StreamPoolC* P, long id, bool no_window) {
    StreamPoolC::Round* R = &P->rd[id & 1];
    const auto t_lead0 = std::chrono::steady_clock::now();
    {
        using clk = std::chrono::steady_clock;
        std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
        const auto t0 = clk::now();
        const bool others = P->other_seen != clk::time_point{} && t0 - P->other_seen < std::chrono::seconds(1);
        if (!no_window && others && P->window_s > 0) {
            auto dur = [](double sec) { return std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(sec)); };
            const auto t_short = t0 + dur(P->window_s), t_long = t0 + dur(P->regular_window_s), t_spin = t0 + dur(P->spin_s);
            // (a slot has at most one request per round: once as many are queued as slots are in use, nobody else can come.)
            // Two windows: slots that rode in the previous round are being fed hop after hop -- their submitters are on their way
            // back (a wake-up, some host code), and a round fired without them costs everybody a second device call: the leader
            // waits `regular_window_s` for those; for in-use slots that sat the last round out only `window_s`.
            for (;;) {
                if (R->n_queued + P->n_blocked >= P->n_in_use) break;
                const auto now = clk::now();
                if (now >= t_long) break;
                if (now >= t_short && R->n_regular_queued + P->n_blocked >= P->n_prev_in_use) break;
                if (now < t_spin) {                    // poll: a futex wake-up costs tens of microseconds per arrival
                    const int seen = P->arrivals.load(std::memory_order_acquire);
                    lk.unlock();
                    while (P->arrivals.load(std::memory_order_acquire) == seen && clk::now() < t_spin) __builtin_ia32_pause();
                    pool_lock(lk);
                    continue;
                }
                ++P->n_cv_waiters; P->cv.wait_until(lk, now < t_short ? t_short : t_long); --P->n_cv_waiters;
            }
        }
    }
    // rounds execute in order: the previous round's leader holds exec_mu until its device call is through; this round stays open
    // (and keeps filling) while we wait for it
    std::lock_guard<std::mutex> ex(P->exec_mu);
    int k;
    {
        std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
        R->state = StreamPoolC::Round::FIRING;
        ++P->open_id;                                  // later submitters fill the other buffer
        P->cv.notify_all();
        while (R->copies_pending > 0) { ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters; }
        k = R->k;
        P->n_prev_in_use = 0;                          // (element-wise into vectors sized at creation: nothing here allocates)
        for (int i = 0; i < s->S; ++i) { P->prev_active[i] = R->active[i] && P->in_use[i]; P->n_prev_in_use += P->prev_active[i]; }
    }
    const auto t_call0 = std::chrono::steady_clock::now();
    // Whatever happens in the device call, the round reaches DONE with a return code: its followers block in pool_collect without
    // a time-out, and no C++ exception may cross the extern "C" boundary above us.  (R->active is not written while the round fires.)
    int rc; std::string call_err;
    try { rc = streams_call(s, nullptr, k, nullptr, R->active.data(), DPDF_HOST_PTRS, R->pin_in, R->pin_out); if (rc) call_err = dpdf_last_error(); }
    catch (const std::exception& e) { rc = DPDF_E_RUNTIME; try { call_err = std::string("exception in the pool's device call: ") + e.what(); } catch (...) {} }
    catch (...) { rc = DPDF_E_RUNTIME; }
    {
        std::lock_guard<std::mutex> lk(P->mu);
        const auto t_call1 = std::chrono::steady_clock::now();
        P->t_wait += std::chrono::duration<double>(t_call0 - t_lead0).count();
        P->t_call += std::chrono::duration<double>(t_call1 - t_call0).count();
        if (P->last_done != std::chrono::steady_clock::time_point{}) P->t_gap += std::chrono::duration<double>(t_call0 - P->last_done).count();
        P->last_done = t_call1;
        R->rc = rc; R->err.swap(call_err);
        R->state = StreamPoolC::Round::DONE;
        R->readers_left = R->n_queued;
        ++P->device_calls; ++P->rounds;
        P->cv.notify_all();
    }
}
// wait for the round the group rode in and copy its rows out (one acquisition of the lock on either side of the copies)
static int pool_collect(dpdf_streams* s, StreamPoolC* P, int n, const int* idx, const int* slots, float* const* out_rows) {
    const dpdf_dims& d = s->m->d;
    std::unique_lock<std::mutex> lk(P->mu, std::defer_lock); pool_lock(lk);
    const long id = P->slot_round[slots[idx[0]]];
    if (id < 0) return set_err(DPDF_E_STATE, "slot %d has no request in flight", slots[idx[0]]);
    StreamPoolC::Round* R = &P->rd[id & 1];
    if (P->spin_s > 0 && !(R->id == id && R->state == StreamPoolC::Round::DONE)) {
        // poll for the round's completion (the leader's device call takes hundreds of microseconds; being woken through the
        // condition variable adds tens more on the way back to the caller, in front of its NEXT submission)
        const auto t_spin = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(P->spin_s));
        lk.unlock();
        while (!(R->id.load(std::memory_order_acquire) == id && R->state.load(std::memory_order_acquire) == StreamPoolC::Round::DONE) &&
               std::chrono::steady_clock::now() < t_spin) __builtin_ia32_pause();
        pool_lock(lk);
    }
    while (!(R->id == id && R->state == StreamPoolC::Round::DONE)) { ++P->n_cv_waiters; P->cv.wait(lk); --P->n_cv_waiters; }
    const int rc = R->rc;
    const std::string err = rc ? R->err : std::string();
    const size_t row = (size_t)R->k * d.hop;
    const float* base = R->pin_out;
    lk.unlock();
    if (!rc) for (int j = 0; j < n; ++j) memcpy(out_rows[idx[j]], base + (size_t)slots[idx[j]] * row, row * sizeof(float));
    pool_lock(lk);
    for (int j = 0; j < n; ++j) P->slot_round[slots[idx[j]]] = -1;
    R->readers_left -= n;
    if (R->readers_left == 0) {                         // last reader recycles the buffer for round id + 2
        R->state = StreamPoolC::Round::OPEN; R->id = id + 2; R->n_queued = 0; R->n_regular_queued = 0; R->k = 0; R->has_leader = false; R->rc = 0;
        std::fill(R->active.begin(), R->active.end(), 0);
        if (P->n_cv_waiters) P->cv.notify_all();
    }
    if (rc) return set_err(rc, "%s", err.c_str());
    return DPDF_OK;
}
// n requests of ONE host thread (n = 1: a StreamEnhancer-shaped object's process()): slots[i] gets ks[i] whole hops from in_rows[i],
// out_rows[i] takes ks[i] * hop samples.  Requests with the same hop count ride in the same round, together with whatever
// other threads submit in the window.  flags bit 0: do not wait for other submitters (the caller knows it is alone).
extern "C" int dpdf_streams_submit_many(dpdf_streams* s, int n, const int* slots, const float* const* in_rows, const int* ks,
                                        float* const* out_rows, int flags) {
    if (!s || !slots || !in_rows || !ks || !out_rows) return set_err(DPDF_E_INVALID, "null argument");
    if (n <= 0) return DPDF_OK;
    {
        std::vector<unsigned char> seen(s->S, 0);
        for (int i = 0; i < n; ++i) {
            if (slots[i] < 0 || slots[i] >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", slots[i], s->S);
            if (ks[i] <= 0 || !in_rows[i] || !out_rows[i]) return set_err(DPDF_E_INVALID, "bad request %d", i);
            if (!s->primed[slots[i]]) return set_err(DPDF_E_STATE, "stream %d not primed: call dpdf_streams_prime with its first hop", slots[i]);
            if (seen[slots[i]]) return set_err(DPDF_E_INVALID, "slot %d appears twice", slots[i]);
            seen[slots[i]] = 1;
        }
    }
    StreamPoolC* P = pool_of(s);
    std::vector<char> done(n, 0);
    std::vector<int> grp; grp.reserve(n);
    int first_rc = DPDF_OK; std::string first_err;
    for (int i0 = 0; i0 < n; ++i0) {
        if (done[i0]) continue;
        const int k = ks[i0];                          // one group (= one round) per distinct hop count, in order of appearance
        grp.clear();
        for (int i = i0; i < n; ++i) if (!done[i] && ks[i] == k) { grp.push_back(i); done[i] = 1; }
        long lid = -1;
        int rc = pool_join(s, P, (int)grp.size(), grp.data(), slots, in_rows, k, &lid);
        if (!rc) {
            if (lid >= 0) pool_lead(s, P, lid, (flags & 1) != 0);
            rc = pool_collect(s, P, (int)grp.size(), grp.data(), slots, out_rows);
        }
        if (rc && !first_rc) { first_rc = rc; first_err = dpdf_last_error(); }
    }
    if (first_rc) return set_err(first_rc, "%s", first_err.c_str());
    return DPDF_OK;
}
// the same for n requests of equal hop count whose rows lie one after the other: in_block / out_block [n][k_hops * hop]
extern "C" int dpdf_streams_submit_block(dpdf_streams* s, int n, const int* slots, const float* in_block, int k_hops, float* out_block, int flags) {
    if (!s || !slots || !in_block || !out_block || n < 0 || k_hops <= 0) return set_err(DPDF_E_INVALID, "bad argument");
    const size_t row = (size_t)k_hops * s->m->d.hop;
    std::vector<const float*> in(n); std::vector<float*> out(n); std::vector<int> ks(n, k_hops);
    for (int i = 0; i < n; ++i) { in[i] = in_block + (size_t)i * row; out[i] = out_block + (size_t)i * row; }
    return dpdf_streams_submit_many(s, n, slots, in.data(), ks.data(), out.data(), flags);
}
extern "C" int dpdf_streams_submit_wait(dpdf_streams* s, int slot, const float* pcm, int k_hops, float* out, int flags) {
    return dpdf_streams_submit_many(s, 1, &slot, &pcm, &k_hops, &out, flags);
}
extern "C" int dpdf_streams_process(dpdf_streams* s, const float* pcm_in, int n_hops, float* pcm_out, int flags) {
    return dpdf_streams_process_masked(s, pcm_in, n_hops, pcm_out, nullptr, flags);
}
// Resume a stream from saved data: `state` is the reference's flat state vector (dpdf_streams_get_state, or a state the
// reference's own session loop produced: onnx_backend.py:52-78); in_tail / ola_tail are the StreamEnhancer's analysis and
// overlap-add buffers (stream.py:62-72; hop floats each, dpdf_streams_get_tails).  Null pointers leave that part as it is;
// a stream that receives an in_tail counts as primed.
extern "C" int dpdf_streams_set_state(dpdf_streams* s, int stream, const float* state, const float* in_tail, const float* ola_tail) {
    if (!s) return set_err(DPDF_E_INVALID, "null streams");
    if (stream < 0 || stream >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", stream, s->S);
    dpdf_model* m = s->m;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (state) HIP_TRY(hipMemcpy(s->state.p + (size_t)stream * m->d.state_size, state, (size_t)m->d.state_size * sizeof(float), hipMemcpyHostToDevice));
    if (in_tail) { HIP_TRY(hipMemcpy(s->in_tail.p + (size_t)stream * m->d.hop, in_tail, (size_t)m->d.hop * sizeof(float), hipMemcpyHostToDevice)); s->primed[stream] = 1; }
    if (ola_tail) HIP_TRY(hipMemcpy(s->ola_tail.p + (size_t)stream * m->d.hop, ola_tail, (size_t)m->d.hop * sizeof(float), hipMemcpyHostToDevice));
    return DPDF_OK;
}
extern "C" int dpdf_streams_get_tails(dpdf_streams* s, int stream, float* in_tail, float* ola_tail) {
    if (!s) return set_err(DPDF_E_INVALID, "null streams");
    if (stream < 0 || stream >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", stream, s->S);
    dpdf_model* m = s->m;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (in_tail) HIP_TRY(hipMemcpy(in_tail, s->in_tail.p + (size_t)stream * m->d.hop, (size_t)m->d.hop * sizeof(float), hipMemcpyDeviceToHost));
    if (ola_tail) HIP_TRY(hipMemcpy(ola_tail, s->ola_tail.p + (size_t)stream * m->d.hop, (size_t)m->d.hop * sizeof(float), hipMemcpyDeviceToHost));
    return DPDF_OK;
}
extern "C" int dpdf_streams_is_primed(dpdf_streams* s, int stream) {
    if (!s || stream < 0 || stream >= s->S) return 0;
    return s->primed[stream];
}
// prime ONE stream with its first hop (host pointer)
extern "C" int dpdf_streams_prime_one(dpdf_streams* s, int stream, const float* pcm_hop) {
    if (!s || !pcm_hop) return set_err(DPDF_E_INVALID, "null argument");
    return dpdf_streams_set_state(s, stream, nullptr, pcm_hop, nullptr);
}
extern "C" long dpdf_recovery_count(const dpdf_model* m) { return m ? m->recoveries : 0; }
extern "C" int dpdf_streams_get_state(dpdf_streams* s, int stream, float* state_host) {
    if (!s || !state_host) return set_err(DPDF_E_INVALID, "null argument");
    if (stream < 0 || stream >= s->S) return set_err(DPDF_E_STATE, "stream %d out of range (have %d)", stream, s->S);
    dpdf_model* m = s->m;
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(state_host, s->state.p + (size_t)stream * m->d.state_size, (size_t)m->d.state_size * sizeof(float), hipMemcpyDeviceToHost));
    return DPDF_OK;
}

// ------------------------------------------------------------------------------------------------
// debug: fetch an intermediate tensor of the LAST chunk (per-stage parity tests).  Layouts are the
// engine's own channels-last forms: e0 [B][Tc][Ec][64], c0 [B][4+Tc][D][64] (halo 4), ...
// ------------------------------------------------------------------------------------------------
extern "C" long dpdf_debug_fetch(dpdf_model* m, const char* name, float* host, long cap) {
    if (!m || !name) return -1;
    std::lock_guard<std::mutex> lk(m->mu);
    if (hipSetDevice(m->device) != hipSuccess) return -1;
    Lane& L0 = m->lanes[0];
    const dpdf_dims& d = m->d; Workspace& w = L0.ws; XSet& x = w.x[L0.dbg_parity];
    const long B = L0.dbg_B, Tc = L0.dbg_Tc, BT = B * Tc;
    const float* src = nullptr; long n = 0;
    std::string s(name);
    if (s == "feat_erb") { src = w.feat_erb.p; n = B * (Tc + 2) * d.E; }
    else if (s == "feat_spec") { src = w.feat_spec.p; n = B * (Tc + 2) * 2 * d.D; }
    else if (s == "e0") { src = x.e0.p; n = BT * d.Ec * 64; }
    else if (s == "e1") { src = x.e1.p; n = BT * d.F1 * 64; }
    else if (s == "e2") { src = x.e2.p; n = BT * d.F2 * 64; }
    else if (s == "e3") { src = x.e3.p; n = BT * d.F3 * 64; }
    else if (s == "e3_dprnn") { src = L0.dbg_e3d; n = BT * d.F3 * 64; }
    else if (s == "c0") { src = x.c0.p; n = B * (Tc + 4) * d.D * 64; }
    else if (s == "c1") { src = x.c1.p; n = BT * d.Fd * 64; }
    else if (s == "c1_dprnn") { src = L0.dbg_c1d; n = BT * d.Fd * 64; }
    else if (s == "emb") { src = L0.dbg_emb ? L0.dbg_emb : w.emb.p; n = BT * 512; }
    else if (s == "m") { src = w.m.p; n = BT * d.E; }
    else if (s == "coefs") { src = w.coefs.p; n = B * (Tc + 2) * d.D * 10; }
    else if (s == "xm") { src = w.xm.p; n = B * (Tc + 4) * d.F * 2; }
    else if (s == "raw_spec") { src = m->raw_spec.p; n = m->dbg_nspec; }        // the last offline call's analysis spectra [B][T][F][2]
    else if (s == "frames") { src = m->frames.p; n = m->dbg_nframes; }          // ... and windowed synthesis frames [B][T][win]
    if (!src) return -1;
    if (host && cap >= n) {
        if (hipStreamSynchronize(m->stream) != hipSuccess) return -1;
        if (hipMemcpy(host, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return n;
}

#ifdef DPDF_PHASE_TRACE
extern "C" int dpdf_debug_stack_trace(unsigned long long* out512) { return hipMemcpyFromSymbol(out512, HIP_SYMBOL(dpdf_stack_trace), 512 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; }
extern "C" int dpdf_debug_trace(unsigned long long* out32) { return hipMemcpyFromSymbol(out32, HIP_SYMBOL(dpdf_trace_buf), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1; }
#endif

#ifdef DPDF_HAZARD_PROBE
// Probe build only (tools/hazard_probe.py): the side buffer df_apply_probe_kernel dumps the taps it consumed into.
extern "C" int dpdf_probe_dump_alloc(dpdf_model* m, int B, int T) {
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    const size_t n = (size_t)B * T * m->d.F * 36;
    if (n > m->probe_dump_n) {
        if (m->probe_dump) (void)hipFree(m->probe_dump);
        HIP_TRY(hipMalloc((void**)&m->probe_dump, n * sizeof(unsigned)));
        m->probe_dump_n = n;
    }
    m->probe_dump_T = T;
    HIP_TRY(hipMemset(m->probe_dump, 0xff, n * sizeof(unsigned)));
    return DPDF_OK;
}
extern "C" int dpdf_probe_dump_fetch(dpdf_model* m, unsigned* host, size_t n) {
    std::lock_guard<std::mutex> lk(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, m->probe_dump, std::min(n, m->probe_dump_n) * sizeof(unsigned), hipMemcpyDeviceToHost));
    return DPDF_OK;
}
#endif
