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

#include "host_model_types.h"
#include "host_schedule.h"

#include "host_create.h"
#include "host_batch_calls.h"
#include "host_streaming.h"
#include "host_stream_pool.h"
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
