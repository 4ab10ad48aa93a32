/*
 * dpdfnet_hip.h -- C ABI of the MI355X-native DPDFNet enhancement engine (libdpdfnet_hip.so).
 *
 * The reference has no C ABI: its runtime seam is the Python `RuntimeModel` around an
 * onnxruntime CPU session (reference package/src/dpdfnet/onnx_backend.py:11-18, 81-99), called
 * once per 10 ms frame from `enhance()` (package/src/dpdfnet/api.py:96-104) and
 * `StreamEnhancer.process()` (package/src/dpdfnet/stream.py:129-135).  This header is what a
 * binding for that seam binds instead (ctypes stub: INTEGRATION.md).  Plain pointers and
 * sizes only; every entry point returns 0 on success or a negative DPDF_E_* code, with a
 * thread-local message available from dpdf_last_error().
 *
 * Threading: a dpdf_model is immutable after creation; calls that run work take the model's
 * internal lock (one HIP stream + one workspace per model).  Use one model per host thread for
 * concurrency (the reference uses one ORT session per thread, cli.py:251-259).
 */
#ifndef DPDFNET_HIP_H
#define DPDFNET_HIP_H

#include <stddef.h>
#include "dpdf_manifest.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DPDF_OK 0
#define DPDF_E_INVALID (-1)   /* bad argument / size mismatch       -> ValueError   */
#define DPDF_E_RUNTIME (-2)   /* HIP runtime / no device / OOM      -> RuntimeError */
#define DPDF_E_STATE   (-3)   /* stream index / state misuse        -> ValueError   */

/* flags for the pointer-taking entry points */
#define DPDF_HOST_PTRS   0    /* wav/spec/state/out are host pointers (copied over PCIe)  */
#define DPDF_DEVICE_PTRS 1    /* ... are device pointers on the model's GPU (HBM-resident) */

typedef struct dpdf_model dpdf_model;
typedef struct dpdf_streams dpdf_streams;

int dpdf_abi_version(void);
const char* dpdf_last_error(void);
int dpdf_device_count(void);

/* weight-blob layout queries (no GPU needed).  dpdf_manifest_text writes one
 * "name offset count d0,d1,.." line per tensor; returns the byte count needed (excl. NUL). */
size_t dpdf_weight_count(const dpdf_cfg* cfg);
size_t dpdf_manifest_text(const dpdf_cfg* cfg, char* buf, size_t cap);
int dpdf_query_dims(const dpdf_cfg* cfg, dpdf_dims* out);

/* Replaces build_runtime_model(onnx_path) (onnx_backend.py:81-99): builds the frame-function
 * handle on GPU `device` from the flat fp32 checkpoint blob (layout: dpdf_manifest.h).
 * BatchNorm folding and MFMA-fragment packing of every matrix happen here, once. */
int dpdf_create(const dpdf_cfg* cfg, const float* weights, size_t n_floats, int device, dpdf_model** out);
void dpdf_destroy(dpdf_model* m);

/* Replaces load_initial_state_from_metadata (onnx_backend.py:52-78).  The two norm init
 * vectors default to the reference's own initial states (include/dpdf_norm_init.h: 16 kHz linspace, 48 kHz empirical
 * tables of onnx_model/init_norms.py); this call overrides them, e.g. from `erb_norm_init` / `spec_norm_init`
 * metadata carried by a weight file. */
int dpdf_set_norm_init(dpdf_model* m, const float* erb_norm_init, int n_erb, const float* spec_norm_init, int n_spec);
int dpdf_state_size(const dpdf_model* m);                  /* S of the reference flat state */
int dpdf_initial_state(const dpdf_model* m, float* state); /* host pointer, S floats         */
int dpdf_win_len(const dpdf_model* m);                     /* infer_win_len (onnx_backend.py:102-107) */
int dpdf_hop(const dpdf_model* m);
int dpdf_freq_bins(const dpdf_model* m);
int dpdf_sample_rate(const dpdf_model* m);

/* Replaces the per-frame session.run loop (api.py:96-104 / stream.py:129-135), for B independent
 * streams and T consecutive frames in ONE call:
 *   spec   [B,T,F,2]  unnormalised STFT frames (what the reference feeds as "spec")
 *   state  [B,S]      reference flat state vectors, updated IN PLACE (state_in -> state_out)
 *   spec_e [B,T,F,2]  enhanced frames (what session.run returns as "spec_e")
 * T = 1 is exactly one session.run per stream. */
int dpdf_run_frames(dpdf_model* m, const float* spec, int B, int T, float* state, float* spec_e, int flags);

/* Replaces enhance() (api.py:51-113) from the padded-waveform stage on, for a batch of B mono
 * clips of N samples each at the model sample rate: tail pad, centre/reflect STFT, T frames of
 * the frame function from the initial state, attenuation limit, iSTFT, 2*win alignment shift,
 * fit to N.  wav/out: [B,N].  attn_limit_db: NaN or +inf = off (reference None/inf). */
int dpdf_enhance_batch(dpdf_model* m, const float* wav, int B, int N, float attn_limit_db, float* out, int flags);
int dpdf_num_frames(const dpdf_model* m, int n_samples);   /* T = 1 + (N + win)/hop */
/* Progress of the dpdf_enhance_batch* call in flight: frames (per clip) whose enhanced spectra are complete, 0 .. T.  No lock,
 * no synchronisation: meant to be polled from another host thread (reference api.py:94-104 progress_callback(t + 1, total)). */
int dpdf_progress(const dpdf_model* m);
/* The same for clips of DIFFERENT lengths in one call -- what a directory of files is (reference cli.py:222-311 runs
 * enhance_file per file on a thread pool; api.py:172-280).  wav/out: [B, n_max] rows; clip b holds lengths[b] <= n_max
 * samples (lengths: HOST array, also with DPDF_DEVICE_PTRS).  Every clip gets exactly the result of dpdf_enhance_batch
 * on it alone: its own win-sample tail pad and reflection point, its own T_b frames, its own 2*win shift / zero tail /
 * fit_length; out[b][lengths[b]:] = 0.  The frame function runs max_b T_b frames for every
 * clip (causal: padding frames cannot reach earlier outputs), so callers bucket by length to bound the waste. */
int dpdf_enhance_batch_ragged(dpdf_model* m, const float* wav, int B, int n_max, const int* lengths,
                              float attn_limit_db, float* out, int flags);
/* The same with every clip in its OWN host buffer -- a list of arrays is what `enhance_batch` / `enhance_dir` hold (reference
 * api.py:51-113 per clip, cli.py:249-259 per file): in_rows[b] holds lengths[b] samples (lengths NULL: n_max each),
 * out_rows[b] receives lengths[b] samples; no [B, n_max] block has to be assembled or taken apart by the caller.  Host
 * pointers only.
 * HOST-pointer calls of all three entry points are pipelined over time slices inside the library: the noisy PCM of time
 * chunk k+2 is gathered into pinned staging by a few copy threads and uploaded, and the enhanced PCM of chunk k-2 downloaded
 * and scattered to the caller's rows, while the GPU computes chunk k (the metric includes H2D and D2H);
 * only the first slice's upload and the last one's download are exposed.  dpdf_set_option "host_pipe" 0 = one upload,
 * compute, one download (A/B); "host_copy_threads" (default 4, the caller's thread included). */
int dpdf_enhance_batch_rows(dpdf_model* m, const float* const* in_rows, const int* lengths, int B, int n_max,
                            float attn_limit_db, float* const* out_rows, int flags);

/* Device-resident streaming (StreamEnhancer.process hot loop, stream.py:116-156, for S
 * concurrent streams): state, analysis tail and overlap-add tail live in HBM.
 *   pcm_in  [S, n_hops*hop]  new samples per stream (host or device per flags)
 *   pcm_out [S, n_hops*hop]  committed output hops
 * The very first call of a stream needs win samples before the first frame: call with the
 * stream's first hop through dpdf_streams_prime (buffers it, emits nothing).
 * Host-pointer calls return as soon as pcm_out is complete; a single hop's last state-FIFO export may still be running on the
 * model's stream at that point.  Nothing observable depends on it: every entry point that reads or writes stream state
 * (process, get_state / set_state / get_tails, reset, destroy) is ordered behind it on that stream or synchronises it first. */
int dpdf_streams_create(dpdf_model* m, int n_streams, dpdf_streams** out);
void dpdf_streams_destroy(dpdf_streams* s);
int dpdf_streams_reset(dpdf_streams* s, int stream /* -1 = all */);
int dpdf_streams_prime(dpdf_streams* s, const float* pcm_in, int flags);            /* [S,hop] */
int dpdf_streams_process(dpdf_streams* s, const float* pcm_in, int n_hops, float* pcm_out, int flags);
int dpdf_streams_get_state(dpdf_streams* s, int stream, float* state_host);          /* S floats */
/* The reference's StreamEnhancer objects are independent of each other (stream.py:13-72): each takes chunks whenever its
 * caller has them.  The masked form advances only the streams with active[i] != 0 (HOST array of S bytes, also with
 * DPDF_DEVICE_PTRS; NULL = all) by n_hops hops in ONE device call: they are packed into a dense batch, processed, unpacked.
 * Rows of pcm_in / pcm_out keep their [S, n_hops*hop] positions; rows of inactive streams are neither read nor written,
 * and those streams' state, analysis tail and overlap-add tail stay exactly as they were. */
int dpdf_streams_process_masked(dpdf_streams* s, const float* pcm_in, int n_hops, float* pcm_out,
                                const unsigned char* active, int flags);
/* Native coalescing for INDEPENDENT stream objects (the reference's usage pattern: N StreamEnhancer instances, each fed by its own
 * caller whenever it has a chunk, package/src/dpdfnet/stream.py:13-72, 74-165; replaces the per-object session.run of :129-135).
 * Any host thread submits whole hops for slots it owns; the first submitter of a round leads it: it waits -- only until every slot
 * marked in use (dpdf_streams_slot_use) has queued, and not at all while it is the only thread that has submitted lately; at most
 * `regular_window_s` (default 2 ms) for the slots that rode in the previous round (their callers are feeding the pool hop after hop
 * and are on their way back), at most `window_s` (default 200 us) for in-use slots that sat the previous round out -- then issues ONE
 * masked device call for the whole round.  dpdf_streams_pool_tune sets both windows and `spin_s`: for that long a round's leader and
 * the submitters waiting for its result poll instead of sleeping on the condition variable (a wake-up through the kernel costs tens of
 * microseconds per thread, in front of that thread's NEXT submission; default 0 = never poll).  Samples go
 * straight into the round's pinned GPU-visible block and results come straight out of it, copied by the submitting threads
 * themselves, in parallel; the next round fills while this one is on the GPU.
 *   dpdf_streams_submit_wait: k_hops whole hops for one slot in, k_hops * hop samples out (blocks until the round is done).
 *   dpdf_streams_submit_many: the same for n slots of one caller (distinct slots; requests with equal hop counts share rounds).
 *   flags bit 0 (DPDF_POOL_NO_WINDOW): the caller knows nobody else is feeding the pool -- never wait for other submitters.
 * Results are exactly those of dpdf_streams_process_masked on the same samples; errors of a round's device call reach every
 * submitter of that round. */
#define DPDF_POOL_NO_WINDOW 1
int dpdf_streams_pool_config(dpdf_streams* s, double window_s);
int dpdf_streams_pool_tune(dpdf_streams* s, double window_s, double regular_window_s, double spin_s);
int dpdf_streams_slot_use(dpdf_streams* s, int slot, int in_use);
int dpdf_streams_submit_wait(dpdf_streams* s, int slot, const float* pcm, int k_hops, float* out, int flags);
int dpdf_streams_submit_many(dpdf_streams* s, int n, const int* slots, const float* const* in_rows, const int* k_hops,
                             float* const* out_rows, int flags);
int dpdf_streams_submit_block(dpdf_streams* s, int n, const int* slots, const float* in_block /* [n][k_hops*hop] */, int k_hops,
                              float* out_block, int flags);
int dpdf_streams_pool_stats(dpdf_streams* s, long* device_calls, long* rounds);   /* masked device calls issued / rounds run; NULL = skip */
/* seconds summed over all rounds so far: [0] leaders waiting for the other submitters, [1] inside the device calls, [2] from the end of one
 * device call to the start of the next (submitters collecting, the caller's own code, coming back, the wait of [0]) */
int dpdf_streams_pool_timing(dpdf_streams* s, double* out3);
/* Resume = the explicit state vector (onnx_backend.py:52-78): `state` is the reference's flat state
 * layout (from dpdf_streams_get_state, dpdf_run_frames or the reference's own session loop); in_tail / ola_tail are the
 * StreamEnhancer's analysis buffer and overlap-add buffer (stream.py:62-72, hop floats each).  Host pointers; a NULL part
 * is left as it is; a stream given an in_tail counts as primed.  dpdf_streams_prime_one = set_state(stream, NULL, hop, NULL). */
int dpdf_streams_set_state(dpdf_streams* s, int stream, const float* state, const float* in_tail, const float* ola_tail);
int dpdf_streams_get_tails(dpdf_streams* s, int stream, float* in_tail, float* ola_tail);   /* hop floats each, NULL = skip */
int dpdf_streams_prime_one(dpdf_streams* s, int stream, const float* pcm_hop);
int dpdf_streams_is_primed(dpdf_streams* s, int stream);
/* Forward progress.  The multi-workgroup GRU-256 kernels wait for their peers under ordinary launches; a wait that times
 * out (~1 s; peers not co-resident: a GPU shared with other processes) raises a device flag.  HOST-pointer calls then
 * recover by themselves: dpdf_enhance_batch* / dpdf_run_frames start again from the caller's (or the initial) state,
 * dpdf_streams_process* from a copy of state and tails taken at the start of the call, with every GRU-256 recurrence on
 * the single-workgroup scan, which has no cross-workgroup waits -- DPDF_OK, results equal to rounding; this counter says
 * how often that happened.  DEVICE-pointer calls are asynchronous: dpdf_sync returns DPDF_E_RUNTIME, the results are
 * invalid and in-place state is half advanced (reset or restore the streams, re-issue batch calls). */
long dpdf_recovery_count(const dpdf_model* m);

/* Wait for the model's stream.  Calls with DPDF_DEVICE_PTRS return as soon as the work is queued: dpdf_sync is where
 * their completion -- and a device-side failure (DPDF_E_RUNTIME: a GRU-256 cluster exchange that timed out) -- is
 * observed.  Host-pointer calls synchronise and check before returning. */
int dpdf_sync(dpdf_model* m);
/* Test hook: raise the device error flag so that the next synchronisation point reports DPDF_E_RUNTIME (and clears it). */
int dpdf_debug_raise_device_error(dpdf_model* m);
/* Enable per-kernel-class timing (HIP events around each launch class on the model stream);
 * dpdf_profile_report writes "name total_ms calls" lines.  Off by default. */
int dpdf_profile_enable(dpdf_model* m, int on);
size_t dpdf_profile_report(dpdf_model* m, char* buf, size_t cap);
/* Set the time-chunk length used by dpdf_enhance_batch (frames per chunk; <=0 = whole clip). */
int dpdf_set_chunk_frames(dpdf_model* m, int frames);
/* Execution-shape mask (default 27 = 1|2|8|16): bit 0 stage 2 of chunk i (GRU-256 scans, decoders) on its own HIP stream
 * underneath stage 1 of chunk i+1; bit 1 the ERB encoder branch on its own stream; bit 3 the DF decoder beside the ERB decoder
 * inside stage 2; bit 4 eight / sixteen (not four) workgroups per tile in the GRU-256 cluster scans of small launches.
 * 0: everything serial on one stream (A/B timing).  Other bits are ignored. */
int dpdf_set_overlap(dpdf_model* m, int mask);
/* Where fc + LayerNorm + residual of every DPRNN block run: 2 always inside the GRU-64 scan kernels;
 * 0 always as separate GEMM kernels; 1 (default) picks per chunk -- fused once streams x frames fills the
 * chip (>= 3072 frame rows), separate below that (single-hop streaming, small batches). */
int dpdf_set_fuse_dprnn(dpdf_model* m, int mode);
/* Engine switches by name: every name, its default, what it selects and the test that exercises it are ONE table, docs/OPTIONS.md
 * (tests/test_options_table.py keeps the table, the library and the tests in step).  Two kinds: MODES a caller may want --
 * "gru64_limbs" (3 default: the GRU-64 throughput kernels on bf16 limbs, gru_limb.h: fp32-exact products on the bf16 matrix pipe; 0: the
 * fp32-MFMA kernels), "host_pipe", "host_copy_threads", "snapshot", "tail_frames" -- and A/B switches between kernel forms that give results
 * equal to rounding (measurement and recovery only).  Unknown name -> DPDF_E_INVALID. */
int dpdf_set_option(dpdf_model* m, const char* name, int value);

/* Rational polyphase resampler on the device, for `ensure_sample_rate` when the caller's rate differs from the
 * model's (reference package/src/dpdfnet/audio.py:20-27 -> librosa.resample(res_type="soxr_hq"); stream.py:112,
 * 163-165 per chunk).  Kaiser(5.0)-windowed-sinc polyphase filter of scipy.signal.resample_poly; parity with soxr
 * is unpinned (its source is not part of the reference).  in [B][n_in] -> out [B][dpdf_resample_len(n_in, ...)].
 * Model-independent; synchronous (returns after the result is complete, also for device pointers). */
long dpdf_resample_len(long n_in, int sr_in, int sr_out);      /* ceil(n_in * sr_out / sr_in); -1 on bad arguments */
int dpdf_resample(int device, const float* in, int B, long n_in, int sr_in, int sr_out, float* out, int flags);

/* Debug/test hook: copy an intermediate tensor of the last processed chunk to the host
 * ("e0","e1","e2","e3","e3_dprnn","c0","c1","c1_dprnn","emb","m","coefs","xm","feat_erb",
 * "feat_spec"; engine-native channels-last layouts).  Returns the element count, -1 if unknown. */
long dpdf_debug_fetch(dpdf_model* m, const char* name, float* host, long cap);

#ifdef __cplusplus
}
#endif
#endif /* DPDFNET_HIP_H */
