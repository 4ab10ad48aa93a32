/*
 * dpdf_oracle.h -- CPU restatement of the DPDFNet enhancement hot path (TEST INFRASTRUCTURE).
 *
 * This is the parity ORACLE: a plain-C, fp32, frame-at-a-time restatement of what the
 * reference's `dpdfnet.enhance()` / `StreamEnhancer` execute.  It is a
 * checker and the CPU baseline timed by bench.py -- never a product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinned by: tests/golden/ (npz files), captured from the reference's own PyTorch streaming modules
 * (onnx_model/dpdfnet.py, onnx_model/dpdfnet_48khz_hr.py) by tests/golden/make_golden.py in the
 * build container (the reference ships no network goldens).
 */
#ifndef DPDF_ORACLE_H
#define DPDF_ORACLE_H

#include <stddef.h>
#include "../include/dpdf_manifest.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dpdf_oracle dpdf_oracle;

/* weights: flat blob in dpdf_manifest.h order.  Returns NULL on size/cfg mismatch. */
dpdf_oracle* dpdf_oracle_create(const dpdf_cfg* cfg, const float* weights, size_t n_floats);
void dpdf_oracle_destroy(dpdf_oracle* o);
size_t dpdf_oracle_weight_count(const dpdf_cfg* cfg);
int dpdf_oracle_state_size(const dpdf_oracle* o);
int dpdf_oracle_win_len(const dpdf_oracle* o);
int dpdf_oracle_freq_bins(const dpdf_oracle* o);
/* override the two norm initial vectors (what the reference reads from ONNX metadata,
 * package/src/dpdfnet/onnx_backend.py:52-78).  NULL keeps the linspace defaults. */
void dpdf_oracle_set_norm_init(dpdf_oracle* o, const float* erb_norm_init, const float* spec_norm_init);
/* reference `initial_state()` in the flat layout (onnx_model/dpdfnet.py:726-746). */
void dpdf_oracle_initial_state(const dpdf_oracle* o, float* state);

/* One call of the exported graph: spec[F,2] (UNNORMALISED STFT frame), state_in[S] ->
 * spec_e[F,2], state_out[S]   (export_dpdfnet_to_onnx.py:14-25 + onnx_model/dpdfnet.py:748-806).
 * state_in and state_out may alias. */
void dpdf_oracle_frame(dpdf_oracle* o, const float* spec, const float* state_in,
                       float* spec_e, float* state_out);

/* Intermediate tensors of the LAST dpdf_oracle_frame call, reference layouts ([C,F'] etc.):
 * "feat_erb","feat_spec","e0","e1","e2","e3","e3_dprnn","c0","c1","c1_dprnn","emb","m","coefs".
 * Returns the element count (0 if unknown), copies min(count,cap) floats. */
int dpdf_oracle_probe(const dpdf_oracle* o, const char* name, float* out, int cap);

/* Host DSP of the offline path. */
int dpdf_oracle_num_frames(const dpdf_oracle* o, int n_samples);           /* T for enhance() */
/* package/src/dpdfnet/audio.py:104-117 on np.pad(wav,(0,win)) (api.py:88): spec[T,F,2] */
void dpdf_oracle_stft(const dpdf_oracle* o, const float* wav, int n, float* spec);
/* audio.py:120-136 + fit_length (audio.py:30-38): spec[T,F,2] -> out[n] */
void dpdf_oracle_istft(const dpdf_oracle* o, const float* spec, int T, float* out, int n);
/* audio.py:41-76.  attn_limit_db < 0 or NaN or +inf => identity. In place on spec_e. */
void dpdf_oracle_attn_limit(const float* spec_noisy, float* spec_e, int T, int F, float attn_limit_db);
/* whole `enhance()` (package/src/dpdfnet/api.py:51-113) for one mono clip at the model rate. */
void dpdf_oracle_enhance(dpdf_oracle* o, const float* wav, int n, float attn_limit_db, float* out);

/* Sample-rate conversion for ensure_sample_rate (package/src/dpdfnet/audio.py:20-27).  The reference calls
 * librosa.resample(res_type="soxr_hq"), a third-party dependency (librosa==0.11.0, soxr unpinned) that is absent from
 * /root/reference and from this image: PARITY UNPINNED against it.  Restated here is the published algorithm of
 * scipy.signal.resample_poly (Kaiser beta=5 windowed sinc, 20*max(up,down)+1 taps, centred), float64 taps and
 * accumulation; tests/test_resample.py pins it against scipy itself.  Returns the output length
 * ceil(n_in*sr_out/sr_in); writes min(len, cap) samples. */
long dpdf_oracle_resample(const float* x, long n_in, int sr_in, int sr_out, float* out, long cap);

/* ERB band widths (model/utils.py:265-324) for goldens; returns number of bands. */
int dpdf_oracle_erb_widths(const dpdf_oracle* o, int* widths, int cap);
void dpdf_oracle_window(const dpdf_oracle* o, float* w);

/* Enumerate the weight manifest into text: one "name offset count d0,d1,.." line per tensor.
 * Returns bytes needed (excluding NUL). */
size_t dpdf_oracle_manifest_text(const dpdf_cfg* cfg, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
