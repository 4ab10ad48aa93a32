/*
 * dpdf_manifest.h -- model dimensions and the canonical weight-blob layout.
 *
 * Part of the C-ABI contract (see dpdfnet_hip.h): `dpdf_create()` receives ONE flat
 * float32 blob holding every checkpoint tensor of a DPDFNet model, in the order this
 * header enumerates.  Tensor names are the reference's *streaming* state_dict keys
 * (reference: onnx_model/dpdfnet.py:28-706, onnx_model/dpdfnet_48khz_hr.py:72-780,
 * key renaming onnx_model/dpdfnet.py:876-888); tensors keep the PyTorch memory layout.
 * The only re-packing is for grouped linears (reference onnx_model/layers.py:976-1050):
 *   "<prefix>.weight" [G, Og, Ig]  (= layers.{g}.weight stacked; einsum form transposed)
 *   "<prefix>.bias"   [G*Og]
 * Buffers that are pure functions of the config (erb_fb, erb_inv_fb, windows,
 * num_batches_tracked) and the dead `enc.lsnr_fc` head (onnx_model/dpdfnet.py:242,
 * never reaches forward's outputs) are NOT part of the blob.
 *
 * Header-only, plain C99; included by the HIP library, the CPU oracle and (via a tiny
 * query entry point) the Python loader, so the three can never disagree on offsets.
 */
#ifndef DPDF_MANIFEST_H
#define DPDF_MANIFEST_H

#include <stddef.h>
#include <stdio.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Model selector.  sample_rate: 16000 | 48000 ("_48khz_hr" family).  nb = number of
 * DPRNN blocks (reference `dprnn_num_blocks`: baseline 0, dpdfnet2/4/8 -> 2/4/8). */
typedef struct dpdf_cfg {
    int sample_rate;
    int nb;
} dpdf_cfg;

/* Derived dimensions (reference defaults: onnx_model/dpdfnet.py:523-565,
 * onnx_model/dpdfnet_48khz_hr.py:586-640, export_dpdfnet_to_onnx.py:86-100). */
typedef struct dpdf_dims {
    int sr, nb;
    int win, hop;      /* 320/160 | 960/480 */
    int F;             /* rfft bins 161 | 481 */
    int E;             /* width of the "erb" feature/norm state: 32 | 481 */
    int Ec;            /* width erb_conv0 runs on: 32 | 480 (48k drops the last bin) */
    int D;             /* deep-filter bins: 96 */
    int C;             /* conv channels: 64 */
    int H;             /* GRU-256 hidden */
    int O;             /* DF order (taps): 5 */
    int s1, s2, s3;    /* erb_conv1..3 frequency strides: 2,2,1 | 3,2,2 */
    int F1, F2, F3;    /* widths after erb_conv1..3: 16,8,8 | 160,80,40 */
    int Fd;            /* width after df_conv1: 48 */
    int emb;           /* embedding width 512 */
    int is48;          /* 48 kHz HR variant: magnitude features + erb_fc_emb layers */
    float wnorm;       /* 1/(win^2/(2 hop)) (model/utils.py:164-167) */
    int state_size;    /* reference flat state vector length (onnx_model/dpdfnet.py:715-724) */
} dpdf_dims;

static inline int dpdf_get_dims(const dpdf_cfg* cfg, dpdf_dims* d) {
    memset(d, 0, sizeof(*d));
    if (!cfg || cfg->nb < 0 || cfg->nb > 64) return -1;
    d->sr = cfg->sample_rate; d->nb = cfg->nb;
    d->D = 96; d->C = 64; d->H = 256; d->O = 5; d->Fd = 48; d->emb = 512;
    if (cfg->sample_rate == 16000) {
        d->win = 320; d->hop = 160; d->F = 161; d->E = 32; d->Ec = 32;
        d->s1 = 2; d->s2 = 2; d->s3 = 1; d->is48 = 0;
    } else if (cfg->sample_rate == 48000) {
        d->win = 960; d->hop = 480; d->F = 481; d->E = 481; d->Ec = 480;
        d->s1 = 3; d->s2 = 2; d->s3 = 2; d->is48 = 1;
    } else {
        return -1;
    }
    d->F1 = d->Ec / d->s1; d->F2 = d->F1 / d->s2; d->F3 = d->F2 / d->s3;
    d->wnorm = 1.0f / ((float)d->win * (float)d->win / (2.0f * (float)d->hop));
    /* state layout, reference order (onnx_model/dpdfnet.py:737-745, 171):
       erb_norm E ; spec_norm D ; erb_conv0_buf 3*E ; dprnn_erb nb*F3*C ; df_conv0_buf 3*2*D ;
       dprnn_df nb*Fd*C ; emb_gru H ; erb_dec 2H ; df_dec 2H + 5*C*D ; mask 3*F*2 ;
       df_op coefs 3*O*D*2 + spec 5*F*2 */
    d->state_size = d->E + d->D + 3 * d->E + d->nb * d->F3 * d->C + 3 * 2 * d->D +
                    d->nb * d->Fd * d->C + d->H + 2 * d->H + 2 * d->H + 5 * d->C * d->D +
                    3 * d->F * 2 + 3 * d->O * d->D * 2 + 5 * d->F * 2;
    return 0;
}

/* Offsets of the segments of the reference flat state vector. */
typedef struct dpdf_state_layout {
    int erb_norm, spec_norm, erb_conv0_buf, dprnn_erb, df_conv0_buf, dprnn_df, emb_gru,
        erb_dec_gru, df_dec_gru, df_convp_buf, mask_buf, df_coefs_buf, df_spec_buf, total;
} dpdf_state_layout;

static inline void dpdf_get_state_layout(const dpdf_dims* d, dpdf_state_layout* L) {
    int o = 0;
    L->erb_norm = o;      o += d->E;
    L->spec_norm = o;     o += d->D;
    L->erb_conv0_buf = o; o += 3 * d->E;
    L->dprnn_erb = o;     o += d->nb * d->F3 * d->C;
    L->df_conv0_buf = o;  o += 3 * 2 * d->D;
    L->dprnn_df = o;      o += d->nb * d->Fd * d->C;
    L->emb_gru = o;       o += d->H;
    L->erb_dec_gru = o;   o += 2 * d->H;
    L->df_dec_gru = o;    o += 2 * d->H;
    L->df_convp_buf = o;  o += 5 * d->C * d->D;
    L->mask_buf = o;      o += 3 * d->F * 2;
    L->df_coefs_buf = o;  o += 3 * d->O * d->D * 2;
    L->df_spec_buf = o;   o += 5 * d->F * 2;
    L->total = o;
}

/* Manifest enumeration.  The callback sees every tensor once, in blob order. */
typedef void (*dpdf_manifest_cb)(void* ud, const char* name, const int* shape, int ndim,
                                 size_t offset, size_t count);

typedef struct dpdf__mctx {
    dpdf_manifest_cb cb; void* ud; size_t off;
} dpdf__mctx;

static inline void dpdf__emit(dpdf__mctx* m, const char* name, int ndim, int s0, int s1, int s2, int s3) {
    int shape[4] = {s0, s1, s2, s3};
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    if (m->cb) m->cb(m->ud, name, shape, ndim, m->off, n);
    m->off += n;
}
static inline void dpdf__emit2(dpdf__mctx* m, const char* prefix, const char* suffix, int ndim,
                               int s0, int s1, int s2, int s3) {
    char name[160];
    snprintf(name, sizeof(name), "%s%s", prefix, suffix);
    dpdf__emit(m, name, ndim, s0, s1, s2, s3);
}
static inline void dpdf__bn(dpdf__mctx* m, const char* prefix, int ch) {
    dpdf__emit2(m, prefix, ".weight", 1, ch, 0, 0, 0);
    dpdf__emit2(m, prefix, ".bias", 1, ch, 0, 0, 0);
    dpdf__emit2(m, prefix, ".running_mean", 1, ch, 0, 0, 0);
    dpdf__emit2(m, prefix, ".running_var", 1, ch, 0, 0, 0);
}
static inline void dpdf__gl(dpdf__mctx* m, const char* prefix, int G, int Og, int Ig) {
    dpdf__emit2(m, prefix, ".weight", 3, G, Og, Ig, 0);
    dpdf__emit2(m, prefix, ".bias", 1, G * Og, 0, 0, 0);
}
static inline void dpdf__gru(dpdf__mctx* m, const char* prefix, const char* sfx, int H, int I) {
    char n[160];
    snprintf(n, sizeof(n), "%s.weight_ih%s", prefix, sfx); dpdf__emit(m, n, 2, 3 * H, I, 0, 0);
    snprintf(n, sizeof(n), "%s.weight_hh%s", prefix, sfx); dpdf__emit(m, n, 2, 3 * H, H, 0, 0);
    snprintf(n, sizeof(n), "%s.bias_ih%s", prefix, sfx);   dpdf__emit(m, n, 1, 3 * H, 0, 0, 0);
    snprintf(n, sizeof(n), "%s.bias_hh%s", prefix, sfx);   dpdf__emit(m, n, 1, 3 * H, 0, 0, 0);
}
/* depthwise(1,3) [or sub-pixel: nsub depthwise convs] + pointwise + BN  (layers.py:761-834, 919-973) */
static inline void dpdf__sepconv(dpdf__mctx* m, const char* prefix, int C, int nsub) {
    char n[160];
    if (nsub <= 1) {
        snprintf(n, sizeof(n), "%s.0.weight", prefix); dpdf__emit(m, n, 4, C, 1, 1, 3);
    } else {
        for (int k = 0; k < nsub; ++k) {
            snprintf(n, sizeof(n), "%s.0.convs.%d.weight", prefix, k); dpdf__emit(m, n, 4, C, 1, 1, 3);
        }
    }
    snprintf(n, sizeof(n), "%s.1.weight", prefix); dpdf__emit(m, n, 4, C, C, 1, 1);
    snprintf(n, sizeof(n), "%s.2", prefix); dpdf__bn(m, n, C);
}
/* pathway conv: depthwise 1x1 (per-channel scale) + BN */
static inline void dpdf__pathconv(dpdf__mctx* m, const char* prefix, int C) {
    char n[160];
    snprintf(n, sizeof(n), "%s.0.weight", prefix); dpdf__emit(m, n, 4, C, 1, 1, 1);
    snprintf(n, sizeof(n), "%s.1", prefix); dpdf__bn(m, n, C);
}
static inline void dpdf__dprnn(dpdf__mctx* m, const char* prefix, int nb, int C) {
    char p[160], q[200];
    for (int b = 0; b < nb; ++b) {
        snprintf(p, sizeof(p), "%s.blocks.%d", prefix, b);
        snprintf(q, sizeof(q), "%s.intra_gru", p);
        dpdf__gru(m, q, "_l0", C, C);
        dpdf__gru(m, q, "_l0_reverse", C, C);
        snprintf(q, sizeof(q), "%s.fc_intra.weight", p); dpdf__emit(m, q, 2, C, 2 * C, 0, 0);
        snprintf(q, sizeof(q), "%s.fc_intra.bias", p);   dpdf__emit(m, q, 1, C, 0, 0, 0);
        snprintf(q, sizeof(q), "%s.ln_intra.weight", p); dpdf__emit(m, q, 1, C, 0, 0, 0);
        snprintf(q, sizeof(q), "%s.ln_intra.bias", p);   dpdf__emit(m, q, 1, C, 0, 0, 0);
        snprintf(q, sizeof(q), "%s.inter_gru.grucell", p);
        dpdf__gru(m, q, "", C, C);
        snprintf(q, sizeof(q), "%s.fc_inter.weight", p); dpdf__emit(m, q, 2, C, C, 0, 0);
        snprintf(q, sizeof(q), "%s.fc_inter.bias", p);   dpdf__emit(m, q, 1, C, 0, 0, 0);
        snprintf(q, sizeof(q), "%s.ln_inter.weight", p); dpdf__emit(m, q, 1, C, 0, 0, 0);
        snprintf(q, sizeof(q), "%s.ln_inter.bias", p);   dpdf__emit(m, q, 1, C, 0, 0, 0);
    }
}

/* Enumerate the blob.  Returns the total number of floats (0 on bad cfg).  cb may be NULL. */
static inline size_t dpdf_manifest(const dpdf_cfg* cfg, dpdf_manifest_cb cb, void* ud) {
    dpdf_dims d;
    if (dpdf_get_dims(cfg, &d) != 0) return 0;
    dpdf__mctx m; m.cb = cb; m.ud = ud; m.off = 0;
    const int C = d.C, H = d.H;
    /* ---- encoder (onnx_model/dpdfnet.py:74-162) ---- */
    dpdf__emit(&m, "enc.erb_conv0.1.weight", 4, C, 1, 3, 3);
    dpdf__bn(&m, "enc.erb_conv0.2", C);
    dpdf__sepconv(&m, "enc.erb_conv1", C, 1);
    dpdf__sepconv(&m, "enc.erb_conv2", C, 1);
    dpdf__sepconv(&m, "enc.erb_conv3", C, 1);
    dpdf__emit(&m, "enc.df_conv0.1.convs.0.weight", 4, C / 2, 1, 3, 3);
    dpdf__emit(&m, "enc.df_conv0.1.convs.1.weight", 4, C / 2, 1, 3, 3);
    dpdf__emit(&m, "enc.df_conv0.2.weight", 4, C, C, 1, 1);
    dpdf__bn(&m, "enc.df_conv0.3", C);
    dpdf__sepconv(&m, "enc.df_conv1", C, 1);
    dpdf__dprnn(&m, "enc.dprnn_erb", d.nb, C);
    dpdf__dprnn(&m, "enc.dprnn_df", d.nb, C);
    if (d.is48) dpdf__gl(&m, "enc.erb_fc_emb.0", 32, d.emb / 32, C * d.F3 / 32);
    dpdf__gl(&m, "enc.df_fc_emb.0", 32, d.emb / 32, C * d.Fd / 32);
    dpdf__gl(&m, "enc.emb_gru.linear_in.0", 16, H / 16, 2 * d.emb / 16);
    dpdf__gru(&m, "enc.emb_gru.gru.0.grucell", "", H, H);
    dpdf__gl(&m, "enc.emb_gru.linear_out.0", 16, d.emb / 16, H / 16);
    /* ---- ERB decoder (onnx_model/dpdfnet.py:288-323) ---- */
    dpdf__gl(&m, "erb_dec.emb_gru.linear_in.0", 16, H / 16, d.emb / 16);
    dpdf__gru(&m, "erb_dec.emb_gru.gru.0.grucell", "", H, H);
    dpdf__gru(&m, "erb_dec.emb_gru.gru.1.grucell", "", H, H);
    dpdf__gl(&m, "erb_dec.emb_gru.linear_out.0", 16, d.emb / 16, H / 16);
    if (d.is48) dpdf__gl(&m, "erb_dec.erb_fc_emb.0", 32, C * d.F3 / 32, d.emb / 32);
    dpdf__pathconv(&m, "erb_dec.conv3p", C);
    dpdf__sepconv(&m, "erb_dec.convt3", C, d.s3 > 1 ? d.s3 : 1);
    dpdf__pathconv(&m, "erb_dec.conv2p", C);
    dpdf__sepconv(&m, "erb_dec.convt2", C, d.s2);
    dpdf__pathconv(&m, "erb_dec.conv1p", C);
    dpdf__sepconv(&m, "erb_dec.convt1", C, d.s1);
    dpdf__pathconv(&m, "erb_dec.conv0p", C);
    dpdf__emit(&m, "erb_dec.conv0_out.0.weight", 4, 1, C, 1, 3);
    dpdf__bn(&m, "erb_dec.conv0_out.1", 1);
    /* ---- DF decoder (onnx_model/dpdfnet.py:424-458) ---- */
    dpdf__emit(&m, "df_dec.df_convp.1.convs.0.weight", 4, d.O, C / 2, 5, 1);
    dpdf__emit(&m, "df_dec.df_convp.1.convs.1.weight", 4, d.O, C / 2, 5, 1);
    dpdf__emit(&m, "df_dec.df_convp.2.weight", 4, 2 * d.O, 2 * d.O, 1, 1);
    dpdf__bn(&m, "df_dec.df_convp.3", 2 * d.O);
    dpdf__gl(&m, "df_dec.df_gru.linear_in.0", 8, H / 8, d.emb / 8);
    dpdf__gru(&m, "df_dec.df_gru.gru.0.grucell", "", H, H);
    dpdf__gru(&m, "df_dec.df_gru.gru.1.grucell", "", H, H);
    dpdf__gl(&m, "df_dec.df_skip", 16, H / 16, d.emb / 16);
    dpdf__gl(&m, "df_dec.df_out.0", 16, d.D * 2 * d.O / 16, H / 16);
    return m.off;
}

#ifdef __cplusplus
}
#endif
#endif /* DPDF_MANIFEST_H */
