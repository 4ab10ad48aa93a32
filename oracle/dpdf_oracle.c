/*
 * dpdf_oracle.c -- CPU restatement of the DPDFNet enhancement hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see dpdf_oracle.h).  Plain C99, fp32, batch 1, one frame per call,
 * explicit flat state in the reference's own layout -- i.e. the reference's execution model
 * (package/src/dpdfnet/api.py:96-104, one session.run per 10 ms frame).
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * repository root).  Nothing here is shared with the HIP product path except the weight-blob
 * layout header include/dpdf_manifest.h.
 */
#include "dpdf_oracle.h"
#include "../include/dpdf_norm_init.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* torch.relu / F.relu propagate NaN (relu(NaN) = NaN: the reference's frame function keeps a NaN that has entered a clip in every state
   of that clip); `v > 0 ? v : 0` would turn it into 0.  tests/golden/nonfinite_*.npz pins this. */
static inline float relu_f(float v) { return v < 0.f ? 0.f : v; }

#define MAXC 64

typedef struct { const float *w, *b, *mean, *var; } bn_t;
typedef struct { const float *w_ih, *w_hh, *b_ih, *b_hh; float *w_ih_t, *w_hh_t; int H, I; } gru_t;
typedef struct { const float *w, *b; int G, Og, Ig; } gl_t;
typedef struct {
    const float* dw[3]; int nsub; const float* pw; float* pw_t; bn_t bn;
} sepconv_t;
typedef struct { const float* scale; bn_t bn; } pathconv_t;
typedef struct {
    gru_t intra_f, intra_b, inter;
    const float *fc_intra_w, *fc_intra_b, *ln_intra_w, *ln_intra_b;
    const float *fc_inter_w, *fc_inter_b, *ln_inter_w, *ln_inter_b;
    float *fc_intra_wt, *fc_inter_wt;
} dprnn_block_t;

struct dpdf_oracle {
    dpdf_cfg cfg;
    dpdf_dims d;
    dpdf_state_layout L;
    float* blob;
    size_t n_blob;
    /* name table */
    char (*names)[160];
    size_t* offs;
    int n_names, cap_names;
    /* derived constants */
    int band_of[512];      /* bin -> erb band */
    int band_w[64];
    float window[960];
    double* dft_cos;       /* [win] cos(2 pi j / win) */
    double* dft_sin;
    float erb_norm_init[512];
    float spec_norm_init[96];
    /* layers */
    const float* erb_conv0_w; bn_t erb_conv0_bn;
    sepconv_t erb_conv1, erb_conv2, erb_conv3, df_conv1;
    const float *df_conv0_w0, *df_conv0_w1, *df_conv0_pw; float* df_conv0_pw_t; bn_t df_conv0_bn;
    dprnn_block_t* dprnn_erb; dprnn_block_t* dprnn_df;
    gl_t enc_erb_fc_emb, df_fc_emb, enc_lin_in, enc_lin_out;
    gru_t enc_gru;
    gl_t erbdec_lin_in, erbdec_lin_out, erbdec_erb_fc_emb;
    gru_t erbdec_gru0, erbdec_gru1;
    pathconv_t conv3p, conv2p, conv1p, conv0p;
    sepconv_t convt3, convt2, convt1;
    const float* conv0_out_w; bn_t conv0_out_bn;
    const float *df_convp_w0, *df_convp_w1, *df_convp_pw; bn_t df_convp_bn;
    gl_t df_lin_in, df_skip, df_out;
    gru_t df_gru0, df_gru1;
    /* probes (last frame) */
    float *p_feat_erb, *p_feat_spec, *p_e0, *p_e1, *p_e2, *p_e3, *p_e3d, *p_c0, *p_c1, *p_c1d,
          *p_emb, *p_m, *p_coefs;
};

/* ------------------------------------------------------------------------------------------ */
/* manifest lookup                                                                             */
/* ------------------------------------------------------------------------------------------ */
static void collect_cb(void* ud, const char* name, const int* shape, int ndim, size_t off, size_t cnt) {
    (void)shape; (void)ndim; (void)cnt;
    dpdf_oracle* o = (dpdf_oracle*)ud;
    if (o->n_names == o->cap_names) {
        o->cap_names = o->cap_names ? o->cap_names * 2 : 256;
        o->names = realloc(o->names, (size_t)o->cap_names * sizeof(*o->names));
        o->offs = realloc(o->offs, (size_t)o->cap_names * sizeof(size_t));
    }
    strncpy(o->names[o->n_names], name, 159);
    o->names[o->n_names][159] = 0;
    o->offs[o->n_names] = off;
    o->n_names++;
}
static const float* W(const dpdf_oracle* o, const char* name) {
    for (int i = 0; i < o->n_names; ++i)
        if (strcmp(o->names[i], name) == 0) return o->blob + o->offs[i];
    fprintf(stderr, "dpdf_oracle: missing tensor %s\n", name);
    abort();
}
static const float* W2(const dpdf_oracle* o, const char* prefix, const char* sfx) {
    char n[200];
    snprintf(n, sizeof(n), "%s%s", prefix, sfx);
    return W(o, n);
}
static bn_t get_bn(const dpdf_oracle* o, const char* prefix) {
    bn_t b;
    b.w = W2(o, prefix, ".weight"); b.b = W2(o, prefix, ".bias");
    b.mean = W2(o, prefix, ".running_mean"); b.var = W2(o, prefix, ".running_var");
    return b;
}
static float* transpose_new(const float* w, int rows, int cols) { /* w[rows][cols] -> t[cols][rows] */
    float* t = malloc(sizeof(float) * (size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return t;
}
static gru_t get_gru(const dpdf_oracle* o, const char* prefix, const char* sfx, int H, int I) {
    gru_t g; char n[200];
    snprintf(n, sizeof(n), "%s.weight_ih%s", prefix, sfx); g.w_ih = W(o, n);
    snprintf(n, sizeof(n), "%s.weight_hh%s", prefix, sfx); g.w_hh = W(o, n);
    snprintf(n, sizeof(n), "%s.bias_ih%s", prefix, sfx);   g.b_ih = W(o, n);
    snprintf(n, sizeof(n), "%s.bias_hh%s", prefix, sfx);   g.b_hh = W(o, n);
    g.H = H; g.I = I;
    g.w_ih_t = transpose_new(g.w_ih, 3 * H, I);
    g.w_hh_t = transpose_new(g.w_hh, 3 * H, H);
    return g;
}
static gl_t get_gl(const dpdf_oracle* o, const char* prefix, int G, int Og, int Ig) {
    gl_t g; g.w = W2(o, prefix, ".weight"); g.b = W2(o, prefix, ".bias"); g.G = G; g.Og = Og; g.Ig = Ig;
    return g;
}
static sepconv_t get_sepconv(const dpdf_oracle* o, const char* prefix, int nsub) {
    sepconv_t s; char n[200];
    memset(&s, 0, sizeof(s));
    s.nsub = nsub;
    if (nsub <= 1) { s.dw[0] = W2(o, prefix, ".0.weight"); s.nsub = 1; }
    else for (int k = 0; k < nsub; ++k) { snprintf(n, sizeof(n), "%s.0.convs.%d.weight", prefix, k); s.dw[k] = W(o, n); }
    s.pw = W2(o, prefix, ".1.weight");
    s.pw_t = transpose_new(s.pw, MAXC, MAXC);
    snprintf(n, sizeof(n), "%s.2", prefix); s.bn = get_bn(o, n);
    return s;
}
static pathconv_t get_pathconv(const dpdf_oracle* o, const char* prefix) {
    pathconv_t p; char n[200];
    p.scale = W2(o, prefix, ".0.weight");
    snprintf(n, sizeof(n), "%s.1", prefix); p.bn = get_bn(o, n);
    return p;
}
static dprnn_block_t* get_dprnn(const dpdf_oracle* o, const char* prefix, int nb) {
    dprnn_block_t* b = calloc((size_t)(nb > 0 ? nb : 1), sizeof(*b));
    char p[200], q[256];
    for (int i = 0; i < nb; ++i) {
        snprintf(p, sizeof(p), "%s.blocks.%d", prefix, i);
        snprintf(q, sizeof(q), "%s.intra_gru", p);
        b[i].intra_f = get_gru(o, q, "_l0", 64, 64);
        b[i].intra_b = get_gru(o, q, "_l0_reverse", 64, 64);
        b[i].fc_intra_w = W2(o, p, ".fc_intra.weight"); b[i].fc_intra_b = W2(o, p, ".fc_intra.bias");
        b[i].ln_intra_w = W2(o, p, ".ln_intra.weight"); b[i].ln_intra_b = W2(o, p, ".ln_intra.bias");
        snprintf(q, sizeof(q), "%s.inter_gru.grucell", p);
        b[i].inter = get_gru(o, q, "", 64, 64);
        b[i].fc_inter_w = W2(o, p, ".fc_inter.weight"); b[i].fc_inter_b = W2(o, p, ".fc_inter.bias");
        b[i].ln_inter_w = W2(o, p, ".ln_inter.weight"); b[i].ln_inter_b = W2(o, p, ".ln_inter.bias");
        b[i].fc_intra_wt = transpose_new(b[i].fc_intra_w, 64, 128);
        b[i].fc_inter_wt = transpose_new(b[i].fc_inter_w, 64, 64);
    }
    return b;
}

/* ------------------------------------------------------------------------------------------ */
/* constants: window, ERB bands                                                                */
/* ------------------------------------------------------------------------------------------ */
/* package/src/dpdfnet/audio.py:84-88 == model/utils.py:153-161 */
static void make_vorbis_window(float* w, int n) {
    const double h = n / 2.0;
    for (int i = 0; i < n; ++i) {
        double s = sin(0.5 * M_PI * (i + 0.5) / h);
        w[i] = (float)sin(0.5 * M_PI * s * s);
    }
}
/* model/utils.py:265-324 (note the hard-coded range(33) at :304 -- n_filters is 32 on every
 * shipped config, so bins[] has 33 entries either way). */
static void make_erb_bands(int nfft, int fs, int n_filters, int min_nb_freqs, int* band_w, int* band_of) {
    const double nyq = fs / 2.0, freq_width = (double)fs / nfft;
    const double erb_low = 9.265 * log1p(0.0 / (24.7 * 9.265));
    const double erb_high = 9.265 * log1p(nyq / (24.7 * 9.265));
    const double step = (erb_high - erb_low) / n_filters;
    int bins[65];
    for (int i = 0; i <= n_filters; ++i) {
        double f = 24.7 * 9.265 * (exp((erb_low + i * step) / 9.265) - 1.0);
        bins[i] = (int)rint(f / freq_width); /* python round(): half-to-even, as rint() */
    }
    bins[n_filters] = nfft / 2 + 1;
    int freq_over = 0;
    for (int f = 0; f < nfft / 2 + 1; ++f) band_of[f] = -1;
    for (int j = 0; j < n_filters; ++j) {
        int alpha = bins[j] + freq_over, beta = bins[j + 1];
        if (beta - alpha < min_nb_freqs) {
            freq_over = min_nb_freqs - (beta - alpha);
            beta = beta + freq_over < nfft / 2 + 1 ? beta + freq_over : nfft / 2 + 1;
        } else {
            freq_over = 0;
        }
        band_w[j] = beta - alpha;
        for (int f = alpha; f < beta; ++f) band_of[f] = j;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* create / destroy                                                                            */
/* ------------------------------------------------------------------------------------------ */
size_t dpdf_oracle_weight_count(const dpdf_cfg* cfg) { return dpdf_manifest(cfg, NULL, NULL); }

static float* probe_alloc(int n) { return calloc((size_t)n, sizeof(float)); }

dpdf_oracle* dpdf_oracle_create(const dpdf_cfg* cfg, const float* weights, size_t n_floats) {
    dpdf_dims d;
    if (dpdf_get_dims(cfg, &d) != 0) return NULL;
    if (dpdf_manifest(cfg, NULL, NULL) != n_floats) return NULL;
    dpdf_oracle* o = calloc(1, sizeof(*o));
    o->cfg = *cfg; o->d = d;
    dpdf_get_state_layout(&d, &o->L);
    o->blob = malloc(sizeof(float) * n_floats);
    memcpy(o->blob, weights, sizeof(float) * n_floats);
    o->n_blob = n_floats;
    dpdf_manifest(cfg, collect_cb, o);

    make_vorbis_window(o->window, d.win);
    if (!d.is48) make_erb_bands(d.win, d.sr, 32, 1, o->band_w, o->band_of);
    o->dft_cos = malloc(sizeof(double) * d.win);
    o->dft_sin = malloc(sizeof(double) * d.win);
    for (int j = 0; j < d.win; ++j) {
        o->dft_cos[j] = cos(2.0 * M_PI * j / d.win);
        o->dft_sin[j] = sin(2.0 * M_PI * j / d.win);
    }
    /* Initial norm states: 16 kHz ErbNorm / SpecNorm linspace (onnx_model/layers.py:455-463, 516-522; torch computes
       init0 + arange(n) * step in float32), 48 kHz MagNorm48 / SpecNorm48 empirical tables (onnx_model/init_norms.py:21-139
       via layers.py:575-730) -- shared constant header include/dpdf_norm_init.h. */
    dpdf_default_norm_init(&d, o->erb_norm_init, o->spec_norm_init);

    o->erb_conv0_w = W(o, "enc.erb_conv0.1.weight"); o->erb_conv0_bn = get_bn(o, "enc.erb_conv0.2");
    o->erb_conv1 = get_sepconv(o, "enc.erb_conv1", 1);
    o->erb_conv2 = get_sepconv(o, "enc.erb_conv2", 1);
    o->erb_conv3 = get_sepconv(o, "enc.erb_conv3", 1);
    o->df_conv0_w0 = W(o, "enc.df_conv0.1.convs.0.weight");
    o->df_conv0_w1 = W(o, "enc.df_conv0.1.convs.1.weight");
    o->df_conv0_pw = W(o, "enc.df_conv0.2.weight");
    o->df_conv0_pw_t = transpose_new(o->df_conv0_pw, 64, 64);
    o->df_conv0_bn = get_bn(o, "enc.df_conv0.3");
    o->df_conv1 = get_sepconv(o, "enc.df_conv1", 1);
    o->dprnn_erb = get_dprnn(o, "enc.dprnn_erb", d.nb);
    o->dprnn_df = get_dprnn(o, "enc.dprnn_df", d.nb);
    if (d.is48) o->enc_erb_fc_emb = get_gl(o, "enc.erb_fc_emb.0", 32, d.emb / 32, d.C * d.F3 / 32);
    o->df_fc_emb = get_gl(o, "enc.df_fc_emb.0", 32, d.emb / 32, d.C * d.Fd / 32);
    o->enc_lin_in = get_gl(o, "enc.emb_gru.linear_in.0", 16, d.H / 16, 2 * d.emb / 16);
    o->enc_gru = get_gru(o, "enc.emb_gru.gru.0.grucell", "", d.H, d.H);
    o->enc_lin_out = get_gl(o, "enc.emb_gru.linear_out.0", 16, d.emb / 16, d.H / 16);
    o->erbdec_lin_in = get_gl(o, "erb_dec.emb_gru.linear_in.0", 16, d.H / 16, d.emb / 16);
    o->erbdec_gru0 = get_gru(o, "erb_dec.emb_gru.gru.0.grucell", "", d.H, d.H);
    o->erbdec_gru1 = get_gru(o, "erb_dec.emb_gru.gru.1.grucell", "", d.H, d.H);
    o->erbdec_lin_out = get_gl(o, "erb_dec.emb_gru.linear_out.0", 16, d.emb / 16, d.H / 16);
    if (d.is48) o->erbdec_erb_fc_emb = get_gl(o, "erb_dec.erb_fc_emb.0", 32, d.C * d.F3 / 32, d.emb / 32);
    o->conv3p = get_pathconv(o, "erb_dec.conv3p");
    o->convt3 = get_sepconv(o, "erb_dec.convt3", d.s3 > 1 ? d.s3 : 1);
    o->conv2p = get_pathconv(o, "erb_dec.conv2p");
    o->convt2 = get_sepconv(o, "erb_dec.convt2", d.s2);
    o->conv1p = get_pathconv(o, "erb_dec.conv1p");
    o->convt1 = get_sepconv(o, "erb_dec.convt1", d.s1);
    o->conv0p = get_pathconv(o, "erb_dec.conv0p");
    o->conv0_out_w = W(o, "erb_dec.conv0_out.0.weight"); o->conv0_out_bn = get_bn(o, "erb_dec.conv0_out.1");
    o->df_convp_w0 = W(o, "df_dec.df_convp.1.convs.0.weight");
    o->df_convp_w1 = W(o, "df_dec.df_convp.1.convs.1.weight");
    o->df_convp_pw = W(o, "df_dec.df_convp.2.weight");
    o->df_convp_bn = get_bn(o, "df_dec.df_convp.3");
    o->df_lin_in = get_gl(o, "df_dec.df_gru.linear_in.0", 8, d.H / 8, d.emb / 8);
    o->df_gru0 = get_gru(o, "df_dec.df_gru.gru.0.grucell", "", d.H, d.H);
    o->df_gru1 = get_gru(o, "df_dec.df_gru.gru.1.grucell", "", d.H, d.H);
    o->df_skip = get_gl(o, "df_dec.df_skip", 16, d.H / 16, d.emb / 16);
    o->df_out = get_gl(o, "df_dec.df_out.0", 16, d.D * 2 * d.O / 16, d.H / 16);

    o->p_feat_erb = probe_alloc(d.E); o->p_feat_spec = probe_alloc(2 * d.D);
    o->p_e0 = probe_alloc(64 * d.Ec); o->p_e1 = probe_alloc(64 * d.F1); o->p_e2 = probe_alloc(64 * d.F2);
    o->p_e3 = probe_alloc(64 * d.F3); o->p_e3d = probe_alloc(64 * d.F3);
    o->p_c0 = probe_alloc(64 * d.D); o->p_c1 = probe_alloc(64 * d.Fd); o->p_c1d = probe_alloc(64 * d.Fd);
    o->p_emb = probe_alloc(d.emb); o->p_m = probe_alloc(d.F); o->p_coefs = probe_alloc(d.O * d.D * 2);
    return o;
}

static void free_gru(gru_t* g) { free(g->w_ih_t); free(g->w_hh_t); }
void dpdf_oracle_destroy(dpdf_oracle* o) {
    if (!o) return;
    free(o->blob); free(o->names); free(o->offs); free(o->dft_cos); free(o->dft_sin);
    free(o->erb_conv1.pw_t); free(o->erb_conv2.pw_t); free(o->erb_conv3.pw_t); free(o->df_conv1.pw_t);
    free(o->convt3.pw_t); free(o->convt2.pw_t); free(o->convt1.pw_t); free(o->df_conv0_pw_t);
    for (int i = 0; i < o->d.nb; ++i) {
        dprnn_block_t* bs[2] = {&o->dprnn_erb[i], &o->dprnn_df[i]};
        for (int k = 0; k < 2; ++k) {
            free_gru(&bs[k]->intra_f); free_gru(&bs[k]->intra_b); free_gru(&bs[k]->inter);
            free(bs[k]->fc_intra_wt); free(bs[k]->fc_inter_wt);
        }
    }
    free(o->dprnn_erb); free(o->dprnn_df);
    free_gru(&o->enc_gru); free_gru(&o->erbdec_gru0); free_gru(&o->erbdec_gru1);
    free_gru(&o->df_gru0); free_gru(&o->df_gru1);
    free(o->p_feat_erb); free(o->p_feat_spec); free(o->p_e0); free(o->p_e1); free(o->p_e2); free(o->p_e3);
    free(o->p_e3d); free(o->p_c0); free(o->p_c1); free(o->p_c1d); free(o->p_emb); free(o->p_m); free(o->p_coefs);
    free(o);
}

int dpdf_oracle_state_size(const dpdf_oracle* o) { return o->d.state_size; }
int dpdf_oracle_win_len(const dpdf_oracle* o) { return o->d.win; }
int dpdf_oracle_freq_bins(const dpdf_oracle* o) { return o->d.F; }
void dpdf_oracle_set_norm_init(dpdf_oracle* o, const float* e, const float* s) {
    if (e) memcpy(o->erb_norm_init, e, sizeof(float) * o->d.E);
    if (s) memcpy(o->spec_norm_init, s, sizeof(float) * o->d.D);
}
/* onnx_model/dpdfnet.py:726-746: zeros except the two norm states. */
void dpdf_oracle_initial_state(const dpdf_oracle* o, float* state) {
    memset(state, 0, sizeof(float) * o->d.state_size);
    memcpy(state + o->L.erb_norm, o->erb_norm_init, sizeof(float) * o->d.E);
    memcpy(state + o->L.spec_norm, o->spec_norm_init, sizeof(float) * o->d.D);
}
int dpdf_oracle_erb_widths(const dpdf_oracle* o, int* w, int cap) {
    if (o->d.is48) return 0;
    for (int i = 0; i < 32 && i < cap; ++i) w[i] = o->band_w[i];
    return 32;
}
void dpdf_oracle_window(const dpdf_oracle* o, float* w) { memcpy(w, o->window, sizeof(float) * o->d.win); }

/* ------------------------------------------------------------------------------------------ */
/* elementary ops                                                                              */
/* ------------------------------------------------------------------------------------------ */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[n] = b[n] + sum_k wt[k][n] * x[k]   (wt = W^T so the inner loop is contiguous) */
static void matvec_t(float* y, const float* wt, const float* b, const float* x, int K, int N) {
    if (b) memcpy(y, b, sizeof(float) * N); else memset(y, 0, sizeof(float) * N);
    for (int k = 0; k < K; ++k) {
        const float xk = x[k];
        const float* w = wt + (size_t)k * N;
        for (int n = 0; n < N; ++n) y[n] += w[n] * xk;
    }
}
/* nn.BatchNorm2d eval (eps 1e-5): SURVEY appendix A.1 */
static inline float bn_apply(const bn_t* bn, int c, float x) {
    return (x - bn->mean[c]) / sqrtf(bn->var[c] + 1e-5f) * bn->w[c] + bn->b[c];
}
/* nn.GRUCell (onnx_model/layers.py:1211, 1257): gates (r,z,n) */
static void gru_cell(const gru_t* g, const float* x, const float* h, float* h_out) {
    float gi[768], gh[768];
    const int H = g->H;
    matvec_t(gi, g->w_ih_t, g->b_ih, x, g->I, 3 * H);
    matvec_t(gh, g->w_hh_t, g->b_hh, h, H, 3 * H);
    for (int j = 0; j < H; ++j) {
        float r = sigmoidf_(gi[j] + gh[j]);
        float z = sigmoidf_(gi[H + j] + gh[H + j]);
        float n = tanhf(gi[2 * H + j] + r * gh[2 * H + j]);
        h_out[j] = (1.0f - z) * n + z * h[j];
    }
}
/* nn.LayerNorm(64), eps 1e-5, biased variance (onnx_model/layers.py:134,139) */
static void layer_norm64(float* y, const float* x, const float* w, const float* b) {
    float mean = 0.f;
    for (int i = 0; i < 64; ++i) mean += x[i];
    mean /= 64.f;
    float var = 0.f;
    for (int i = 0; i < 64; ++i) { float d = x[i] - mean; var += d * d; }
    var /= 64.f;
    float inv = 1.0f / sqrtf(var + 1e-5f);
    for (int i = 0; i < 64; ++i) y[i] = (x[i] - mean) * inv * w[i] + b[i];
}
/* GroupedLinear / GroupedLinearEinsum (onnx_model/layers.py:1008-1013, 1035-1046) */
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
static void grouped_linear(const gl_t* g, const float* x, float* y, int act) {
    for (int gi = 0; gi < g->G; ++gi)
        for (int o = 0; o < g->Og; ++o) {
            const float* w = g->w + ((size_t)gi * g->Og + o) * g->Ig;
            const float* xi = x + (size_t)gi * g->Ig;
            float acc = 0.f;
            for (int i = 0; i < g->Ig; ++i) acc += w[i] * xi[i];
            acc += g->b[gi * g->Og + o];
            if (act == ACT_RELU) acc = relu_f(acc);
            else if (act == ACT_TANH) acc = tanhf(acc);
            y[gi * g->Og + o] = acc;
        }
}
/* CyclicBuffer.forward (onnx_model/layers.py:68-107): next = cat(buf[1:], x) */
static void fifo_push(const float* buf_in, float* buf_out, const float* x, int cap, int frame) {
    if (buf_out != buf_in) memmove(buf_out, buf_in + frame, sizeof(float) * (size_t)(cap - 1) * frame);
    else memmove(buf_out, buf_out + frame, sizeof(float) * (size_t)(cap - 1) * frame);
    memcpy(buf_out + (size_t)(cap - 1) * frame, x, sizeof(float) * frame);
}
/* depthwise k(1,3) conv, pad 1, stride s on [C][Fin] -> [C][Fout] (Conv2d groups=C,
 * onnx_model/layers.py:813-824) */
static int dwconv3(const float* w /*[C][3]*/, const float* x, float* y, int Fin, int s) {
    const int Fout = (Fin + 2 - 3) / s + 1;
    for (int c = 0; c < 64; ++c)
        for (int fo = 0; fo < Fout; ++fo) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) {
                int fi = fo * s + k - 1;
                if (fi >= 0 && fi < Fin) acc += w[c * 3 + k] * x[c * Fin + fi];
            }
            y[c * Fout + fo] = acc;
        }
    return Fout;
}
/* pointwise 64->64 + BN + ReLU on [C][F] */
static void pw_bn_relu(const float* pw_t, const bn_t* bn, const float* x, float* y, int F) {
    float col[64], out[64];
    for (int f = 0; f < F; ++f) {
        for (int c = 0; c < 64; ++c) col[c] = x[c * F + f];
        matvec_t(out, pw_t, NULL, col, 64, 64);
        for (int oc = 0; oc < 64; ++oc) {
            float v = bn_apply(bn, oc, out[oc]);
            y[oc * F + f] = relu_f(v);
        }
    }
}
/* Conv2dNormAct separable: depthwise(1,3,stride) + pointwise + BN + ReLU (layers.py:761-834) */
static int sepconv(const sepconv_t* s, const float* x, float* y, int Fin, int stride) {
    float tmp[64 * 480];
    int Fout = dwconv3(s->dw[0], x, tmp, Fin, stride);
    pw_bn_relu(s->pw_t, &s->bn, tmp, y, Fout);
    return Fout;
}
/* SubPixelConv2dNormAct (layers.py:895-973): nsub depthwise convs interleaved along F */
static int subpixel_conv(const sepconv_t* s, const float* x, float* y, int Fin) {
    float tmp[64 * 480], up[64 * 480];
    const int S = s->nsub, Fout = Fin * S;
    for (int k = 0; k < S; ++k) {
        dwconv3(s->dw[k], x, tmp, Fin, 1);
        for (int c = 0; c < 64; ++c)
            for (int f = 0; f < Fin; ++f) up[c * Fout + f * S + k] = tmp[c * Fin + f];
    }
    pw_bn_relu(s->pw_t, &s->bn, up, y, Fout);
    return Fout;
}
/* pathway conv: depthwise 1x1 + BN + ReLU, then + add */
static void pathconv_add(const pathconv_t* p, const float* e, const float* add, float* y, int F) {
    for (int c = 0; c < 64; ++c)
        for (int f = 0; f < F; ++f) {
            float v = bn_apply(&p->bn, c, p->scale[c] * e[c * F + f]);
            v = relu_f(v);
            y[c * F + f] = v + add[c * F + f];
        }
}

/* DPRNNBlock.forward (onnx_model/layers.py:159-196).  x: [C=64][F'] in/out, h_state: [F'][64]. */
static void dprnn_block(const dprnn_block_t* b, float* x, int Fp, const float* h_in, float* h_out) {
    float xf[480][64];       /* (f, c) view */
    static __thread float hf[480][64], hb[480][64];
    float h[64], hn[64], cat[128], fc[64], ln[64];
    for (int f = 0; f < Fp; ++f)
        for (int c = 0; c < 64; ++c) xf[f][c] = x[c * Fp + f];
    /* intra bi-GRU over f, h0 = 0 */
    memset(h, 0, sizeof(h));
    for (int f = 0; f < Fp; ++f) { gru_cell(&b->intra_f, xf[f], h, hn); memcpy(h, hn, sizeof(h)); memcpy(hf[f], hn, sizeof(h)); }
    memset(h, 0, sizeof(h));
    for (int f = Fp - 1; f >= 0; --f) { gru_cell(&b->intra_b, xf[f], h, hn); memcpy(h, hn, sizeof(h)); memcpy(hb[f], hn, sizeof(h)); }
    for (int f = 0; f < Fp; ++f) {
        memcpy(cat, hf[f], sizeof(float) * 64); memcpy(cat + 64, hb[f], sizeof(float) * 64);
        matvec_t(fc, b->fc_intra_wt, b->fc_intra_b, cat, 128, 64);
        layer_norm64(ln, fc, b->ln_intra_w, b->ln_intra_b);
        for (int c = 0; c < 64; ++c) xf[f][c] += ln[c];
    }
    /* inter GRUCell: one hidden state per f, shared weights */
    for (int f = 0; f < Fp; ++f) {
        gru_cell(&b->inter, xf[f], h_in + f * 64, hn);
        memcpy(h_out + f * 64, hn, sizeof(float) * 64);
        matvec_t(fc, b->fc_inter_wt, b->fc_inter_b, hn, 64, 64);
        layer_norm64(ln, fc, b->ln_inter_w, b->ln_inter_b);
        for (int c = 0; c < 64; ++c) xf[f][c] += ln[c];
    }
    for (int f = 0; f < Fp; ++f)
        for (int c = 0; c < 64; ++c) x[c * Fp + f] = xf[f][c];
}

/* ------------------------------------------------------------------------------------------ */
/* the frame function                                                                          */
/* ------------------------------------------------------------------------------------------ */
void dpdf_oracle_frame(dpdf_oracle* o, const float* spec_in, const float* state_in,
                       float* spec_e, float* state_out) {
    const dpdf_dims* d = &o->d;
    const dpdf_state_layout* L = &o->L;
    const int F = d->F, D = d->D, E = d->E;
    /* work on a private copy of the state so state_in/state_out may alias */
    float* st = malloc(sizeof(float) * d->state_size);
    memcpy(st, state_in, sizeof(float) * d->state_size);
    float* so = state_out;

    /* export wrapper: spec * wnorm (export_dpdfnet_to_onnx.py:22) */
    static __thread float spec[481 * 2];
    for (int i = 0; i < 2 * F; ++i) spec[i] = spec_in[i] * d->wnorm;

    /* ---- _feature_extraction (onnx_model/dpdfnet.py:815-852; 48k: dpdfnet_48khz_hr.py:887-924) */
    float feat_erb[481];
    const float a = 0.98f, b1 = (float)(1.0 - 0.98);
    if (!d->is48) {
        float band[32];
        for (int e = 0; e < 32; ++e) band[e] = 0.f;
        for (int f = 0; f < F; ++f) {
            float p = spec[2 * f] * spec[2 * f] + spec[2 * f + 1] * spec[2 * f + 1]; /* get_pow utils.py:9-11 */
            int e = o->band_of[f];
            band[e] += p * (1.0f / (float)o->band_w[e]);           /* @ erb_fb (dpdfnet.py:584-591) */
        }
        for (int e = 0; e < 32; ++e) feat_erb[e] = 10.0f * log10f(band[e] + 1e-10f); /* to_db utils.py:84 */
    } else {
        for (int f = 0; f < F; ++f) {
            float mag = sqrtf(spec[2 * f] * spec[2 * f] + spec[2 * f + 1] * spec[2 * f + 1]); /* get_mag */
            feat_erb[f] = 10.0f * log10f(mag + 1e-10f);
        }
    }
    /* ErbNorm (layers.py:498-506) / MagNorm48 (layers.py:637-661, var0 = 40^2, eps 1e-12) */
    for (int e = 0; e < E; ++e) {
        float mu = a * st[L->erb_norm + e] + b1 * feat_erb[e];
        so[L->erb_norm + e] = mu;
        feat_erb[e] = d->is48 ? (feat_erb[e] - mu) / (sqrtf(1600.0f) + 1e-12f) : (feat_erb[e] - mu) / 40.0f;
    }
    /* SpecNorm (layers.py:561-572) -> channel-first [2][D] (dpdfnet.py:769) */
    float feat_spec[2 * 96];
    for (int f = 0; f < D; ++f) {
        float re = spec[2 * f], im = spec[2 * f + 1];
        float mag = sqrtf(re * re + im * im);
        float s = a * st[L->spec_norm + f] + b1 * mag;
        so[L->spec_norm + f] = s;
        float den = sqrtf(s + 1e-12f);
        feat_spec[f] = re / den;
        feat_spec[D + f] = im / den;
    }
    memcpy(o->p_feat_erb, feat_erb, sizeof(float) * E);
    memcpy(o->p_feat_spec, feat_spec, sizeof(float) * 2 * D);

    /* ---- Encoder.forward (onnx_model/dpdfnet.py:193-246) ---- */
    static __thread float e0[64 * 480], e1[64 * 160], e2[64 * 80], e3[64 * 40], e3d[64 * 40];
    static __thread float c0[64 * 96], c1[64 * 48], tmp[64 * 480];
    const int Ec = d->Ec;
    /* erb_conv0_buffer: FIFO of 3 frames of [E] */
    fifo_push(st + L->erb_conv0_buf, so + L->erb_conv0_buf, feat_erb, 3, E);
    {
        const float* buf = so + L->erb_conv0_buf; /* [3][E]; 48k uses bins [:-1] (hr.py:263) */
        for (int oc = 0; oc < 64; ++oc)
            for (int f = 0; f < Ec; ++f) {
                float acc = 0.f;
                for (int kt = 0; kt < 3; ++kt)
                    for (int kf = 0; kf < 3; ++kf) {
                        int fi = f + kf - 1;
                        if (fi >= 0 && fi < Ec) acc += o->erb_conv0_w[oc * 9 + kt * 3 + kf] * buf[kt * E + fi];
                    }
                float v = bn_apply(&o->erb_conv0_bn, oc, acc);
                e0[oc * Ec + f] = relu_f(v);
            }
    }
    sepconv(&o->erb_conv1, e0, e1, Ec, d->s1);
    sepconv(&o->erb_conv2, e1, e2, d->F1, d->s2);
    sepconv(&o->erb_conv3, e2, e3, d->F2, d->s3);
    memcpy(e3d, e3, sizeof(float) * 64 * d->F3);
    for (int bi = 0; bi < d->nb; ++bi)
        dprnn_block(&o->dprnn_erb[bi], e3d, d->F3, st + L->dprnn_erb + bi * d->F3 * 64,
                    so + L->dprnn_erb + bi * d->F3 * 64);
    /* df_conv0_buffer: FIFO of 3 frames of [2][D] */
    fifo_push(st + L->df_conv0_buf, so + L->df_conv0_buf, feat_spec, 3, 2 * D);
    {
        const float* buf = so + L->df_conv0_buf; /* [3][2][D] */
        /* GroupedConv2D groups=2 (layers.py:1083-1114): ch0 -> out 0..31, ch1 -> out 32..63 */
        for (int oc = 0; oc < 64; ++oc) {
            const int g = oc / 32;
            const float* w = (g == 0 ? o->df_conv0_w0 : o->df_conv0_w1) + (oc % 32) * 9;
            for (int f = 0; f < D; ++f) {
                float acc = 0.f;
                for (int kt = 0; kt < 3; ++kt)
                    for (int kf = 0; kf < 3; ++kf) {
                        int fi = f + kf - 1;
                        if (fi >= 0 && fi < D) acc += w[kt * 3 + kf] * buf[(kt * 2 + g) * D + fi];
                    }
                tmp[oc * D + f] = acc;
            }
        }
        pw_bn_relu(o->df_conv0_pw_t, &o->df_conv0_bn, tmp, c0, D);
    }
    sepconv(&o->df_conv1, c0, c1, D, 2);
    memcpy(o->p_c1, c1, sizeof(float) * 64 * d->Fd);
    for (int bi = 0; bi < d->nb; ++bi)
        dprnn_block(&o->dprnn_df[bi], c1, d->Fd, st + L->dprnn_df + bi * d->Fd * 64,
                    so + L->dprnn_df + bi * d->Fd * 64);
    memcpy(o->p_e0, e0, sizeof(float) * 64 * Ec); memcpy(o->p_e1, e1, sizeof(float) * 64 * d->F1);
    memcpy(o->p_e2, e2, sizeof(float) * 64 * d->F2); memcpy(o->p_e3, e3, sizeof(float) * 64 * d->F3);
    memcpy(o->p_e3d, e3d, sizeof(float) * 64 * d->F3);
    memcpy(o->p_c0, c0, sizeof(float) * 64 * D); memcpy(o->p_c1d, c1, sizeof(float) * 64 * d->Fd);

    /* flatten (f, c) (dpdfnet.py:233,235), df_fc_emb, concat */
    static __thread float flat[64 * 48], eflat[64 * 40];
    float embin[1024], emb[512], hid[256], hid2[256];
    for (int f = 0; f < d->Fd; ++f)
        for (int c = 0; c < 64; ++c) flat[f * 64 + c] = c1[c * d->Fd + f];
    grouped_linear(&o->df_fc_emb, flat, embin + 512, ACT_RELU);
    for (int f = 0; f < d->F3; ++f)
        for (int c = 0; c < 64; ++c) eflat[f * 64 + c] = e3d[c * d->F3 + f];
    if (d->is48) grouped_linear(&o->enc_erb_fc_emb, eflat, embin, ACT_RELU); /* hr.py:288 */
    else memcpy(embin, eflat, sizeof(float) * 512);
    /* emb_gru: SqueezedGRU_S (layers.py:1168-1188) */
    grouped_linear(&o->enc_lin_in, embin, hid, ACT_RELU);
    gru_cell(&o->enc_gru, hid, st + L->emb_gru, hid2);
    memcpy(so + L->emb_gru, hid2, sizeof(float) * 256);
    grouped_linear(&o->enc_lin_out, hid2, emb, ACT_RELU);
    memcpy(o->p_emb, emb, sizeof(float) * 512);

    /* ---- ErbDecoder.forward (onnx_model/dpdfnet.py:343-368; 48k hr.py:405-432) ---- */
    float m[481];
    {
        float dh[256], dh1[256], dh2[256], demb[64 * 40];
        static __thread float embc[64 * 40], t3[64 * 80], t2[64 * 160], t1[64 * 480], u[64 * 480];
        grouped_linear(&o->erbdec_lin_in, emb, dh, ACT_RELU);
        gru_cell(&o->erbdec_gru0, dh, st + L->erb_dec_gru, dh1);
        memcpy(so + L->erb_dec_gru, dh1, sizeof(float) * 256);
        gru_cell(&o->erbdec_gru1, dh1, st + L->erb_dec_gru + 256, dh2);
        memcpy(so + L->erb_dec_gru + 256, dh2, sizeof(float) * 256);
        grouped_linear(&o->erbdec_lin_out, dh2, demb, ACT_RELU);            /* [512] */
        if (d->is48) {
            float demb2[64 * 40];
            grouped_linear(&o->erbdec_erb_fc_emb, demb, demb2, ACT_RELU);   /* [C*F3] */
            memcpy(demb, demb2, sizeof(float) * 64 * d->F3);
        }
        /* view [f8][c] -> [c][f] (dpdfnet.py:360) */
        for (int f = 0; f < d->F3; ++f)
            for (int c = 0; c < 64; ++c) embc[c * d->F3 + f] = demb[f * 64 + c];
        pathconv_add(&o->conv3p, e3, embc, u, d->F3);
        if (d->s3 > 1) subpixel_conv(&o->convt3, u, t3, d->F3); else sepconv(&o->convt3, u, t3, d->F3, 1);
        pathconv_add(&o->conv2p, e2, t3, u, d->F2);
        subpixel_conv(&o->convt2, u, t2, d->F2);
        pathconv_add(&o->conv1p, e1, t2, u, d->F1);
        subpixel_conv(&o->convt1, u, t1, d->F1);
        pathconv_add(&o->conv0p, e0, t1, u, Ec);
        /* conv0_out: dense 64->1 k(1,3) pad 1 + BN(1) + Sigmoid */
        for (int f = 0; f < Ec; ++f) {
            float acc = 0.f;
            for (int c = 0; c < 64; ++c)
                for (int k = 0; k < 3; ++k) {
                    int fi = f + k - 1;
                    if (fi >= 0 && fi < Ec) acc += o->conv0_out_w[c * 3 + k] * u[c * Ec + fi];
                }
            m[f] = sigmoidf_(bn_apply(&o->conv0_out_bn, 0, acc));
        }
        if (d->is48) m[480] = m[478]; /* F.pad reflect (0,1) (hr.py:428) */
    }

    /* ---- DfDecoder.forward (onnx_model/dpdfnet.py:486-519) ---- */
    float coefs[96 * 10];
    {
        float dh[256], dh1[256], dh2[256], skip[256], outv[960];
        grouped_linear(&o->df_lin_in, emb, dh, ACT_RELU);
        gru_cell(&o->df_gru0, dh, st + L->df_dec_gru, dh1);
        memcpy(so + L->df_dec_gru, dh1, sizeof(float) * 256);
        gru_cell(&o->df_gru1, dh1, st + L->df_dec_gru + 256, dh2);
        memcpy(so + L->df_dec_gru + 256, dh2, sizeof(float) * 256);
        grouped_linear(&o->df_skip, emb, skip, ACT_NONE);
        for (int i = 0; i < 256; ++i) dh2[i] += skip[i];
        /* df_convp_buffer: FIFO 5 frames of c0 [C][D]; df_convp = grouped(2) 32->5 k(5,1) + pw 10->10 + BN + ReLU */
        fifo_push(st + L->df_convp_buf, so + L->df_convp_buf, c0, 5, 64 * D);
        const float* buf = so + L->df_convp_buf; /* [5][64][D] */
        float g10[10 * 96];
        for (int oc = 0; oc < 10; ++oc) {
            const int g = oc / 5;
            const float* w = (g == 0 ? o->df_convp_w0 : o->df_convp_w1) + (oc % 5) * 32 * 5;
            for (int f = 0; f < D; ++f) {
                float acc = 0.f;
                for (int ci = 0; ci < 32; ++ci)
                    for (int kt = 0; kt < 5; ++kt) acc += w[ci * 5 + kt] * buf[((size_t)kt * 64 + g * 32 + ci) * D + f];
                g10[oc * D + f] = acc;
            }
        }
        grouped_linear(&o->df_out, dh2, outv, ACT_TANH);
        for (int f = 0; f < D; ++f)
            for (int oc = 0; oc < 10; ++oc) {
                float acc = 0.f;
                for (int ci = 0; ci < 10; ++ci) acc += o->df_convp_pw[oc * 10 + ci] * g10[ci * D + f];
                float v = bn_apply(&o->df_convp_bn, oc, acc);
                v = relu_f(v);
                coefs[f * 10 + oc] = outv[f * 10 + oc] + v; /* view(b,t,F,O*2) + c0 (dpdfnet.py:515) */
            }
    }
    memcpy(o->p_m, m, sizeof(float) * (d->is48 ? F : E));
    /* probe coefs in the reference's [O][F][2] layout (DfOutputReshapeMF dpdfnet.py:382-389) */
    for (int n = 0; n < 5; ++n)
        for (int f = 0; f < D; ++f)
            for (int p = 0; p < 2; ++p) o->p_coefs[(n * D + f) * 2 + p] = coefs[f * 10 + 2 * n + p];

    /* ---- Mask.forward (layers.py:414-445) / MagnitudeMask (hr.py:55-69): delay 2 ---- */
    static __thread float spec_m[481 * 2];
    fifo_push(st + L->mask_buf, so + L->mask_buf, spec, 3, 2 * F);
    {
        const float* old = so + L->mask_buf; /* next[0] = frame from two steps ago */
        for (int f = 0; f < F; ++f) {
            float g = d->is48 ? m[f] : m[o->band_of[f]];
            spec_m[2 * f] = old[2 * f] * g;
            spec_m[2 * f + 1] = old[2 * f + 1] * g;
        }
    }
    /* ---- DF.forward + df_real (onnx_model/multiframe.py:200-232, 140-154) ---- */
    fifo_push(st + L->df_coefs_buf, so + L->df_coefs_buf, o->p_coefs, 3, 5 * D * 2);
    fifo_push(st + L->df_spec_buf, so + L->df_spec_buf, spec_m, 5, 2 * F);
    {
        const float* cd = so + L->df_coefs_buf;  /* next[0]: coefs of two steps ago [5][D][2] */
        const float* sb = so + L->df_spec_buf;   /* [5][F][2] */
        for (int f = 0; f < D; ++f) {
            float rr = 0.f, ii = 0.f, ri = 0.f, ir = 0.f;
            for (int n = 0; n < 5; ++n) {
                float sr = sb[(n * F + f) * 2], si = sb[(n * F + f) * 2 + 1];
                float cr = cd[(n * D + f) * 2], ci = cd[(n * D + f) * 2 + 1];
                rr += sr * cr; ii += si * ci; ri += sr * ci; ir += si * cr;
            }
            spec_e[2 * f] = rr - ii;
            spec_e[2 * f + 1] = ri + ir;
        }
        for (int f = D; f < F; ++f) {
            spec_e[2 * f] = sb[(2 * F + f) * 2];
            spec_e[2 * f + 1] = sb[(2 * F + f) * 2 + 1];
        }
    }
    /* export wrapper: * inv_wnorm (export_dpdfnet_to_onnx.py:24) */
    const float inv_wnorm = (float)(1.0 / (double)d->wnorm);
    for (int i = 0; i < 2 * F; ++i) spec_e[i] *= inv_wnorm;
    free(st);
}

int dpdf_oracle_probe(const dpdf_oracle* o, const char* name, float* out, int cap) {
    const dpdf_dims* d = &o->d;
    const float* src = NULL; int n = 0;
    if (!strcmp(name, "feat_erb")) { src = o->p_feat_erb; n = d->E; }
    else if (!strcmp(name, "feat_spec")) { src = o->p_feat_spec; n = 2 * d->D; }
    else if (!strcmp(name, "e0")) { src = o->p_e0; n = 64 * d->Ec; }
    else if (!strcmp(name, "e1")) { src = o->p_e1; n = 64 * d->F1; }
    else if (!strcmp(name, "e2")) { src = o->p_e2; n = 64 * d->F2; }
    else if (!strcmp(name, "e3")) { src = o->p_e3; n = 64 * d->F3; }
    else if (!strcmp(name, "e3_dprnn")) { src = o->p_e3d; n = 64 * d->F3; }
    else if (!strcmp(name, "c0")) { src = o->p_c0; n = 64 * d->D; }
    else if (!strcmp(name, "c1")) { src = o->p_c1; n = 64 * d->Fd; }
    else if (!strcmp(name, "c1_dprnn")) { src = o->p_c1d; n = 64 * d->Fd; }
    else if (!strcmp(name, "emb")) { src = o->p_emb; n = d->emb; }
    else if (!strcmp(name, "m")) { src = o->p_m; n = d->is48 ? d->F : d->E; }
    else if (!strcmp(name, "coefs")) { src = o->p_coefs; n = d->O * d->D * 2; }
    if (!src) return 0;
    memcpy(out, src, sizeof(float) * (size_t)(n < cap ? n : cap));
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* host DSP of the offline path                                                                */
/* ------------------------------------------------------------------------------------------ */
/* T = 1 + len(padded)//hop with padded = N + win (api.py:88, librosa center=True) */
int dpdf_oracle_num_frames(const dpdf_oracle* o, int n) { return 1 + (n + o->d.win) / o->d.hop; }

/* preprocess_waveform (package/src/dpdfnet/audio.py:104-117): librosa.stft(center=True,
 * pad_mode="reflect", window=vorbis) on the tail-padded waveform; unnormalised rfft. */
void dpdf_oracle_stft(const dpdf_oracle* o, const float* wav, int n, float* spec) {
    const int win = o->d.win, hop = o->d.hop, F = o->d.F;
    const int np = n + win;            /* np.pad(waveform, (0, win)) */
    const int T = 1 + np / hop;
    const int half = win / 2;
    float* xp = malloc(sizeof(float) * (size_t)(np + win));
    for (int i = 0; i < np + win; ++i) {
        int j = i - half;              /* index into the tail-padded signal */
        if (j < 0) j = -j;             /* reflect (no edge repeat) */
        if (j >= np) j = 2 * (np - 1) - j;
        xp[i] = j < n ? wav[j] : 0.0f;
    }
    double* fr = malloc(sizeof(double) * win);
    for (int t = 0; t < T; ++t) {
        for (int k = 0; k < win; ++k) fr[k] = (double)(xp[t * hop + k] * o->window[k]);
        for (int f = 0; f < F; ++f) {
            double re = 0.0, im = 0.0;
            for (int k = 0; k < win; ++k) {
                int idx = (int)(((long)f * k) % win);
                re += fr[k] * o->dft_cos[idx];
                im -= fr[k] * o->dft_sin[idx];
            }
            spec[((size_t)t * F + f) * 2] = (float)re;
            spec[((size_t)t * F + f) * 2 + 1] = (float)im;
        }
    }
    free(fr); free(xp);
}

/* postprocess_spec (audio.py:120-136): librosa.istft(center=True) = irfft * window, overlap-add,
 * divide by the window sum-square where it is non-negligible, trim win/2 both ends; then drop
 * the first 2*win samples and append 2*win zeros; fit_length (audio.py:30-38) to n. */
void dpdf_oracle_istft(const dpdf_oracle* o, const float* spec, int T, float* out, int n) {
    const int win = o->d.win, hop = o->d.hop, F = o->d.F;
    const size_t len = (size_t)win + (size_t)hop * (T - 1);
    float* y = calloc(len, sizeof(float));
    float* wss = calloc(len, sizeof(float));
    double* fr = malloc(sizeof(double) * win);
    for (int t = 0; t < T; ++t) {
        const float* S = spec + (size_t)t * F * 2;
        for (int k = 0; k < win; ++k) {
            /* irfft: imaginary parts of DC and Nyquist bins are ignored */
            double acc = S[0] + ((k & 1) ? -1.0 : 1.0) * S[2 * (F - 1)];
            for (int f = 1; f < F - 1; ++f) {
                int idx = (int)(((long)f * k) % win);
                acc += 2.0 * (S[2 * f] * o->dft_cos[idx] - S[2 * f + 1] * o->dft_sin[idx]);
            }
            fr[k] = acc / win;
        }
        for (int k = 0; k < win; ++k) {
            y[(size_t)t * hop + k] += (float)fr[k] * o->window[k];
            wss[(size_t)t * hop + k] += o->window[k] * o->window[k];
        }
    }
    for (size_t i = 0; i < len; ++i)
        if (wss[i] > 1.17549435e-38f) y[i] /= wss[i];
    const long ylen = (long)hop * (T - 1);          /* after trimming win/2 each side */
    const float* yt = y + win / 2;
    for (int i = 0; i < n; ++i) {
        long src = (long)i + 2L * win;               /* waveform_e[2*win:] ++ zeros(2*win) */
        out[i] = (src < ylen) ? yt[src] : 0.0f;
    }
    free(fr); free(wss); free(y);
}

/* apply_attn_limit (audio.py:41-76) */
void dpdf_oracle_attn_limit(const float* noisy, float* enh, int T, int F, float db) {
    if (!(db >= 0.0f) || isinf(db)) return;
    const float alpha = (float)pow(10.0, -(double)db / 20.0);
    const float beta = (float)(1.0 - (double)alpha);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 2 * F; ++i) {
            float nz = t >= 4 ? noisy[((size_t)(t - 4) * F) * 2 + i] : 0.0f;
            enh[(size_t)t * F * 2 + i] = alpha * nz + beta * enh[(size_t)t * F * 2 + i];
        }
}

/* enhance() (package/src/dpdfnet/api.py:51-113) at the model sample rate, mono */
void dpdf_oracle_enhance(dpdf_oracle* o, const float* wav, int n, float attn_db, float* out) {
    const int F = o->d.F;
    const int T = dpdf_oracle_num_frames(o, n);
    float* spec = malloc(sizeof(float) * (size_t)T * F * 2);
    float* spec_e = malloc(sizeof(float) * (size_t)T * F * 2);
    float* state = malloc(sizeof(float) * o->d.state_size);
    dpdf_oracle_stft(o, wav, n, spec);
    dpdf_oracle_initial_state(o, state);
    for (int t = 0; t < T; ++t)
        dpdf_oracle_frame(o, spec + (size_t)t * F * 2, state, spec_e + (size_t)t * F * 2, state);
    dpdf_oracle_attn_limit(spec, spec_e, T, F, attn_db);
    dpdf_oracle_istft(o, spec_e, T, out, n);
    free(state); free(spec_e); free(spec);
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { char* buf; size_t cap, len; } txt_t;
static void text_cb(void* ud, const char* name, const int* shape, int ndim, size_t off, size_t cnt) {
    txt_t* t = (txt_t*)ud;
    char line[256];
    int k = snprintf(line, sizeof(line), "%s %zu %zu ", name, off, cnt);
    for (int i = 0; i < ndim; ++i) k += snprintf(line + k, sizeof(line) - k, i ? ",%d" : "%d", shape[i]);
    k += snprintf(line + k, sizeof(line) - k, "\n");
    if (t->buf && t->len + k < t->cap) memcpy(t->buf + t->len, line, (size_t)k);
    t->len += (size_t)k;
}
size_t dpdf_oracle_manifest_text(const dpdf_cfg* cfg, char* buf, size_t cap) {
    txt_t t = {buf, cap, 0};
    dpdf_manifest(cfg, text_cb, &t);
    if (buf && t.len < cap) buf[t.len] = 0;
    return t.len;
}


/* ---------------------------------------------------------------------------------------------
 * ensure_sample_rate for mismatched rates (package/src/dpdfnet/audio.py:20-27): see the header for why this
 * follows scipy.signal.resample_poly rather than soxr.  Steps of resample_poly restated:
 *   up, down reduced by gcd;  n_out = ceil(n_in*up/down);  half_len = 10*max(up,down)
 *   h = firwin(2*half_len+1, 1/max(up,down), window=('kaiser',5.0)) * up     (unit DC gain before *up)
 *   n_pre_pad = down - half_len % down;  n_pre_remove = (half_len + n_pre_pad) / down
 *   y = upfirdn([0]*n_pre_pad + h, x, up, down)[n_pre_remove : n_pre_remove + n_out]
 * and upfirdn is y[m] = sum_i x[i] * hp[m*down - i*up].
 * ------------------------------------------------------------------------------------------- */
static double oracle_i0(double x) {
    double sum = 1.0, term = 1.0, q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) { term *= q / ((double)k * k); sum += term; if (term < 1e-18 * sum) break; }
    return sum;
}
long dpdf_oracle_resample(const float* x, long n_in, int sr_in, int sr_out, float* out, long cap) {
    long a = sr_in, b = sr_out;
    while (b) { long t = a % b; a = b; b = t; }
    const long up = sr_out / a, down = sr_in / a;
    const long n_out = (n_in * up) / down + ((n_in * up) % down ? 1 : 0);
    if (n_in <= 0) return 0;
    const long max_rate = up > down ? up : down, half_len = 10 * max_rate, M = 2 * half_len + 1;
    const double fc = 1.0 / (double)max_rate, beta = 5.0, pi = 3.14159265358979323846;
    const long n_pre_pad = down - half_len % down, n_pre_remove = (half_len + n_pre_pad) / down;
    const long lh = n_pre_pad + M;
    double* hp = (double*)calloc((size_t)lh, sizeof(double));
    double sum = 0.0;
    for (long n = 0; n < M; ++n) {
        const double m = (double)n - (double)half_len, xa = fc * m, r = m / (double)half_len;
        const double sinc = xa == 0.0 ? 1.0 : sin(pi * xa) / (pi * xa);
        double rad = 1.0 - r * r; if (rad < 0.0) rad = 0.0;
        hp[n_pre_pad + n] = fc * sinc * oracle_i0(beta * sqrt(rad)) / oracle_i0(beta);
        sum += hp[n_pre_pad + n];
    }
    for (long n = 0; n < M; ++n) hp[n_pre_pad + n] = hp[n_pre_pad + n] / sum * (double)up;
    for (long n = 0; n < n_out && n < cap; ++n) {
        const long t = (n + n_pre_remove) * down;
        long i_hi = t / up; if (i_hi > n_in - 1) i_hi = n_in - 1;
        long i_lo = t - lh + 1 <= 0 ? 0 : (t - lh + 1 + up - 1) / up;
        double acc = 0.0;
        for (long i = i_lo; i <= i_hi; ++i) acc += (double)x[i] * hp[t - i * up];
        out[n] = (float)acc;
    }
    free(hp);
    return n_out;
}
