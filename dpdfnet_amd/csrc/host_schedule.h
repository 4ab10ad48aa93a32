// host_schedule.h -- part of dpdf_model.hip (included there, in this order; one translation unit): the frame function as launches: weight packing helpers, layer launchers, the DPRNN walk, stage 1 / stage 2 of a chunk, the chunk loop (run_chunks).

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
        if (!(m->hop_stack && hop_glue && m->hop_fused && m->use_gru256_cluster)) return false;
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
                if (scan4 && hop_glue && m->hop_fused && m->use_gru256_cluster && hop_block(bi, ai)) return;
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
                // eight waves per tile: half the dependent MFMAs and operand loads per wave (fcln_gi.h; the four-wave form of round 3 is gone)
                if (next) hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue8_kernel<true>), dim3((M + 15) / 16), dim3(512), 0, m->cur, ha);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(dprnn_hop_glue8_kernel<false>), dim3((M + 15) / 16), dim3(512), 0, m->cur, ha);
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
        const bool one = m->dec_seg >= 3 && BT >= m->dec_seg_all_frames;   // (few frames per workgroup: three launches fill and drain faster)
        auto grid = [&](long) { return dim3((unsigned)std::min<long>(BT, m->dec_seg_grid)); };
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
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), grid((long)BT * (d.F2 / 80)), dim3(512), 0, st, a3);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<2, 80, false>), grid((long)BT * (d.F1 / 80)), dim3(512), 0, st, a2);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dec_seg2_kernel<3, 96, true>), grid((long)BT * (d.Ec / 96)), dim3(512), 0, st, a1);
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
    } else if (m->fuse_mask && d.s1 == 3) {
        // 48 kHz geometry (stride 3; the 16 kHz one is dec_last_kernel above): w.d1 holds the three tap sums per row ([rows][4]) instead of the 64-channel d1 rows
        run_subpix_mask<3>(m, m->convt1, m->conv1p, e1v, d2v, x.e0.p, w.d1.p, d.Ec, B, Tc);
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
