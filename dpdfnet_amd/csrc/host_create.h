// host_create.h -- part of dpdf_model.hip (included there, in this order; one translation unit): C ABI: create / destroy (weight preparation), geometry queries, options, profiling.

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
    // DPDF_GRU64_LIMBS=0 (or 1, 2, 3): every engine handle of the process starts with that GRU-64 kernel family -- 3, the bf16-limb kernels, is the
    // default; 0 the fp32-MFMA kernels (the whole GPU suite is run under both: profiles/r6_gpu_suite_*.txt); dpdf_set_option("gru64_limbs", ...) overrides per handle
    if (const char* e = getenv("DPDF_GRU64_LIMBS")) { const int v = atoi(e); if (v >= 0 && v <= 3) m->gru64_limbs = v; }
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
        std::vector<float> fa;             // (synthesis only: the analysis is float64, dft64.h)
        for (int k1 = 0; k1 < 32; ++k1) {
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
    DevBuf* bufs[] = {&m->io_spec, &m->io_spec_e, &m->io_state, &m->io_wav, &m->io_out, &m->frames, &m->raw_spec, &m->enh_spec, &m->batch_state, &m->stft_part, &m->dft_mid_i};
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
    else if (n == "dec_seg") m->dec_seg = value <= 0 ? 0 : (value >= 3 ? 3 : 2);     // 0 gemm_rows producers (any geometry), 2 dec_seg2 per stage, 3 in one launch
    else if (n == "dec_seg_grid") m->dec_seg_grid = value > 0 ? value : 256;
    else if (n == "dec_seg_all_frames") m->dec_seg_all_frames = value;
    else if (n == "df_ring") m->df_ring = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "gru256_stack") m->gru256_stack = value != 0;
    else if (n == "tail_frames") m->tail_frames = value < 0 ? 0 : std::min(value, 48);
    else if (n == "gru64_limbs") m->gru64_limbs = value;
    else if (n == "hop_fused") m->hop_fused = value != 0;
    else if (n == "hop_stack") m->hop_stack = value != 0;
    else if (n == "hop_spin_join") m->hop_spin_join = value != 0;
    else if (n == "fuse_small") m->fuse_small = value != 0;
    else if (n == "fuse_enc") m->fuse_enc = value != 0;
    else if (n == "hop_pconv") m->hop_pconv = value != 0;
    else if (n == "dual_step") m->dual_step = value != 0;
    else if (n == "hop_prologue") m->hop_prologue = value != 0;
    else if (n == "late_export") m->late_export = value != 0;
    else if (n == "snapshot") m->snapshot = value != 0;
    else if (n == "host_pipe") m->host_pipe = value != 0;
    else if (n == "dft2") m->dft2 = value != 0;
    else if (n == "chunk_io") m->chunk_io = value != 0;
    else if (n == "gru256_fused_x") m->gru256_fused_x = value != 0;
    else if (n == "gru256_fused_x_tiles") m->gru256_fused_x_tiles = value < 1 ? 1 : value;
    else if (n == "host_copy_threads") m->host_copy_threads = value < 1 ? 1 : value;
    else if (n == "single_chunk_inline") m->single_chunk_inline = value != 0;
    else if (n == "fuse_dec") m->fuse_dec = value != 0;
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


