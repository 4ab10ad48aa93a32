// host_batch_calls.h -- part of dpdf_model.hip (included there, in this order; one translation unit): C ABI: dpdf_run_frames and the batch calls (dpdf_enhance_batch*): STFT -> chunks -> iSTFT, the pipelined host-pointer path.

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
    // big launches: the synthesis DFT as two small matrix stages (dft2stage.h) -- at 960 only: at 320 its two launches measure 0.96 ms
    // against 0.88 ms for the one GEMM + overlap-add
    const bool dft2_inv = d.win == 960 && m->dft2 && (long)B * T > SMALL_M_ROWS;
    if (dft2_inv) {
        const size_t rows = piped ? (size_t)B * *std::max_element(sizes.begin(), sizes.end()) : (size_t)B * T;
        const size_t per_frame = (size_t)(d.win / 32) * 64;
        if ((rc = m->dft_mid_i.ensure(rows * per_frame))) return rc;
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
            {       // (float64 on every call path: section 3c of DESIGN.md -- the fp32 analysis forms of rounds 1-4 are gone)
                Dft64Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, seg, 0, nullptr, m->raw_spec.p, B * Tc,
                             (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
                launch_dft64_forward(m->stream, da, d.win);
            }
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
            launch_gemm_rows_wn<5, 48>(hp.s_down, ap, m->C(m->istft_frag), ep, B * Tc, m->istft_K, m->istft_groups / 4);      // win / 80 = 4 or 12 column groups: always a multiple of 4
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
        Dft64Args da{d_wav, N, T, d.hop, m->C(m->window), d_lens, RowSeg{T, T, 0}, 0, nullptr, m->raw_spec.p, B * T,
                     (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
        launch_dft64_forward(m->stream, da, d.win);
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

