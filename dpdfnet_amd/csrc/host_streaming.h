// host_streaming.h -- part of dpdf_model.hip (included there, in this order; one translation unit): C ABI: device-resident stream sets (dpdf_streams_*): priming, hops, masked calls, snapshot recovery.

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
        Dft64Args da{in_place ? src : xbuf, (T + 1) * d.hop, T, d.hop, m->C(m->window), nullptr, RowSeg{T, T, 0}, 1, in_place ? v.in_tail : nullptr,
                     s->spec.p, S * T, (const double*)m->C(m->dft64_tw1), (const double*)m->C(m->dft64_twm), (const double*)m->C(m->dft64_tw2)};
        launch_dft64_forward(m->stream, da, d.win);
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

