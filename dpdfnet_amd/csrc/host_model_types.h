// host_model_types.h -- part of dpdf_model.hip (included there, in this order; one translation unit): the engine's host-side types: weight blob / constant arena, prepared-weight records, device buffers, workspace, lane (streams, events), host pipe, dpdf_model, dpdf_streams, recovery helpers.

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
                                       // 0 = gemm_rows producers (the general path); the first tile form (dec_seg_kernel, 1) is gone
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
    size_t window, istft_frag;
    int istft_groups, istft_K;
    DevBuf io_spec, io_spec_e, io_state, io_wav, io_out, frames, raw_spec, enh_spec, batch_state, stft_part;
    size_t dft_iA = 0, dft_iB = 0;     // operands of the two-stage synthesis DFT (dft2stage.h)
    DevBuf dft_mid_i;                  // its intermediates [frames][30][64]
    long dbg_nspec = 0, dbg_nframes = 0;
    int dft2 = 1;                      // big launches at 48 kHz: the synthesis DFT as two small matrix stages (0: one [2F x win] GEMM; A/B)
    size_t dft64_tw1 = 0, dft64_twm = 0, dft64_tw2 = 0;   // operands of the float64 analysis DFT (dft64.h; doubles stored in the float arena)
    int gru64_limbs = 3;               // DEFAULT since the end of round 6 (the corruption seen beside these kernels in round 5 was packed FP32 arithmetic; the library has none now): bit 0 intra-band pair, bit 1 inter-band scan -- the GRU-64 throughput kernels
                                       // on bf16 limbs (gru_limb.h: every fp32 product from three bf16 limbs per operand, six bf16 MFMAs per term, fp32 accumulation -- fp32-exact products at 2.67 x the fp32 matrix rate, closer to a float64 recurrence than the fp32 MFMA);
                                       // 0 = the fp32-MFMA kernels of gru_scan.h (dpdf_set_option, or DPDF_GRU64_LIMBS=0 for every handle of the process)
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


