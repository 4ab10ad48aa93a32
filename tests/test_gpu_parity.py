"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against (a) the CPU oracle on the
same seeded inputs, (b) the committed reference goldens, and (c) size-independent properties at
BASELINE.json's full sizes.  Tolerance from BASELINE.json north_star: waveform RMS error < 1e-4 vs
the reference; we assert 50x tighter (2e-6) on waveforms and relative 1e-4 on every stage."""
import numpy as np
import pytest

from tests.util import GOLDEN, MODEL_TAGS, golden_blob, load_golden, make_oracle, norm_inits, rms, synth_clip

pytestmark = pytest.mark.gpu

WAVE_TOL = 2e-6          # RMS, signals are O(0.05)
STAGE_REL_TOL = 1e-4


def K(meta) -> float:
    """Tolerance factor of a fixture: x 10 for the weight-robustness goldens (weights.stress_blob: three-fold GRU gain,
    five-fold LayerNorm gain, BatchNorm var ~ eps).  Different kernel forms of one recurrence / different fp32
    implementations differ by an ulp per operation, and those weights amplify it 10-20 x through the recurrences
    (tests/test_oracle_golden.py:_slack has the measured figures); x 10 keeps every check >= 30 x inside north_star's 1e-4."""
    return 10.0 if meta.get("stress") else 1.0


@pytest.fixture(scope="module")
def be():
    from dpdfnet_amd import backend
    assert backend.device_count() >= 1, "no GPU visible: the HIP engine has no CPU fallback"
    return backend


@pytest.fixture(scope="module", params=MODEL_TAGS)
def case(request, be):
    g, meta = load_golden(request.param)
    blob = golden_blob(meta)
    m = be.HipModel(meta["sample_rate"], meta["nb"], blob, 0)      # engine-default initial state, nothing injected
    yield g, meta, make_oracle(meta, blob), m
    m.close()


def test_initial_state_and_geometry(case):
    g, meta, o, m = case
    assert m.state_size == meta["state_size"] == o.state_size
    np.testing.assert_array_equal(m.initial_state(), g["init_state"])
    e, s = norm_inits(meta["sample_rate"])      # the reference's tables (constants.npz) with no extras passed in
    np.testing.assert_array_equal(m.initial_state()[:e.size], e)
    np.testing.assert_array_equal(m.initial_state()[e.size:e.size + s.size], s)
    np.testing.assert_array_equal(o.initial_state(), g["init_state"])
    assert (m.win_len, m.hop, m.freq_bins) == (o.win_len, o.hop, o.freq_bins)
    assert m.num_frames(meta["n"]) == meta["T"]


def test_run_frames_matches_oracle_and_golden(case):
    g, meta, o, m = case
    spec = o.stft(g["wav"])
    ref, st_ref = o.run_frames(spec)
    m.set_chunk_frames(0)
    out, st = m.run_frames(spec, m.initial_state())
    scale = float(np.abs(ref).max())
    assert np.abs(out - ref).max() < STAGE_REL_TOL * scale
    assert rms(out - ref) < 1e-5 * K(meta) * scale
    assert np.abs(st - st_ref).max() < 2e-4 * K(meta)
    assert np.abs(out[:64] - g["spec_e_head"]).max() < STAGE_REL_TOL * scale
    assert np.abs(st - g["state_out"]).max() < 2e-4 * K(meta)


def test_stage_tensors_match_reference_probes(case, be):
    g, meta, o, m = case
    spec = o.stft(g["wav"])
    T = spec.shape[0]
    m.set_chunk_frames(-1)                      # the whole sequence as ONE chunk: debug_fetch returns the last chunk's tensors
    m.run_frames(spec, m.initial_state())
    m.set_chunk_frames(0)
    d = be.query_dims(meta["sample_rate"], meta["nb"])
    shapes = {"e0": d.Ec, "e1": d.F1, "e2": d.F2, "e3": d.F3, "e3_dprnn": d.F3, "c1": d.Fd, "c1_dprnn": d.Fd}
    checked = 0
    for t in meta["probe_frames"]:
        for name, Fp in shapes.items():
            key = f"f{t}_{name}"
            if key not in g.files:
                continue
            mine = m.debug_fetch(name).reshape(T, Fp, 64)[t].T.reshape(-1)     # -> reference [C][F]
            ref = g[key]
            assert np.abs(mine - ref).max() < STAGE_REL_TOL * max(1.0, float(np.abs(ref).max())), (t, name)
            checked += 1
        c0 = m.debug_fetch("c0").reshape(T + 4, d.D, 64)[4 + t].T.reshape(-1)
        assert np.abs(c0 - g[f"f{t}_c0"]).max() < STAGE_REL_TOL * max(1.0, float(np.abs(g[f"f{t}_c0"]).max()))
        emb = m.debug_fetch("emb").reshape(T, 512)[t]
        assert np.abs(emb - g[f"f{t}_emb"]).max() < STAGE_REL_TOL * max(1.0, float(np.abs(g[f"f{t}_emb"]).max()))
        mk = m.debug_fetch("m").reshape(T, d.E)[t]
        ref_m = g[f"f{t}_m"]
        assert np.abs(mk[: ref_m.size] - ref_m).max() < 1e-5 * K(meta)
        ck = m.debug_fetch("coefs").reshape(T + 2, d.D * 10)[2 + t]
        assert np.abs(ck - g[f"f{t}_coefs_fk"]).max() < STAGE_REL_TOL * max(1.0, float(np.abs(ck).max()))
        fe = m.debug_fetch("feat_erb").reshape(T + 2, d.E)[2 + t]
        # log-domain feature of near-silent bins amplifies the (oracle-double vs torch-fp32) STFT rounding
        assert np.abs(fe - g[f"f{t}_feat_erb"]).max() < 5e-5 * K(meta)
        fs = m.debug_fetch("feat_spec").reshape(T + 2, 2, d.D)[2 + t]          # engine [re|im][D]; reference [D][re,im]
        ref_fs = g[f"f{t}_feat_spec_ri"].reshape(d.D, 2).T
        assert np.abs(fs - ref_fs).max() < STAGE_REL_TOL * max(1.0, float(np.abs(ref_fs).max())), (t, "feat_spec")
    assert checked >= 10


@pytest.mark.parametrize("chunk", [1, 3, 16, 50])
def test_time_chunking_and_state_carry_invariance(case, chunk):
    """Splitting T frames into chunks (state carried in the reference flat layout) changes nothing;
    chunk = 1 is literally one session.run per frame."""
    g, meta, o, m = case
    spec = o.stft(g["wav"])[:40]
    m.set_chunk_frames(0)
    ref, st_ref = m.run_frames(spec, m.initial_state())
    m.set_chunk_frames(chunk)
    out, st = m.run_frames(spec, m.initial_state())
    m.set_overlap(0)                            # the same chunks with every kernel on one stream
    out0, st0 = m.run_frames(spec, m.initial_state())
    m.set_overlap(27)
    m.set_chunk_frames(0)
    np.testing.assert_allclose(out0, out, atol=1e-5 * K(meta) * float(np.abs(ref).max()))
    np.testing.assert_allclose(st0, st, rtol=5e-6, atol=2e-5 * K(meta) * max(1.0, meta["nb"] / 4.0))
    np.testing.assert_allclose(out, ref, atol=1e-5 * K(meta) * float(np.abs(ref).max()))
    # different chunk lengths pick different kernel forms of the recurrences (fused / hoisted-input / plain): equal to
    # rounding, and the rounding differences of a DPRNN stack grow with its depth (8 blocks: 2.1e-5 seen on one state value)
    np.testing.assert_allclose(st, st_ref, rtol=5e-6, atol=2e-5 * K(meta) * max(1.0, meta["nb"] / 4.0))


def test_fused_and_unfused_dprnn_paths_agree(case):
    """fc + LayerNorm + residual inside the GRU-64 scans (large chunks) vs as separate GEMM kernels
    (small chunks); both forced here on the same small input."""
    g, meta, o, m = case
    spec = o.stft(g["wav"])[:30]
    ref, st_ref = o.run_frames(spec)
    outs = []
    # fused on bf16 limbs (gru_limb.h, the default), fused on the fp32-MFMA kernels (gru_scan.h), unfused -- every golden tag, incl. the
    # hot / stiff weights: each against the oracle, and against each other
    for fuse, limbs in ((True, 3), (True, 0), (False, 3)):
        m.set_fuse_dprnn(fuse)
        m.set_option("gru64_limbs", limbs)
        out, st = m.run_frames(spec, m.initial_state())
        assert np.abs(out - ref).max() < STAGE_REL_TOL * float(np.abs(ref).max()), (fuse, limbs)
        assert np.abs(st - st_ref).max() < 2e-4 * K(meta), (fuse, limbs)
        outs.append(out)
    m.set_fuse_dprnn("auto")
    m.set_option("gru64_limbs", 3)
    assert np.abs(outs[0] - outs[2]).max() < 2e-5 * K(meta) * float(np.abs(ref).max())
    assert np.abs(outs[0] - outs[1]).max() < 2e-5 * K(meta) * float(np.abs(ref).max())


def test_single_frame_chunks_are_race_free_under_stream_overlap(case):
    """Regression: with one frame per launch the 4-workgroup GRU-256 cluster has no inter-step
    hand-off, and a fast workgroup used to overwrite the carried state before a late peer had read
    it (seen only with stage-2 and ERB streams both active).  Hammer that path."""
    g, meta, o, m = case
    spec = o.stft(g["wav"])[:4]
    m.set_chunk_frames(0)
    ref, st_ref = m.run_frames(spec, m.initial_state())
    m.set_chunk_frames(1)
    try:
        for mask in (27, 19):                   # 19: without the decoder fork
            m.set_overlap(mask)
            for _ in range(40):
                out, st = m.run_frames(spec, m.initial_state())
                assert np.abs(out - ref).max() < 1e-5 * K(meta) * float(np.abs(ref).max())
                np.testing.assert_allclose(st, st_ref, rtol=5e-6, atol=2e-5 * K(meta))
    finally:
        m.set_chunk_frames(0)
        m.set_overlap(27)


def test_frame_by_frame_host_loop_is_a_drop_in_for_session_run(case):
    """The reference's hot loop verbatim (api.py:96-104): one call per frame, state through the host."""
    g, meta, o, m = case
    spec = o.stft(g["wav"])[:12]
    ref, st_ref = o.run_frames(spec)
    st = m.initial_state()
    frames = []
    for t in range(spec.shape[0]):
        y, st = m.run_frames(spec[t:t + 1], st)
        frames.append(y)
    out = np.concatenate(frames)
    assert np.abs(out - ref).max() < STAGE_REL_TOL * float(np.abs(ref).max())
    assert np.abs(st - st_ref).max() < 2e-4 * K(meta)


def test_enhance_batch_matches_reference_waveforms(case):
    g, meta, o, m = case
    wav = g["wav"]
    for chunk in (0, 37):
        m.set_chunk_frames(chunk)
        for key, db in (("enhanced", None), ("enhanced_attn0", 0.0), ("enhanced_attn12", 12.0)):
            out = m.enhance_batch(wav[None], db)[0]
            assert rms(out - g[key]) < WAVE_TOL * K(meta), (key, chunk, rms(out - g[key]))
            assert np.all(out[-m.win_len:] == 0.0)            # reference quirk: last 2 hops are zero
    m.set_chunk_frames(0)


def test_batch_rows_are_independent_and_ragged_sizes(case):
    g, meta, o, m = case
    sr = meta["sample_rate"]
    for n in (1, m.hop - 1, m.win_len + 7, 3 * m.hop):          # tiny / non-multiple-of-hop lengths
        wav = np.stack([synth_clip(n, sr, 50 + i) for i in range(3)])
        out = m.enhance_batch(wav)
        for i in range(3):
            assert rms(out[i] - o.enhance(wav[i])) < WAVE_TOL, (n, i)
    wav = np.stack([synth_clip(2000, sr, 60 + i) for i in range(5)] + [synth_clip(2000, sr, 60)])
    out = m.enhance_batch(wav, 6.0)
    np.testing.assert_array_equal(out[0], out[5])                # same clip, different batch slot
    for i in (1, 4):
        assert rms(out[i] - o.enhance(wav[i], 6.0)) < WAVE_TOL


def test_error_behaviour(case):
    g, meta, o, m = case
    with pytest.raises(ValueError, match="attn_limit_db"):
        m.enhance_batch(g["wav"][None], -2.0)
    with pytest.raises(ValueError):
        m.run_frames(np.zeros((2, 3, m.freq_bins + 1, 2), np.float32), m.initial_state())
    with pytest.raises(ValueError):
        m.run_frames(np.zeros((1, 3, m.freq_bins, 2), np.float32), np.zeros(5, np.float32))
    assert m.enhance_batch(np.zeros((2, 0), np.float32)).shape == (2, 0)


# ----- StreamEnhancer on the device-resident streaming path ------------------------------------
@pytest.mark.parametrize("tag", ["16k_nb2", "48k_nb1", "48k_nb8"])
def test_stream_enhancer_matches_reference_stream_goldens(tag, be, tmp_path, monkeypatch):
    """Reference StreamEnhancer driven by the real frame function (stream_*.npz 'real_*')."""
    from dpdfnet_amd import stream, weights
    from dpdfnet_amd.models import ModelInfo, ResolvedModel
    g, meta = load_golden(tag)
    sr = meta["sample_rate"]
    wfile = weights.save_blob(tmp_path / "w.npz", golden_blob(meta))
    info = ModelInfo(name=f"test_{tag}", sample_rate=sr, frame_ms=20.0, description="", onnx_filename="w.onnx",
                     dprnn_num_blocks=meta["nb"])
    monkeypatch.setattr(stream, "resolve_model", lambda **_k: ResolvedModel(info=info, onnx_path=wfile))
    G = np.load(GOLDEN / f"stream_{tag}.npz")
    wav = G["wav"]
    hop = 160 if sr == 16000 else 480
    for chunk in (7, hop, 171, 512, len(wav)):
        se = stream.StreamEnhancer(model="ignored")
        parts = [se.process(wav[i:i + chunk]) for i in range(0, len(wav), chunk)]
        parts.append(se.flush())
        got = np.concatenate(parts)
        ref = G[f"real_chunk{chunk}"]
        assert got.shape == ref.shape, chunk
        assert rms(got - ref) < WAVE_TOL, (chunk, rms(got - ref))
        se.reset()
        assert len(se.process(wav[: 2 * hop - 1])) == 0 and len(se.process(wav[:1])) == hop


def test_package_enhance_with_synthetic_weights_matches_oracle(be):
    import dpdfnet_amd
    from oracle import oracle as orc
    from dpdfnet_amd.weights import synth_blob
    wav = synth_clip(12000, 16000, 5)
    out = dpdfnet_amd.enhance(wav, 16000, model="dpdfnet2", onnx_path="synthetic:77", attn_limit_db=9.0)
    ref = orc.Oracle(16000, 2, synth_blob(be.manifest(16000, 2), 77)).enhance(wav, 9.0)
    assert out.shape == wav.shape and rms(out - ref) < WAVE_TOL
    # PESQ/STOI libraries are absent: the stand-in for "PESQ delta <= 0.001" is the SI-SNR of our output
    # against the oracle's (reference pesq_stoi_sisnr_calc.py:16-27); its eps = 1e-8 caps the figure near 93 dB here
    from dpdfnet_amd.evalkit import si_snr
    assert si_snr(ref, out) > 85.0
    outs = dpdfnet_amd.enhance_batch([wav, wav[:5000]], 16000, model="dpdfnet2", onnx_path="synthetic:77")
    assert rms(outs[1] - orc.Oracle(16000, 2, synth_blob(be.manifest(16000, 2), 77)).enhance(wav[:5000])) < WAVE_TOL


@pytest.mark.parametrize("kind", ["silence", "full_scale_square", "dc", "impulse", "tiny"])
def test_extreme_inputs_match_the_oracle(be, kind):
    """Inputs at the edges of the numeric range: digital silence (log of the 1e-10 floor, zero gradients in the norms),
    a clipped square wave, DC, a single impulse, 1e-7-amplitude noise.  Fast exp/rcp, the folded exponent scales and
    the saturating tanh must stay finite and on the oracle."""
    from oracle import oracle as orc
    from dpdfnet_amd.weights import synth_blob
    sr, nb, n = 16000, 2, 16000
    rng = np.random.default_rng(3)
    t = np.arange(n)
    wav = {
        "silence": np.zeros(n, np.float32),
        "full_scale_square": np.where((t // 40) % 2 == 0, 1.0, -1.0).astype(np.float32),
        "dc": np.full(n, 0.5, np.float32),
        "impulse": np.eye(1, n, 3000, dtype=np.float32)[0],
        "tiny": (1e-7 * rng.standard_normal(n)).astype(np.float32),
    }[kind]
    blob = synth_blob(be.manifest(sr, nb), 4711)
    m = be.HipModel(sr, nb, blob, 0)
    out = m.enhance_batch(np.stack([wav, wav[::-1].copy()]), None)
    m.close()
    assert np.isfinite(out).all()
    ref = orc.Oracle(sr, nb, blob).enhance(wav)
    scale = max(float(np.abs(ref).max()), 1e-6)
    assert float(np.abs(out[0] - ref).max()) < 2e-5 * scale + 1e-9, (kind, float(np.abs(out[0] - ref).max()), scale)


def test_models_in_concurrent_host_threads(be):
    """The reference runs one session per host thread (cli.py:249-259).  Here: four threads, each with its own engine
    handle (own streams, own workspace), plus all four sharing ONE handle (calls serialise on its mutex) -- every result
    bit-identical to the single-threaded run."""
    import threading
    from dpdfnet_amd.weights import synth_blob
    sr, nb = 16000, 2
    blob = synth_blob(be.manifest(sr, nb), 99)
    clips = [np.stack([synth_clip(8000 + 640 * i, sr, 50 + 10 * i + j) for j in range(3)]) for i in range(4)]
    base = be.HipModel(sr, nb, blob, 0)
    want = [base.enhance_batch(c, None) for c in clips]
    got_own, got_shared, errs = [None] * 4, [None] * 4, []

    def own(i):
        try:
            m = be.HipModel(sr, nb, blob, 0)
            for _ in range(3):
                got_own[i] = m.enhance_batch(clips[i], None)
            m.close()
        except Exception as exc:   # surfaced below
            errs.append(exc)

    def shared(i):
        try:
            for _ in range(3):
                got_shared[i] = base.enhance_batch(clips[i], None)
        except Exception as exc:
            errs.append(exc)

    for fn in (own, shared):
        ths = [threading.Thread(target=fn, args=(i,)) for i in range(4)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    base.close()
    assert not errs, errs
    for i in range(4):
        np.testing.assert_array_equal(got_own[i], want[i])
        np.testing.assert_array_equal(got_shared[i], want[i])


def test_single_frame_calls_with_changing_stream_counts(be):
    """Regression (gru256_step_kernel): the per-tile arrival counters that guard the in-place state update must stay
    consistent when consecutive one-frame calls cover different numbers of 16-row tiles (64 -> 16 -> 64 streams used to
    leave tiles 1-3 one round behind: spin timeout)."""
    g, meta = load_golden("16k_nb1")
    blob = golden_blob(meta)
    m = be.HipModel(meta["sample_rate"], meta["nb"], blob, 0)
    o = make_oracle(meta, blob)
    frames = o.stft(synth_clip(3000, 16000, 77))[:3]
    for B in (64, 16, 64, 5, 40, 64):
        spec = np.repeat(frames[None, :1], B, axis=0) * (1.0 + 0.01 * np.arange(B, dtype=np.float32))[:, None, None, None]
        st = np.tile(m.initial_state(), (B, 1))
        out, st1 = m.run_frames(spec, st)
        out2, st2 = m.run_frames(np.repeat(frames[None, 1:2], B, axis=0), st1)
        ref, s_ref = o.run_frames(np.concatenate([spec[B - 1], frames[1:2]], axis=0))
        scale = float(np.abs(ref).max()) + 1e-9
        assert np.abs(np.concatenate([out[B - 1], out2[B - 1]], axis=0) - ref).max() < STAGE_REL_TOL * scale, B
        assert np.abs(st2[B - 1] - s_ref).max() < 2e-4, B
    m.close()


def test_throughput_regime_chunk_schedule_matches_whole_sequence_and_oracle(be):
    """>= 96 streams: 192-frame chunks, and a last chunk of >= 96 frames gives up a 32-frame tail chunk (pipeline drain).
    300 frames = 192 + 76 + 32 here; the result must equal the whole sequence run as one chunk, and the oracle."""
    g, meta = load_golden("16k_nb1")
    blob = golden_blob(meta)
    m = be.HipModel(meta["sample_rate"], meta["nb"], blob, 0)
    o = make_oracle(meta, blob)
    B, T = 96, 300
    spec = np.stack([o.stft(synth_clip(T * 160, 16000, 4000 + (i % 7)))[:T] * (0.5 + (i % 3) / 2.0) for i in range(B)]).astype(np.float32)
    st0 = np.tile(m.initial_state(), (B, 1))
    out, st = m.run_frames(spec, st0)                    # automatic schedule
    m.set_option("tail_frames", 0)
    out_nt, st_nt = m.run_frames(spec, st0)              # 192 + 108
    m.set_option("tail_frames", 32)
    m.set_chunk_frames(-1)
    ref, st_ref = m.run_frames(spec, st0)                # one chunk
    m.set_chunk_frames(0)
    scale = float(np.abs(ref).max())
    for a, sa in ((out, st), (out_nt, st_nt)):
        assert np.abs(a - ref).max() < 2e-5 * scale
        np.testing.assert_allclose(sa, st_ref, rtol=5e-6, atol=2e-5)
    np.testing.assert_array_equal(out[0], out[21])       # identical clips (seed i % 7, scale i % 3) in different slots and tiles
    r0, s0 = o.run_frames(spec[5])
    assert np.abs(out[5] - r0).max() < STAGE_REL_TOL * scale and np.abs(st[5] - s0).max() < 2e-4
    m.close()


def test_48khz_decoder_stage_forms_agree_and_match_the_oracle(be):
    """The 48 kHz decoder stages (40 -> 80 -> 160 -> 480 bands + the mask head's tap sums) exist three times: gemm_rows launches with
    sub-pixel producers (dec_seg = 0), band-segment tiles with inputs read once (1: dec_last.h dec_seg_kernel) and the tile pipeline of
    dec_seg2.h (2, the default from 1024 frames per chunk on; 3: its three stages in one launch).  97 clips x 11-frame chunks (1067 frames: 1067 / 2134 / 5335 tiles over 256
    workgroups -- uneven shares, a ragged last chunk of 4 frames falls back to the small forms), stage tensors and waveforms against
    each other, three clips against the oracle."""
    from oracle import oracle as orc
    from dpdfnet_amd.weights import synth_blob
    sr, nb = 48000, 1
    blob = synth_blob(be.manifest(sr, nb), 4242)
    m = be.HipModel(sr, nb, blob, 0)
    B, n = 97, int(0.37 * sr) + 5
    wav = np.stack([synth_clip(n, sr, 5100 + i) * (0.4 + (i % 5) / 5.0) for i in range(B)]).astype(np.float32)
    m.set_chunk_frames(11)
    outs, masks = {}, {}
    m.set_option("dec_seg_all_frames", 0)
    for form in (2, 0, 2, 3):
        m.set_option("dec_seg", form)
        out = m.enhance_batch(wav, 6.0)
        if form in outs:
            np.testing.assert_array_equal(out, outs[form])          # run to run
        outs[form], masks[form] = out, m.debug_fetch("m")
    np.testing.assert_array_equal(outs[3], outs[2])                  # the one-launch form runs the same code on the same tiles
    for form in (0,):
        assert rms(outs[2] - outs[form]) < 1e-6, form
        assert np.abs(masks[2] - masks[form]).max() < 2e-5, form
    o = orc.Oracle(sr, nb, blob)
    for b in (0, 50, 96):
        assert rms(outs[2][b] - o.enhance(wav[b], 6.0)) < WAVE_TOL, b
    m.set_option("dec_seg_grid", 7)                                   # many tiles per workgroup, odd shares
    np.testing.assert_array_equal(m.enhance_batch(wav, 6.0), outs[2])
    m.close()


@pytest.mark.parametrize("sr,nb", [(16000, 2), (48000, 1)])
def test_big_batch_kernel_forms_match_small_batch_forms_and_oracle(be, sr, nb):
    """Launch shapes that only big batches select -- the time-walking DF pass over c0 (df_ring_kernel: df_conv1 + pathway
    conv, >= 64 clips), the mask head folded into the last decoder GEMM, the round-robin GRU-256 ring scan (opt-in) --
    against the small-batch forms of the same engine (stage tensors) and the oracle (waveforms), on 192 distinct clips
    (12 GRU-256 tiles: divisible by 2, 3 and 4)."""
    from oracle import oracle as orc
    from dpdfnet_amd.weights import synth_blob
    blob = synth_blob(be.manifest(sr, nb), 777)
    m = be.HipModel(sr, nb, blob, 0)
    B, n = 192, int(0.25 * sr) + 17
    wav = np.stack([synth_clip(n, sr, 2000 + i) * (0.5 + (i % 7) / 7.0) for i in range(B)]).astype(np.float32)
    m.set_chunk_frames(16)                      # several chunks: the ring's halo frames come through the state FIFO
    ref_probe = {}
    outs = {}
    for tag, opts in (("small_forms", {"df_ring": 0, "fuse_mask": 0}), ("ring_reads_c0", {"df_ring": 1, "fuse_mask": 1}),
                      ("big_forms", {"df_ring": 2, "fuse_mask": 1})):
        for k, v in opts.items():
            m.set_option(k, v)
        outs[tag] = m.enhance_batch(wav, 6.0)
        ref_probe[tag] = {k: m.debug_fetch(k) for k in ("c1", "coefs", "m")}
    for tag in ("ring_reads_c0", "big_forms"):
        assert rms(outs[tag] - outs["small_forms"]) < 1e-6, tag
        for k in ("c1", "coefs", "m"):
            a_, b_ = ref_probe[tag][k], ref_probe["small_forms"][k]
            assert np.abs(a_ - b_).max() < 2e-5 * max(1.0, float(np.abs(b_).max())), (tag, k)
    o = orc.Oracle(sr, nb, blob)
    for b in (0, 41, 191):
        assert rms(outs["big_forms"][b] - o.enhance(wav[b], 6.0)) < WAVE_TOL, b
    m.close()


@pytest.mark.parametrize("sr,nb", [(16000, 2), (48000, 1)])
def test_small_launch_kernel_forms_equal_the_plain_forms(be, sr, nb):
    """Launch shapes that only small launches select (<= 512 frame rows: the ERB encoder pyramid as one launch -- enc_seg.h --
    the grouped linears chained per tile, mask + deep filter in one launch) against the plain forms, each on a FRESH model
    (a row one form forgets to write must not be inherited from the other's run).  The encoder pyramid repeats the plain
    kernels' arithmetic exactly, so its outputs are bit-identical -- which is what makes a clip enhanced alone equal the
    same clip inside a big batch; the others agree to rounding."""
    from dpdfnet_amd.weights import synth_blob
    blob = synth_blob(be.manifest(sr, nb), 31337)
    wav = np.stack([synth_clip(int(0.3 * sr) + 11, sr, 4000 + i) for i in range(5)]).astype(np.float32)
    outs, probes = {}, {}
    for tag, opts in (("fused", {}), ("plain_enc", {"fuse_enc": 0}), ("plain", {"fuse_small": 0})):
        m = be.HipModel(sr, nb, blob, 0)
        m.set_chunk_frames(16)
        for k, v in opts.items():
            m.set_option(k, v)
        outs[tag] = m.enhance_batch(wav, 100.0)
        probes[tag] = {k: m.debug_fetch(k) for k in ("e0", "e1", "e2", "e3", "m", "coefs")}
        m.close()
    np.testing.assert_array_equal(outs["fused"], outs["plain_enc"])
    for k in ("e0", "e1", "e2", "e3"):
        np.testing.assert_array_equal(probes["fused"][k], probes["plain_enc"][k])
    assert rms(outs["fused"] - outs["plain"]) < 1e-6
    for k in ("m", "coefs"):
        a_, b_ = probes["fused"][k], probes["plain"][k]
        assert np.abs(a_ - b_).max() < 2e-5 * max(1.0, float(np.abs(b_).max())), k


@pytest.mark.parametrize("sr,nb,S", [(16000, 2, 1), (16000, 4, 8), (48000, 1, 5), (48000, 2, 64), (48000, 8, 64), (16000, 8, 3)])
def test_streaming_hop_forms_equal_the_plain_chain(be, sr, nb, S):
    """Everything a single streaming hop does differently from a general call -- stage 2 on the main stream, staging + FIFO import
    + state copy as one prologue launch, one combined FIFO export behind the overlap-add (the host waits for the output event
    only), the decoders' GRU-256 steps pairwise in one launch, the DF decoder's pathway conv inside the front-end launch, the
    encoder / decoder pyramids, a DPRNN block's scan and glue as one launch -- against the same streams with all of it switched off, on fresh models: audio equal to rounding
    hop by hop, and so is every stream's state afterwards (read right after the last hop: the getter must see the late export)."""
    from dpdfnet_amd.weights import synth_blob
    blob = synth_blob(be.manifest(sr, nb), 4711)
    rng = np.random.default_rng(5)
    runs = {}
    for tag, opts in (("hop", {}), ("stack_launch", {"hop_stack": 1}), ("two_launch", {"hop_fused": 0}), ("event_join", {"hop_spin_join": 0}),
                      ("plain", {"single_chunk_inline": 0, "hop_prologue": 0, "late_export": 0, "dual_step": 0, "hop_pconv": 0,
                                 "fuse_enc": 0, "fuse_dec": 0, "snapshot": 0, "hop_fused": 0})):
        m = be.HipModel(sr, nb, blob, 0)
        for k, v in opts.items():
            m.set_option(k, v)
        st = be.HipStreams(m, S)
        r = np.random.default_rng(5)
        st.prime((0.05 * r.standard_normal((S, m.hop))).astype(np.float32))
        outs = [st.process((0.05 * r.standard_normal((S, m.hop))).astype(np.float32)).copy() for _ in range(9)]
        outs.append(st.process((0.05 * r.standard_normal((S, 3 * m.hop))).astype(np.float32)).copy())      # a three-hop call in between
        outs.append(st.process((0.05 * r.standard_normal((S, m.hop))).astype(np.float32)).copy())
        states = np.stack([st.get_state(i) for i in range(S)])
        runs[tag] = (outs, states)
        st.close(); m.close()
    for a_, b_ in zip(runs["hop"][0], runs["plain"][0]):
        assert a_.shape == b_.shape and rms(a_ - b_) < 1e-6
    assert np.abs(runs["hop"][1] - runs["plain"][1]).max() < 5e-5
    # a whole DPRNN stack as ONE persistent launch (dprnn_hop_stack.h, opt-in) against one launch per block: the same per-row arithmetic
    # whatever rows share a tile, bit for bit
    for a_, b_ in zip(runs["hop"][0], runs["stack_launch"][0]):
        np.testing.assert_array_equal(a_, b_)
    np.testing.assert_array_equal(runs["hop"][1], runs["stack_launch"][1])
    # scan + glue of a DPRNN block as one launch (dprnn_hop_block.h) against the two launches: the same arithmetic, bit for bit
    for a_, b_ in zip(runs["hop"][0], runs["two_launch"][0]):
        np.testing.assert_array_equal(a_, b_)
    np.testing.assert_array_equal(runs["hop"][1], runs["two_launch"][1])
    # ... and stage 2 waiting for the ERB stack by counter (emb_in spins) against the cross-stream event
    for a_, b_ in zip(runs["hop"][0], runs["event_join"][0]):
        np.testing.assert_array_equal(a_, b_)
    np.testing.assert_array_equal(runs["hop"][1], runs["event_join"][1])


def test_every_gru256_scan_form_agrees(be):
    """The forms of the GRU-256 recurrence -- single-workgroup scan (`gru256_cluster` = 0), 4-workgroup cluster with the input
    projection hoisted or inside the scan, 8- and 16-workgroup clusters for small launches, the two cells of a decoder stack as one wavefront launch (gru_stack.h) -- on
    the same input: equal to rounding, state included."""
    g, meta = load_golden("16k_nb1")
    blob = golden_blob(meta)
    m = be.HipModel(meta["sample_rate"], meta["nb"], blob, 0)
    o = make_oracle(meta, blob)
    spec = np.stack([o.stft(synth_clip(3000, 16000, 900 + i)) for i in range(40)])      # 40 streams = 3 tiles
    st0 = np.tile(m.initial_state(), (40, 1))
    outs = {}
    for tag, opts, ov in (("cluster", {"gru256_c16_tiles": 0, "gru256_cluster": 1, "gru256_stack": 0, "gru256_fused_x": 0}, 27 & ~16),
                          ("cluster_x", {"gru256_fused_x": 1, "gru256_fused_x_tiles": 1}, 27 & ~16),      # the input projection inside the four-workgroup scan (gru_clusterx.h)
                          ("cluster8", {"gru256_c16_tiles": 0}, 27), ("cluster16", {"gru256_c16_tiles": 4}, 27),
                          ("stack16", {"gru256_c16_tiles": 4, "gru256_stack": 1}, 27), ("stack16_serial", {}, 16),
                          ("single_wg", {"gru256_cluster": 0}, 27)):
        for k, v in opts.items():
            m.set_option(k, v)
        m.set_overlap(ov)
        outs[tag] = m.run_frames(spec, st0)
    m.set_option("gru256_cluster", 1); m.set_option("gru256_c16_tiles", 2); m.set_overlap(27)
    assert not np.array_equal(outs["stack16"][0], outs["cluster16"][0])         # the stacked form really ran
    assert not np.array_equal(outs["cluster_x"][0], outs["cluster"][0])         # ... and so did the fused-projection form
    ref, st_ref = outs["cluster"]
    scale = float(np.abs(ref).max())
    for tag, (out, st) in outs.items():
        assert np.abs(out - ref).max() < 2e-5 * scale, tag
        assert np.abs(st - st_ref).max() < 5e-5, tag
    r0, s0 = o.run_frames(spec[7])
    assert np.abs(ref[7] - r0).max() < STAGE_REL_TOL * scale and np.abs(st_ref[7] - s0).max() < 2e-4
    m.close()


@pytest.mark.parametrize("tag", ["48k_nb1", "16k_nb2"])
def test_two_stage_dft_equals_the_one_gemm_form_and_the_reference(tag, be):
    """dft2stage.h: the analysis / synthesis transforms of big launches as two small matrix stages (960 = 32 x 30, 320 = 32 x 10)
    against the one-GEMM form of the same engine (spectra, synthesis frames, waveforms), the reference's own STFT of the golden
    clip (`spec_head` / `spec_tail`, torch.stft of the reference modules) and the oracle -- plain and pipelined host paths,
    ragged lengths, a frame count that is no multiple of the frame tiles."""
    g, meta = load_golden(tag)
    blob = golden_blob(meta)
    sr, nb = meta["sample_rate"], meta["nb"]
    win = 960 if sr == 48000 else 320
    hop, F = win // 2, win // 2 + 1
    n = len(g["wav"])
    B = 19
    wav = np.stack([g["wav"]] + [synth_clip(n, sr, 700 + i) for i in range(B - 1)])
    T = 1 + (n + win) // hop
    o = make_oracle(meta, blob)
    res = {}
    for dft2 in (1, 0):
        m = be.HipModel(sr, nb, blob, 0)
        m.set_option("dft2", dft2)                # the two synthesis forms (the analysis is float64 on every path: test_float64_analysis_*)
        for pipe in (0, 1):
            m.set_option("host_pipe", pipe)
            y = m.enhance_batch(wav, None)
            spec = m.debug_fetch("raw_spec").reshape(B, T, F, 2)
            frames = m.debug_fetch("frames").reshape(B, T, win)
            res[(dft2, pipe)] = (y, spec, frames)
        m.close()
    y_ref, spec_ref, frames_ref = res[(0, 0)]
    scale = float(np.abs(spec_ref).max())
    for key in ((1, 0), (1, 1)):
        y, spec, frames = res[key]
        assert np.abs(spec - spec_ref).max() < 3e-6 * scale, (key, np.abs(spec - spec_ref).max() / scale)
        assert np.abs(frames - frames_ref).max() < 3e-6 * max(1.0, float(np.abs(frames_ref).max())), key
        assert rms(y - y_ref) < 5e-7 and np.abs(y - y_ref).max() < 6e-6, (key, rms(y - y_ref), np.abs(y - y_ref).max())
        # the reference's own analysis of the golden clip (unnormalised STFT, first 8 and last 4 frames)
        assert np.abs(spec[0, :8] - g["spec_head"]).max() < 3e-6 * scale and np.abs(spec[0, -4:] - g["spec_tail"]).max() < 3e-6 * scale
        assert rms(y[0] - g["enhanced"]) < 2e-6
        for b in (1, B - 1):
            assert rms(y[b] - o.enhance(wav[b])) < 2e-6
    np.testing.assert_array_equal(res[(1, 0)][0], res[(1, 1)][0])       # plain and pipelined host paths: bit-identical
    # ragged: per-clip reflection point / frame count inside the two-stage analysis
    lens = np.array([n, n - 1, 7 * hop + 3, 2 * hop, 100, 1] + [n - 13 * i for i in range(B - 6)], dtype=np.int32)
    m = be.HipModel(sr, nb, blob, 0)
    rows = m.enhance_batch_ragged([wav[b, : lens[b]].copy() for b in range(B)])
    m.set_option("dft2", 0)
    rows0 = m.enhance_batch_ragged([wav[b, : lens[b]].copy() for b in range(B)])
    m.close()
    for b in range(B):
        assert np.abs(rows[b] - rows0[b]).max() < 6e-6, (b, lens[b])
    for b in (2, 4):
        assert rms(rows[b] - o.enhance(wav[b, : lens[b]])) < 2e-6


# ----- float64 analysis DFT (dft64.h) and spectrally sparse inputs at 48 kHz ----------------------------------------------
from tests.util import SPARSE_MAIN, SPARSE_TAGS, SPARSE_TORTURE   # noqa: E402


@pytest.mark.parametrize("tag", ["48k_nb1", "16k_nb2"])
def test_float64_analysis_equals_the_float64_dft_to_the_last_bit_on_every_call_path(tag, be):
    """dft64.h: the analysis spectra of the engine = float64 DFT of the float32 product frame * window, rounded once -- on the
    whole-batch launch, on the per-chunk launches of the pipelined host path, for ragged lengths and for streaming hops.
    Compared with numpy's float64 rfft of the same float32 frames: at most one float32 ulp per value (two correctly rounded
    float64 transforms may round a value on a tie boundary differently), and the plain / pipelined paths bit-identical."""
    g, meta = load_golden(tag)
    blob = golden_blob(meta)
    sr, nb = meta["sample_rate"], meta["nb"]
    win = 960 if sr == 48000 else 320
    hop, F = win // 2, win // 2 + 1
    n = len(g["wav"])
    B = 19
    wav = np.stack([g["wav"]] + [synth_clip(n, sr, 900 + i) for i in range(B - 1)])
    wav[3] = np.round(wav[3] * 32768.0) / 32768.0
    wav[4, :] = 0.25
    T = 1 + (n + win) // hop
    o = make_oracle(meta, blob)
    w = o.window()

    def ref_spec(x):
        xp = np.pad(np.pad(x, (0, win)), (hop, hop), mode="reflect")
        fr = np.stack([(xp[t * hop: t * hop + win] * w).astype(np.float32) for t in range(1 + (len(x) + win) // hop)])
        c = np.fft.rfft(fr.astype(np.float64), axis=1)
        return np.stack([c.real.astype(np.float32), c.imag.astype(np.float32)], axis=-1)

    def ulp_close(a, b):
        # one float32 ulp of the value, or the float64 transform's own residue (1e-16 of the loudest bin: values that are
        # exactly zero in one summation order -- Im of DC / Nyquist -- and 1e-15 in another)
        return np.all(np.abs(a - b) <= 1.2e-7 * np.abs(b) + 1e-14 * float(np.abs(b).max()))

    m = be.HipModel(sr, nb, blob, 0)
    specs, ys = {}, {}
    for pipe in (0, 1):
        m.set_option("host_pipe", pipe)
        ys[pipe] = m.enhance_batch(wav, None)
        specs[pipe] = m.debug_fetch("raw_spec").reshape(B, T, F, 2)
    np.testing.assert_array_equal(specs[0], specs[1])
    np.testing.assert_array_equal(ys[0], ys[1])
    for b in (0, 3, 4, B - 1):
        r = ref_spec(wav[b])
        assert ulp_close(specs[0][b], r), (b, np.abs(specs[0][b] - r).max())
        big = np.abs(r) > 1e-9 * float(np.abs(r).max())         # (below: the float64 residue itself, e.g. every bin but 0 of the DC clip)
        assert (specs[0][b][big] != r[big]).mean() < 0.05, b      # (a handful of values land on the other side of a rounding boundary)
        assert rms(ys[0][b] - o.enhance(wav[b])) < 2e-6, b
    # ragged: per-clip reflection point and frame count
    lens = np.array([n, n - 1, 7 * hop + 3, 2 * hop, 100, 1] + [n - 13 * i for i in range(B - 6)], dtype=np.int32)
    rows = m.enhance_batch_ragged([wav[b, : lens[b]].copy() for b in range(B)])
    for b in (1, 2, 4, 5):
        assert rms(rows[b] - o.enhance(wav[b, : lens[b]])) < 2e-6, (b, lens[b])
    # a small launch (one clip: fewer frames than a frame tile row) takes the same kernel
    y1 = m.enhance_batch(wav[:1, : 3 * hop + 5], None)
    assert rms(y1[0] - o.enhance(wav[0, : 3 * hop + 5])) < 2e-6
    m.close()


@pytest.mark.parametrize("tag", SPARSE_TAGS)
def test_sparse_spectra_offline_paths_match_the_oracle_and_the_reference(tag, be):
    """Round-4 review weak #1: band-limited float32 audio, the same as 16-bit PCM, silence -> signal, DC and a +-1 square wave
    at 48 kHz, through enhance_batch on the whole-batch launch, the pipelined host path, small time chunks and a one-clip call.
    HIP vs oracle and vs the reference's float64-fed golden: < 1e-5 RMS (north_star budget 1e-4), i.e. orders of magnitude
    inside the reference's own fp32-FFT-vs-float64 spread stored in the fixture."""
    g, meta = load_golden(tag)
    blob = golden_blob(meta)
    o = make_oracle(meta, blob)
    classes = list(meta["classes"])
    wav = np.stack([g[f"wav_{c}"] for c in classes])
    want = [o.enhance(w) for w in wav]
    m = be.HipModel(48000, meta["nb"], blob, 0)
    runs = {}
    m.set_option("host_pipe", 0); runs["batch"] = m.enhance_batch(wav, None)
    m.set_option("host_pipe", 1); runs["pipelined"] = m.enhance_batch(np.concatenate([wav] * 8), None)[: len(classes)]
    m.set_chunk_frames(7); runs["chunks7"] = m.enhance_batch(wav, None); m.set_chunk_frames(0)
    runs["one_clip"] = np.stack([m.enhance_batch(w[None], None)[0] for w in wav])
    worst = {}
    for name, y in runs.items():
        assert np.isfinite(y).all(), name
        for i, cls in enumerate(classes):
            e_or, e_ref = rms(y[i] - want[i]), rms(y[i] - g[f"enh_f64_{cls}"])
            spread = meta["spread"][cls]["torch_vs_f64"]
            worst[cls] = max(worst.get(cls, 0.0), e_or, e_ref)
            assert e_or < 1e-5 and e_ref < 1e-5, (name, cls, e_or, e_ref)
            assert e_ref < 0.05 * spread, (name, cls, e_ref, spread)     # no further from the float64 golden than torch's is: 20 x closer
    print(f"[sparse {tag}] worst RMS vs oracle / float64 golden per class: " + ", ".join(f"{c} {v:.1e}" for c, v in worst.items()))
    m.close()


@pytest.mark.parametrize("S", [3, 6])
def test_sparse_spectra_streaming_hops_match_the_reference_stream_enhancer(S, be, tmp_path, monkeypatch):
    """The same for config 5's model on the persistent-state path: single hops through the C ABI for 3 streams (analysis buffers
    read in place) and 6 streams (staging prologue), and the public StreamEnhancer for chunk sizes hop and 171, against the
    reference StreamEnhancer's float64-rfft goldens (stream_48k_nb8_sparse.npz)."""
    from dpdfnet_amd import stream, weights
    from dpdfnet_amd.models import ModelInfo, ResolvedModel
    g, meta = load_golden("48k_nb8_sparse")
    blob = golden_blob(meta)
    G = np.load(GOLDEN / "stream_48k_nb8_sparse.npz")
    hop = 480
    rows = np.stack([G[f"wav_{c}"] for c in SPARSE_MAIN] * (S // 3))
    m = be.HipModel(48000, 8, blob, 0)
    st = m.open_streams(S)
    k = rows.shape[1] // hop
    st.prime(rows[:, :hop].copy())
    outs = [st.process(rows[:, j * hop:(j + 1) * hop].copy()) for j in range(1, k)]
    got = np.concatenate(outs, axis=1)
    st.close(); m.close()
    for i in range(S):
        ref = G[f"f64_{SPARSE_MAIN[i % 3]}_chunk480"]
        e = rms(got[i] - ref[: got.shape[1]])
        assert e < 1e-5, (i, e)
    if S != 3:
        return
    wfile = weights.save_blob(tmp_path / "w.npz", blob)
    info = ModelInfo(name="test_sparse", sample_rate=48000, frame_ms=20.0, description="", onnx_filename="w.onnx", dprnn_num_blocks=8)
    monkeypatch.setattr(stream, "resolve_model", lambda **_k: ResolvedModel(info=info, onnx_path=wfile))
    for cls in SPARSE_MAIN:
        wav = G[f"wav_{cls}"]
        for chunk in (hop, 171):
            se = stream.StreamEnhancer(model="ignored")
            y = np.concatenate([se.process(wav[i:i + chunk]) for i in range(0, len(wav), chunk)] + [se.flush()])
            ref = G[f"f64_{cls}_chunk{chunk}"]
            assert y.shape == ref.shape and rms(y - ref) < 1e-5, (cls, chunk, rms(y - ref))


# ----- non-finite inputs (round 6) --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["16k_nb2", "48k_nb8"])
def test_nonfinite_inputs_match_the_reference_and_stay_in_their_slot(be, tag):
    """A NaN / +Inf / -Inf sample (and a denormal-only clip) in SOME slots of a batch (reference goldens tests/golden/nonfinite_*.npz,
    generated by make_golden.py:make_nonfinite_fixture from onnx_model/dpdfnet.py:748-852 / dpdfnet_48khz_hr.py:820-924): the
    enhanced waveform of such a slot is non-finite in EXACTLY the reference's samples (torch.relu keeps a NaN: it stays in the clip's
    states to the end) and equal elsewhere; every other slot of the batch is BIT-IDENTICAL to a batch without the poisoned clips; the
    streaming path does the same hop by hop."""
    import json
    g = np.load(GOLDEN / f"nonfinite_{tag}.npz")
    meta = json.loads(bytes(g["meta_json"]).decode())
    sr, nb, n = meta["sample_rate"], meta["nb"], meta["n"]
    blob = golden_blob(meta)
    m = be.HipModel(sr, nb, blob, 0)
    classes = meta["classes"]
    clean = [synth_clip(n, sr, 900 + i) for i in range(len(classes) + 3)]
    base = m.enhance_batch(np.stack(clean), None)
    assert np.isfinite(base).all()
    # poisoned clips in the odd slots, clean clips around them
    batch = [c.copy() for c in clean]
    slots = {}
    for i, cls in enumerate(classes):
        slot = 1 + i if i < len(clean) - 1 else i
        batch[slot] = g[f"{cls}_wav"]; slots[cls] = slot
    out = m.enhance_batch(np.stack(batch), None)
    for b in range(len(clean)):
        if b not in slots.values():
            np.testing.assert_array_equal(out[b], base[b], err_msg=f"clean slot {b} changed beside poisoned clips")
    for cls, slot in slots.items():
        ref = g[f"{cls}_enhanced"]
        bad_o, bad_r = ~np.isfinite(out[slot]), ~np.isfinite(ref)
        assert np.array_equal(bad_o, bad_r), (tag, cls, int(bad_o.sum()), int(bad_r.sum()), int(np.argmax(bad_o)), int(np.argmax(bad_r)))
        fin = ~bad_r
        if fin.any():
            assert float(np.abs(out[slot][fin] - ref[fin]).max()) < 1e-5, (tag, cls)
    # chunked in time (stage 2 of a chunk under stage 1 of the next) and as a one-clip call: the same samples go bad
    m.set_chunk_frames(7)
    out7 = m.enhance_batch(np.stack(batch), None)
    m.set_chunk_frames(0)
    assert np.array_equal(np.isfinite(out7), np.isfinite(out))
    fin = np.isfinite(out)
    assert float(np.abs(out7[fin] - out[fin]).max()) < 2e-6
    # streaming: one poisoned stream between two clean ones, hop by hop; the clean streams stay bit-identical to a run of their own
    hop = m.hop
    k = (n // hop) * hop
    S = be.HipStreams(m, 3)
    trio = np.stack([clean[0][:k], g["nan_sample_wav"][:k], clean[2][:k]])
    S.prime(trio[:, :hop].copy())
    got = [S.process(trio[:, j:j + hop].copy()) for j in range(hop, k, hop)]
    got = np.concatenate(got, axis=1)
    S2 = be.HipStreams(m, 3)
    solo = np.stack([clean[0][:k], clean[1][:k], clean[2][:k]])
    S2.prime(solo[:, :hop].copy())
    want = np.concatenate([S2.process(solo[:, j:j + hop].copy()) for j in range(hop, k, hop)], axis=1)
    np.testing.assert_array_equal(got[0], want[0]); np.testing.assert_array_equal(got[2], want[2])
    assert not np.isfinite(got[1]).all() and not np.isfinite(got[1][-hop:]).any(), "the poisoned stream must stay poisoned to the end"
    st = S.get_state(1)
    assert not np.isfinite(st).all()
    S.close(); S2.close(); m.close()
