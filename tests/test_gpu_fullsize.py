"""GPU (-m gpu): BASELINE.json configurations at FULL size, through size-independent properties
(the oracle needs ~4.5 ms per frame, so it checks a few whole clips, not all 256):
  * spot clips of the big batch against the CPU oracle on the same clip (full 10 s),
  * batch-slot invariance (the same clip in two slots gives bit-identical output),
  * causality/prefix property: enhance(x[:n0]) == enhance(x)[:n0 - tail] (the model only sees 2+2
    frames of look-ahead),
  * time-chunk invariance, finiteness, the reference's zero tail,
  * config 5: 64 concurrent device-resident 48 kHz streams, hop by hop, against the oracle."""
import numpy as np
import pytest

from tests.util import rms, synth_clip

pytestmark = pytest.mark.gpu
WAVE_TOL = 2e-6


def _model(be, sr, nb, seed=20260417):
    from dpdfnet_amd.weights import synth_blob
    blob = synth_blob(be.manifest(sr, nb), seed)
    return be.HipModel(sr, nb, blob, 0), blob


@pytest.fixture(scope="module")
def be():
    from dpdfnet_amd import backend
    assert backend.device_count() >= 1
    return backend


@pytest.mark.parametrize("nb,B,check_oracle", [(2, 256, True), (4, 256, True), (8, 256, True)])
def test_baseline_batch_configs_full_size(be, nb, B, check_oracle):
    """configs[1] dpdfnet2 / configs[2] per-GPU shard dpdfnet4 / configs[3] dpdfnet8: 256 x 10 s."""
    from oracle import oracle as orc
    sr, n = 16000, 160000
    m, blob = _model(be, sr, nb)
    # AUTOMATIC chunk schedule (5 x 192 + 43 frames at 256 clips, tail-chunk rule armed): the execution shape bench.py times
    base = [synth_clip(n, sr, 9000 + i) for i in range(6)]
    # every slot holds a DIFFERENT signal (6 clips x a slot-dependent gain and circular shift), so a slot-dependent bug
    # shows up in the oracle spot checks below, not only in the bit-identity asserts
    gain = 0.4 + 0.6 * ((np.arange(B) * 37) % 101) / 100.0
    wav = np.stack([np.roll(base[b % 6], 131 * b) * gain[b] for b in range(B)]).astype(np.float32)
    wav[200] = wav[3]                                   # slots 3 and 200 hold the same clip
    wav[6] = wav[0]
    wav[7, 40000:] = 0.0                                # a clip that goes silent
    out = m.enhance_batch(wav, None)
    assert out.shape == wav.shape and np.isfinite(out).all()
    np.testing.assert_array_equal(out[3], out[200])
    np.testing.assert_array_equal(out[0], out[6])
    assert rms(out[1] - out[7]) > 1e-3                  # different slots really carry different results
    assert np.all(out[:, -m.win_len:] == 0.0)
    assert m.num_frames(n) == 1003
    if check_oracle:
        o = orc.Oracle(sr, nb, blob)
        for b in ((7, 131, 255) if nb < 8 else (7, 255)):    # the oracle needs ~8 s per dpdfnet8 clip
            err = rms(out[b] - o.enhance(wav[b]))
            assert err < WAVE_TOL, (b, err)
    # prefix/causality on a slice of the batch
    n0 = 48000
    pre = m.enhance_batch(wav[:4, :n0], None)
    keep = n0 - 4 * m.win_len
    assert rms(pre[:, :keep] - out[:4, :keep]) < WAVE_TOL
    # chunk invariance at size: forced 128- and 200-frame schedules against the automatic one
    m.set_chunk_frames(128)
    out1 = m.enhance_batch(wav, None)
    assert rms(out1 - out) < 1e-6
    assert rms(out1[255] - out[255]) < 1e-6 and rms(out1[131] - out[131]) < 1e-6
    m.set_chunk_frames(200)
    out2 = m.enhance_batch(wav[:32], None)
    assert rms(out2 - out[:32]) < 1e-6
    m.close()


def test_every_slot_of_the_headline_batch_against_the_oracle(be):
    """The headline workload (dpdfnet4, 256 x 10 s, automatic chunk schedule) with EVERY slot checked against the CPU oracle --
    256 different signals, one oracle per host thread.  The spot checks above would miss an error confined to a slot, a row tile or
    a band of the batch that they do not sample.  Every sixteenth slot is compared over the whole 10 s; the others over their first
    2.5 s: the frame function is causal, so the oracle run on a PREFIX of a clip equals the prefix of the oracle run on the clip up to
    five hops before the cut (checked bit for bit on the CPU) -- a third of the oracle's time, still every slot, every row tile and every
    chunk boundary of the first quarter of the clip."""
    import os
    import threading
    from oracle import oracle as orc
    sr, n, nb, B = 16000, 160000, 4, 256
    m, blob = _model(be, sr, nb)
    base = [synth_clip(n, sr, 9300 + i) for i in range(8)]
    gain = 0.3 + 0.7 * ((np.arange(B) * 53) % 97) / 96.0
    wav = np.stack([np.roll(base[b % 8], 977 * b) * gain[b] for b in range(B)]).astype(np.float32)
    out = m.enhance_batch(wav, None)
    m.close()
    assert np.isfinite(out).all()
    nthr = max(1, min(len(os.sched_getaffinity(0)), 64))
    errs = np.full(B, np.inf)

    n_pre, cut = 40000, 40000 - 5 * 160

    def work(k):
        o = orc.Oracle(sr, nb, blob)
        for b in range(k, B, nthr):
            if b % 16 == 0:
                errs[b] = rms(out[b] - o.enhance(wav[b]))
            else:
                errs[b] = rms(out[b][:cut] - o.enhance(wav[b][:n_pre])[:cut])

    ths = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    worst = int(np.argmax(errs))
    assert errs[worst] < WAVE_TOL, (worst, float(errs[worst]))
    print(f"256 slots vs oracle: worst {errs.max():.2e} (slot {worst}), median {np.median(errs):.2e}, {nthr} host threads")


def test_whole_config3_batch_on_one_gpu(be):
    """configs[2] (dpdfnet4, 2048 clips x 10 s = 2.05 M frames) UNSHARDED through ONE engine call -- what a single
    288 GB GPU is sized for; guards the 64-bit indexing of spectra / frame buffers (B*T*win = 657 M floats), the auto
    chunking at large B (16-frame chunks) and the 512-workgroup GRU-256 cluster launches on 256 CUs."""
    from oracle import oracle as orc
    sr, n, nb, B = 16000, 160000, 4, 2048
    m, blob = _model(be, sr, nb)
    base = [synth_clip(n, sr, 9100 + i) for i in range(4)]
    wav = np.empty((B, n), np.float32)
    for b in range(B):
        wav[b] = base[b % 4] * np.float32(0.5 + 0.5 * ((b * 29) % 64) / 63.0)
    wav[2044] = wav[0]; wav[1025] = wav[1]
    wav[B - 1, :] = base[1][::-1]                        # the very last slot holds something unique
    out = m.enhance_batch(wav, None)
    assert out.shape == wav.shape and np.isfinite(out).all()
    np.testing.assert_array_equal(out[0], out[2044])
    np.testing.assert_array_equal(out[1], out[1025])
    o = orc.Oracle(sr, nb, blob)
    for b in (1337, B - 1):
        err = rms(out[b] - o.enhance(wav[b]))
        assert err < WAVE_TOL, (b, err)
    m.close()


def test_one_long_clip(be):
    """A single 60 s utterance (6001 frames): the whole clip is ONE time chunk at B=1 (scans of 6001 dependent steps,
    GRU-256 granule epochs far past one launch), against the oracle and against a 1000-frame chunking."""
    from oracle import oracle as orc
    sr, nb, n = 16000, 2, 60 * 16000
    m, blob = _model(be, sr, nb)
    wav = synth_clip(n, sr, 4242)[None, :]
    out = m.enhance_batch(wav, 12.0)
    assert m.num_frames(n) == 6003 and np.isfinite(out).all()
    ref = orc.Oracle(sr, nb, blob).enhance(wav[0], 12.0)
    assert rms(out[0] - ref) < WAVE_TOL
    m.set_chunk_frames(1000)
    assert rms(m.enhance_batch(wav, 12.0) - out) < 1e-6
    m.close()


def test_config5_64_streams_48khz_dpdfnet8(be):
    """configs[4]: 64 concurrent StreamEnhancer states on dpdfnet8_48khz_hr, 10 ms hops."""
    from oracle import oracle as orc
    sr, nb, S, hops = 48000, 8, 64, 40
    m, blob = _model(be, sr, nb)
    hop, win = m.hop, m.win_len
    streams = be.HipStreams(m, S)
    pcm = np.stack([synth_clip((hops + 1) * hop, sr, 700 + i) for i in range(S)])
    streams.prime(pcm[:, :hop])
    outs = []
    for j0, k in ((0, 1), (1, 1), (2, 8), (10, 30)):                      # hop-by-hop then multi-hop calls
        outs.append(streams.process(pcm[:, (1 + j0) * hop:(1 + j0 + k) * hop]))
    got = np.concatenate(outs, axis=1)
    assert got.shape == (S, hops * hop) and np.isfinite(got).all()
    # oracle: causal STFT (stream.py:119-126) -> frame function -> OLA (stream.py:138-155), numpy host DSP
    o = orc.Oracle(sr, nb, blob)
    w = o.window()
    for s in (0, 63):
        x = pcm[s]
        spec = np.stack([np.fft.rfft(x[j * hop: j * hop + win] * w) for j in range(hops)])
        spec_ri = np.stack([spec.real, spec.imag], axis=-1).astype(np.float32)
        spec_e, st = o.run_frames(spec_ri)
        ola = np.zeros(win, dtype=np.float32)
        ref = []
        for j in range(hops):
            fr = (np.fft.irfft(spec_e[j, :, 0] + 1j * spec_e[j, :, 1], n=win) * w).astype(np.float32)
            ola += fr
            ref.append(ola[:hop].copy())
            ola[:hop] = ola[hop:]; ola[hop:] = 0.0
        ref = np.concatenate(ref)
        assert rms(got[s] - ref) < WAVE_TOL, (s, rms(got[s] - ref))
        assert np.abs(streams.get_state(s) - st).max() < 5e-4
    streams.reset(5)
    np.testing.assert_array_equal(streams.get_state(5), m.initial_state())
    streams.close(); m.close()


def test_limb_and_fp32_gru64_kernels_agree_on_every_clip_of_a_pipelined_batch(be):
    """The default GRU-64 throughput kernels on bf16 limbs (gru_limb.h, gru64_limbs = 3) against the fp32-MFMA kernels of gru_scan.h (gru64_limbs = 0), 256 clips through
    the multi-chunk pipeline (stage 2 of a chunk under stage 1 of the next), several times over: every clip within fp32 rounding of the
    other path, each path bit-identical to itself run to run AND to its own serial schedule (one stream, nothing side by side).  This is the
    dynamic guard of DESIGN.md section 6: in round 5 the deep-filter kernel's packed FP32 products (v_pk_fma_f32) lost half of a result
    beside the limb kernels' bf16 MFMAs -- single wrong low-band frames in 23-36 clips per run, never the same twice; the library is built
    without packed FP32 instructions since round 6 (tests/test_no_packed_fp32.py is the static guard)."""
    m, blob = _model(be, 16000, 4)
    rng = np.random.default_rng(3)
    B, n = 256, 160 * 64 * 8
    wav = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
    m.set_chunk_frames(64)
    m.set_option("gru64_limbs", 0)
    y0 = m.enhance_batch(wav, None)
    for rep in range(3):
        assert np.array_equal(y0, m.enhance_batch(wav, None)), rep
    m.set_overlap(0)
    assert np.array_equal(y0, m.enhance_batch(wav, None)), "fp32-MFMA kernels: pipelined schedule differs from the serial one"
    m.set_option("gru64_limbs", 3)
    y1_serial = m.enhance_batch(wav, None)
    m.set_overlap(27)
    for rep in range(6):
        y1 = m.enhance_batch(wav, None)
        d = np.sqrt(np.mean((y1.astype(np.float64) - y0) ** 2, axis=1))
        assert d.max() < 5e-7, (rep, int(d.argmax()), float(d.max()))
        assert np.array_equal(y1, y1_serial), (rep, "limb kernels: pipelined schedule differs from the serial one")
    m.set_chunk_frames(0)
    y2 = m.enhance_batch(wav, None)                     # the automatic schedule
    assert np.sqrt(np.mean((y2.astype(np.float64) - y0) ** 2, axis=1)).max() < 5e-7
    # and against the oracle, at the one tolerance both families are held to
    from oracle import oracle as orc
    for b in (0, 131, 255):
        assert rms(y2[b] - orc.Oracle(16000, 4, blob).enhance(wav[b])) < WAVE_TOL, b
    m.close()


@pytest.mark.parametrize("tag,limbs", [("16k_nb2", 3), ("16k_nb2", 0), ("48k_nb8", 3)])
def test_nonfinite_clips_in_a_throughput_batch_match_the_reference_and_stay_in_their_slot(be, tag, limbs):
    """The reference's NaN / +-Inf / denormal goldens (tests/golden/nonfinite_*.npz, make_golden.py:make_nonfinite_fixture from
    onnx_model/dpdfnet.py:748-852 / dpdfnet_48khz_hr.py:820-924) inside a 256-clip batch: this is the execution shape of the headline -- the
    FUSED GRU-64 throughput kernels, on bf16 limbs (the default) and as fp32 MFMAs -- where tests/test_gpu_parity.py's non-finite test runs the
    small-batch forms.  A poisoned clip shares its 16-row tiles with its neighbours' rows; a NaN must not cross a row of a tile (the limb split
    of a NaN or an Inf is NaN in every limb: it stays in its own row of the matrix product), so: the poisoned slots are non-finite in EXACTLY
    the reference's samples and equal elsewhere; every other slot is BIT-IDENTICAL to the batch without poisoned clips."""
    import json
    from tests.util import GOLDEN, golden_blob
    g = np.load(GOLDEN / f"nonfinite_{tag}.npz")
    meta = json.loads(bytes(g["meta_json"]).decode())
    sr, nb, n = meta["sample_rate"], meta["nb"], meta["n"]
    m = be.HipModel(sr, nb, golden_blob(meta), 0)
    m.set_option("gru64_limbs", limbs)
    B = 256
    clean = np.stack([synth_clip(n, sr, 4000 + i) for i in range(B)])
    m.profile(True)
    base = m.enhance_batch(clean, None)
    rep = m.profile_report(); m.profile(False)
    want = "gru64_l3_kernel" if limbs else "gru64_epi_kernel"
    assert any(k.startswith(want) for k in rep), sorted(rep)           # the fused throughput kernels of that family did run
    assert np.isfinite(base).all()
    batch = clean.copy()
    slots = {cls: s for cls, s in zip(meta["classes"], (1, 16, 127, 200, 255))}      # first / last rows of tiles, the last slot of the batch
    for cls, s in slots.items():
        batch[s] = g[f"{cls}_wav"]
    out = m.enhance_batch(batch, None)
    for b in range(B):
        if b not in slots.values():
            assert np.array_equal(out[b], base[b]), f"clean slot {b} changed beside poisoned clips"
    for cls, s in slots.items():
        ref = g[f"{cls}_enhanced"]
        bad_o, bad_r = ~np.isfinite(out[s]), ~np.isfinite(ref)
        assert np.array_equal(bad_o, bad_r), (tag, limbs, cls, int(bad_o.sum()), int(bad_r.sum()))
        fin = ~bad_r
        if fin.any():
            assert float(np.abs(out[s][fin] - ref[fin]).max()) < 1e-5, (tag, limbs, cls)
    m.close()
