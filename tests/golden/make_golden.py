#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs ONLY in the build container, where /root/reference is mounted: it imports the reference's
PyTorch streaming modules (onnx_model/dpdfnet.py, onnx_model/dpdfnet_48khz_hr.py) and host code
(package/src/dpdfnet/{audio,stream}.py), loads OUR portable seeded synthetic weights into them
(the reference ships no checkpoints and no network goldens) and records
inputs + expected outputs as small .npz files.  Nothing of the reference itself is stored:
fixtures are data only (waveforms, spectra, state vectors, per-stage activations).

    python tests/golden/make_golden.py            # regenerates every fixture

The fixtures pin (a) the CPU oracle (tests/test_oracle_golden.py, no GPU) and (b) through the
oracle and directly, the HIP path (tests/test_gpu_parity.py, -m gpu).
"""
from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True          # importing the reference must not drop __pycache__ into its (read-only) checkout
REPO = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

SEED = 20260417
PROBE_FRAMES = (0, 1, 2, 5, 50)


def synth_clip(n: int, sr: int, seed: int) -> np.ndarray:
    """SURVEY.md section 8(d) synthetic input: 0.05 N(0,1) + AM sinusoid, clipped to [-1,1]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    f0 = rng.uniform(100.0, 1000.0)
    x = 0.05 * rng.standard_normal(n) + 0.1 * np.sin(2 * np.pi * f0 * t) * (1.0 + np.sin(2 * np.pi * 3.0 * t))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def _stub_modules():
    for name in ("soundfile", "librosa", "onnxruntime"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    ort = sys.modules["onnxruntime"]
    if not hasattr(ort, "InferenceSession"):
        ort.InferenceSession = object
        ort.SessionOptions = object
        ort.GraphOptimizationLevel = types.SimpleNamespace(ORT_ENABLE_ALL=0)


def build_reference_model(sr: int, nb: int, blob: np.ndarray, entries):
    import torch
    from dpdfnet_amd.weights import unpack_to_streaming_state_dict

    if sr == 16000:
        from onnx_model.dpdfnet import DPDFNet as Net
    else:
        from onnx_model.dpdfnet_48khz_hr import DPDFNet48HR as Net
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = Net(dprnn_num_blocks=nb).eval()
    sd = {k: torch.from_numpy(v) for k, v in unpack_to_streaming_state_dict(entries, blob).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    allowed = ("erb_fb", "erb_inv_fb", "mask.erb_inv_fb", "stft.w", "istft.w_inv", "istft_norm.w_inv",
               "enc.lsnr_fc.0.weight", "enc.lsnr_fc.0.bias")
    bad = [k for k in missing if k not in allowed and "num_batches_tracked" not in k]
    assert not bad, f"manifest does not cover reference tensors: {bad}"
    return model


def run_reference_frames(model, spec_unnorm: np.ndarray, probe_frames=PROBE_FRAMES, state=None):
    """spec_unnorm [T,F,2] (librosa-style unnormalised STFT) -> dict of goldens.  Applies the export
    wrapper's x wnorm / / wnorm (reference onnx_model/export_dpdfnet_to_onnx.py:14-25)."""
    import torch

    acts = {}

    def grab(name, pick=lambda o: o):
        def hook(_m, _i, o):
            acts[name] = pick(o).detach().cpu().numpy().copy()
        return hook

    first = lambda o: o[0]
    hooks = [
        model.erb_norm.register_forward_hook(grab("feat_erb", first)),      # [1,1,E]
        model.spec_norm.register_forward_hook(grab("feat_spec_ri", first)),  # [1,1,D,2]
        model.enc.erb_conv0.register_forward_hook(grab("e0")),
        model.enc.erb_conv1.register_forward_hook(grab("e1")),
        model.enc.erb_conv2.register_forward_hook(grab("e2")),
        model.enc.erb_conv3.register_forward_hook(grab("e3")),
        model.enc.df_conv0.register_forward_hook(grab("c0")),
        model.enc.df_conv1.register_forward_hook(grab("c1")),
        model.enc.emb_gru.register_forward_hook(grab("emb", first)),
        model.erb_dec.register_forward_hook(grab("m", first)),
        model.df_dec.register_forward_hook(grab("coefs_fk", first)),        # [1,1,D,10]
    ]
    if hasattr(model.enc.dprnn_erb, "blocks"):
        hooks.append(model.enc.dprnn_erb.register_forward_hook(grab("e3_dprnn", first)))
        hooks.append(model.enc.dprnn_df.register_forward_hook(grab("c1_dprnn", first)))

    wnorm = np.float32(model.wnorm)
    inv_wnorm = np.float32(1.0 / float(model.wnorm))
    T = spec_unnorm.shape[0]
    st = model.initial_state(dtype=torch.float32) if state is None else torch.from_numpy(state.copy())
    init_state = st.numpy().copy()
    out = np.zeros_like(spec_unnorm)
    probes = {}
    with torch.no_grad():
        for t in range(T):
            x = torch.from_numpy(spec_unnorm[t:t + 1][None]) * torch.tensor(wnorm)
            y, st = model(x, st)
            out[t] = (y * torch.tensor(inv_wnorm)).numpy()[0, 0]
            if t in probe_frames:
                for k, v in acts.items():
                    probes[f"f{t}_{k}"] = v.reshape(-1).astype(np.float32)
    for h in hooks:
        h.remove()
    return dict(spec_e=out, state_out=st.numpy().copy(), init_state=init_state, **probes)


def make_model_fixture(tag: str, sr: int, nb: int, seconds: float, seed: int, attn_dbs=(0.0, 12.0), stress: str = ""):
    """stress: "" = the plain seeded weights; "hot" / "stiff" = dpdfnet_amd.weights.stress_blob (saturating GRU gates;
    BatchNorm running_var ~ eps and LayerNorm gains x 5) -- weight-robustness goldens."""
    import torch
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, synth_blob, stress_blob

    entries = parse_manifest_text(orc.manifest_text(sr, nb))
    blob = stress_blob(entries, seed, stress) if stress else synth_blob(entries, seed)
    model = build_reference_model(sr, nb, blob, entries)
    n = int(seconds * sr)
    wav = synth_clip(n, sr, seed + 1)
    win = model.stft.win_len

    # reference analysis: onnx_model/dpdfnet.py:854-860 (pad win, torch.stft center/reflect), but
    # kept UNNORMALISED as package/src/dpdfnet/audio.py:104-117 hands it to session.run
    with torch.no_grad():
        audio_pad = torch.nn.functional.pad(torch.from_numpy(wav)[None], (0, win))
        spec_c = model.stft(audio_pad).transpose(1, 2)[0]           # [T,F] complex64
    spec = torch.view_as_real(spec_c).numpy().astype(np.float32)     # [T,F,2]
    g = run_reference_frames(model, spec)

    # reference synthesis: torch.istft == librosa.istft semantics (dpdfnet.py:862-873, audio.py:120-136)
    def synth(spec_e: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            c = torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(spec_e)))[None]  # [1,T,F]
            audio = model.istft(c.transpose(1, 2))
            audio = torch.nn.functional.pad(audio[:, win * 2:], (0, win * 2))
        a = audio[0].numpy()
        out = np.zeros(n, dtype=np.float32)
        m = min(n, a.shape[0])
        out[:m] = a[:m]                                              # fit_length audio.py:30-38
        return out

    fix = dict(
        wav=wav, spec_head=spec[:8].copy(), spec_tail=spec[-4:].copy(),
        enhanced=synth(g["spec_e"]),
        spec_e_head=g["spec_e"][:64].copy(), state_out=g["state_out"], init_state=g["init_state"],
    )
    fix.update({k: v for k, v in g.items() if k.startswith("f")})
    # attenuation limit: the reference's own apply_attn_limit (package/src/dpdfnet/audio.py:41-76)
    from dpdfnet.audio import apply_attn_limit
    for db in attn_dbs:
        se = apply_attn_limit(spec[None], g["spec_e"][None], db)[0]
        fix[f"enhanced_attn{int(db)}"] = synth(se)
    meta = dict(tag=tag, sample_rate=sr, nb=nb, seed=seed, n=n, T=int(spec.shape[0]),
                state_size=int(model.state_size()), n_weights=int(blob.size),
                probe_frames=list(PROBE_FRAMES))
    if stress:
        meta["stress"] = stress
    fix["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"model_{tag}.npz", **fix)
    print(f"[golden] model_{tag}.npz  T={spec.shape[0]} S={model.state_size()} "
          f"rms_in={np.sqrt(np.mean(wav**2)):.4f} rms_out={np.sqrt(np.mean(fix['enhanced']**2)):.4f} "
          f"m~[{g['f50_m'].min():.3f},{g['f50_m'].max():.3f}] coefs_rms={np.sqrt(np.mean(g['f50_coefs_fk']**2)):.3f}")
    return model, entries, blob


def make_constants_fixture():
    """Config-derived constants the reference computes: vorbis window, ERB band widths, norm inits,
    48 kHz empirical init tables (onnx_model/init_norms.py), state sizes, manifest shapes."""
    import io, contextlib
    import torch
    from model.utils import vorbis_window, erb_filter_banks, get_wnorm
    from onnx_model.layers import ErbNorm, SpecNorm, MagNorm48, SpecNorm48
    from dpdfnet.audio import vorbis_window as pkg_window

    fb = erb_filter_banks(nfft=320, low_freq=0, fs=16000, n_filters=32, min_nb_freqs=1)
    out = dict(
        window_320=vorbis_window(320).numpy(), window_960=vorbis_window(960).numpy(),
        pkg_window_320=pkg_window(320), pkg_window_960=pkg_window(960),
        erb_widths_16k=fb.sum(axis=1).astype(np.int32),
        wnorm_16k=np.float32(get_wnorm(320, 160)), wnorm_48k=np.float32(get_wnorm(960, 480)),
        erb_norm_init_16k=ErbNorm(32, 0.98).initial_state().numpy(),
        spec_norm_init_16k=SpecNorm(96, 0.98).initial_state().numpy(),
        erb_norm_init_48k=MagNorm48(481, 0.98).initial_state().numpy(),
        spec_norm_init_48k=SpecNorm48(96, 0.98).initial_state().numpy(),
    )
    sizes = {}
    from onnx_model.dpdfnet import DPDFNet
    from onnx_model.dpdfnet_48khz_hr import DPDFNet48HR
    with contextlib.redirect_stdout(io.StringIO()):
        for nb in (0, 1, 2, 4, 8):
            sizes[f"16000_{nb}"] = int(DPDFNet(dprnn_num_blocks=nb).state_size())
        for nb in (1, 2, 8):
            sizes[f"48000_{nb}"] = int(DPDFNet48HR(dprnn_num_blocks=nb).state_size())
    out["state_sizes_json"] = np.frombuffer(json.dumps(sizes).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / "constants.npz", **out)
    print("[golden] constants.npz", sizes)


def make_host_dsp_fixture():
    """Known answers for the host DSP helpers of the path (package/src/dpdfnet/audio.py)."""
    from dpdfnet.audio import apply_attn_limit, fit_length, to_mono, pcm16_safe
    rng = np.random.default_rng(SEED + 77)
    noisy = rng.standard_normal((1, 9, 5, 2)).astype(np.float32)
    enh = rng.standard_normal((1, 9, 5, 2)).astype(np.float32)
    stereo = rng.standard_normal((50, 2)).astype(np.float32)
    x = rng.uniform(-1.5, 1.5, 64).astype(np.float32)
    np.savez_compressed(
        OUT / "host_dsp.npz", noisy=noisy, enh=enh,
        attn0=apply_attn_limit(noisy, enh, 0.0), attn6=apply_attn_limit(noisy, enh, 6.0),
        attn_inf=apply_attn_limit(noisy, enh, float("inf")), attn_none=apply_attn_limit(noisy, enh, None),
        stereo=stereo, mono=to_mono(stereo), x=x, pcm16=pcm16_safe(x),
        fit_short=fit_length(x, 40), fit_long=fit_length(x, 80),
    )
    print("[golden] host_dsp.npz")


def make_stream_fixture(model, sr: int, tag: str):
    """StreamEnhancer goldens: the reference's stream.py driven by (a) a passthrough session and
    (b) the real frame function, for several chunk sizes (package/src/dpdfnet/stream.py:74-200)."""
    import torch
    import dpdfnet.stream as ref_stream
    from dpdfnet.models import ModelInfo, ResolvedModel

    win = 320 if sr == 16000 else 960
    F = win // 2 + 1

    class _In:
        def __init__(self, name, shape): self.name, self.shape = name, shape

    class _Session:
        def __init__(self, fn): self.fn = fn
        def get_inputs(self): return [_In("spec", [1, 1, F, 2]), _In("state_in", [model.state_size()])]
        def get_outputs(self): return [_In("spec_e", None), _In("state_out", None)]
        def run(self, _names, feeds): return self.fn(feeds["spec"], feeds["state_in"])

    wnorm = torch.tensor(np.float32(model.wnorm)); inv = torch.tensor(np.float32(1.0 / float(model.wnorm)))

    def real_fn(spec, state):
        with torch.no_grad():
            y, st = model(torch.from_numpy(np.ascontiguousarray(spec)) * wnorm, torch.from_numpy(state.copy()))
        return [(y * inv).numpy()[None] if y.dim() == 3 else (y * inv).numpy(), st.numpy()]

    def pass_fn(spec, state):
        return [spec.copy(), state.copy()]

    def make(fn):
        init = model.initial_state(dtype=torch.float32).numpy()
        rt = ref_stream.RuntimeModel(session=_Session(fn), init_state=init, in_spec_name="spec",
                                     in_state_name="state_in", out_spec_name="spec_e", out_state_name="state_out")
        ref_stream.resolve_model = lambda **kw: ResolvedModel(
            info=ModelInfo(name="x", sample_rate=sr, frame_ms=20.0, description="", onnx_filename="x.onnx"),
            onnx_path=Path("/dev/null"))
        ref_stream.build_runtime_model = lambda _p: rt
        ref_stream.infer_win_len = lambda _s, _sr: win
        return ref_stream.StreamEnhancer(model="x")

    n = int(0.35 * sr) + 37
    wav = synth_clip(n, sr, SEED + 5)
    out = dict(wav=wav)
    for kind, fn in (("pass", pass_fn), ("real", real_fn)):
        for chunk in (7, win // 2, 171, 512, n):
            se = make(fn)
            pieces = [se.process(wav[i:i + chunk]) for i in range(0, n, chunk)]
            pieces.append(se.flush())
            out[f"{kind}_chunk{chunk}"] = np.concatenate(pieces).astype(np.float32)
            out[f"{kind}_chunk{chunk}_nflush"] = np.int32(len(pieces[-1]))
    np.savez_compressed(OUT / f"stream_{tag}.npz", **out)
    print(f"[golden] stream_{tag}.npz", {k: v.shape for k, v in out.items() if not k.endswith('nflush')})


def make_checkpoint_keys_fixture():
    """N1: the checkpoint key names a user's HF `checkpoints/*.pth` carries.  Instantiates the reference's OFFLINE twins
    (model/dpdfnet.py, model/dpdfnet_48khz_hr.py), records every state_dict key with its shape, and the streaming-module key the
    reference's own `correct_state_dict` maps it to (onnx_model/dpdfnet.py:876-888, onnx_model/dpdfnet_48khz_hr.py:948-963; None
    = dropped, i.e. `mask.erb_inv_fb` at 48 kHz).  Also proves the naming both ways: our synthetic weights, renamed to the
    offline names through that mapping, load into the offline twin with strict=True.  Data only: names and shapes."""
    import contextlib, importlib.util, io
    import torch
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, synth_blob, unpack_to_streaming_state_dict
    from onnx_model.dpdfnet import correct_state_dict as fix16, DPDFNet as S16
    from onnx_model.dpdfnet_48khz_hr import correct_state_dict as fix48, DPDFNet48HR as S48

    sys.path.insert(0, str(REF / "model"))      # the twins use bare `import multiframe`, `from modules import ...`
    def load(fname, modname):
        spec = importlib.util.spec_from_file_location(modname, REF / "model" / fname)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    out = {}
    for tag, sr, fname, cls, fix, Stream in (("16k", 16000, "dpdfnet.py", "DPDFNet", fix16, S16),
                                              ("48k", 48000, "dpdfnet_48khz_hr.py", "DPDFNet48HR", fix48, S48)):
        mod = load(fname, f"ref_offline_{tag}")
        nb = 1
        with contextlib.redirect_stdout(io.StringIO()):
            off = getattr(mod, cls)(dprnn_num_blocks=nb).eval()
            stream = Stream(dprnn_num_blocks=nb).eval()
        sd = off.state_dict()
        # map each offline key individually through the reference's renaming
        mapping = {}
        for k, v in sd.items():
            r = fix({k: v})
            mapping[k] = next(iter(r.keys())) if r else None
        # our synthetic weights under the offline names -> strict load into the offline twin
        entries = parse_manifest_text(orc.manifest_text(sr, nb))
        ssd = unpack_to_streaming_state_dict(entries, synth_blob(entries, SEED + 900))
        renamed = {}
        for k, v in sd.items():
            sk = mapping[k]
            renamed[k] = torch.from_numpy(ssd[sk]).reshape(v.shape) if sk in ssd else v
        off.load_state_dict(renamed, strict=True)
        # and the reference's own conversion of that offline checkpoint loads into the streaming module
        missing, unexpected = stream.load_state_dict(fix(renamed), strict=False)
        assert not unexpected, unexpected
        covered = [k for k in sd if mapping[k] in ssd]
        out[tag] = dict(sample_rate=sr, nb=nb,
                        keys=[[k, list(v.shape), mapping[k], bool(mapping[k] in ssd)] for k, v in sd.items()])
        print(f"[golden] checkpoint keys {tag}: {len(sd)} offline keys, {len(covered)} carried into the blob, "
              f"{sum(1 for k in sd if mapping[k] is None)} dropped by correct_state_dict")
    sys.path.remove(str(REF / "model"))
    (OUT / "checkpoint_keys.json").write_text(json.dumps(out, indent=0))


def make_einsum_fixture():
    """N1: the grouped linears of a checkpoint in the reference's EINSUM storage.  Our seeded weights are loaded into the
    reference's streaming module, the module is converted by the reference's own convert_grouped_linear_to_einsum
    (onnx_model/layers.py:1053-1080) and the converted module's state_dict entries of every grouped linear are recorded
    ([G, Ig, Og] weights + merged biases: data only).  tests/test_checkpoint_keys.py feeds them to pack_state_dict and
    expects the original blob back, and checks that the converted module still computes the same frame."""
    import torch
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, synth_blob
    from onnx_model.layers import convert_grouped_linear_to_einsum
    out = {}
    for tag, sr, nb in (("16k", 16000, 0), ("48k", 48000, 1)):
        entries = parse_manifest_text(orc.manifest_text(sr, nb))
        blob = synth_blob(entries, SEED + 700)
        model = build_reference_model(sr, nb, blob, entries)
        before = {k for k in model.state_dict()}
        spec = (np.random.default_rng(SEED + 701).standard_normal((1, 1, model.stft.win_len // 2 + 1, 2)) * 3.0).astype(np.float32)
        with torch.no_grad():
            y0, _ = model(torch.from_numpy(spec), model.initial_state(dtype=torch.float32))
        convert_grouped_linear_to_einsum(model)
        with torch.no_grad():
            y1, _ = model(torch.from_numpy(spec), model.initial_state(dtype=torch.float32))
        assert float((y0 - y1).abs().max()) < 1e-5
        sd = model.state_dict()
        new = [k for k in sd if k not in before]
        assert new and all(sd[k].dim() in (1, 3) for k in new), new
        for k in new:
            out[f"{tag}:{k}"] = sd[k].numpy().copy()
        out[f"{tag}:meta_json"] = np.frombuffer(json.dumps(dict(sample_rate=sr, nb=nb, seed=SEED + 700, keys=new)).encode(), dtype=np.uint8)
        print(f"[golden] einsum {tag}: {len(new)} converted tensors, e.g. {new[0]} {tuple(sd[new[0]].shape)}")
    np.savez_compressed(OUT / "einsum_checkpoint.npz", **out)


def make_eval_fixture():
    """N4: known answers of the reference's SI-SNR and cross-correlation alignment (pesq_stoi_sisnr_calc.py:16-27, 101-146)."""
    for name in ("pystoi", "pystoi.stoi", "pesq", "pandas"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sys.modules["pystoi.stoi"].stoi = getattr(sys.modules["pystoi.stoi"], "stoi", None)
    sys.modules["pesq"].pesq = getattr(sys.modules["pesq"], "pesq", None)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_eval_calc", REF / "pesq_stoi_sisnr_calc.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(SEED + 31)
    clean = synth_clip(6000, 16000, SEED + 32)
    noisy = (clean + 0.03 * rng.standard_normal(6000)).astype(np.float32)
    scaled = (0.37 * clean + 0.01).astype(np.float32)
    out = dict(clean=clean, noisy=noisy, scaled=scaled,
               sisnr_noisy=np.float64(mod.si_snr(clean, noisy)), sisnr_rev=np.float64(mod.si_snr(noisy, clean)),
               sisnr_scaled=np.float64(mod.si_snr(clean, scaled)), sisnr_self=np.float64(mod.si_snr(clean, clean)))
    for i, (shift, n_b) in enumerate(((37, 6000), (-113, 5200), (0, 6000), (250, 4000))):
        b = np.zeros(n_b, dtype=np.float32)
        src = noisy
        if shift >= 0:
            m = min(n_b - shift, len(src)); b[shift:shift + m] = src[:m]
        else:
            m = min(n_b, len(src) + shift); b[:m] = src[-shift:-shift + m]
        a_al, b_al, lag = mod.align_by_xcorr_trim(clean, b)
        out[f"xc{i}_b"] = b; out[f"xc{i}_a_al"] = a_al; out[f"xc{i}_b_al"] = b_al; out[f"xc{i}_lag"] = np.int32(lag)
        a_al, b_al, lag = mod.align_by_xcorr_trim(b, clean)
        out[f"xc{i}_rev_lag"] = np.int32(lag); out[f"xc{i}_rev_len"] = np.int32(len(a_al))
    np.savez_compressed(OUT / "evalkit.npz", **out)
    print("[golden] evalkit.npz", {k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items() if k.startswith(("sisnr", "xc0"))})


def make_stress_fixtures():
    """Weight-robustness goldens (round-3 review item 5): the same probes and waveforms on weights in the corners."""
    make_model_fixture("16k_nb2_hot", 16000, 2, 0.5, SEED + 202, stress="hot")
    make_model_fixture("16k_nb2_stiff", 16000, 2, 0.5, SEED + 203, stress="stiff")
    make_model_fixture("48k_nb1_hot", 48000, 1, 0.6, SEED + 248, stress="hot")
    make_model_fixture("48k_nb1_stiff", 48000, 1, 0.6, SEED + 249, stress="stiff")


# ---------------------------------------------------------------------------------------------------------------------------
# Spectrally SPARSE inputs at 48 kHz (round-4 review, weak #1).  The 48 kHz feature path is 10 log10(|X| + 1e-10) PER BIN
# (onnx_model/dpdfnet_48khz_hr.py:887-924, MagNorm48 onnx_model/layers.py:575-661): a bin that holds only the rounding noise of
# the analysis transform becomes a -60 .. -100 dB feature that differs between STFT implementations.  The reference itself has
# two analyses on its path -- an fp32 FFT (torch.stft here; librosa -> scipy.fft single precision in the package's offline
# path) and, under its pinned numpy 1.26.4, a float64 FFT of the float32 windowed frame (np.fft.rfft in stream.py:119-126).
# Each class below is run through the reference frame function fed BOTH ways and both enhanced waveforms are stored, so the
# fixture carries the reference's own spread next to the float64 variant the oracle restates.
SPARSE_CLASSES = ("bl_f32", "bl_i16", "dc", "square", "sil_sig")


def sparse_clip(cls: str, n: int, sr: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    def band_limited(m):
        spec = np.fft.rfft(rng.standard_normal(m))
        f = np.fft.rfftfreq(m, 1.0 / sr)
        spec[f > 6000.0] = 0.0                                          # nothing above 6 kHz: 3/4 of the 481 bins hold no signal
        x = np.fft.irfft(spec, m)
        t = np.arange(m) / sr
        x = x / np.sqrt(np.mean(x ** 2)) * 0.14 * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t))
        return np.clip(x, -1.0, 1.0)
    if cls == "bl_f32":
        return band_limited(n).astype(np.float32)
    if cls == "bl_i16":                                                 # the same audio as a 16-bit PCM file would deliver it
        return (np.round(band_limited(n) * 32768.0) / 32768.0).astype(np.float32)
    if cls == "dc":
        return np.full(n, 0.5, dtype=np.float32)
    if cls == "square":                                                 # +-1, period 100 samples (480 Hz)
        return np.where((np.arange(n) // 50) % 2 == 0, 1.0, -1.0).astype(np.float32)
    if cls == "sil_sig":                                                # 0.25 s of exact zeros, then the band-limited signal
        x = band_limited(n)
        x[: int(0.25 * sr)] = 0.0
        return x.astype(np.float32)
    raise ValueError(cls)


def _f64_stft_center(wav: np.ndarray, win: int, hop: int, window: np.ndarray) -> np.ndarray:
    """preprocess_waveform's framing (tail pad win, centre / reflect pad: package/src/dpdfnet/audio.py:104-117, api.py:88) with
    the transform of stream.py:119-126 under numpy 1.26.4: float32 product frame * window, float64 rfft, result cast to f32."""
    x = np.pad(wav.astype(np.float32), (0, win))
    xp = np.pad(x, (win // 2, win // 2), mode="reflect")
    T = 1 + len(x) // hop
    out = np.zeros((T, win // 2 + 1, 2), dtype=np.float32)
    for t in range(T):
        fr = (xp[t * hop: t * hop + win] * window).astype(np.float32)
        c = np.fft.rfft(fr.astype(np.float64), n=win)
        out[t, :, 0] = c.real.astype(np.float32); out[t, :, 1] = c.imag.astype(np.float32)
    return out


def make_sparse_model_fixture(tag: str, nb: int, seconds: float, seed: int):
    import torch
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, synth_blob

    sr = 48000
    entries = parse_manifest_text(orc.manifest_text(sr, nb))
    blob = synth_blob(entries, seed)
    model = build_reference_model(sr, nb, blob, entries)
    win = model.stft.win_len
    window = model.stft.w.numpy().astype(np.float32)
    n = int(seconds * sr)

    def synth(spec_e):
        with torch.no_grad():
            c = torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(spec_e)))[None]
            audio = model.istft(c.transpose(1, 2))
            audio = torch.nn.functional.pad(audio[:, win * 2:], (0, win * 2))
        a = audio[0].numpy()
        out = np.zeros(n, dtype=np.float32)
        m = min(n, a.shape[0]); out[:m] = a[:m]
        return out

    fix, report = {}, {}
    for i, cls in enumerate(SPARSE_CLASSES):
        wav = sparse_clip(cls, n, sr, seed + 10 + i)
        with torch.no_grad():
            audio_pad = torch.nn.functional.pad(torch.from_numpy(wav)[None], (0, win))
            spec_t = torch.view_as_real(model.stft(audio_pad).transpose(1, 2)[0]).numpy().astype(np.float32)
        spec_d = _f64_stft_center(wav, win, win // 2, window)
        g_t = run_reference_frames(model, spec_t, probe_frames=())
        g_d = run_reference_frames(model, spec_d, probe_frames=())
        e_t, e_d = synth(g_t["spec_e"]), synth(g_d["spec_e"])
        fix[f"wav_{cls}"] = wav
        fix[f"enh_torch_{cls}"] = e_t
        fix[f"enh_f64_{cls}"] = e_d
        fix[f"spec_f64_head_{cls}"] = spec_d[28:32].copy()             # pins the oracle's analysis on these inputs
        fix[f"spec_e_f64_head_{cls}"] = g_d["spec_e"][28:40].copy()
        fix[f"state_out_f64_{cls}"] = g_d["state_out"]
        rms = lambda v: float(np.sqrt(np.mean(np.square(v, dtype=np.float64))))
        report[cls] = dict(rms_in=rms(wav), rms_out=rms(e_d), torch_vs_f64=rms(e_t - e_d))
    meta = dict(tag=tag, sample_rate=sr, nb=nb, seed=seed, n=n, classes=list(SPARSE_CLASSES), spread=report)
    fix["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"model_{tag}.npz", **fix)
    print(f"[golden] model_{tag}.npz", {k: f"{v['torch_vs_f64']:.2e}" for k, v in report.items()})
    return model


def make_sparse_stream_fixture(model, tag: str, seed: int):
    """The reference's StreamEnhancer (package/src/dpdfnet/stream.py:74-200) on sparse inputs, with np.fft.rfft as numpy 1.26.4
    (the reference's pin, requirements.txt:4) computes it -- float64 -- and as this container's numpy 2.x does (float32)."""
    import torch
    import dpdfnet.stream as ref_stream
    from dpdfnet.models import ModelInfo, ResolvedModel

    sr, win = 48000, 960
    F = win // 2 + 1

    class _In:
        def __init__(self, name, shape): self.name, self.shape = name, shape

    class _Session:
        def get_inputs(self): return [_In("spec", [1, 1, F, 2]), _In("state_in", [model.state_size()])]
        def get_outputs(self): return [_In("spec_e", None), _In("state_out", None)]
        def run(self, _names, feeds):
            with torch.no_grad():
                y, st = model(torch.from_numpy(np.ascontiguousarray(feeds["spec"])) * wnorm, torch.from_numpy(feeds["state_in"].copy()))
            return [(y * inv).numpy(), st.numpy()]

    wnorm = torch.tensor(np.float32(model.wnorm)); inv = torch.tensor(np.float32(1.0 / float(model.wnorm)))

    class _FFT64:                                   # numpy 1.x: rfft of a float32 array is computed in double precision
        @staticmethod
        def rfft(x, n=None): return np.fft.rfft(np.asarray(x, dtype=np.float64), n=n)
        @staticmethod
        def irfft(x, n=None): return np.fft.irfft(np.asarray(x, dtype=np.complex128), n=n)

    class _NP64:
        fft = _FFT64
        def __getattr__(self, k): return getattr(np, k)

    def make(f64: bool):
        init = model.initial_state(dtype=torch.float32).numpy()
        rt = ref_stream.RuntimeModel(session=_Session(), init_state=init, in_spec_name="spec",
                                     in_state_name="state_in", out_spec_name="spec_e", out_state_name="state_out")
        ref_stream.resolve_model = lambda **kw: ResolvedModel(
            info=ModelInfo(name="x", sample_rate=sr, frame_ms=20.0, description="", onnx_filename="x.onnx"),
            onnx_path=Path("/dev/null"))
        ref_stream.build_runtime_model = lambda _p: rt
        ref_stream.infer_win_len = lambda _s, _sr: win
        ref_stream.np = _NP64() if f64 else np
        return ref_stream.StreamEnhancer(model="x")

    n = int(0.4 * sr) + 37
    out = {}
    try:
        for i, cls in enumerate(("bl_f32", "bl_i16", "sil_sig")):
            wav = sparse_clip(cls, n, sr, seed + 20 + i)
            out[f"wav_{cls}"] = wav
            for name, f64 in (("f64", True), ("f32", False)):
                for chunk in (win // 2, 171):
                    if name == "f32" and chunk != win // 2:
                        continue
                    se = make(f64)
                    pieces = [se.process(wav[j:j + chunk]) for j in range(0, n, chunk)]
                    pieces.append(se.flush())
                    out[f"{name}_{cls}_chunk{chunk}"] = np.concatenate(pieces).astype(np.float32)
    finally:
        ref_stream.np = np
    np.savez_compressed(OUT / f"stream_{tag}.npz", **out)
    rms = lambda v: float(np.sqrt(np.mean(np.square(v, dtype=np.float64))))
    print(f"[golden] stream_{tag}.npz", {c: f"{rms(out[f'f64_{c}_chunk480'] - out[f'f32_{c}_chunk480']):.2e}" for c in ("bl_f32", "bl_i16", "sil_sig")})


def make_sparse_fixtures():
    m1 = make_sparse_model_fixture("48k_nb1_sparse", 1, 0.6, SEED + 301)
    m8 = make_sparse_model_fixture("48k_nb8_sparse", 8, 0.6, SEED + 308)
    make_sparse_stream_fixture(m8, "48k_nb8_sparse", SEED + 318)


NONFINITE_CLASSES = ("nan_sample", "pos_inf_sample", "neg_inf_sample", "denormal_noise", "nan_first_sample")


def nonfinite_clip(cls: str, n: int, sr: int, seed: int) -> np.ndarray:
    """A clean synthetic clip with ONE non-finite sample (or denormal-only content): what the reference's frame function does with it
    is the contract -- a NaN / Inf reaches the EMA norm states and every recurrent state of that clip and never leaves them."""
    wav = synth_clip(n, sr, seed)
    k = int(0.35 * n)
    if cls == "nan_sample":
        wav[k] = np.nan
    elif cls == "pos_inf_sample":
        wav[k] = np.inf
    elif cls == "neg_inf_sample":
        wav[k] = -np.inf
    elif cls == "nan_first_sample":
        wav[0] = np.nan
    elif cls == "denormal_noise":
        rng = np.random.default_rng(seed + 5)
        wav = (1e-40 * rng.standard_normal(n)).astype(np.float32)
    else:
        raise ValueError(cls)
    return wav


def make_nonfinite_fixture(tag: str, sr: int, nb: int, seconds: float, seed: int):
    """Round 6: NaN / +-Inf / denormal inputs through the reference's frame function (onnx_model/dpdfnet.py:748-852, 48 kHz twin
    dpdfnet_48khz_hr.py:820-924) with torch's own analysis and synthesis, as make_model_fixture does for clean clips.  Stored per class:
    the input, the enhanced waveform (NaNs included), the index of the first non-finite output sample and the non-finite mask of the
    final state vector."""
    import torch
    from oracle import oracle as orc
    from dpdfnet_amd.weights import parse_manifest_text, synth_blob

    entries = parse_manifest_text(orc.manifest_text(sr, nb))
    blob = synth_blob(entries, seed)
    model = build_reference_model(sr, nb, blob, entries)
    n = int(seconds * sr)
    win = model.stft.win_len
    fix = {}
    for i, cls in enumerate(NONFINITE_CLASSES):
        wav = nonfinite_clip(cls, n, sr, seed + 30 + i)
        with torch.no_grad():
            audio_pad = torch.nn.functional.pad(torch.from_numpy(wav)[None], (0, win))
            spec_c = model.stft(audio_pad).transpose(1, 2)[0]
        spec = torch.view_as_real(spec_c).numpy().astype(np.float32)
        g = run_reference_frames(model, spec, probe_frames=())
        with torch.no_grad():
            c = torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(g["spec_e"])))[None]
            audio = model.istft(c.transpose(1, 2))
            audio = torch.nn.functional.pad(audio[:, win * 2:], (0, win * 2))
        a = audio[0].numpy()
        out = np.zeros(n, dtype=np.float32)
        m = min(n, a.shape[0])
        out[:m] = a[:m]
        bad = np.nonzero(~np.isfinite(out))[0]
        fix[f"{cls}_wav"] = wav
        fix[f"{cls}_enhanced"] = out
        fix[f"{cls}_first_bad"] = np.int64(bad[0] if len(bad) else -1)
        fix[f"{cls}_state_bad"] = ~np.isfinite(g["state_out"])
        fr = np.nonzero(~np.isfinite(g["spec_e"]).all(axis=(1, 2)))[0]
        print(f"[golden] nonfinite_{tag} {cls}: first bad output sample {fix[f'{cls}_first_bad']} of {n}, "
              f"{int((~np.isfinite(out)).sum())} bad samples, first bad enhanced frame {fr[0] if len(fr) else -1} of {spec.shape[0]}, "
              f"{int(fix[f'{cls}_state_bad'].sum())} of {g['state_out'].size} state entries non-finite")
    meta = dict(tag=tag, sample_rate=sr, nb=nb, seed=seed, n=n, classes=list(NONFINITE_CLASSES))
    fix["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / f"nonfinite_{tag}.npz", **fix)


def make_nonfinite_fixtures():
    make_nonfinite_fixture("16k_nb2", 16000, 2, 0.5, SEED + 402)
    make_nonfinite_fixture("48k_nb8", 48000, 8, 0.3, SEED + 408)


def main():
    assert REF.is_dir(), "reference checkout not mounted; goldens can only be regenerated in the build container"
    _stub_modules()
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "package" / "src"))
    import torch
    torch.set_num_threads(1)
    torch.manual_seed(0)
    if "--stress-only" in sys.argv:          # the round-3 additions alone (the other fixtures regenerate bit-for-bit anyway)
        make_stress_fixtures()
        make_einsum_fixture()
        return
    if "--sparse-only" in sys.argv:          # the round-5 additions alone
        make_sparse_fixtures()
        return
    if "--nonfinite-only" in sys.argv:       # the round-6 additions alone
        make_nonfinite_fixtures()
        return

    make_constants_fixture()
    make_host_dsp_fixture()
    make_eval_fixture()
    make_model_fixture("16k_nb0", 16000, 0, 0.6, SEED + 100)
    make_model_fixture("16k_nb1", 16000, 1, 1.0, SEED + 101)
    m2, _, _ = make_model_fixture("16k_nb2", 16000, 2, 1.0, SEED + 102)
    make_stream_fixture(m2, 16000, "16k_nb2")
    make_model_fixture("16k_nb4", 16000, 4, 2.0, SEED + 104)
    make_model_fixture("16k_nb8", 16000, 8, 0.7, SEED + 108)        # BASELINE config 4 (deep DPRNN stack)
    m48, _, _ = make_model_fixture("48k_nb1", 48000, 1, 0.6, SEED + 148)
    make_stream_fixture(m48, 48000, "48k_nb1")
    make_model_fixture("48k_nb2", 48000, 2, 0.6, SEED + 150)        # dpdfnet2_48khz_hr
    m488, _, _ = make_model_fixture("48k_nb8", 48000, 8, 0.6, SEED + 152)   # dpdfnet8_48khz_hr (BASELINE config 5)
    make_stream_fixture(m488, 48000, "48k_nb8")
    make_checkpoint_keys_fixture()
    make_stress_fixtures()
    make_einsum_fixture()
    make_sparse_fixtures()
    make_nonfinite_fixtures()


if __name__ == "__main__":
    main()
