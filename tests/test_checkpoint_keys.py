"""N1: checkpoint key handling pinned to key names CAPTURED from the reference's offline twins
(tests/golden/checkpoint_keys.json, written by make_golden.py from model/dpdfnet.py and
model/dpdfnet_48khz_hr.py state_dicts and the reference's own correct_state_dict,
onnx_model/dpdfnet.py:876-888, onnx_model/dpdfnet_48khz_hr.py:948-963)."""
import json

import numpy as np
import pytest

from tests.util import GOLDEN, golden_blob, load_golden, rms

KEYS = json.loads((GOLDEN / "checkpoint_keys.json").read_text())


def offline_state_dict(tag: str, blob: np.ndarray, entries):
    """A checkpoint as a user would hold it: every key of the offline twin (incl. buffers, num_batches_tracked, the
    lsnr head and mask.erb_inv_fb), values of the carried tensors from `blob`."""
    from dpdfnet_amd import weights
    ssd = weights.unpack_to_streaming_state_dict(entries, blob)
    sd = {}
    for k, shape, sk, carried in KEYS[tag]["keys"]:
        sd[k] = ssd[sk].reshape(shape) if carried else np.full(shape, 0.5, dtype=np.float32)
    return sd


@pytest.mark.parametrize("tag", ["16k", "48k"])
def test_streaming_key_equals_reference_renaming(tag):
    from dpdfnet_amd import backend, weights
    info = KEYS[tag]
    ents = backend.manifest(info["sample_rate"], info["nb"])
    names = {e.name for e in ents}
    gl = {e.name[:-len(".weight")] for e in ents if len(e.shape) == 3}
    carried = 0
    for k, shape, sk, is_carried in info["keys"]:
        if sk is None:                       # dropped by the reference (48 kHz mask.erb_inv_fb)
            assert k == "mask.erb_inv_fb"
            continue
        assert weights.streaming_key(k) == sk, k
        if is_carried:
            carried += 1
            base = sk.split(".layers.")[0]
            assert sk in names or base in gl, sk
    # every manifest tensor is fed by the checkpoint: plain tensors 1:1, grouped linears G weights + G biases each
    n_plain = sum(1 for e in ents if e.name.rsplit(".", 1)[0] not in gl)
    n_gl = sum(2 * e.shape[0] for e in ents if len(e.shape) == 3)
    assert carried == n_plain + n_gl


@pytest.mark.parametrize("tag", ["16k", "48k"])
def test_offline_checkpoint_packs_to_the_same_blob(tag, tmp_path, monkeypatch):
    import torch
    from dpdfnet_amd import backend, weights
    info = KEYS[tag]
    ents = backend.manifest(info["sample_rate"], info["nb"])
    blob = weights.synth_blob(ents, 4242)
    sd = offline_state_dict(tag, blob, ents)
    np.testing.assert_array_equal(weights.pack_state_dict(ents, sd), blob)
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    for name, obj in (("plain.pth", tsd), ("wrapped.ckpt", {"state_dict": tsd, "epoch": 3})):
        torch.save(obj, tmp_path / name)
        b2, extras = weights.load_weight_file(tmp_path / name, ents)
        np.testing.assert_array_equal(b2, blob)
        assert extras == {}
    # a pickle that is not a plain state_dict is refused unless explicitly allowed (no silent unsafe fallback)
    class Evil:
        def __reduce__(self):
            return (print, ("unsafe pickle executed",))
    torch.save({"state_dict": tsd, "hook": Evil()}, tmp_path / "evil.pth")
    monkeypatch.delenv("DPDFNET_ALLOW_UNSAFE_PICKLE", raising=False)
    with pytest.raises(ValueError, match="DPDFNET_ALLOW_UNSAFE_PICKLE"):
        weights.load_weight_file(tmp_path / "evil.pth", ents)
    sd.pop(next(k for k, _s, sk, c in info["keys"] if c))
    with pytest.raises(KeyError):
        weights.pack_state_dict(ents, sd)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["16k", "48k"])
def test_offline_pth_through_hip_matches_reference_waveform(tag, tmp_path):
    """load_weight_file(.pth in offline-twin naming) -> HIP engine -> the reference's golden waveform."""
    import torch
    from dpdfnet_amd import backend, weights
    g, meta = load_golden(f"{tag}_nb1")
    ents = backend.manifest(meta["sample_rate"], meta["nb"])
    sd = offline_state_dict(tag, golden_blob(meta), ents)
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, tmp_path / "ckpt.pth")
    blob, extras = weights.load_weight_file(tmp_path / "ckpt.pth", ents)
    m = backend.HipModel(meta["sample_rate"], meta["nb"], blob, 0, extras.get("erb_norm_init"), extras.get("spec_norm_init"))
    try:
        out = m.enhance_batch(g["wav"][None])[0]
        assert rms(out - g["enhanced"]) < 2e-6
    finally:
        m.close()


@pytest.mark.parametrize("tag", ["16k", "48k"])
def test_reference_einsum_conversion_packs_to_the_same_blob(tag):
    """Grouped linears in the reference's EINSUM storage: tests/golden/einsum_checkpoint.npz holds the tensors the
    reference's own convert_grouped_linear_to_einsum (onnx_model/layers.py:1053-1080) produced from our seeded weights
    loaded into its streaming module.  pack_state_dict must give the original blob back -- and must check SHAPES: a tensor
    of the right size in the wrong layout is an error, not a silent reinterpretation."""
    from dpdfnet_amd import backend, weights
    E = np.load(GOLDEN / "einsum_checkpoint.npz")
    meta = json.loads(bytes(E[f"{tag}:meta_json"]).decode())
    ents = backend.manifest(meta["sample_rate"], meta["nb"])
    blob = weights.synth_blob(ents, meta["seed"])
    sd = {k: v for k, v in weights.unpack_to_streaming_state_dict(ents, blob).items() if ".layers." not in k}
    n3 = 0
    for k in meta["keys"]:
        sd[k] = E[f"{tag}:{k}"]
        if sd[k].ndim == 3:
            e = next(e for e in ents if e.name == k)
            assert sd[k].shape == (e.shape[0], e.shape[2], e.shape[1])        # [G, Ig, Og]: the transpose of the manifest's layout
            n3 += 1
    assert n3 == sum(1 for e in ents if len(e.shape) == 3)
    np.testing.assert_array_equal(weights.pack_state_dict(ents, sd), blob)
    # shape, not size: the same einsum tensor flattened, or with Ig and Og regrouped, is refused
    k3 = next(k for k in meta["keys"] if sd[k].ndim == 3)
    bad = dict(sd); bad[k3] = sd[k3].reshape(sd[k3].shape[0], -1, 2 * sd[k3].shape[2])
    with pytest.raises(ValueError, match="expected"):
        weights.pack_state_dict(ents, bad)
    k2 = next(e.name for e in ents if len(e.shape) == 2)
    bad = dict(sd); bad[k2] = np.ascontiguousarray(sd[k2].T) if sd[k2].shape[0] != sd[k2].shape[1] else sd[k2].reshape(-1)
    with pytest.raises(ValueError, match="expected"):
        weights.pack_state_dict(ents, bad)
