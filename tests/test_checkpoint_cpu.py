"""Checkpoint container handling and the key-map hook (host logic only)."""
import pytest
import torch


def test_unwrap_containers_and_dataparallel_prefix(s3r):
    m = s3r.Stereo2Voxel()
    sd = s3r.seeded_state_dict(m, 3)
    ck = s3r.checkpoint
    assert ck.unwrap(sd).keys() == sd.keys()
    assert ck.unwrap({"epoch": 7, "state_dict": sd}).keys() == sd.keys()
    assert ck.unwrap({"model": {"module." + k: v for k, v in sd.items()}}).keys() == sd.keys()
    with pytest.raises(KeyError):
        ck.unwrap({"epoch": 1, "optimizer": {"lr": 0.1}})
    with pytest.raises(TypeError):
        ck.unwrap([1, 2, 3])


def test_keymap_prefix_and_full_key(s3r, tmp_path):
    m = s3r.Stereo2Voxel()
    sd = s3r.seeded_state_dict(m, 3)
    # a "reference-style" checkpoint: different prefixes, DataParallel-wrapped, inside a container
    foreign = {}
    for k, v in sd.items():
        nk = k.replace("encoder.e1.", "module.feat.stem.").replace("decoder.", "module.recnet.")
        foreign[nk if nk.startswith("module.") else "module." + nk] = v
    path = tmp_path / "ref.pth"
    torch.save({"epoch_idx": 150, "network": foreign}, path)
    with pytest.raises(RuntimeError, match="keymap"):
        s3r.checkpoint.load_checkpoint(s3r.Stereo2Voxel(), str(path), keymap={})
    keymap = {"feat.stem.": "encoder.e1.", "recnet.": "decoder."}
    m2 = s3r.Stereo2Voxel()
    missing, unexpected = s3r.checkpoint.load_checkpoint(m2, str(path), keymap=keymap)
    assert not missing and not unexpected
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_shape_mismatch_is_reported(s3r, tmp_path):
    m = s3r.Stereo2Voxel()
    sd = s3r.seeded_state_dict(m, 3)
    sd["encoder.e2.conv.weight"] = torch.zeros(64, 32, 5, 5)
    path = tmp_path / "bad.pth"
    torch.save(sd, path)
    with pytest.raises(RuntimeError, match="shapes"):
        s3r.checkpoint.load_checkpoint(s3r.Stereo2Voxel(), str(path), keymap={})


def test_shipped_keymap_is_empty_and_documented(s3r):
    assert s3r.checkpoint.load_keymap() == {}


def test_synthetic_eval_set_is_seeded(s3r):
    a = s3r.evaluate.synthetic_eval_set(3, 5)
    b = s3r.evaluate.synthetic_eval_set(3, 5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a[2].shape == (3, 32, 32, 32) and set(a[2].unique().tolist()) <= {0.0, 1.0} and a[2].sum() > 0


def test_suggest_keymap_from_order_and_shape(s3r, tmp_path):
    """A checkpoint with this build's tensors under other names: the map is derived, saved as JSON, and loads."""
    import json
    import torch
    model = s3r.Stereo2Voxel()
    s3r.seed_module(model, 3)
    renamed = {}
    for k, v in model.state_dict().items():
        nk = k.replace("encoder.", "feat_net.").replace("decoder.", "vol_net.").replace(".conv.", ".c.").replace(".bn.", ".norm.")
        renamed["module." + nk] = v.clone()
    km = s3r.checkpoint.suggest_keymap({"state_dict": renamed}, model)
    assert km["feat_net.e1.c.weight"] == "encoder.e1.conv.weight"
    # BatchNorm's step counters are paired in order as well (one per BatchNorm on both sides): the suggested map is complete
    assert km["feat_net.e1.norm.num_batches_tracked"] == "encoder.e1.bn.num_batches_tracked"
    assert s3r.checkpoint.suggest_keymap(model.state_dict(), model) == {}
    ck, kmf = tmp_path / "c.pth", tmp_path / "km.json"
    torch.save({"state_dict": renamed}, ck)
    kmf.write_text(json.dumps(km))
    fresh = s3r.Stereo2Voxel()
    missing, unexpected = s3r.checkpoint.load_checkpoint(fresh, str(ck), json.loads(kmf.read_text()))
    assert not missing and not unexpected
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))
    other = dict(renamed)
    first = next(iter(other))
    other[first] = torch.zeros(5)
    import pytest
    with pytest.raises(ValueError, match="pairing breaks"):
        s3r.checkpoint.suggest_keymap(other, model)


def test_renamed_batchnorm_counters_do_not_break_the_load(s3r, tmp_path):
    """A checkpoint whose every key is renamed — BatchNorm's num_batches_tracked included — loads through the map
    suggest_keymap derives (found by the GPU test of the `.pth` row against the oracle, round 2); counters that keep a
    foreign name, or are absent, are not a mismatch either (they play no part in an eval-mode forward)."""
    m = s3r.Stereo2Voxel()
    sd = s3r.seeded_state_dict(m, 3)
    foreign = {"module." + k.replace("encoder.", "feat.").replace(".bn.", ".norm."): v for k, v in sd.items()}
    assert any(k.endswith("norm.num_batches_tracked") for k in foreign)
    path = tmp_path / "renamed.pth"
    torch.save({"network": foreign}, path)
    keymap = s3r.checkpoint.suggest_keymap(torch.load(path, weights_only=True), m)
    assert keymap["feat.e1.norm.num_batches_tracked"] == "encoder.e1.bn.num_batches_tracked"
    m2 = s3r.Stereo2Voxel()
    assert s3r.checkpoint.load_checkpoint(m2, str(path), keymap) == ([], [])
    assert all(torch.equal(v, sd[k]) for k, v in m2.state_dict().items())
    weights_only = {k: v for k, v in keymap.items() if not k.endswith("num_batches_tracked")}
    m3 = s3r.Stereo2Voxel()
    assert s3r.checkpoint.load_checkpoint(m3, str(path), weights_only) == ([], [])      # counters left unmapped
    no_counters = {k: v for k, v in foreign.items() if not k.endswith("num_batches_tracked")}
    torch.save(no_counters, path)
    m4 = s3r.Stereo2Voxel()
    assert s3r.checkpoint.load_checkpoint(m4, str(path), weights_only) == ([], [])      # counters absent
