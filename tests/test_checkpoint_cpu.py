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
