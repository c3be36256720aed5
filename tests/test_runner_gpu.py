"""runner.py --test end to end (the reference's eval entry point, /root/reference/README.md:88-92): a checkpoint in a
training-style container, an .npz eval list with ground-truth disparity, one JSON line out."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_runner_test_mode_matches_in_process_eval(tmp_path):
    import s3r
    left, right, gt = s3r.evaluate.synthetic_eval_set(3, 5)
    g = torch.Generator().manual_seed(2)
    dl, dr = torch.rand(3, 28, 28, generator=g) * 200, torch.rand(3, 28, 28, generator=g) * 200
    dl[0, :7] = float("inf")                                   # background, as the EXR maps mark it
    data = tmp_path / "eval.npz"
    np.savez(data, left=left.numpy(), right=right.numpy(), volume=gt.numpy(), disp_left=dl.numpy(), disp_right=dr.numpy())
    model = s3r.Stereo2Voxel()
    s3r.seed_module(model, 4)
    ckpt = tmp_path / "ckpt.pth"                               # DataParallel-style keys inside a container
    torch.save({"epoch": 7, "state_dict": {"module." + k: v for k, v in model.state_dict().items()}}, ckpt)

    r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py"), "--test", "--weights", str(ckpt), "--data", str(data),
                        "--batch", "2"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["samples"] == 3 and out["n_gpus"] == 1 and len(out["mean_iou"]) == len(out["thresholds"])

    model.to("cuda:0")
    want = s3r.evaluate.test_net(model, left, right, gt, batch=2, device="cuda:0")
    assert out["mean_iou"] == [round(x, 6) for x in want["mean_iou"]]
    wd = s3r.evaluate.test_disparity(model, left, right, dl, dr, batch=2, device="cuda:0")
    assert out["disparity_epe_left_px"] == round(wd["epe_left"], 4)
    assert out["disparity_epe_right_px"] == round(wd["epe_right"], 4)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_renamed_checkpoint_through_keymap_vs_oracle(tmp_path):
    """The `.pth` row against the ORACLE, not against the HIP path itself: a container with RENAMED keys (other
    prefixes, DataParallel's `module.`) is loaded (a) into the HIP model through `suggest_keymap` + `load_checkpoint`,
    as `runner.py --test --weights X --keymap K` does, and (b) into the oracle by plain tensor order; the two forwards
    must agree within the fp32 tolerance, and runner.py's IoU must be the oracle's IoU on the same data."""
    import s3r
    from oracle import s2v_oracle as O
    src = O.OracleStereo2Voxel().eval()
    s3r.seed_module(src, 21)
    own = src.state_dict()
    renamed = {}
    for k, v in own.items():                                   # a "reference-style" naming this build has never seen
        nk = k.replace("encoder.", "feature_net.").replace("decoder.", "recon3d.").replace(".conv.", ".c.").replace(".bn.", ".norm.")
        renamed["module." + nk] = v.clone()
    ckpt = tmp_path / "renamed.pth"
    torch.save({"epoch_idx": 250, "best_iou": 0.7, "network": renamed}, ckpt)

    hip = s3r.Stereo2Voxel()
    s3r.seed_module(hip, 99)                                   # different weights until the checkpoint is in
    keymap = s3r.checkpoint.suggest_keymap(torch.load(ckpt, map_location="cpu", weights_only=True), hip)
    assert keymap and all(k not in hip.state_dict() for k in keymap)
    km = tmp_path / "keymap.json"
    km.write_text(json.dumps(keymap))
    missing, unexpected = s3r.checkpoint.load_checkpoint(hip, str(ckpt), keymap)
    assert not missing and not unexpected
    hip.to("cuda:0")

    left, right, gt = s3r.evaluate.synthetic_eval_set(3, 8)
    with torch.no_grad():
        want = src(left, right)
    got = hip(left.to("cuda:0"), right.to("cuda:0")).cpu()
    assert ((got - want).norm() / want.norm()).item() < 1e-5   # north_star: 1e-4 relative
    assert (got - want).abs().max().item() < 1e-4

    data = tmp_path / "eval.npz"
    np.savez(data, left=left.numpy(), right=right.numpy(), volume=gt.numpy())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py"), "--test", "--weights", str(ckpt), "--keymap", str(km),
                        "--data", str(data), "--batch", "2"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for t, iou in zip(out["thresholds"], out["mean_iou"]):
        ref_iou = O.voxel_iou(want, gt, t).mean().item()
        assert abs(iou - ref_iou) < 1e-3, (t, iou, ref_iou)    # north_star: voxel IoU within 1e-3


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_runner_on_a_dataset_tree_with_disparity(tmp_path):
    import s3r
    from tests.test_data_cpu import _make_tree
    _make_tree(str(tmp_path), n_models=2, views=(0,), size=224)
    rng = np.random.default_rng(1)
    for m in range(2):
        rdir = os.path.join(str(tmp_path), "ShapeNetStereoRendering", "02691156", f"model{m:02d}")
        for side in "lr":
            s3r.exr.write_exr(os.path.join(rdir, "disp_00_%s.exr" % side),
                              {"Z": (rng.random((224, 224), dtype=np.float32) * 100)}, "ZIP")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py"), "--test", "--dataset-root", str(tmp_path), "--disparity",
                        "--batch", "2", "--seed", "3"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["samples"] == 2 and out["disparity_epe_left_px"] > 0 and out["disparity_epe_right_px"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_runner_point_variant_and_bf16_precision():
    for extra, key in ((["--variant", "point", "--samples", "3"], "mean_chamfer"),
                       (["--precision", "bf16", "--samples", "3"], "mean_iou")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py"), "--test", "--batch", "2"] + extra,
                           capture_output=True, text=True, timeout=240, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["samples"] == 3 and key in out and out["precision"] in ("fp32", "bf16")


def test_runner_refuses_what_it_does_not_implement():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py")], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "only --test is implemented" in (r.stderr + r.stdout)
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "runner.py"), "--test"], capture_output=True, text=True,
                           timeout=300, cwd=ROOT)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_runner_eight_ranks_share_one_gpu():
    """`runner.py --test --gpus 8` at the world size of BASELINE configs[4], eight ranks sharing cuda:0 over gloo (the RCCL run
    needs an 8-GPU node): the eval list — 19 samples, so shards of 3 and 2 — is sharded over the ranks, the per-sample metrics are
    all-gathered, and rank 0's line must carry exactly the single-process result on the same list."""
    base = [sys.executable, os.path.join(ROOT, "runner.py"), "--test", "--samples", "19", "--batch", "4", "--seed", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run(base, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    many = subprocess.run(base + ["--gpus", "8", "--backend", "gloo", "--same-device"], capture_output=True, text=True, timeout=580,
                          cwd=ROOT, env=env)
    assert many.returncode == 0, many.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in many.stdout.splitlines() if l.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 8 and b["n_ranks_seen"] == 8 and b["collective_backend"] == "gloo"
    assert a["samples"] == b["samples"] == 19
    assert a["mean_iou"] == b["mean_iou"], (a["mean_iou"], b["mean_iou"])
