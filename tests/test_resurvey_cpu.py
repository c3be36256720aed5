"""tools/resurvey.py (SURVEY.md §9 as a script) end to end against a STAND-IN module tree: the real reference branches
are not mounted (/root/reference/README.md:5), so the tool is exercised on a tiny tree laid out the way README.md
describes the reference (runner.py, config.py using easydict, models/*.py) — written here by the test, sharing nothing
with the reference but its file names."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIG = '''
from easydict import EasyDict as edict     # not installed in this image: resurvey installs a stand-in
import cv2                                 # idem (imported, never called)
__C = edict()
cfg = __C
__C.DATASETS = edict()
__C.DATASETS.SHAPENET = edict()
__C.DATASETS.SHAPENET.VOLUME_PATH = '/path/to/ShapeNetVox32/%s/%s.mat'
__C.NETWORK = edict()
__C.NETWORK.LEAKY_VALUE = 0.2
'''

MODEL = '''
import torch
import torchvision.models          # a stand-in here; a real reference may take a backbone from it


class Encoder(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer1 = torch.nn.Sequential(torch.nn.Conv2d(3, 32, 3, 2, 1), torch.nn.BatchNorm2d(32),
                                          torch.nn.LeakyReLU(cfg.NETWORK.LEAKY_VALUE))
        self.layer2 = torch.nn.Sequential(torch.nn.Conv2d(32, 48, 5, 4, 2), torch.nn.BatchNorm2d(48), torch.nn.ReLU())

    def forward(self, left, right):
        return self.layer2(self.layer1(left)) - self.layer2(self.layer1(right))


class Head(torch.nn.Module):          # a sub-network whose input is not a render: static table only
    def __init__(self, cfg):
        super().__init__()
        self.fc = torch.nn.Linear(48, 7, bias=False)

    def forward(self, feats, unused, also_unused):
        return self.fc(feats.mean((2, 3)))
'''


def test_resurvey_on_a_stand_in_tree(tmp_path):
    ref = tmp_path / "Stereo2Voxel"
    (ref / "models").mkdir(parents=True)
    (ref / "config.py").write_text(CONFIG)
    (ref / "runner.py").write_text("raise SystemExit('runner.py must not be imported by the survey')\n")
    (ref / "models" / "__init__.py").write_text("")
    (ref / "models" / "net.py").write_text(MODEL)
    out = tmp_path / "golden"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resurvey.py"), "--ref", str(ref), "--out", str(out)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    text = r.stdout
    assert "stand-ins installed for:" in text and "easydict" in text
    assert "models.net:Encoder" in text and "models.net:Head" in text
    # layer table in execution order with output shapes at 224x224
    assert "layer1.0" in text and "Conv2d" in text and "kernel_size=3 stride=2 padding=1" in text and "(2, 32, 112, 112)" in text
    assert "kernel_size=5 stride=4 padding=2" in text and "(2, 48, 28, 28)" in text
    # state_dict keys with shapes; the diff against arch_spec names what differs
    assert "layer2.0.weight" in text and "[48, 32, 5, 5]" in text
    assert "diff against arch_spec (voxel)" in text and "cout 64 (ref 48)" in text and "k 3 (ref 5)" in text
    assert "(no reference layer)" in text                       # arch_spec rows the stand-in does not have
    assert "not traceable" in text                              # Head: static table, no golden
    summary = json.loads(text.strip().splitlines()[-1].split("summary: ", 1)[1])
    by = {s["class"].split(":")[-1]: s for s in summary}
    assert by["Encoder"]["traced"] and by["Encoder"]["golden"] and not by["Head"]["traced"]
    z = np.load(out / "ref_Encoder.npz")
    assert z["output0"].shape == (2, 48, 28, 28) and np.isfinite(z["output0"]).all()
    assert json.loads(str(z["weights"])) == {"seed": 0}
    keys = json.loads((out / "ref_Encoder_keys.json").read_text())["keys"]
    assert keys[0] == ["layer1.0.weight", [32, 3, 3, 3]]


def test_resurvey_refuses_the_mount_as_it_is():
    """Today's /root/reference (README.md + requirements.txt) has nothing to survey: the tool says so."""
    if not os.path.isdir("/root/reference"):
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resurvey.py"), "--ref", "/root/reference"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "holds no Python source" in (r.stderr + r.stdout)
