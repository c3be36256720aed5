"""`.pth` loading for the HIP modules.

`runner.py --test --weights=<file>` (/root/reference/README.md:91) hands a torch checkpoint to the model.
The released checkpoints (README.md:35-36) and the reference's state_dict key names are NOT in the mount
(SURVEY.md §0), so compatibility with them cannot be claimed or tested here.  What this module provides
is the mechanism a maintainer needs once the keys are known:

  * container unwrapping: a bare state_dict, or a dict holding one under a usual key
    ("state_dict", "model", "net", "network" ...), with or without DataParallel's "module." prefix;
  * a key map (JSON, reference key -> build key; `keymap.json` beside this file, shipped EMPTY) applied
    before load_state_dict — filling that file in is the only step needed to adopt reference weights
    whose tensors have this build's shapes;
  * strict shape checking with a readable report of what did not match;
  * `suggest_keymap`: when a checkpoint's tensors have this build's shapes in this build's order under other names
    (the usual situation after a refactor), the key map is derived by walking both state_dicts in order.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

KEYMAP_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "keymap.json")
_CONTAINER_KEYS = ("state_dict", "model_state_dict", "model", "net", "network", "weights")


def load_keymap(path: Optional[str] = None) -> Dict[str, str]:
    path = path or KEYMAP_PATH
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        m = json.load(f)
    return {k: v for k, v in m.items() if not k.startswith("_")}


def unwrap(obj) -> Dict[str, torch.Tensor]:
    """Find the state_dict inside whatever torch.load returned."""
    if isinstance(obj, torch.nn.Module):
        obj = obj.state_dict()
    if not isinstance(obj, dict):
        raise TypeError(f"checkpoint holds a {type(obj).__name__}, not a state_dict")
    if obj and all(isinstance(v, torch.Tensor) for v in obj.values()):
        sd = obj
    else:
        for k in _CONTAINER_KEYS:
            if k in obj and isinstance(obj[k], dict):
                sd = unwrap(obj[k])
                break
        else:
            raise KeyError(f"no state_dict found under any of {_CONTAINER_KEYS}; top-level keys: {list(obj)[:8]}")
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def remap(sd: Dict[str, torch.Tensor], keymap: Dict[str, str]) -> Dict[str, torch.Tensor]:
    """Rename keys.  A map entry may name a full key or a prefix ending in '.' (longest prefix wins)."""
    if not keymap:
        return dict(sd)
    prefixes = sorted((k for k in keymap if k.endswith(".")), key=len, reverse=True)
    out = {}
    for k, v in sd.items():
        if k in keymap:
            nk = keymap[k]
        else:
            nk = k
            for p in prefixes:
                if k.startswith(p):
                    nk = keymap[p] + k[len(p):]
                    break
        if nk in out:
            raise KeyError(f"key map sends two checkpoint keys to {nk!r}")
        out[nk] = v
    return out


def suggest_keymap(sd: Dict[str, torch.Tensor], model: torch.nn.Module) -> Dict[str, str]:
    """Checkpoint key -> model key by ORDER and SHAPE: both state_dicts are walked in their own order (BatchNorm's
    `num_batches_tracked` counters are skipped on both sides) and paired while every pair has equal shapes.  Raises with
    the first mismatch otherwise — a map is only ever suggested when it is unambiguous in this sense.  Keys that already
    agree are left out of the result, so an identical naming yields {}."""
    def items(d, counters):
        return [(k, tuple(v.shape)) for k, v in d.items() if k.endswith("num_batches_tracked") == counters]
    src, own = unwrap(sd), model.state_dict()
    a, b = items(src, False), items(own, False)
    if len(a) != len(b):
        raise ValueError(f"checkpoint has {len(a)} tensors, the model {len(b)}: no order-based map exists")
    out = {}
    for (ka, sa), (kb, sb) in zip(a, b):
        if sa != sb:
            raise ValueError(f"order-based pairing breaks at {ka!r} {sa} vs {kb!r} {sb}")
        if ka != kb:
            out[ka] = kb
    # the BatchNorm step counters ride along when both sides carry one per BatchNorm (a checkpoint written by an old
    # torch has none: load_checkpoint does not miss them)
    ca, cb = items(src, True), items(own, True)
    if len(ca) == len(cb):
        out.update({ka: kb for (ka, _), (kb, _) in zip(ca, cb) if ka != kb})
    return out


def load_checkpoint(model: torch.nn.Module, path: str, keymap: Optional[Dict[str, str]] = None, strict: bool = True):
    """torch.load(path) -> unwrap -> key map -> model.load_state_dict.  Returns (missing, unexpected)."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    sd = remap(unwrap(obj), load_keymap() if keymap is None else keymap)
    own = model.state_dict()
    bad = [f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}"
           for k, v in sd.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError("checkpoint tensors do not have this build's shapes (arch_spec is build-specified, "
                           "SURVEY.md §0):\n  " + "\n  ".join(bad[:12]))
    res = model.load_state_dict(sd, strict=False)
    # BatchNorm's num_batches_tracked counters play no part in an eval-mode forward: neither a checkpoint without
    # them nor one whose counters kept their foreign names is a mismatch
    missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")]
    unexpected = [k for k in res.unexpected_keys if not k.endswith("num_batches_tracked")]
    if strict and (missing or unexpected):
        raise RuntimeError(f"state_dict mismatch: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected {unexpected[:8]}; extend {KEYMAP_PATH}")
    return missing, unexpected
