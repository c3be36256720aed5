#!/usr/bin/env python3
"""tools/resurvey.py — the re-survey of SURVEY.md §9, as a script: run it the day the reference's `Stereo2Voxel` /
`Stereo2Point` branches are mounted (today /root/reference holds README.md + requirements.txt only, README.md:5).

    python tools/resurvey.py --ref /root/reference/Stereo2Voxel [--weights X.pth] [--out tests/golden]
                             [--entry models.stereo2voxel:Stereo2Voxel] [--seed 0]

It runs IN THE BUILD CONTAINER ONLY (the reference's Python is imported here and never copied: what it leaves behind
are data files) and, for every torch.nn.Module class the tree defines (or the one --entry names):

  1. prints the LAYER TABLE — every leaf module in execution order with op, Cin, Cout, kernel, stride, padding,
     bias / BatchNorm / activation, and (when a forward could be traced on 224x224 stereo input) its output shape;
  2. prints every `state_dict` key with its shape (and writes them to <out>/ref_<Class>_keys.json);
  3. prints a DIFF against this build's arch_spec: the conv / deconv / linear layers of both, paired in order, with
     every field that differs, and the parameter totals — the list of what arch_spec.py has to become;
  4. writes GOLDEN VECTORS <out>/ref_<Class>.npz: the module's outputs (eval mode, no_grad) on
     s3r.synthetic_pairs(2, seed) with weights from s3r.seed_module(module, seed) — that init is a function of the
     state_dict's key order and shapes only, so a HIP module with the same key layout regenerates the same weights on
     the GPU box from the seed alone; the file also records the key layout it was made for — or, with --weights, on
     the released checkpoint (then the npz records the checkpoint's sha256 instead of a seed).

Modules the image lacks (easydict, cv2, pyexr, tensorboardX, torchvision, matplotlib; requirements.txt:2-10) are
replaced by inert stand-ins for the duration of the import, so that `config.py` and the model files load; anything that
really CALLS into one of them fails loudly with the stand-in's name.
"""
import argparse
import hashlib
import importlib
import inspect
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STUBS = ("easydict", "cv2", "pyexr", "tensorboardX", "torchvision", "torchvision.models", "torchvision.transforms",
         "matplotlib", "matplotlib.pyplot", "mpl_toolkits", "mpl_toolkits.mplot3d")


class _EasyDict(dict):
    """What `from easydict import EasyDict as edict` needs: a dict whose items are attributes, recursively."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self[k] = _EasyDict(v) if isinstance(v, dict) and not isinstance(v, _EasyDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _Inert(types.ModuleType):
    """Stand-in for a module the image lacks: importable, every attribute is a callable that raises when CALLED."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        mod = self.__name__

        def missing(*a, **k):
            raise RuntimeError(f"{mod}.{name} was called, but {mod} is a stand-in (not installed in this image)")
        missing.__name__ = name
        return missing


def install_stubs(log):
    made = []
    for name in STUBS:
        try:
            importlib.import_module(name)
        except Exception:
            m = _Inert(name)
            if name == "easydict":
                m.EasyDict = _EasyDict
            if name == "tensorboardX":
                m.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None,
                                                             "add_scalar": lambda self, *a, **k: None,
                                                             "add_image": lambda self, *a, **k: None,
                                                             "close": lambda self: None})
            sys.modules[name] = m
            parent, _, child = name.rpartition(".")
            if parent and parent in sys.modules:
                setattr(sys.modules[parent], child, m)
            made.append(name)
    if made:
        log("stand-ins installed for: " + ", ".join(made))
    return made


def discover(ref, entry, log):
    """[(qualified name, class)] of the torch.nn.Module subclasses defined under `ref`."""
    import torch
    out = []
    if entry:
        modname, _, cls = entry.partition(":")
        out.append((entry, getattr(importlib.import_module(modname), cls)))
        return out
    skip = ("runner", "setup", "test")
    for dirpath, dirnames, files in os.walk(ref):
        dirnames[:] = [d for d in dirnames if not d.startswith(".") and d not in ("extensions", "__pycache__", "output", "datasets")]
        for f in sorted(files):
            if not f.endswith(".py") or f.split(".")[0] in skip:
                continue
            rel = os.path.relpath(os.path.join(dirpath, f), ref)[:-3].replace(os.sep, ".")
            if rel.endswith(".__init__"):
                rel = rel[:-9]
            try:
                mod = importlib.import_module(rel)
            except Exception as e:
                log(f"  import {rel}: {type(e).__name__}: {e}")
                continue
            for name, obj in vars(mod).items():
                if inspect.isclass(obj) and issubclass(obj, torch.nn.Module) and obj.__module__ == mod.__name__:
                    out.append((f"{rel}:{name}", obj))
    return out


def find_cfg(log):
    """The reference's config tree (config.py: `__C` / `cfg`, README.md:68-78), if it imports."""
    try:
        c = importlib.import_module("config")
    except Exception as e:
        log(f"  config.py not importable: {type(e).__name__}: {e}")
        return None
    for k in ("cfg", "__C", "config", "CFG"):
        if hasattr(c, k):
            return getattr(c, k)
    return None


def instantiate(cls, cfg):
    errs = []
    for args in ((), (cfg,)):
        if args == (cfg,) and cfg is None:
            continue
        try:
            return cls(*args), None
        except Exception as e:
            errs.append(f"{type(e).__name__}: {e}")
    return None, "; ".join(errs)


def _field(m, name):
    v = getattr(m, name, None)
    if isinstance(v, (tuple, list)):
        return v[0] if len(set(v)) == 1 else tuple(v)
    return v


def leaf_rows(model, trace_inputs=None):
    """[(name, class, fields dict, output shape or None)] of the leaf modules, in execution order when traced."""
    import torch
    leaves = [(n, m) for n, m in model.named_modules() if n and not list(m.children())]
    shapes, order = {}, []
    if trace_inputs is not None:
        hooks = []
        for n, m in leaves:
            def hook(mod, inp, out, n=n):
                if n not in shapes:
                    order.append(n)
                shapes[n] = tuple(out.shape) if isinstance(out, torch.Tensor) else type(out).__name__
            hooks.append(m.register_forward_hook(hook))
        try:
            with torch.no_grad():
                model.eval()(*trace_inputs)
        finally:
            for h in hooks:
                h.remove()
    by_name = dict(leaves)
    names = order + [n for n, _ in leaves if n not in shapes]
    rows = []
    for n in names:
        m = by_name[n]
        f = {}
        for k in ("in_channels", "out_channels", "in_features", "out_features", "num_features", "kernel_size", "stride",
                  "padding", "dilation", "output_padding"):
            v = _field(m, k)
            if v is not None:
                f[k] = v
        if hasattr(m, "bias") and not isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            f["bias"] = m.bias is not None
        if hasattr(m, "negative_slope"):
            f["negative_slope"] = m.negative_slope
        rows.append((n, type(m).__name__, f, shapes.get(n)))
    return rows


def conv_like(rows):
    """The layers arch_spec describes: (op, cin, cout, k, s, p) of every conv / transposed conv / linear."""
    ops = {"Conv2d": "conv2d", "Conv3d": "conv3d", "ConvTranspose3d": "deconv3d", "ConvTranspose2d": "deconv2d",
           "Linear": "linear"}
    out = []
    for n, c, f, shape in rows:
        if c in ops:
            out.append({"name": n, "op": ops[c], "cin": f.get("in_channels", f.get("in_features")),
                        "cout": f.get("out_channels", f.get("out_features")), "k": f.get("kernel_size", 1),
                        "s": f.get("stride", 1), "p": f.get("padding", 0), "out": shape})
    return out


def diff_vs_arch_spec(ref_layers, variant, log):
    import s3r
    spec = s3r.arch_spec
    mine = list(spec.ENCODER) + (list(spec.DECODER) if variant == "voxel" else list(spec.DECODER_DOWN) + list(spec.POINT_HEAD))
    log(f"  diff against arch_spec ({variant}): {len(ref_layers)} reference conv/linear layers vs {len(mine)} in arch_spec")
    n_diff = 0
    for i in range(max(len(ref_layers), len(mine))):
        r = ref_layers[i] if i < len(ref_layers) else None
        a = mine[i] if i < len(mine) else None
        if r is None:
            log(f"   - [{i:2d}] arch_spec {a.name}: {a.op} {a.cin}->{a.cout} k{a.k} s{a.s} p{a.p}   (no reference layer)")
            n_diff += 1
            continue
        if a is None:
            log(f"   + [{i:2d}] reference {r['name']}: {r['op']} {r['cin']}->{r['cout']} k{r['k']} s{r['s']} p{r['p']}   (not in arch_spec)")
            n_diff += 1
            continue
        fields = [(k, r[k], getattr(a, k)) for k in ("op", "cin", "cout", "k", "s", "p") if r[k] != getattr(a, k)]
        mark = "=" if not fields else "!"
        n_diff += bool(fields)
        log(f"   {mark} [{i:2d}] {r['name']:32s} {r['op']} {r['cin']}->{r['cout']} k{r['k']} s{r['s']} p{r['p']}"
            + ("" if not fields else "   arch_spec " + a.name + ": " + ", ".join(f"{k} {mv} (ref {rv})" for k, rv, mv in fields)))
    log(f"  {n_diff} of {max(len(ref_layers), len(mine))} rows differ")
    return n_diff


def guess_inputs(model, seed):
    """(left, right) / (left,) on 224x224 synthetic renders, by the forward's positional arity."""
    import s3r
    left, right = s3r.synthetic_pairs(2, seed=seed)
    try:
        params = [p for p in inspect.signature(model.forward).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty]
    except (TypeError, ValueError):
        params = [None, None]
    return (left, right) if len(params) >= 2 else (left,)


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ref", required=True, help="directory of ONE checked-out reference branch (holds runner.py, config.py, models/ ...)")
    ap.add_argument("--entry", default=None, help="module:Class to survey instead of every nn.Module under --ref")
    ap.add_argument("--weights", default=None, help="released .pth: survey its container and run the goldens on it")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"), help="where ref_*.npz / ref_*_keys.json go")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", default="voxel", choices=["voxel", "point"], help="which arch_spec table to diff against")
    args = ap.parse_args(argv)

    def log(*a):
        print(*a, flush=True)

    import numpy as np
    import torch
    import s3r
    ref = os.path.abspath(args.ref)
    if not os.path.isdir(ref):
        sys.exit(f"{ref}: not a directory")
    py = [f for _, _, fs in os.walk(ref) for f in fs if f.endswith(".py")]
    if not py:
        sys.exit(f"{ref} holds no Python source (the mounted reference is README.md + requirements.txt only: "
                 f"README.md:5) — nothing to survey")
    os.makedirs(args.out, exist_ok=True)
    install_stubs(log)
    sys.path.insert(0, ref)
    cfg = find_cfg(log)
    log(f"config tree: {'found' if cfg is not None else 'none'}")
    ckpt = None
    if args.weights:
        obj = torch.load(args.weights, map_location="cpu", weights_only=False)
        log(f"checkpoint {args.weights}: top-level {type(obj).__name__}"
            + (f" keys {list(obj)[:12]}" if isinstance(obj, dict) else ""))
        ckpt = {"sha256": sha256_file(args.weights), "obj": obj}
    classes = discover(ref, args.entry, log)
    log(f"{len(classes)} nn.Module class(es): " + ", ".join(q for q, _ in classes))
    summary = []
    for qual, cls in classes:
        log(f"\n=== {qual}")
        model, err = instantiate(cls, cfg)
        if model is None:
            log(f"  could not be constructed ({err}); pass --entry with a constructible class")
            summary.append({"class": qual, "constructed": False})
            continue
        weights_src = {"seed": args.seed}
        if ckpt is not None:
            try:
                sd = s3r.checkpoint.unwrap(ckpt["obj"])
                res = model.load_state_dict(sd, strict=False)
                log(f"  checkpoint loaded: {len(res.missing_keys)} missing, {len(res.unexpected_keys)} unexpected keys")
                weights_src = {"checkpoint_sha256": ckpt["sha256"]}
            except Exception as e:
                log(f"  checkpoint does not load into this class ({type(e).__name__}: {e}); seeded init instead")
                s3r.seed_module(model, args.seed)
        else:
            try:
                s3r.seed_module(model, args.seed)
            except Exception as e:       # a key this build's seeded init does not know: keep the class's own init
                log(f"  s3r.seed_module: {type(e).__name__}: {e}; torch.manual_seed({args.seed}) init kept")
                weights_src = {"torch_manual_seed": args.seed}
        inputs = guess_inputs(model, args.seed)
        traced = True
        try:
            rows = leaf_rows(model, inputs)
        except Exception as e:
            log(f"  forward on {len(inputs)} x (2,3,224,224) not traceable ({type(e).__name__}: {e}); static table")
            rows, traced = leaf_rows(model, None), False
        log("  layer table (" + ("execution order, output shapes at 224x224" if traced else "definition order") + "):")
        for n, c, f, shape in rows:
            log(f"    {n:40s} {c:18s} " + " ".join(f"{k}={v}" for k, v in f.items()) + (f"  -> {shape}" if shape else ""))
        sd = model.state_dict()
        keys = [(k, list(v.shape)) for k, v in sd.items()]
        log(f"  state_dict: {len(keys)} tensors, {sum(v.numel() for v in sd.values()) / 1e6:.2f} M values")
        for k, shp in keys:
            log(f"    {k:56s} {shp}")
        short = qual.split(":")[-1]
        with open(os.path.join(args.out, f"ref_{short}_keys.json"), "w") as f:
            json.dump({"class": qual, "keys": keys}, f, indent=1)
        n_diff = diff_vs_arch_spec(conv_like(rows), args.variant, log)
        wrote = None
        if traced:
            with torch.no_grad():
                out = model.eval()(*inputs)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            arrays = {f"output{i}": o.detach().cpu().numpy() for i, o in enumerate(outs) if isinstance(o, torch.Tensor)}
            wrote = os.path.join(args.out, f"ref_{short}.npz")
            np.savez_compressed(wrote, inputs=np.array(f"s3r.synthetic_pairs(2, seed={args.seed})[:{len(inputs)}]"),
                                weights=np.array(json.dumps(weights_src)), keys=np.array(json.dumps(keys)), **arrays)
            log(f"  golden vectors -> {wrote} ({', '.join(f'{k}{list(v.shape)}' for k, v in arrays.items())})")
        summary.append({"class": qual, "constructed": True, "traced": traced, "tensors": len(keys), "rows_differing": n_diff,
                        "golden": wrote})
    log("\nsummary: " + json.dumps(summary))
    return summary


if __name__ == "__main__":
    main()
