#!/usr/bin/env python3
"""Per-layer microbenchmark of the MFMA conv kernel at BASELINE batch (B=32 pairs), interleaved A/B of
tile configurations / gather widths in ONE process (cdna guide §5.4 rule 24).

  python tools/layer_bench.py [--layers v1,d3] [--tiles -1,0,1] [--variants 0,1] [--rounds 5] [--batch 32]

`tile` passed to the C-ABI = tile_cfg + 16*gather_width (tile_cfg -1 = library heuristic; width 0 = widest
legal, 1 = dword LDS-DMA, 4 = 16-byte LDS-DMA).  Configurations a layer cannot run are skipped.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import s3r  # noqa: E402

PEAK = 157.3   # fp32; bf16 runs are shown against the same number for convenience


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="")
    ap.add_argument("--tiles", default="-1")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--ksplits", default="0", help="split-K factors to try (0 = library heuristic)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--shapes", default="", help="bf16: MFMA shapes to A/B, e.g. 32,16 (S3R_BF16_MFMA; weights are re-packed per shape)")
    ap.add_argument("--algo", type=int, default=0, help="fp32: s3r_algo of the layer (0 auto, 1 direct, 2 Winograd: --tiles are then "
                    "launch-FORM codes: -1 the library's plan, 0 serial, 1 class-parallel, 2 dual)")
    ap.add_argument("--zeros", action="store_true", help="all-zero inputs and weights (how much of the rate is power: the\n"
                    "chip holds a higher clock on zeros, MI355X_MICROARCH.md DVFS notes)")
    args = ap.parse_args()
    spec = s3r.arch_spec
    dev = torch.device("cuda:0")
    cases = []
    for layers, n0, mult in ((spec.ENCODER, spec.IMG_HW, 2), (spec.DECODER, spec.MAX_DISP, 1)):
        for l, n_in, _ in spec.trace(layers, n0):
            if l.name in ("e1", "d4"):
                continue
            if args.layers and l.name not in args.layers.split(","):
                continue
            cases.append((l, n_in, mult * args.batch))
    tiles = [int(t) for t in args.tiles.split(",")]
    variants = [int(v) for v in args.variants.split(",")]
    ksplits = [int(v) for v in args.ksplits.split(",")]
    total = {}
    dbg_clock = None
    try:                                   # only present in -DS3R_ABLATE diagnostic builds
        import ctypes
        dbg_clock = s3r.load_library().s3r_debug_clock_ghz
        dbg_clock.restype, dbg_clock.argtypes = ctypes.c_double, [ctypes.c_int]
    except AttributeError:
        pass
    shapes = [int(v) for v in args.shapes.split(",")] if args.shapes else [0]
    for l, n_in, B in cases:
        ch = s3r.modules._HipChain([l], n_in, precision=args.dtype)
        s3r.seed_module(ch, 1)
        ch.to(dev)
        x = torch.randn((B, l.cin) + (n_in,) * spec.ndim(l), device=dev)
        if args.zeros:
            x.zero_()
            with torch.no_grad():
                for prm in ch.parameters():
                    prm.zero_()
        if args.dtype == "bf16":
            x = x.to(torch.bfloat16).permute(0, *range(2, x.dim()), 1).contiguous()
        run_kw = {}
        flops = 2.0 * spec.layer_macs(l, n_in) * B
        res = {}
        ran = {}
        clk = {}
        ref = None
        for rnd in range(args.rounds + 1):
          for shp in shapes:
            if shp:
                os.environ["S3R_BF16_MFMA"] = str(shp)      # (the module's pack cache is keyed on it: re-packs)
            for t in tiles:
                for v0 in variants:
                  for ks in ksplits:
                    v = (v0 if not shp else shp, ks)
                    code = (15 if t < 0 else t) + 16 * v0
                    if args.dtype == "bf16":
                        code = t            # -1 = library heuristic; 1, 2, 4 per-tap; 9, 10 row-reuse
                    if args.algo == 2:
                        code = t            # -1 = the library's own plan
                    ch.tile_override[l.name] = code
                    ch.ksplit_override[l.name] = ks
                    if args.algo:
                        ch.algo_override[l.name] = args.algo
                    s3r.profile_enable(8)
                    try:
                        y = ch._run(x, None, **run_kw)
                    except s3r.S3RError:
                        s3r.profile_enable(0)
                        continue
                    rec = s3r.profile_read(8)
                    s3r.profile_enable(0)
                    if rnd == 0:
                        if ref is None:
                            ref = y.clone()
                        elif args.algo == 2 and not torch.equal(y, ref):      # every launch form must give the same bits
                            print(f"!! {l.name} form {t}: NOT bit-identical to the first form (max {float((y - ref).abs().max()):.3e})")
                        elif not torch.allclose(y.float(), ref.float(), rtol=1e-4 if args.dtype == "fp32" else 2e-2,
                                                atol=1e-4 if args.dtype == "fp32" else 2e-2):
                            print(f"!! {l.name} tile {t} variant {v}: output differs from first config "
                                  f"(max {float((y - ref).abs().max()):.3e})")
                        continue
                    mrec = [r for r in rec if r["family"] == "conv_mfma"][0]
                    res.setdefault((t, v), []).append(mrec["ms"])
                    ran[(t, v)] = f"{mrec['ran']}/{mrec['launches']}"
                    if dbg_clock is not None:
                        torch.cuda.synchronize()
                        clk[(t, v)] = dbg_clock(256)
        line = f"{l.name:4s}"
        best = None
        for (t, v), ms in sorted(res.items()):
            ms.sort()
            med = ms[len(ms) // 2]
            tf = flops / med / 1e9
            if best is None or med < best[0]:
                best = (med, t, v)
            line += f" | t{t} v{v[0]} k{v[1]}: {med:7.4f} ms {tf:6.1f} TF {tf / PEAK:5.3f} [{ran.get((t, v), '')}]"
            if (t, v) in clk:
                line += f" @{clk[(t, v)]:.2f}GHz"
            total.setdefault((t, v), [0.0, 0.0])
            total[(t, v)][0] += med
            total[(t, v)][1] += flops
        print(line, flush=True)
        print(f"{l.name:4s} BEST tile {best[1]} vec {best[2][0]} ksplit {best[2][1]}: {best[0]:.4f} ms "
              f"{flops / best[0] / 1e9:6.1f} TF", flush=True)
    for (t, v), (ms, fl) in sorted(total.items()):
        print(f"TOTAL t{t} v{v[0]} k{v[1]}: {ms:8.4f} ms  {fl / ms / 1e9:6.1f} TF  frac {fl / ms / 1e9 / PEAK:5.3f}")


if __name__ == "__main__":
    main()
