#!/usr/bin/env python3
"""Experiment: software-pipeline ONE batch through the forward in P parts on two HIP streams — the encoder + cost
volume of part i+1 (HBM-heavy: stem, e2, e3, e5, cost volume) running beside the 3D hourglass of part i (MFMA-heavy)
— captured as one HIP graph.  python tools/pipeline_exp.py [--batch 256] [--precision bf16] [--parts 1 2 4]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s3r

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
ap.add_argument("--eager", action="store_true", help="time eager launches instead of a captured graph")
ap.add_argument("--lockstep", action="store_true", help="parts side by side on P streams instead of enc/dec pipelining")
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch
base = s3r.Stereo2Voxel(a.precision)
s3r.seed_module(base, 0)
base.to(dev)
left, right = (t.to(dev) for t in s3r.synthetic_pairs(B, seed=1000))
ref = base(left, right).clone()


def replica():
    m = s3r.Stereo2Voxel(a.precision)
    for chain in ("encoder", "decoder"):
        src, dst = getattr(base, chain), getattr(m, chain)
        for n in src.names:
            setattr(dst, n, getattr(src, n))              # the SAME parameter holders: only the arenas are per replica
    return m.to(dev)


for P in a.parts:
    per = B // P
    models = [base if (i == 0 and P == 1) else replica() for i in range(P)]
    ls, rs = [left[i * per:(i + 1) * per] for i in range(P)], [right[i * per:(i + 1) * per] for i in range(P)]
    s_enc, s_dec = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    extra = [torch.cuda.Stream(device=dev) for _ in range(P)]
    out = torch.empty(B, 32, 32, 32, device=dev)

    def fwd():
        cur = torch.cuda.current_stream(dev)
        if P == 1:
            out.copy_(models[0](left, right))
            return
        if a.lockstep:
            for i, st in enumerate(extra):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    out[i * per:(i + 1) * per].copy_(models[i](ls[i], rs[i]))
            for st in extra:
                cur.wait_stream(st)
            return
        s_enc.wait_stream(cur), s_dec.wait_stream(cur)
        vols, evs = [], []
        with torch.cuda.stream(s_enc):
            for i in range(P):
                f = models[i].encoder.forward_pair(ls[i], rs[i])
                vols.append(models[i].cost_volume.forward_padded(f[:per], f[per:]))
                ev = torch.cuda.Event()
                ev.record(s_enc)
                evs.append(ev)
        with torch.cuda.stream(s_dec):
            for i in range(P):
                s_dec.wait_event(evs[i])
                out[i * per:(i + 1) * per].copy_(models[i].decoder.forward_padded(vols[i]))
        cur.wait_stream(s_enc), cur.wait_stream(s_dec)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    if a.eager:
        class g:                                          # noqa: N801  (same .replay() as the graph)
            replay = staticmethod(fwd)
    else:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fwd()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / a.steps)
    dt = min(ts)
    print(f"{a.precision} B={B} parts={P}{' lockstep' if a.lockstep else ''}{' eager' if a.eager else ''}: {dt * 1e3:7.3f} ms/step  {B / dt:8.1f} pairs/s  "
          f"bit-identical to one part: {same and torch.equal(out, ref)}", flush=True)
