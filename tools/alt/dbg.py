import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import s3r
spec, L = s3r.arch_spec, s3r._lib
dec = {l.name: l for l in spec.DECODER}
head = spec.Layer("d4", "conv3d", 64, 1, 1, 1, 0, bn=False, act="none")
B = 3
ch = s3r.modules._HipChain([dec["d3"], head], 16, precision="fp32"); s3r.seed_module(ch, 7); ch.to("cuda:0")
c1 = s3r.modules._HipChain([dec["d3"]], 16, precision="fp32"); c1.d3.load_state_dict(ch.d3.state_dict()); c1.to("cuda:0")
x = torch.randn((B, 128, 16, 16, 16), generator=torch.Generator().manual_seed(3)).cuda()
ch.algo_override["d3"] = L.ALGO_WINOGRAD; c1.algo_override["d3"] = L.ALGO_WINOGRAD
outs = {}
for code in (7, 8):
    ch.tile_override["d3"] = code
    outs[code] = ch._run(x).clone().cpu().numpy()
c1.tile_override["d3"] = 8
a = c1._run(x).cpu().numpy()          # (B,64,32,32,32) after BN + ReLU
hw = ch.d4.conv.weight.detach().cpu().numpy().reshape(64).astype(np.float32)
hb = ch.d4.conv.bias.detach().cpu().numpy().astype(np.float32)
def fma(a, b, c): return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)
def chain(order_groups):
    tot = None
    for wm in (0, 1):
        th = []
        for h in (0, 1):
            t = np.zeros_like(a[:, 0])
            for r in range(16):
                dm = wm * 32 + 4 * h + (r & 3) + 8 * (r >> 2)
                t = fma(a[:, dm], hw[dm], t)
            th.append(t)
        s = (th[0] + th[1]).astype(np.float32)
        tot = s if tot is None else (tot + s).astype(np.float32)
    return tot
want = chain(None)
want = (want.astype(np.float64) * 1.0 + hb[0]).astype(np.float32)
for code in (7, 8):
    d = np.abs(outs[code].reshape(want.shape) - want)
    print(code, "vs emulated order: max", d.max(), "n diff", int((d > 0).sum()), "of", d.size)
print("head scale/shift:", ch.d4.conv.bias[:1].tolist())
ser = outs[8].reshape(want.shape)
def fin(tot): return (tot.astype(np.float64) + hb[0]).astype(np.float32)
def rep(name, tot):
    d = np.abs(fin(tot) - ser); print(name, "vs serial: max", d.max(), "n diff", int((d > 0).sum()))
# O2: one chain over dm = 0..63
t = np.zeros_like(a[:, 0])
for dm in range(64): t = fma(a[:, dm], hw[dm], t)
rep("O2 sequential", t)
# O3: per (wm): chain r-major over both halves interleaved
tot = None
for wm in (0, 1):
    t = np.zeros_like(a[:, 0])
    for r in range(16):
        for h in (0, 1):
            dm = wm * 32 + 4 * h + (r & 3) + 8 * (r >> 2); t = fma(a[:, dm], hw[dm], t)
    tot = t if tot is None else (tot + t).astype(np.float32)
rep("O3", tot)
# O5: (h0: wm0 + wm1) + (h1: wm0 + wm1)
th = {}
for wm in (0, 1):
    for h in (0, 1):
        t = np.zeros_like(a[:, 0])
        for r in range(16):
            dm = wm * 32 + 4 * h + (r & 3) + 8 * (r >> 2); t = fma(a[:, dm], hw[dm], t)
        th[(wm, h)] = t
rep("O5 (wm sum first)", ((th[(0,0)] + th[(1,0)]).astype(np.float32) + (th[(0,1)] + th[(1,1)]).astype(np.float32)).astype(np.float32))
# O6: fp64 exact
t = np.zeros(a[:, 0].shape, np.float64)
for dm in range(64): t += a[:, dm].astype(np.float64) * np.float64(hw[dm])
rep("O6 fp64", t.astype(np.float32))
# O7: documented order but with activations recomputed unfused: mul then add (no fma)
tot = None
for wm in (0, 1):
    thh = []
    for h in (0, 1):
        t = np.zeros_like(a[:, 0])
        for r in range(16):
            dm = wm * 32 + 4 * h + (r & 3) + 8 * (r >> 2); t = ((a[:, dm] * hw[dm]).astype(np.float32) + t).astype(np.float32)
        thh.append(t)
    s = (thh[0] + thh[1]).astype(np.float32); tot = s if tot is None else (tot + s).astype(np.float32)
rep("O7 no fma", tot)
print("sample", ser.flat[:4], fin(chain(None)).flat[:4])
