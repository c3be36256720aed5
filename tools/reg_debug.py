#!/usr/bin/env python3
"""hipHostRegister behaviour on sub-ranges of one host tensor (what evaluate._device_batches relies on)."""
import ctypes, sys, os, time
import torch
torch.cuda.init()
rt = torch.cuda.cudart()
hip = ctypes.CDLL("libamdhip64.so")
t = torch.zeros(64 << 20, dtype=torch.uint8)          # 64 MB
p = t.data_ptr()
print("data_ptr % 4096 =", p % 4096)
PAGE = 4096
def reg(lo, n, tag):
    rc = int(rt.cudaHostRegister(lo, n, 0))
    le = hip.hipGetLastError()
    print(f"{tag}: register({lo % PAGE}+, {n}) rc={rc} lastError={le}", flush=True)
    return rc
def unreg(lo, tag):
    rc = int(rt.cudaHostUnregister(lo))
    le = hip.hipGetLastError()
    print(f"{tag}: unregister rc={rc} lastError={le}", flush=True)
# (a) exact whole tensor
reg(p, t.numel(), "whole"); print("pinned:", t.is_pinned(), t[1000:2000].is_pinned()); unreg(p, "whole")
# (b) page-aligned sub-range in the middle
lo = (p + (8 << 20)) // PAGE * PAGE
reg(lo, 4 << 20, "aligned-mid"); print("pinned mid:", t[(8 << 20) + 5000:(9 << 20)].is_pinned()); 
x = t[(8 << 20) + 5000:(9 << 20)].to("cuda:0", non_blocking=True); torch.cuda.synchronize(); print("copy ok")
# (c) the next aligned range right behind it
reg(lo + (4 << 20), 4 << 20, "aligned-next")
y = t[(8 << 20) + 5000:(15 << 20)].to("cuda:0", non_blocking=True); torch.cuda.synchronize(); print("copy across two registrations ok")
unreg(lo, "aligned-mid"); unreg(lo + (4 << 20), "aligned-next")
# (d) aligned range starting BEFORE the tensor (first page)
lo0 = p // PAGE * PAGE
reg(lo0, 1 << 20, "aligned-first"); unreg(lo0, "aligned-first")
# (e) unaligned exact sub-ranges sharing a page
a = p + 150528 * 3
reg(a, 150528 * 5, "exact-1"); reg(a + 150528 * 5, 150528 * 5, "exact-2 (shares a page)")
unreg(a, "exact-1"); unreg(a + 150528 * 5, "exact-2")
# (f) timing: 38 MB registrations
big = torch.zeros(400 << 20, dtype=torch.uint8)
q = big.data_ptr() // PAGE * PAGE + PAGE
for k in range(4):
    t0 = time.perf_counter(); rc = int(rt.cudaHostRegister(q + k * (38 << 20), 38 << 20, 0)); t1 = time.perf_counter()
    print(f"register 38 MB #{k}: rc={rc} {1e3 * (t1 - t0):.2f} ms")
for k in range(4):
    rt.cudaHostUnregister(q + k * (38 << 20))
