"""Minimal OpenEXR reader for the dataset's ground-truth disparity maps.

The reference reads `disp_%02d_{l,r}.exr` (/root/reference/README.md:75-76) through `pyexr`/OpenEXR
(/root/reference/requirements.txt), neither of which is in this image.  This module decodes the subset those files
can be expected to use — single-part scan-line images, HALF / FLOAT / UINT channels, compression NONE, RLE, ZIPS or
ZIP (Blender's default for EXR output is ZIP) — from the published file layout, with numpy + zlib only:

    magic 20000630 | version (low byte 2; flag bits: 0x200 tiled, 0x800 deep, 0x1000 multi-part)
    header: attributes  name\\0 type\\0 size:int32 value...   ended by an empty name
        channels (chlist): name\\0 pixelType:int32 pLinear:u8 pad[3] xSampling:int32 ySampling:int32 ... \\0
        compression: u8 (0 NONE, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ, ...),  dataWindow: box2i,  lineOrder: u8
    offset table: one uint64 per chunk (1 scan line per chunk for NONE/RLE/ZIPS, 16 for ZIP)
    chunk: y:int32 size:int32 data;  uncompressed data = per scan line, per channel (alphabetical), the row's pixels
    ZIP / RLE data is additionally byte-delta-predicted and split into even / odd byte halves.

PIZ, PXR24, B44 and DWA files are rejected with a clear error (convert them with any OpenEXR tool).
`write_exr` (NONE / ZIPS / ZIP, one pixel type for all channels) exists for tests and fixtures.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Tuple

import numpy as np

MAGIC = 20000630
_PIX = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16}
_NAMES = {4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}


def _cstr(buf: bytes, pos: int) -> Tuple[str, int]:
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unpredict(t: np.ndarray) -> np.ndarray:
    """Inverse of OpenEXR's byte predictor + even/odd split (the post-processing shared by ZIP and RLE)."""
    n = t.size
    if n == 0:
        return t
    d = t.astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) & 0xFF).astype(np.uint8)                 # t[i] = t[i-1] + t[i] - 128  (mod 256)
    out = np.empty(n, np.uint8)
    half = (n + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out


def _predict(raw: np.ndarray) -> np.ndarray:
    n = raw.size
    half = (n + 1) // 2
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int64)
    assert t.size == n and half == raw[0::2].size
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 0xFF).astype(np.uint8)


def _unrle(src: bytes, n: int) -> np.ndarray:
    out = bytearray()
    i = 0
    while i < len(src) and len(out) < n:
        c = struct.unpack_from("b", src, i)[0]
        i += 1
        if c < 0:                                               # -c literal bytes
            out += src[i:i - c]
            i += -c
        else:                                                   # c + 1 copies of the next byte
            out += src[i:i + 1] * (c + 1)
            i += 1
    if len(out) != n:
        raise ValueError("corrupt RLE chunk")
    return np.frombuffer(bytes(out), np.uint8)


def read_exr(path: str) -> Dict[str, np.ndarray]:
    """Channel name -> (H, W) float32 array (HALF / UINT channels are widened)."""
    buf = open(path, "rb").read()
    if len(buf) < 8 or struct.unpack_from("<i", buf, 0)[0] != MAGIC:
        raise ValueError(f"{path}: not an OpenEXR file")
    version = struct.unpack_from("<i", buf, 4)[0]
    if (version & 0xFF) != 2 or version & (0x200 | 0x800 | 0x1000):
        raise ValueError(f"{path}: only single-part scan-line OpenEXR 2 files are supported (version word {version:#x})")
    pos, attrs = 8, {}
    while True:
        name, pos = _cstr(buf, pos)
        if not name:
            break
        typ, pos = _cstr(buf, pos)
        size = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ValueError(f"{path}: header has no `{need}` attribute")
    chans, cb, p = [], attrs["channels"][1], 0
    while cb[p] != 0:
        cname, p = _cstr(cb, p)
        ptype, _, xs, ys = struct.unpack_from("<iB3xii", cb, p)
        p += 16
        if ptype not in _PIX or xs != 1 or ys != 1:
            raise ValueError(f"{path}: channel {cname}: pixel type {ptype} / sampling {xs}x{ys} not supported")
        chans.append((cname, _PIX[ptype]))
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ValueError(f"{path}: {_NAMES.get(comp, comp)} compression is not supported by this reader "
                         "(NONE, RLE, ZIPS, ZIP are); re-save the file with ZIP compression")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    if W <= 0 or H <= 0:
        raise ValueError(f"{path}: empty data window")
    lines = _LINES[comp]
    nchunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{nchunks}Q", buf, pos)
    row_bytes = sum(W * dt.itemsize for _, dt in chans)
    out = {n: np.empty((H, W), np.float32) for n, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        r0 = y - y0
        nl = min(lines, H - r0)
        if r0 < 0 or nl <= 0:
            raise ValueError(f"{path}: chunk outside the data window")
        raw_n = nl * row_bytes
        data = buf[off + 8:off + 8 + size]
        if size == raw_n or comp == 0:
            raw = np.frombuffer(data, np.uint8, raw_n)          # stored as is (compression did not pay)
        elif comp == 1:
            raw = _unpredict(_unrle(data, raw_n))
        else:
            raw = _unpredict(np.frombuffer(zlib.decompress(data), np.uint8))
            if raw.size != raw_n:
                raise ValueError(f"{path}: chunk at line {y} inflates to {raw.size} bytes, expected {raw_n}")
        p = 0
        for line in range(nl):
            for cname, dt in chans:
                nb = W * dt.itemsize
                out[cname][r0 + line] = raw[p:p + nb].view(dt).astype(np.float32)
                p += nb
    return out


def write_exr(path: str, channels: Dict[str, np.ndarray], compression: str = "ZIP", half: bool = False) -> None:
    """Write (H, W) arrays as a single-part scan-line file (tests / fixtures)."""
    comp = {"NONE": 0, "ZIPS": 2, "ZIP": 3}[compression]
    names = sorted(channels)
    H, W = channels[names[0]].shape
    dt = np.dtype("<f2") if half else np.dtype("<f4")
    ptype = 1 if half else 2

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    head = struct.pack("<ii", MAGIC, 2)
    head += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = _LINES[comp]
    chunks = []
    for r0 in range(0, H, lines):
        raw = b"".join(np.ascontiguousarray(channels[n][r], dtype=dt).tobytes()
                       for r in range(r0, min(H, r0 + lines)) for n in names)
        data = raw
        if comp:
            z = zlib.compress(_predict(np.frombuffer(raw, np.uint8)).tobytes())
            data = z if len(z) < len(raw) else raw
        chunks.append(struct.pack("<ii", r0, len(data)) + data)
    table_at = len(head)
    off, offsets = table_at + 8 * len(chunks), []
    for c in chunks:
        offsets.append(off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(head + struct.pack(f"<{len(chunks)}Q", *offsets) + b"".join(chunks))


def disparity_channel(channels: Dict[str, np.ndarray]) -> np.ndarray:
    """The one map of a disparity file: a depth-like channel if named so, else the first colour channel (renderers that
    store a scalar in an RGB(A) file repeat it), else the only channel."""
    for k in ("Z", "V", "Y", "disparity", "R"):
        if k in channels:
            return channels[k]
    return channels[sorted(channels)[0]]
