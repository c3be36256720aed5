"""Writes tests/golden/disp_zip_half.exr + disp_zip_half.npz: an OpenEXR file assembled byte by byte WITHOUT this
package's exr.py (struct + zlib + byte loops only), so that the decoder is checked against an encoder that shares no
code with it (ZIP blocks and a HALF channel; the RLE and ZIPS cases have such files in
tests/test_exr_cpu.py already).

    python tests/golden/make_exr_fixture.py

Layout facts used (OpenEXR file format, single-part scan-line files): magic 20000630, version 2; header = attributes
(name\\0 type\\0 int32 size, value) ended by \\0; `channels` = chlist of (name\\0, int32 pixel type 0 UINT / 1 HALF /
2 FLOAT, uint8 pLinear, 3 pad bytes, int32 xSampling, int32 ySampling) ended by \\0, channels in ALPHABETICAL order;
compression 3 = ZIP: blocks of 16 scan lines; a block holds, line by line, every channel's pixels of that line
(channel-major within the line); the block's bytes are split into even / odd bytes, delta-predicted with bias 128 and
deflated; a block whose deflated size is not smaller is stored raw; then the line-offset table (one uint64 per block).
"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def attr(name, typ, val):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val


def predict_loops(raw: bytes) -> bytes:
    t = bytes(raw[0::2]) + bytes(raw[1::2])              # even bytes, then odd bytes
    d = bytearray(t)
    for i in range(len(t) - 1, 0, -1):
        d[i] = (t[i] - t[i - 1] + 128) & 0xFF
    return bytes(d)


def main():
    h, w = 21, 13                                         # two ZIP blocks: 16 lines + 5 lines
    rng = np.random.default_rng(20261003)
    z = (rng.random((h, w)) * 180.0).astype(np.float16)   # HALF disparity-like channel
    z[2, 3:9] = np.inf                                    # background marker
    z[20, :] = 0.0
    a = rng.standard_normal((h, w)).astype(np.float32)    # a FLOAT channel beside it ("A" sorts before "Z")
    a[5, 5] = -np.inf
    chl = b""
    for name, ptype in (("A", 2), ("Z", 1)):
        chl += name.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1)
    chl += b"\0"
    head = struct.pack("<ii", 20000630, 2)
    head += attr("channels", "chlist", chl)
    head += attr("compression", "compression", b"\x03")                       # ZIP
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    head += attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    head += attr("lineOrder", "lineOrder", b"\x00")
    head += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    blocks = []
    for y0 in range(0, h, 16):
        raw = b""
        for y in range(y0, min(h, y0 + 16)):
            raw += a[y].tobytes() + z[y].tobytes()        # channel-major inside a line, alphabetical
        comp = zlib.compress(predict_loops(raw), 6)
        data = comp if len(comp) < len(raw) else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    off = len(head) + 8 * len(blocks)
    table = b""
    for b in blocks:
        table += struct.pack("<Q", off)
        off += len(b)
    with open(os.path.join(HERE, "disp_zip_half.exr"), "wb") as f:
        f.write(head + table + b"".join(blocks))
    np.savez(os.path.join(HERE, "disp_zip_half.npz"), Z=z.astype(np.float32), A=a)
    print("wrote disp_zip_half.exr", len(head) + len(table) + sum(map(len, blocks)), "bytes")


if __name__ == "__main__":
    main()
