"""The package's own OpenEXR decoder (exr.py: SURVEY.md §8f rows 3-4 — the dataset's disp_%02d_{l,r}.exr ground truth,
/root/reference/README.md:75-76): files written here and read back, the ZIP post-processing checked against a direct
transcription of the published byte loops, header / error handling."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch


def _loops_unpredict(buf: bytes) -> bytes:
    """OpenEXR's decode post-processing, written as the byte loops of the file-format description: running sum with
    bias 128, then the first half of the buffer supplies the even bytes, the second half the odd bytes."""
    t = bytearray(buf)
    for i in range(1, len(t)):
        t[i] = (t[i - 1] + t[i] - 128) & 0xFF
    out = bytearray(len(t))
    half = (len(t) + 1) // 2
    a, b = 0, half
    for i in range(len(t)):
        if i % 2 == 0:
            out[i] = t[a]; a += 1
        else:
            out[i] = t[b]; b += 1
    return bytes(out)


@pytest.mark.parametrize("comp", ["NONE", "ZIPS", "ZIP"])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(224, 224), (17, 5), (1, 1), (33, 100)])
def test_exr_round_trip(s3r, tmp_path, comp, half, shape):
    rng = np.random.default_rng(sum(shape))
    ch = {"R": (rng.random(shape, dtype=np.float32) * 200),
          "Z": np.where(rng.random(shape) < 0.3, np.inf, rng.random(shape) * 50).astype(np.float32)}
    p = str(tmp_path / "t.exr")
    s3r.exr.write_exr(p, ch, comp, half)
    back = s3r.exr.read_exr(p)
    assert sorted(back) == ["R", "Z"]
    for k in ch:
        want = ch[k].astype(np.float16).astype(np.float32) if half else ch[k]
        assert back[k].dtype == np.float32 and np.array_equal(back[k], want)
    assert s3r.exr.disparity_channel(back) is back["Z"]


@pytest.mark.parametrize("n", [1, 2, 3, 16, 1001, 4096])
def test_zip_post_processing_matches_the_byte_loops(s3r, n):
    rng = np.random.default_rng(n)
    data = rng.integers(0, 256, n, dtype=np.uint8)
    assert s3r.exr._unpredict(data).tobytes() == _loops_unpredict(data.tobytes())
    assert np.array_equal(s3r.exr._unpredict(s3r.exr._predict(data)), data)


def test_zip_chunk_written_by_hand(s3r, tmp_path):
    """A 2 x 3 single-channel FLOAT file assembled byte by byte with an independent encoder (loops + zlib)."""
    img = np.array([[0.5, 1.5, -2.0], [3.25, np.inf, 7.0]], np.float32)

    def encode(raw: bytes) -> bytes:
        t = bytes(raw[0::2]) + bytes(raw[1::2])
        d = bytearray(t)
        for i in range(len(t) - 1, 0, -1):
            d[i] = (t[i] - t[i - 1] + 128) & 0xFF
        return zlib.compress(bytes(d))

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    head = struct.pack("<ii", 20000630, 2)
    head += attr("channels", "chlist", b"Z\0" + struct.pack("<iB3xii", 2, 0, 1, 1) + b"\0")
    head += attr("compression", "compression", b"\x02")                      # ZIPS: one scan line per chunk
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, 2, 1)) + b"\0"
    chunks = []
    for y in range(2):
        z = encode(img[y].tobytes())
        assert _loops_unpredict(zlib.decompress(z)) == img[y].tobytes()
        chunks.append(struct.pack("<ii", y, len(z)) + z)
    off = len(head) + 16
    table = struct.pack("<2Q", off, off + len(chunks[0]))
    p = tmp_path / "hand.exr"
    p.write_bytes(head + table + b"".join(chunks))
    assert np.array_equal(s3r.exr.read_exr(str(p))["Z"], img)


def test_committed_zip_half_fixture_from_an_independent_encoder(s3r, golden_dir):
    """tests/golden/disp_zip_half.exr was assembled by tests/golden/make_exr_fixture.py, which shares no code with exr.py
    (struct + zlib + byte loops): two channels (FLOAT `A`, HALF `Z`), ZIP compression in 16-scan-line blocks (16 + 5
    lines), infinities as the dataset's background marker.  The decoder must reproduce the stored arrays exactly."""
    back = s3r.exr.read_exr(os.path.join(golden_dir, "disp_zip_half.exr"))
    want = np.load(os.path.join(golden_dir, "disp_zip_half.npz"))
    assert sorted(back) == ["A", "Z"]
    for k in ("A", "Z"):
        assert back[k].dtype == np.float32 and back[k].shape == (21, 13) and np.array_equal(back[k], want[k]), k
    assert s3r.exr.disparity_channel(back) is back["Z"] and np.isinf(back["Z"][2, 3:9]).all()


def test_rle_file(s3r, tmp_path):
    """compression 1: signed run lengths (n >= 0: n + 1 copies of the next byte; n < 0: -n literal bytes) over the same
    predicted / split byte stream as ZIP.  Encoded here with an independent greedy run-length encoder."""
    img = np.zeros((5, 40), np.float32)
    img[1, 3:30] = 7.5
    img[3] = np.arange(40, dtype=np.float32) / 3
    img[4, 20:] = np.inf

    def predict_loops(raw: bytes) -> bytes:
        t = bytes(raw[0::2]) + bytes(raw[1::2])
        d = bytearray(t)
        for i in range(len(t) - 1, 0, -1):
            d[i] = (t[i] - t[i - 1] + 128) & 0xFF
        return bytes(d)

    def rle(b: bytes) -> bytes:
        out, i = bytearray(), 0
        while i < len(b):
            j = i
            while j + 1 < len(b) and b[j + 1] == b[i] and j - i < 126:
                j += 1
            if j - i >= 2:                                         # a run of j - i + 1 equal bytes
                out += struct.pack("b", j - i) + b[i:i + 1]
                i = j + 1
            else:                                                  # literals up to the next run of >= 3
                k = i
                while k < len(b) and k - i < 127 and not (k + 2 < len(b) and b[k] == b[k + 1] == b[k + 2]):
                    k += 1
                out += struct.pack("b", -(k - i)) + b[i:k]
                i = k
        return bytes(out)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    head = struct.pack("<ii", 20000630, 2)
    head += attr("channels", "chlist", b"Z\0" + struct.pack("<iB3xii", 2, 0, 1, 1) + b"\0")
    head += attr("compression", "compression", b"\x01")
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, 39, 4)) + b"\0"
    chunks, off = [], len(head) + 8 * 5
    offsets = []
    for y in range(5):
        enc = rle(predict_loops(img[y].tobytes()))
        data = enc if len(enc) < 160 else img[y].tobytes()        # stored raw when RLE does not pay (as OpenEXR does)
        offsets.append(off)
        chunks.append(struct.pack("<ii", y, len(data)) + data)
        off += len(chunks[-1])
    p = tmp_path / "rle.exr"
    p.write_bytes(head + struct.pack("<5Q", *offsets) + b"".join(chunks))
    assert np.array_equal(s3r.exr.read_exr(str(p))["Z"], img)
    assert s3r.exr._unrle(bytes([0xFD, 1, 2, 3, 4, 9]), 8).tolist() == [1, 2, 3, 9, 9, 9, 9, 9]


def test_exr_errors(s3r, tmp_path):
    p = tmp_path / "x.exr"
    p.write_bytes(b"not an exr file at all")
    with pytest.raises(ValueError, match="not an OpenEXR"):
        s3r.exr.read_exr(str(p))
    good = tmp_path / "g.exr"
    s3r.exr.write_exr(str(good), {"Z": np.zeros((4, 4), np.float32)}, "NONE")
    b = bytearray(good.read_bytes())
    i = b.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    b[i] = 4                                                                  # PIZ
    (tmp_path / "piz.exr").write_bytes(bytes(b))
    with pytest.raises(ValueError, match="PIZ"):
        s3r.exr.read_exr(str(tmp_path / "piz.exr"))
    b = bytearray(good.read_bytes())
    b[5] |= 0x02                                                              # the tiled flag
    (tmp_path / "tiled.exr").write_bytes(bytes(b))
    with pytest.raises(ValueError, match="single-part scan-line"):
        s3r.exr.read_exr(str(tmp_path / "tiled.exr"))


def test_downsample_disparity(s3r):
    d = torch.full((1, 224, 224), float("inf"))
    d[0, :8, :8] = 16.0
    d[0, 0, 0] = -1.0                                                         # invalid pixels do not enter the mean
    d[0, 8:16, :8] = torch.arange(64, dtype=torch.float32).reshape(8, 8)
    out = s3r.data.downsample_disparity(d, 28)
    assert out.shape == (1, 28, 28)
    assert out[0, 0, 0].item() == 16.0 and out[0, 1, 0].item() == 31.5 and torch.isinf(out[0, 2:, :]).all()
    with pytest.raises(ValueError):
        s3r.data.downsample_disparity(torch.zeros(1, 30, 30), 28)
