"""CPU checks of the PNG ingest's host half (csrc/vsf_png_host.cc: chunk walk, CRCs, upload plan; reference counterpart:
cv::imdecode at slam_frontend_main.cc:98-100) and of the committed fixtures.

* tests/golden/png (tools/make_png_golden.py): every file decodes with the Python standard library alone (zlib + the five
  filters restated in numpy) to the image PIL = libpng read when the fixture was made -- the vectors are self-consistent
  and the GPU test's reference for them is a real third party's output.
* `make asan` builds the parser with -fsanitize=address,undefined; a child process runs the fixtures, hand-made refusals and
  3000 seeded mutations (bit flips anywhere, cuts, chunk lengths rewritten, chunks reordered / duplicated / dropped) through
  vsf_png_plan + vsf_png_fill and must exit cleanly; 3000 more, one file at a time, are held against the real libpng driven
  as cv::imdecode drives it (tests/png_ref.py): what it reads the parser accepts, what the parser accepts with the compressed
  data whole it reads."""
import os
import subprocess
import sys
import zlib
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "vision_slam_frontend_amd" / "csrc"
LIB = ROOT / "vision_slam_frontend_amd" / "libvsf_jpeg_host_asan.so"
GOLD = Path(__file__).resolve().parent / "golden" / "png"
sys.path.insert(0, str(Path(__file__).resolve().parent))
import png_craft as pc  # noqa: E402


def unfilter_numpy(raw: bytes, h: int, row_bytes: int, bpp: int) -> np.ndarray:
    rows = np.frombuffer(raw, np.uint8).reshape(h, row_bytes + 1)
    out = np.zeros((h, row_bytes), np.uint8)
    prev = np.zeros(row_bytes, np.int32)
    for y in range(h):
        t = int(rows[y, 0])
        cur = np.zeros(row_bytes, np.int32)
        f = rows[y, 1:].astype(np.int32)
        for x in range(row_bytes):
            a = cur[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            if t == 0:
                p = 0
            elif t == 1:
                p = a
            elif t == 2:
                p = b
            elif t == 3:
                p = (a + b) >> 1
            else:
                pa, pb, pcc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pcc) else (b if pb <= pcc else c)
            cur[x] = (f[x] + p) & 255
        out[y] = cur
        prev = cur
    return out


def gray_from_rows(rows: np.ndarray, w: int, depth: int, channels: int) -> np.ndarray:
    if depth == 8:
        return rows[:, ::channels].copy()
    if depth == 16:
        return rows[:, ::2 * channels].copy()
    ppb = 8 // depth
    out = np.zeros((rows.shape[0], rows.shape[1] * ppb), np.uint8)
    mask = (1 << depth) - 1
    for p in range(ppb):
        out[:, p::ppb] = ((rows >> (8 - depth * (p + 1))) & mask) * (255 // mask)
    return out[:, :w]


def libpng_table(gamma_val: int) -> np.ndarray:
    """png_build_8bit_table: floor(255 * pow(i / 255, gamma) + .5) where the exponent differs from 1 by more than 5 %."""
    import math
    t = np.arange(256, dtype=np.uint8)
    if gamma_val < 95000 or gamma_val > 105000:
        for i in range(1, 255):
            t[i] = int(math.floor(255 * math.pow(i / 255., gamma_val * .00001) + .5))
    return t


def rgb_to_gray_numpy(r, g, b, file_gamma: int):
    """png_do_rgb_to_gray for 8-bit samples with the coefficients of png_set_rgb_to_gray(1, 0.299, 0.587)."""
    import math
    r, g, b = (x.astype(np.uint32) for x in (r, g, b))
    screen = int(math.floor(1e10 / file_gamma + .5))
    if not (file_gamma < 95000 or file_gamma > 105000 or screen < 95000 or screen > 105000):
        return ((9797 * r + 19234 * g + 3737 * b) >> 15).astype(np.uint8)
    to_1 = libpng_table(int(math.floor(1e10 / file_gamma + .5))).astype(np.uint32)
    from_1 = libpng_table(int(math.floor(1e10 / screen + .5)))
    mixed = from_1[(9797 * to_1[r] + 19234 * to_1[g] + 3737 * to_1[b] + 16384) >> 15]
    return np.where((r == g) & (r == b), r.astype(np.uint8), mixed)


def chunks_of(data: bytes):
    pos, out = 8, []
    while pos + 12 <= len(data):
        n = int.from_bytes(data[pos:pos + 4], "big")
        out.append((data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]))
        pos += 12 + n
    return out


def test_png_fixtures_are_self_consistent():
    """Every committed fixture decodes with the standard library alone (zlib, the filters and -- for the colour files -- libpng's
    rgb_to_gray restated in numpy) to the image libpng read when the fixture was made."""
    expected = dict(np.load(GOLD / "expected_gray.npz"))
    n_gray = len(expected)
    expected.update(np.load(GOLD / "expected_gray_colour.npz"))
    files = sorted(GOLD.glob("*.png"))
    assert len(files) == len(expected) and n_gray >= 20 and len(expected) - n_gray >= 15
    for f in files:
        data = f.read_bytes()
        want = expected[f.stem]
        h, w = want.shape
        depth, ctype, lace = data[24], data[25], data[28]
        channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        bpp = max(1, depth * channels // 8)
        raw = zlib.decompress(pc.idat_stream(data))
        # the file's samples as one array of scanline bytes per pixel position: for an interlaced file pass by pass
        if lace:
            sample_rows = np.zeros((h, w * max(1, depth * channels // 8)), np.uint8) if depth >= 8 else np.zeros((h, w), np.uint8)
            at = 0
            for x0, y0, dx, dy in pc.ADAM7:
                wp, hp = len(range(x0, w, dx)), len(range(y0, h, dy))
                if wp == 0 or hp == 0:
                    continue
                rbp = (wp * depth * channels + 7) // 8
                sub = unfilter_numpy(raw[at:at + (rbp + 1) * hp], hp, rbp, bpp)
                at += (rbp + 1) * hp
                if depth >= 8:
                    sample_rows.reshape(h, w, bpp)[y0::dy, x0::dx] = sub.reshape(hp, wp, bpp)
                else:   # unpack to one sample per byte
                    ppb = 8 // depth
                    un = np.zeros((hp, rbp * ppb), np.uint8)
                    for p in range(ppb):
                        un[:, p::ppb] = (sub >> (8 - depth * (p + 1))) & ((1 << depth) - 1)
                    sample_rows[y0::dy, x0::dx] = un[:, :wp]
            assert at == len(raw), f.name
            if depth < 8:   # back to the packed form the code below expects
                rows = pc.pack_samples(sample_rows, depth)
            else:
                rows = sample_rows
        else:
            row_bytes = (w * depth * channels + 7) // 8
            assert len(raw) == (row_bytes + 1) * h, f.name
            rows = unfilter_numpy(raw, h, row_bytes, bpp)
        if ctype in (0, 4):
            np.testing.assert_array_equal(gray_from_rows(rows, w, depth, channels), want, err_msg=f.name)
            continue
        cs = chunks_of(data)
        gamma = 100000
        for kind, body in cs:
            if kind == b"gAMA":
                gamma = int.from_bytes(body, "big")
        if any(kind == b"sRGB" for kind, _ in cs):
            gamma = 45455
        if ctype == 3:
            pal = np.zeros((256, 3), np.uint8)
            body = [b for k, b in cs if k == b"PLTE"][0]
            n = min(len(body) // 3, 1 << depth)
            pal[:n] = np.frombuffer(body, np.uint8)[:3 * n].reshape(n, 3)
            lut = rgb_to_gray_numpy(pal[:, 0], pal[:, 1], pal[:, 2], gamma)
            idx = gray_from_rows(rows, w, depth, 1) // (255 // ((1 << depth) - 1)) if depth < 8 else rows
            got = lut[idx]
        elif depth == 8:
            px = rows.reshape(h, w, channels)
            got = rgb_to_gray_numpy(px[..., 0], px[..., 1], px[..., 2], gamma)
        else:
            px = rows.reshape(h, w, channels, 2).astype(np.uint32)
            v = (px[..., 0] << 8) | px[..., 1]
            got = (((9797 * v[..., 0] + 19234 * v[..., 1] + 3737 * v[..., 2] + 16384) >> 15) >> 8).astype(np.uint8)
        np.testing.assert_array_equal(got, want, err_msg=f.name)


DRIVER = r'''
import ctypes as C, struct, sys, zlib
from pathlib import Path
import numpy as np
sys.path.insert(0, sys.argv[3])
import png_craft as pc
lib = C.CDLL(sys.argv[1])
lib.vsf_png_host_check.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
gold = Path(sys.argv[2])
expected = dict(np.load(gold / "expected_gray.npz"))
expected.update(np.load(gold / "expected_gray_colour.npz"))
files = {p.stem: p.read_bytes() for p in sorted(gold.glob("*.png"))}

def check(batch, w, h):
    n = len(batch)
    ptrs = (C.c_char_p * n)(*batch)
    sizes = (C.c_size_t * n)(*[len(b) for b in batch])
    total, csum = C.c_uint64(), C.c_uint32()
    return lib.vsf_png_host_check(ptrs, sizes, n, w, h, C.byref(total), C.byref(csum)), total.value

def chunks(data):
    pos, out = 8, []
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        out.append(data[pos:pos + 12 + n])
        pos += 12 + n
    return out

for name, data in files.items():
    h, w = expected[name].shape
    st, total = check([data], w, h)
    assert st == 0 and total >= len(pc.idat_stream(data)), (name, st, total)
    assert check([data], w + 1, h)[0] == 1 and check([data], w, h + 1)[0] == 1, name
    assert check([data[:-12]], w, h)[0] == 1, name          # no IEND
    b = bytearray(data); b[len(data) // 2] ^= 1               # inside IDAT: its CRC no longer matches
    assert check([bytes(b)], w, h)[0] == 1, name
img = expected["pil_scene_level6"]
h, w = img.shape
good = pc.gray8(img)
assert check([good], w, h)[0] == 0
assert check([pc.gray8(img, extra_before=[pc.chunk(b"tEXt", b"k\x00v", bad_crc=True)])], w, h)[0] == 0      # ancillary, damaged: skipped
assert check([pc.gray8(img, extra_before=[pc.chunk(b"ABCD", b"")])], w, h)[0] == 1                          # unknown critical chunk
assert check([pc.write_png(pc.pack_samples(img, 8), w, h, 8, 0, interlace=1)], w, h)[0] == 0               # Adam7 (the device finds the data short)
assert check([pc.write_png(np.zeros((h, 3 * w), np.uint8), w, h, 8, 2)], w, h)[0] == 0                     # RGB
assert check([pc.write_png(np.zeros((h, w), np.uint8), w, h, 8, 3)], w, h)[0] == 1                         # palette without a PLTE chunk
assert check([pc.write_png(np.zeros((h, w), np.uint8), w, h, 8, 3, extra_before=[pc.chunk(b"PLTE", bytes(30))])], w, h)[0] == 0
assert check([pc.write_png(np.zeros((h, w), np.uint8), w, h, 8, 3, extra_before=[pc.chunk(b"PLTE", bytes(31))])], w, h)[0] == 1
assert check([pc.write_png(np.zeros((h, 3 * w), np.uint8), w, h, 8, 2, extra_before=[pc.chunk(b"iCCP", b"x")])], w, h)[0] == 4   # colour + iCCP: unsupported
assert check([pc.write_png(pc.pack_samples(img, 8), w, h, 3, 0)], w, h)[0] == 1                            # 3-bit gray does not exist
for hdr in (b"\x78\x9d", b"\x88\x1c", b"\x78\xbb", b"\x79\x9c"):                                            # zlib headers that fail their checks
    assert check([pc.replace_idat(good, hdr + pc.idat_stream(good)[2:])], w, h)[0] == 1, hdr
c = chunks(good)
assert check([good[:8] + c[0] + c[1] + pc.chunk(b"tIME", bytes(7)) + c[1] + c[-1]], w, h)[0] == 1          # IDAT, other chunk, IDAT
assert check([good[:8] + c[0] + c[0] + b"".join(c[1:])], w, h)[0] == 1                                     # IHDR twice
assert check([good[:8] + b"".join(c[1:])], w, h)[0] == 1                                                   # no IHDR
assert check([good[:8] + c[0] + c[-1]], w, h)[0] == 1                                                      # no IDAT
# a chunk's PLACE counts whatever its CRC says (round-5 advice; both reproduced against libpng 1.6.37 there):
assert check([good[:8] + pc.chunk(b"tEXt", b"k\x00v", bad_crc=True) + b"".join(c)], w, h)[0] == 1          # damaged tEXt in front of IHDR: "missing IHDR"
assert check([good[:8] + c[0] + c[1] + pc.chunk(b"tEXt", b"k\x00v", bad_crc=True) + c[1] + c[-1]], w, h)[0] == 1   # ... between two IDATs: the run is over
assert check([good[:8] + c[0] + pc.chunk(b"a1Bc", b"x") + b"".join(c[1:])], w, h)[0] == 1                   # png_check_chunk_name: "invalid chunk type"
assert check([good[:8] + c[0] + pc.chunk(b"tEXt", bytes(9000000)) + b"".join(c[1:])], w, h)[0] == 1         # png_check_chunk_length: "chunk data is too large"
assert check([good[:8] + c[0] + pc.chunk(b"tEXt", bytes(7999999)) + b"".join(c[1:])], w, h)[0] == 0
# a zlib header that declares a smaller window than 32 KiB (CINFO 6, check bits right): libpng would hold every match
# distance against it; the device's window is the full one, so such a file is refused as unsupported, never decoded leniently
assert check([pc.replace_idat(good, b"\x68\x05" + pc.idat_stream(good)[2:])], w, h)[0] == 4

def mutate(data, rng):
    b = bytearray(data)
    kind = int(rng.integers(10))
    if kind >= 7:     # framing intact, CRCs right: damage inside the compressed data, IDAT cut anew, ancillary chunks added
        stream = pc.idat_stream(data)
        if kind == 7:
            stream = pc.mutate_stream(stream, rng)
        cs = chunks(data)
        extra = [pc.chunk(b"tEXt", bytes(rng.integers(32, 127, int(rng.integers(0, 30)), dtype=np.uint8)), bad_crc=bool(rng.integers(2)))
                 for _ in range(int(rng.integers(0, 3)))]
        piece = max(1, int(rng.integers(1, len(stream) + 2)))
        idat = b"".join(pc.chunk(b"IDAT", stream[i:i + piece]) for i in range(0, len(stream), piece))
        return data[:8] + cs[0] + b"".join(extra) + idat + cs[-1]
    if kind == 0:
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(len(b)))] ^= 1 << int(rng.integers(8))
    elif kind == 1:
        del b[int(rng.integers(1, len(b))):]
    elif kind == 2:   # a chunk length rewritten (CRC left alone)
        cs = chunks(data); k = int(rng.integers(len(cs))); off = 8 + sum(len(x) for x in cs[:k])
        b[off:off + 4] = struct.pack(">I", int(rng.integers(0, 1 << int(rng.integers(1, 32)))))
    elif kind == 3:   # chunks shuffled / duplicated / dropped, framing intact
        cs = chunks(data); order = list(rng.permutation(len(cs)))[:int(rng.integers(1, len(cs) + 2))]
        b = bytearray(data[:8] + b"".join(cs[i % len(cs)] for i in order))
    elif kind == 4:   # IHDR fields rewritten with a correct CRC
        f = bytearray(data[16:29]); f[int(rng.integers(13))] = int(rng.integers(256))
        b[8:33] = pc.chunk(b"IHDR", bytes(f))
    elif kind == 5:
        i = int(rng.integers(len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
    else:
        b = bytearray(bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8)))
    return bytes(b)

rng = np.random.Generator(np.random.PCG64(20261004))
names = sorted(files)
ok = bad = 0
for it in range(3000):
    name = names[int(rng.integers(len(names)))]
    h, w = expected[name].shape
    batch = [mutate(files[name], rng) for _ in range(int(rng.integers(1, 4)))]
    if it % 7 == 0:
        batch.append(files[name])
    st, _ = check(batch, w, h)
    assert st in (0, 1, 4), st
    ok += st == 0
    bad += st != 0
# The same kinds of damage, one file at a time, held against the real libpng driven as cv::imdecode drives it
# (tests/png_ref.py): what libpng reads the parser must accept, and what the parser accepts with its compressed data whole
# libpng must read (a damaged stream is the device's to judge: tests/test_gpu_png.py, tools/stress_png.py).
import png_ref
agree = 0
if png_ref.available():
    for it in range(3000):
        name = names[int(rng.integers(len(names)))]
        h, w = expected[name].shape
        f = mutate(files[name], rng)
        st, _ = check([f], w, h)
        ref, _, _ = png_ref.imdecode_gray(f, w, h)
        if ref == 0:  # (4: a zlib header declaring a window below 32 KiB -- refused as unsupported whatever its matches do)
            assert st in (0, 4), (it, name, st, "libpng reads it")
        if st == 0:
            try:
                whole = zlib.decompress(pc.idat_stream(f)) == zlib.decompress(pc.idat_stream(files[name]))
            except zlib.error:
                whole = False
            if whole:
                assert ref == 0, (it, name, ref, png_ref.last_error())
                agree += 1
    print("libpng %s agrees on %d accepted files" % (png_ref.version(), agree))
for junk in (b"", b"\x89", b"\x89PNG\r\n\x1a\n", b"\x89PNG\r\n\x1a\n" + bytes(40), bytes(100)):
    assert check([junk], 8, 8)[0] == 1
print("done ok=%d refused=%d" % (ok, bad))
'''


def test_png_host_parser_under_asan_and_ubsan(tmp_path):
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (asan_rt and Path(asan_rt).exists() and ubsan_rt and Path(ubsan_rt).exists()):
        pytest.skip("no sanitizer runtime in this toolchain")
    r = subprocess.run(["make", "-s", "-C", str(CSRC), "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and LIB.exists(), r.stderr[-2000:]
    script = tmp_path / "drive.py"
    script.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD="%s %s" % (asan_rt, ubsan_rt),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, str(script), str(LIB), str(GOLD), str(Path(__file__).resolve().parent)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-4000:])
    assert p.stdout.strip().splitlines()[-1].startswith("done")
    if "agrees on" in p.stdout:  # (the real libpng was there to be asked)
        assert int(p.stdout.split("agrees on")[1].split()[0]) > 300, p.stdout
    ok = int(p.stdout.split("ok=")[1].split()[0])
    refused = int(p.stdout.split("refused=")[1].split()[0])
    assert ok > 300 and refused > 500, (ok, refused)


def test_the_stated_rule_for_damaged_streams_is_libpngs():
    """tests/png_craft.py zlib_reference (the rule the device decoder follows, written out in zlib calls: fatal while rows are
    wanted and in the rest of the piece behind the last byte, a drain afterwards in which only running out of IDAT data is
    fatal) against the real libpng driven as cv::imdecode drives it, on 800 damaged files."""
    import png_ref
    if not png_ref.available():
        pytest.skip("no libpng16.so.16 to build tests/cpp/png_ref.c against")
    rng = np.random.Generator(np.random.PCG64(77))
    refused = read = 0
    for it in range(800):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8) if it % 3 else np.tile(np.arange(w, dtype=np.uint8), (h, 1))
        rows = pc.filter_rows(img, 1, rng.integers(0, 5, h))
        extra = [0, 0, 1, 600, 40000][int(rng.integers(5))]
        stream = pc.deflate(rows + bytes(rng.integers(0, 256, extra, dtype=np.uint8)), int(rng.choice([0, 1, 6, 9])),
                            [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY][int(rng.integers(4))])
        how = int(rng.integers(5))
        if how == 0:
            stream = pc.mutate_stream(stream, rng)
        elif how == 1:
            cut = max(2, len(stream) - int(rng.integers(1, 300)))
            stream = stream[:cut] + pc.mutate_stream(b"xx" + stream[cut:], rng)[2:]
        elif how == 2:
            stream += bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        elif how == 3:
            stream = stream[:max(2, len(stream) - int(rng.integers(1, 12)))]
        if len(stream) < 8:
            continue
        hdr_ok = (stream[0] & 15) == 8 and (stream[0] >> 4) <= 7 and not (stream[1] & 32) and ((stream[0] << 8) | stream[1]) % 31 == 0
        piece = [None, max(1, len(stream) - 4), max(1, len(stream) - 7), 100, 5000, 8192, 9000][int(rng.integers(7))]
        f = pc.write_png(None, w, h, 8, 0, stream=stream, idat_piece=piece)
        ref_status, ref_img, _ = png_ref.imdecode_gray(f, w, h)
        mine = pc.zlib_reference(pc.idat_pieces(f), (w + 1) * h) if hdr_ok else None
        if mine is not None and np.frombuffer(mine, np.uint8).reshape(h, w + 1)[:, 0].max() > 4:
            mine = None  # a filter type that does not exist: png_error in png_read_row
        assert (mine is None) == (ref_status != 0), (it, how, piece, ref_status, png_ref.last_error())
        if mine is not None:
            np.testing.assert_array_equal(unfilter_numpy(mine, h, w, 1), ref_img, err_msg="file %d" % it)
            read += 1
        else:
            refused += 1
    assert read > 150 and refused > 150, (read, refused)
