#!/usr/bin/env python3
"""Randomised stress of vsf_png_decode_gray_batch (run by hand on a GPU box): random sizes, colour types (gray, gray + alpha,
RGB, RGBA, palette) and bit depths (1 / 2 / 4 / 8 / 16), gamma / sRGB / cHRM chunks beside the colour ones, Adam7 for one file in five, row filters, zlib
strategies and levels, IDAT chunkings, and -- for two files in three -- damage: bit flips,
cuts, zeroed runs and insertions in the compressed data, data that goes on behind the image, trailing garbage, streams cut
inside their last bytes.

Reference for every file: the real libpng, driven as cv::imdecode(IMREAD_GRAYSCALE) drives it (tests/png_ref.py:
png_read_info ... png_read_image, png_read_end); for undamaged files PIL as well.  A file must be flagged (vsf_sync ->
VSF_ERR_INVALID_ARG, or refused by the host parser) exactly when libpng refuses it, and decode to libpng's bytes otherwise.
    python tools/stress_png.py [n_cases] [seed]"""
import io
import struct
import sys
import zlib
from pathlib import Path

import numpy as np
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import png_craft as pc  # noqa: E402
import png_ref  # noqa: E402
from vision_slam_frontend_amd import capi  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 17))
dev = torch.device("cuda", 0)
bad = flagged = clean = 0
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=100))
for c in range(n_cases):
    w = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(1, 12))
    h = int(rng.integers(1, 500)) if rng.random() < 0.8 else int(rng.integers(1, 6))
    depth, ctype = [(8, 0), (8, 0), (8, 0), (16, 0), (1, 0), (2, 0), (4, 0), (8, 4), (16, 4), (8, 2), (8, 2), (8, 6), (16, 2), (16, 6),
                    (8, 3), (4, 3), (2, 3), (1, 3)][int(rng.integers(18))]
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    smooth = rng.random() < 0.5
    shape = (h, w) + ((channels,) if channels > 1 else ())
    if depth <= 8:
        base = rng.integers(0, 1 << depth, shape)
        if smooth and depth == 8:
            yy, xx = np.mgrid[0:h, 0:w]
            ramp = ((xx * 2 + yy) % 256)
            base = (ramp[..., None] + base // 16) % 256 if channels > 1 else (ramp + base // 16) % 256
        raw = pc.pack_samples(base.astype(np.uint8), depth)
    else:
        base = rng.integers(0, 65536, shape)
        raw = pc.pack_samples(base.astype(np.uint16), 16)
    # colour files: the chunks that decide how libpng's rgb_to_gray weights the samples (16-bit: only where no table is built)
    before = []
    if ctype in (2, 3, 6):
        pick = int(rng.integers(8))
        gam = lambda v: pc.chunk(b"gAMA", struct.pack(">I", v))
        srgb_chrm = pc.chunk(b"cHRM", struct.pack(">8I", 31270, 32900, 64000, 33000, 30000, 60000, 15000, 6000))
        if depth == 16:
            before = [[], [gam(100000)], [gam(int(rng.integers(95300, 104900)))], [pc.chunk(b"cHRM", bytes(rng.integers(0, 256, 32, dtype=np.uint8)))]][pick % 4]
        else:
            before = [[], [gam(45455)], [gam(int(rng.integers(16, 400000)))], [pc.chunk(b"sRGB", bytes([int(rng.integers(4))]))],
                      [srgb_chrm, gam(int(rng.integers(20000, 300000)))], [pc.chunk(b"sRGB", b"\x00"), gam(int(rng.integers(20000, 300000)))],
                      [gam(45455), pc.chunk(b"sRGB", b"\x01"), srgb_chrm], [pc.chunk(b"cHRM", bytes(rng.integers(0, 256, 32, dtype=np.uint8)))]][pick]
        if ctype == 3:
            entries = int(rng.integers(1, (1 << depth) + 1))
            before = before + [pc.chunk(b"PLTE", bytes(rng.integers(0, 256, 3 * entries, dtype=np.uint8)))]
            if rng.random() < 0.3:
                before.append(pc.chunk(b"tRNS", bytes(rng.integers(0, 256, entries, dtype=np.uint8))))
    row_bytes = raw.shape[1]
    bpp = max(1, depth * channels // 8)
    filters = [np.arange(h) % 5, rng.integers(0, 5, h), np.full(h, int(rng.integers(5)))][int(rng.integers(3))]
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED][int(rng.integers(5))]
    level = int(rng.choice([0, 1, 6, 9]))
    lace = int(rng.random() < 0.2)
    rows = pc.adam7_filtered(base, depth, ctype, rng) if lace else pc.filter_rows(raw, bpp, filters)
    how = int(rng.integers(7))
    extra = bytes(rng.integers(0, 256, [0, 0, 0, 1, 500, 70000][int(rng.integers(6))], dtype=np.uint8)) if how in (3, 4, 5) else b""
    stream = pc.deflate(rows + extra, level, strategy)
    if how == 1 or how == 4:
        stream = pc.mutate_stream(stream, rng)
    elif how == 2:
        cut = max(2, len(stream) - int(rng.integers(1, 200)))
        stream = stream[:cut] + pc.mutate_stream(b"xx" + stream[cut:], rng)[2:]
    elif how == 5:
        stream += bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8))
    elif how == 6:  # cut inside the last bytes: the end-of-block code, the check value
        stream = stream[:max(2, len(stream) - int(rng.integers(1, 12)))]
    if len(stream) < 8:
        continue
    piece = [None, max(1, len(stream) - 4), max(1, len(stream) - 9), int(rng.integers(1, 400)) if w * h < 20000 else 3000, 8192, 8193, 20000][int(rng.integers(7))]
    f = pc.write_png(None, w, h, depth, ctype, stream=stream, idat_piece=piece, extra_before=before, interlace=lace)
    ref_status, want, _ = png_ref.imdecode_gray(f, w, h)
    if ref_status != 0:
        want = None
    elif how == 0:  # an undamaged file: PIL agrees
        im = Image.open(io.BytesIO(f))
        im.load()
        a = np.asarray(im)
        if ctype in (0, 4):
            pil = (a.astype(np.uint32) >> 8).astype(np.uint8) if im.mode.startswith("I") else \
                a[:, :, 0] if a.ndim == 3 else (a.astype(np.uint8) * 255 if im.mode == "1" else a)
            assert np.array_equal(pil, want), "PIL disagrees with libpng (case %d)" % c
    pitch = (w + 3) // 4 * 4
    d = torch.full((h, pitch), 0x5A, dtype=torch.uint8, device=dev)
    st = ctx.png_decode_gray_batch([f], w, h, d.data_ptr(), h * pitch, pitch, allow_status=(capi.VSF_ERR_INVALID_ARG,))
    if st != capi.VSF_OK:  # the host refused it
        ok = want is None
        note = "refused by the host" if ok else "REFUSED BY THE HOST, READ BY LIBPNG"
    else:
        try:
            sync = ctx.sync()
        except capi.VsfError as e:
            sync = e.status
        if want is None:
            ok = sync == capi.VSF_ERR_INVALID_ARG
            flagged += 1
            note = "flagged" if ok else "NOT FLAGGED"
        else:
            ok = sync == capi.VSF_OK and np.array_equal(d.cpu().numpy()[:, :w], want)
            clean += 1
            note = "ok" if ok else ("FLAGGED" if sync != capi.VSF_OK else "WRONG BYTES")
    bad += not ok
    if not ok or c % 50 == 0:
        print("case %4d %3dx%-3d depth %2d type %d%s level %d strategy %d piece %s damage %d: %s" %
              (c, w, h, depth, ctype, " Adam7" if lace else "", level, strategy, piece, how, note), flush=True)
ctx.close()
print("libpng %s; decoded and equal: %d, flagged where libpng refuses: %d, mismatches: %d of %d" % (png_ref.version(), clean, flagged, bad, n_cases))
sys.exit(1 if bad else 0)
