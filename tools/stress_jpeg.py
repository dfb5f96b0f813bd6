#!/usr/bin/env python3
"""Randomised JPEG parity stress (run by hand on a GPU box): random sizes up to 1500 x 1100, gray / 4:4:4 / 4:2:2 / 4:2:0,
baseline and progressive, qualities 1..100, optimised tables, restart intervals, smooth / noisy / flat content, batches of
1..6 files of one size per call -- vsf_jpeg_decode_gray_batch against libjpeg-turbo itself (Pillow, JCS_GRAYSCALE), bit for
bit.  python tools/stress_jpeg.py [n_cases] [seed]"""
import io
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image, ImageFile

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vision_slam_frontend_amd import capi, synth  # noqa: E402

ImageFile.MAXBLOCK = 1 << 25  # (libjpeg's progressive / optimising encoder wants the whole file in one buffer)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500))
bad = 0
for c in range(n_cases):
    big = rng.random() < 0.2
    w = int(rng.integers(1, 1500 if big else 400))
    h = int(rng.integers(1, 1100 if big else 300))
    files, want = [], []
    for _ in range(int(rng.integers(1, 7))):
        kind = rng.random()
        if kind < 0.4:
            img = synth.stereo_pair(max(w, 16), max(h, 16), int(rng.integers(0, 1000)), n_objects=int(rng.integers(5, 400)))[0][:h, :w]
        elif kind < 0.7:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind < 0.9:
            yy, xx = np.mgrid[0:h, 0:w]
            img = ((np.sin(xx / 9.0) + np.cos(yy / 13.0)) * 60 + 128).clip(0, 255).astype(np.uint8)
        else:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        kw = dict(quality=int(rng.integers(1, 101)))
        if rng.random() < 0.5:
            kw["progressive"] = True
        elif rng.random() < 0.5:
            kw["optimize"] = True
        r = rng.random()
        if r < 0.25:
            kw["restart_marker_blocks"] = int(rng.integers(1, 40))
        elif r < 0.35:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        b = io.BytesIO()
        if rng.random() < 0.35:
            rgb = np.stack([img, np.roll(img, 3, 0), 255 - img], 2)
            kw["subsampling"] = int(rng.integers(0, 3))
            Image.fromarray(np.ascontiguousarray(rgb), "RGB").save(b, "JPEG", **kw)
        else:
            Image.fromarray(np.ascontiguousarray(img), "L").save(b, "JPEG", **kw)
        f = b.getvalue()
        im = Image.open(io.BytesIO(f))
        im.draft("L", im.size)
        files.append(f)
        want.append(np.asarray(im.convert("L") if im.mode != "L" else im))
    pitch = (w + 3) // 4 * 4
    d = torch.full((len(files), h, pitch), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.jpeg_decode_gray_batch(files, w, h, d.data_ptr(), h * pitch, pitch)
    st = ctx.sync()
    got = d.cpu().numpy()
    ok = st == capi.VSF_OK and all(np.array_equal(got[i, :, :w], want[i]) for i in range(len(files)))
    bad += not ok
    n_prog = sum(b"\xff\xc2" in f[:1000] for f in files)
    if not ok or c % 25 == 0 or c == n_cases - 1:
        print("case %3d %4dx%-4d files %d (progressive %d) status %d %s" % (c, w, h, len(files), n_prog, st, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches: %d of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
