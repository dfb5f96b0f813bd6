#!/usr/bin/env python3
"""Throughput of the progressive-JPEG decode (row f4, SOF2 files): a batch of 640x480 progressive files (libjpeg-turbo's
default scan script, synthetic scenes) through vsf_jpeg_decode_gray_batch, next to the same images as baseline files.
python tools/time_jpeg_progressive.py [n_images] [serial]      (serial: vsf_debug_jpeg_serial, every file scan after scan in one wave)"""
import ctypes as C
import io
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from PIL import Image

from vision_slam_frontend_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SERIAL = len(sys.argv) > 2 and sys.argv[2] == "serial"
W, H = 640, 480
dev = torch.device("cuda", 0)
base = synth.bench_batch(16, W, H, n_scenes=4).reshape(-1, H, W)
for label, mode, kw in (("baseline gray q80", "L", dict(quality=80)),
                        ("progressive gray q80", "L", dict(quality=80, progressive=True)),
                        ("progressive gray q95", "L", dict(quality=95, progressive=True)),
                        ("progressive 4:2:0 q80", "RGB", dict(quality=80, progressive=True, subsampling=2))):
    files = []
    for i in range(N):
        img = base[i % len(base)]
        b = io.BytesIO()
        Image.fromarray(img if mode == "L" else np.stack([img, np.roll(img, 9, 1), 255 - img], 2), mode).save(b, "JPEG", **kw)
        files.append(b.getvalue())
    kb = sum(len(f) for f in files) / N / 1024
    ctx = capi.Context(capi.default_params(W, H, max_images=2, nfeatures=2000))
    if SERIAL:
        ctx.debug_jpeg_serial(1)
    d = torch.zeros((N, H, W), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * N)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * N)(*[len(b) for b in bufs])
    call = lambda: capi.lib().vsf_jpeg_decode_gray_batch(ctx._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), N,
                                                         W, H, C.c_void_p(d.data_ptr()), W * H, W)
    for _ in range(2):
        assert call() == 0
    ctx.sync()
    got = d[:4].cpu().numpy()
    for i in range(4):
        im = Image.open(io.BytesIO(files[i]))
        im.draft("L", im.size)
        assert np.array_equal(got[i], np.asarray(im.convert("L") if im.mode != "L" else im)), (label, i)
    reps = 5
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(reps):
        h0 = time.perf_counter()
        assert call() == 0
        host += time.perf_counter() - h0
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print("%-24s (%.0f KB per image): %d images in %.2f ms = %.0f images/s = %.0f stereo frames/s (inside the calls "
          "%.2f ms of it, waits for the staging buffer included); first 4 == libjpeg-turbo" % (label, kb, N, dt * 1e3, N / dt, N / dt / 2, host / reps * 1e3), flush=True)
    ctx.close()
