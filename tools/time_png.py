#!/usr/bin/env python3
"""Throughput of the PNG ingest (row f4): a batch of 640x480 gray PNGs (the bench's synthetic scenes and photographs from
tests/golden/real, written by PIL = libpng at compression levels 1 / 6 / 9) through vsf_png_decode_gray_batch, outputs
compared with the source images; per-kernel time through the context's stream.  python tools/time_png.py [n_images]"""
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
W, H = 640, 480
dev = torch.device("cuda", 0)
root = Path(__file__).resolve().parent.parent / "tests" / "golden" / "real"
photos = [np.asarray(Image.open(f)) for f in sorted(root.glob("*.png"))]
photos = np.stack([a for a in photos if a.shape == (H, W) and a.dtype == np.uint8])
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import zlib

import png_craft as pc
import png_ref

sets = {"synthetic scenes": synth.bench_batch(16, W, H, n_scenes=4).reshape(-1, H, W), "photographs": photos}
for name, base in sets.items():
    # levels 1 / 6 / 9: PIL = libpng with adaptive filters and zlib's default strategy; "cv": what cv::imencode writes when no
    # level is given (grfmt_png.cpp: Z_BEST_SPEED, the Sub filter, Z_RLE -- matches of distance 1 only)
    for level in (1, 6, 9, "cv"):
        files, srcs = [], []
        cache = {}
        for i in range(N):
            k = i % len(base)
            if k not in cache:
                if level == "cv":
                    cache[k] = pc.gray8(base[k], filters=np.full(H, 1), level=1, strategy=zlib.Z_RLE, idat_piece=8192)
                else:
                    b = io.BytesIO()
                    Image.fromarray(base[k], "L").save(b, "PNG", compress_level=level)
                    cache[k] = b.getvalue()
            files.append(cache[k])
            srcs.append(k)
        kb = sum(len(f) for f in files) / N / 1024
        ctx = capi.Context(capi.default_params(W, H, max_images=2, nfeatures=2000))
        s = torch.cuda.Stream(device=dev)
        ctx.set_stream(s.cuda_stream)
        d = torch.zeros((N, H, W), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        bufs = [np.frombuffer(f, np.uint8) for f in files]
        ptrs = (C.c_void_p * N)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * N)(*[len(b) for b in bufs])
        call = lambda: capi.lib().vsf_png_decode_gray_batch(ctx._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), N,
                                                            W, H, C.c_void_p(d.data_ptr()), W * H, W)
        for _ in range(2):
            assert call() == 0
        assert ctx.sync() == capi.VSF_OK
        got = d.cpu().numpy()
        assert all(np.array_equal(got[i], base[srcs[i]]) for i in range(N)), "decode differs from the source images"
        reps = 5
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(reps):
            h0 = time.perf_counter()
            assert call() == 0
            host += time.perf_counter() - h0
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        # the same files through the real libpng on one host core, driven as cv::imdecode drives it (tests/png_ref.py)
        cpu = ""
        if png_ref.available():
            t0 = time.perf_counter()
            for f in files[:32]:
                assert png_ref.imdecode_gray(f, W, H)[0] == 0
            cpu = "; libpng %s on one host core: %.0f images/s" % (png_ref.version(), 32 / (time.perf_counter() - t0))
        print("%s, level %s (%.0f KB per image): %d images in %.2f ms = %.0f images/s (host chunk walk + CRC + staging %.2f ms of it)%s"
              % (name, level, kb, N, dt * 1e3, N / dt, host / reps * 1e3, cpu), flush=True)
        ctx.close()
