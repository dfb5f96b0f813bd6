#!/usr/bin/env python3
"""Throughput of the JPEG ingest (row f4): a batch of 640x480 gray JPEGs (quality 80 / 90, libjpeg-turbo-encoded
synthetic scenes) through vsf_jpeg_decode_gray_batch, alone and beside the extraction of another batch on a second
context / stream.  python tools/time_jpeg.py [n_images]"""
import io
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from PIL import Image

from vision_slam_frontend_amd import capi, synth

import ctypes as C

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, H = 640, 480
dev = torch.device("cuda", 0)
base = synth.bench_batch(16, W, H, n_scenes=4).reshape(-1, H, W)
for q, extra in ((80, {}), (90, {}), (80, {"restart_marker_rows": 1})):
    files = []
    for i in range(N):
        b = io.BytesIO()
        Image.fromarray(base[i % len(base)], "L").save(b, "JPEG", quality=q, **extra)
        files.append(b.getvalue())
    kb = sum(len(f) for f in files) / N / 1024
    ctx = capi.Context(capi.default_params(W, H, max_images=2, nfeatures=2000))
    s = torch.cuda.Stream(device=dev)
    ctx.set_stream(s.cuda_stream)
    d = torch.zeros((N, H, W), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.jpeg_decode_gray_batch(files, W, H, d.data_ptr(), W * H, W)
    ctx.sync()
    reps = 5
    # the C call itself (the ctypes convenience wrapper copies every file once more)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * N)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * N)(*[len(b) for b in bufs])
    call = lambda: capi.lib().vsf_jpeg_decode_gray_batch(ctx._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), N,
                                                         W, H, C.c_void_p(d.data_ptr()), W * H, W)
    assert call() == 0
    ctx.sync()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(reps):
        h0 = time.perf_counter()
        assert call() == 0
        host += time.perf_counter() - h0
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(("restart interval = one MCU row, " if extra else "") + "quality %d (%.0f KB per image): %d images in %.2f ms = %.0f images/s = %.0f stereo frames/s "
          "(host parse + staging %.2f ms of it, not overlapped here)" % (q, kb, N, dt * 1e3, N / dt, N / dt / 2, host / reps * 1e3))
    # beside the extraction of another batch (different context and stream)
    ectx = capi.Context(capi.default_params(W, H, max_images=N, nfeatures=2000))
    es = torch.cuda.Stream(device=dev)
    ectx.set_stream(es.cuda_stream)
    K = ectx.params.max_keypoints
    imgs = torch.from_numpy(np.ascontiguousarray(synth.bench_batch(N // 2, W, H, n_scenes=4))).to(dev)
    kp = torch.zeros((N, K, 28), dtype=torch.uint8, device=dev)
    de = torch.zeros((N, K, 32), dtype=torch.uint8, device=dev)
    cn = torch.zeros(N, dtype=torch.int32, device=dev)
    m = torch.zeros((N // 2, K, 16), dtype=torch.uint8, device=dev)
    nm = torch.zeros(N // 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    run_e = lambda: ectx.stereo_batch_dev(imgs.data_ptr(), N // 2, W * H, W, kp.data_ptr(), de.data_ptr(), cn.data_ptr(),
                                          m.data_ptr(), nm.data_ptr())
    for _ in range(2):
        run_e()
    ectx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run_e()
    ectx.sync()
    alone = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        assert call() == 0
        run_e()
    ectx.sync()
    ctx.sync()
    both = (time.perf_counter() - t0) / reps
    print("   extraction + stereo match of %d frames alone %.2f ms; with the JPEG decode of %d images beside it %.2f ms"
          % (N // 2, alone * 1e3, N, both * 1e3))
    ctx.close()
    ectx.close()
