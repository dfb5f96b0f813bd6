#!/usr/bin/env python3
"""Times vsf_bayer_bg_to_gray_batch_dev (row f4 behind imdecode) on a batch resident in HBM:
python tools/time_ingest.py [n_images] [width] [height]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vision_slam_frontend_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = int(sys.argv[2]) if len(sys.argv) > 2 else 640
h = int(sys.argv[3]) if len(sys.argv) > 3 else 480
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500))
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
src = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
dst = torch.zeros_like(src)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for _ in range(3):
        ctx.bayer_bg_to_gray_batch_dev(src.data_ptr(), n, w, h, w * h, w, dst.data_ptr(), w * h, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    reps = 20
    for _ in range(reps):
        ctx.bayer_bg_to_gray_batch_dev(src.data_ptr(), n, w, h, w * h, w, dst.data_ptr(), w * h, w)
    e1.record(st)
st.synchronize()
ms = e0.elapsed_time(e1) / reps
print("%d x %dx%d: %.3f ms per batch, %.0f GB/s algorithmic (1 B read + 1 B written per pixel), %.2f us per image" % (
    n, w, h, ms, 2.0 * n * w * h / ms / 1e6, 1e3 * ms / n))
# reference point: a plain device copy of the same bytes
with torch.cuda.stream(st):
    for _ in range(3):
        dst.copy_(src)
    e0.record(st)
    for _ in range(reps):
        dst.copy_(src)
    e1.record(st)
st.synchronize()
ms_c = e0.elapsed_time(e1) / reps
print("  (torch copy of the same batch: %.3f ms, %.0f GB/s)" % (ms_c, 2.0 * n * w * h / ms_c / 1e6))
