"""Matcher micro-benchmark: 128 pairs of 2000 x 2000 random descriptors, and 8 pairs (the split-train
path of small launches), ms per launch via hipEvents."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from vision_slam_frontend_amd import capi, synth

dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=2000))
K = ctx.params.max_keypoints
for npairs in (128, 8):
    d = torch.zeros((2 * npairs, K, 32), dtype=torch.uint8, device=dev)
    d[:, :2000] = torch.from_numpy(synth.random_descriptors(2 * npairs * 2000).reshape(2 * npairs, 2000, 32)).to(dev)
    counts = torch.full((2 * npairs,), 2000, dtype=torch.int32, device=dev)
    m = torch.zeros((npairs, K, 16), dtype=torch.uint8, device=dev)
    nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    run = lambda: ctx.match_batch_dev(d.data_ptr(), counts.data_ptr(), K * 32, 0, 0, npairs, 0, 0, m.data_ptr(), nm.data_ptr())
    for _ in range(3): run()
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(20): run()
    st = ctx.profile_read()
    ctx.profile_enable(False)
    ms = st["hamming_knn2"][0] / 20
    print("%3d pairs of 2000 x 2000: knn2 %.4f ms per launch, %.2f T pair-distances/s, %.0f int8 TOP/s" %
          (npairs, ms, npairs * 4e6 / ms / 1e9, npairs * 4e6 * 512 / ms / 1e9))
ctx.close()
