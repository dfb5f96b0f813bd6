"""Matcher micro-benchmark: pairs of N x N random descriptors -- 256 pairs of 2000 x 2000 (the bench's stereo
launch), 8 pairs (the split-train path of small launches), 64 pairs of 10000 x 10000 (the reference's own nfeatures); ms per
launch via hipEvents.  (Until round 5 it also ran round 2's int8 form, since retired: tools/exp/retired/.)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from vision_slam_frontend_amd import capi, synth

dev = torch.device("cuda", 0)
for n, npairs in ((2000, 256), (2000, 8), (10000, 64)):
    ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=n))
    K = ctx.params.max_keypoints
    d = torch.zeros((2 * npairs, K, 32), dtype=torch.uint8, device=dev)
    rnd = synth.random_descriptors(2 * min(npairs, 16) * n).reshape(2 * min(npairs, 16), n, 32)
    d[:, :n] = torch.from_numpy(np.tile(rnd, (npairs // min(npairs, 16), 1, 1))).to(dev)
    counts = torch.full((2 * npairs,), n, dtype=torch.int32, device=dev)
    out = {}
    for name in ("fp4",):
        m = torch.zeros((npairs, K, 16), dtype=torch.uint8, device=dev)
        nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
        idx = torch.zeros((npairs, K, 2), dtype=torch.int32, device=dev)
        dist = torch.zeros((npairs, K, 2), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        run = lambda: ctx.match_batch_dev(d.data_ptr(), counts.data_ptr(), K * 32, 0, 0, npairs, idx.data_ptr(), dist.data_ptr(),
                                          m.data_ptr(), nm.data_ptr())
        for _ in range(3):
            run()
        ctx.sync()
        ctx.profile_enable(True)
        for _ in range(10):
            run()
        st = ctx.profile_read()
        ctx.profile_enable(False)
        ms = st["hamming_knn2"][0] / 10
        out[name] = (idx[:, :n].cpu().numpy(), dist[:, :n].cpu().numpy(), nm.cpu().numpy())
        print("%3d pairs of %5d x %5d, %-4s: knn2 %.4f ms per launch, %.2f T pair-distances/s (%.0f TOP/s of 1-bit multiply-adds x 512)"
              % (npairs, n, n, name, ms, npairs * n * n / ms / 1e9, npairs * n * n * 512 / ms / 1e9))
    print("    matches per pair ~%d" % int(out["fp4"][2].mean()))
    ctx.close()
