"""64 pairs of 10000 x 10000 through the matcher, a few launches: for rocprofv3 --pmc runs (tools/pmc_summary.py reads them)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
from vision_slam_frontend_amd import capi, synth
n, npairs = 10000, 64
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=n))
K = ctx.params.max_keypoints
d = torch.zeros((2 * npairs, K, 32), dtype=torch.uint8, device=dev)
rnd = synth.random_descriptors(2 * 16 * n).reshape(32, n, 32)
d[:, :n] = torch.from_numpy(np.tile(rnd, (npairs // 16, 1, 1))).to(dev)
counts = torch.full((2 * npairs,), n, dtype=torch.int32, device=dev)
m = torch.zeros((npairs, K, 16), dtype=torch.uint8, device=dev)
nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(4):
    ctx.match_batch_dev(d.data_ptr(), counts.data_ptr(), K * 32, 0, 0, npairs, 0, 0, m.data_ptr(), nm.data_ptr())
ctx.sync()
ctx.close()
