#!/usr/bin/env python3
"""Device time of the SURVEY 8(f) f1 entry points on a 64-frame batch (640x480, 2000 kp): RemoveAmbigStereo on the
stereo batch's outputs, then GetFeatureMatches of every frame against its up-to-8 predecessors (one call)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vision_slam_frontend_amd import capi, synth  # noqa: E402

B, W, H, NF, WIN = 64, 640, 480, 2000, 8
dev = torch.device("cuda", 0)
p = capi.default_params(W, H, max_images=2 * B, nfeatures=NF)
ctx = capi.Context(p)
K = ctx.params.max_keypoints
stream = torch.cuda.Stream(device=dev)  # (the default stream's handle 0 would mean "the context's own stream")
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
frames = synth.bench_batch(B, W, H)
img = torch.from_numpy(frames).to(dev)
z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device=dev)
kp, desc, counts = z(2 * B, K, 28), z(2 * B, K, 32), z(2 * B, dt=torch.int32)
m, nm = z(B, K, 16), z(B, dt=torch.int32)
means, thr = z(B, dt=torch.float32), z(B + 1, dt=torch.float32)
kp2, desc2, counts2 = z(2 * B, K, 28), z(2 * B, K, 32), z(2 * B, dt=torch.int32)
F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
qs, ts = [], []
for f in range(B):
    for g in range(max(0, f - WIN), f):
        qs.append(2 * g)
        ts.append(2 * f)
q_set = torch.tensor(qs, dtype=torch.int32, device=dev)
t_set = torch.tensor(ts, dtype=torch.int32, device=dev)
npairs = len(qs)
pairs, npr = z(npairs, K, 2, dt=torch.int64), z(npairs, dt=torch.int32)
torch.cuda.synchronize()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_stereo = timed(lambda: ctx.stereo_batch_dev(img.data_ptr(), B, W * H, W, kp.data_ptr(), desc.data_ptr(),
                                              counts.data_ptr(), m.data_ptr(), nm.data_ptr()))
t_filter = timed(lambda: ctx.remove_ambig_stereo_batch_dev(kp.data_ptr(), desc.data_ptr(), m.data_ptr(), nm.data_ptr(), B,
                                                           F, 10000.0, 0, means.data_ptr(), thr.data_ptr(),
                                                           kp2.data_ptr(), desc2.data_ptr(), counts2.data_ptr()))
t_fm = timed(lambda: ctx.feature_matches_batch_dev(desc2.data_ptr(), counts2.data_ptr(), K * 32, q_set.data_ptr(),
                                                   t_set.data_ptr(), npairs, float(np.float32(0.3)), pairs.data_ptr(),
                                                   npr.data_ptr()))
ctx.sync()
print("frames %d, stereo matches/frame %.0f, kept/frame %.0f, temporal pairs %d, factors/pair %.0f" %
      (B, nm.float().mean().item(), counts2[::2].float().mean().item(), npairs, npr.float().mean().item()))
print("stereo_batch_dev %.3f ms | remove_ambig_stereo_batch_dev %.3f ms | feature_matches_batch_dev (%d pairs) %.3f ms"
      % (t_stereo, t_filter, npairs, t_fm))
ctx.set_stream(None)
ctx.close()
