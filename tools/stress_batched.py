#!/usr/bin/env python3
"""Randomised parity stress of the BATCHED entry point (run by hand on a GPU box): random sizes / feature counts / batch
sizes, vsf_stereo_batch_dev with FAST in its resident form (2..4 waves per SIMD), cross-call pipelining on or off and
repeated calls, against (a) the grid form in a fresh context, every output byte, and (b) the oracle on two frames.
    python tools/stress_batched.py [n_cases] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import binding as ob  # noqa: E402
from vision_slam_frontend_amd import capi, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ob.build()
dev = torch.device("cuda", 0)


def run(frames, nf, resident, pipeline, repeats):
    B, _, H, W = frames.shape
    p = capi.default_params(W, H, max_images=2 * B, nfeatures=nf)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
        kp = torch.zeros((2 * B, K, 28), dtype=torch.uint8, device=dev)
        desc = torch.zeros((2 * B, K, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(2 * B, dtype=torch.int32, device=dev)
        m = torch.zeros((B, K, 16), dtype=torch.uint8, device=dev)
        nm = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.set_pipeline(pipeline)
        if resident is not None:
            ctx.set_fast_resident(resident)
        torch.cuda.synchronize()
        for _ in range(repeats):
            ctx.stereo_batch_dev(d_img.data_ptr(), B, W * H, W, kp.data_ptr(), desc.data_ptr(), cnt.data_ptr(),
                                 m.data_ptr(), nm.data_ptr())
        st = ctx.sync(allow_capacity=True)
        return st, [t.cpu().numpy() for t in (kp, desc, cnt, m, nm)]


bad = 0
for c in range(n_cases):
    w = 16 * int(rng.integers(8, 50))  # (device batches take rows and images at 16-byte multiples)
    h = int(rng.integers(90, 560))
    nf = int(rng.choice([100, 500, 1000, 2000, 4000]))
    B = int(rng.integers(16, 28))
    if c % 4 == 3:  # a large batch of a wide geometry: the batched selection's 9 216-entry class (>= 1 536 wide-level workgroups)
        w = 16 * int(rng.integers(36, 50))
        h = int(rng.integers(430, 560))
        B = int(rng.integers(72, 100))
    resident = int(rng.choice([2, 3, 3, 4]))
    pipeline = bool(rng.integers(0, 2))
    repeats = int(rng.integers(1, 4))
    frames = synth.bench_batch(B, w, h, seed=int(rng.integers(0, 1 << 30)), n_scenes=3)
    st0, ref = run(frames, nf, 0, False, 1)
    st1, out = run(frames, nf, resident, pipeline, repeats)
    st2, auto = run(frames, nf, None, pipeline, 4)
    same = st0 == st1 == st2 and all(np.array_equal(a, b) for a, b in zip(ref, out)) and \
        all(np.array_equal(a, b) for a, b in zip(ref, auto))
    ok = same
    for f in (0, B - 1):
        for side in (0, 1):
            o = ob.Orb(nfeatures=nf)
            o.run(frames[f, side])
            rk, rd = o.result()
            n = int(ref[2][2 * f + side])
            ok = ok and n == len(rk) and ref[0][2 * f + side, :n].tobytes() == rk.tobytes() and \
                np.array_equal(ref[1][2 * f + side, :n], rd)
    bad += not ok
    print("case %2d %3dx%-3d nf %4d B %2d resident %d pipeline %d repeats %d: kp %5d status %d %s" %
          (c, w, h, nf, B, resident, pipeline, repeats, int(ref[2][0]), st1, "ok" if ok else
           ("MISMATCH between forms" if not same else "MISMATCH vs oracle")), flush=True)
print("mismatches: %d of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
