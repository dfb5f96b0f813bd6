#!/usr/bin/env python3
"""Randomised parity stress (run by hand on a GPU box): random sizes / ORB parameters / scenes, GPU vs oracle bit for bit
for keypoints, descriptors and stereo matches.  python tools/stress_parity.py [n_cases] [seed]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import binding as ob  # noqa: E402
from vision_slam_frontend_amd import capi, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ob.build()
bad = 0
for c in range(n_cases):
    w = int(rng.integers(80, 900))
    h = int(rng.integers(70, 700))
    nf = int(rng.choice([50, 300, 1000, 2000, 5000]))
    thr = int(rng.choice([5, 10, 20, 20, 20, 40]))
    nlev, sf = [(50, 1.04), (50, 1.04), (8, 1.2), (20, 1.1), (1, 1.04), (33, 1.06)][int(rng.integers(0, 6))]
    nobj = int(rng.integers(5, max(6, w * h // 150)))
    left, right = synth.stereo_pair(w, h, int(rng.integers(0, 1000)), seed=int(rng.integers(0, 1 << 30)), n_objects=nobj)
    if rng.random() < 0.15:
        left = rng.integers(0, 256, (h, w), dtype=np.uint8)  # pure noise
    try:
        p = capi.default_params(w, h, max_images=1, nfeatures=nf, nlevels=nlev, scale_factor=sf, fast_threshold=thr)
        ctx = capi.Context(p)
    except capi.VsfError as e:
        print("case %d %dx%d nlevels %d scale %.2f: %s (skipped)" % (c, w, h, nlev, sf, e))
        continue
    try:
        res = []
        for img in (left, right):
            o = ob.Orb(nfeatures=nf, nlevels=nlev, scale_factor=sf, fast_threshold=thr)
            o.run(img)
            rk, rd = o.result()
            kp, desc = ctx.extract(img, cap=max(len(rk) + 8, 8))
            ok = len(kp) == len(rk) and kp.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
            res.append((ok, len(rk), desc, rd))
        m_ok = ctx.get_matches(res[0][2], res[1][2]).tobytes() == ob.get_matches(res[0][3], res[1][3]).tobytes()
        ok = res[0][0] and res[1][0] and m_ok
        bad += not ok
        print("case %2d %3dx%-3d nf %4d thr %2d levels %2d scale %.2f objects %5d: kp %5d/%5d %s" %
              (c, w, h, nf, thr, nlev, sf, nobj, res[0][1], res[1][1], "ok" if ok else "MISMATCH"))
    finally:
        ctx.close()
print("mismatches: %d of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
