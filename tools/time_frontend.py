#!/usr/bin/env python3
"""Latency of the one-frame-at-a-time path (C++ slam::Frontend): ObserveImage per stereo frame at 640x480, window 10,
as one GPU submission (vsf_observe_stereo) and call by call (one C-ABI call per reference call).
    python tools/time_frontend.py [nfeatures ...]     (default 2000 10000)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vision_slam_frontend_amd import frontend, synth  # noqa: E402


_FRAMES = {}


def observe_image_ms(nfeatures: int, fused: bool = True, n_frames: int = 56, width: int = 640, height: int = 480):
    """Median ObserveImage time in ms in the steady state: the window of 10 kept frames is full and every launch chain
    the fused path replays as a hipGraph has been captured once (one per ring slot: the first 21 frames)."""
    if (width, height) not in _FRAMES:
        sc = synth.Scene(width, height)
        base = [(sc.render(f, 0), sc.render(f, 1)) for f in range(14)]
        _FRAMES[(width, height)] = base
    base = _FRAMES[(width, height)]
    frames = [base[f % len(base)] for f in range(n_frames)]
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    fe = frontend.Frontend(width, height, nfeatures=nfeatures, fundamental=F)
    fe.set_fused(fused)
    q = np.array([1, 0, 0, 0], np.float32)
    fe.observe_odometry([0, 0, 0], q, 0.0)
    ts = []
    for f, (l, r) in enumerate(frames):
        fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.0 + f)
        t0 = time.perf_counter()
        added = fe.observe_image(l, r)
        ts.append(time.perf_counter() - t0)
        assert added
    feats = [len(n["features"]) for n in fe.nodes()]
    fe.close()
    return 1e3 * float(np.median(ts[32:])), ts, feats


if __name__ == "__main__":
    for nf in [int(a) for a in sys.argv[1:]] or [2000, 10000]:
        for fused in (True, False):
            ms, ts, feats = observe_image_ms(nf, fused)
            print("nfeatures %5d %-12s ObserveImage median %.3f ms (window full); first %.1f ms; features/frame ~%d" %
                  (nf, "fused" if fused else "call-by-call", ms, 1e3 * ts[0], int(np.median(feats))))
