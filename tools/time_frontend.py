#!/usr/bin/env python3
"""The drop-in API (C++ slam::Frontend): ObserveImage per stereo frame at 640x480, window 10 -- as one GPU submission per
call (vsf_observe_stereo: synchronous latency), call by call (one C-ABI call per reference call), and queued
(Frontend::set_pipelined: frames wait in the context's queue and leave for the GPU in batches; frames per second of the
drop-in API, driven by the C++ loop vsfh_time_sequence -- the reference's driver loop, no Python between the calls).
    python tools/time_frontend.py [--json] [nfeatures ...]     (default 2000 10000)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vision_slam_frontend_amd import frontend, synth  # noqa: E402


_FRAMES = {}


def observe_image_ms(nfeatures: int, fused: bool = True, n_frames: int = 56, width: int = 640, height: int = 480,
                     pipelined: bool = False, in_flight: int = 0):
    """Median ObserveImage time in ms in the steady state (the window of 10 kept frames is full; the first 32 calls are
    left out).  pipelined: the call returns once the frame is queued, so the per-call time is host work only; the
    frames-per-second figure is the wall time of the steady-state calls INCLUDING the final flush."""
    if (width, height) not in _FRAMES:
        sc = synth.Scene(width, height)
        base = [(sc.render(f, 0), sc.render(f, 1)) for f in range(14)]
        _FRAMES[(width, height)] = base
    base = _FRAMES[(width, height)]
    frames = [base[f % len(base)] for f in range(n_frames)]
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    fe = frontend.Frontend(width, height, nfeatures=nfeatures, fundamental=F)
    fe.set_fused(fused)
    fe.set_pipelined(pipelined)
    if in_flight:
        fe.set_frames_in_flight(in_flight)
    q = np.array([1, 0, 0, 0], np.float32)
    fe.observe_odometry([0, 0, 0], q, 0.0)
    ts = []
    t_steady = None
    for f, (l, r) in enumerate(frames):
        if f == 32:
            fe.flush()
            t_steady = time.perf_counter()
        fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.0 + f)
        t0 = time.perf_counter()
        added = fe.observe_image(l, r)
        ts.append(time.perf_counter() - t0)
        assert added
    fe.flush()
    fps = (n_frames - 32) / (time.perf_counter() - t_steady)
    feats = [len(n["features"]) for n in fe.nodes()]
    fe.close()
    return 1e3 * float(np.median(ts[32:])), ts, feats, fps


def _stereo_frames(width: int, height: int, n: int = 32) -> np.ndarray:
    sc = synth.Scene(width, height)
    return np.stack([np.stack([sc.render(f, 0), sc.render(f, 1)]) for f in range(n)]).astype(np.uint8)


_STACKS = {}


def queued_fps(nfeatures: int, n_frames: int = 3232, width: int = 640, height: int = 480, depth: int = 0, batch_frames: int = 0,
               min_batch: int = 0, read_every: int = 0, pipelined: bool = True, repeats: int = 1):
    """Frames per second of ObserveOdometry + ObserveImage called from C++ (vsfh_time_sequence: the reference's driver loop,
    slam_frontend_main.cc:271-328) with the queue on: the steady state after 32 warm-up frames, final Flush inside the clock.
    depth / batch_frames / min_batch 0: the class's defaults (256 / 128 / half a batch).  read_every = 1: GetSLAMProblem
    after every node, as the reference's driver does (it drains the queue: the synchronous rate).  Returns the list of
    (frames per second, mean ms inside ObserveImage, max ms) of `repeats` runs, each on a Frontend of its own."""
    if (width, height) not in _STACKS:
        _STACKS[(width, height)] = _stereo_frames(width, height)
    frames = _STACKS[(width, height)]
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    runs = []
    for _ in range(repeats):
        fe = frontend.Frontend(width, height, nfeatures=nfeatures, fundamental=F)
        fe.set_pipelined(pipelined)
        if pipelined:
            fe.set_queue(depth, batch_frames, min_batch)
        runs.append(fe.time_sequence(frames, n_frames, warm=32, read_every=read_every))
        fe.close()
    return runs


if __name__ == "__main__":
    if "--dump" in sys.argv[1:]:  # frames for tools/time_frontend (the C++ driver): --dump FILE [n_frames]
        i = sys.argv.index("--dump")
        n = int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 16
        sc = synth.Scene(640, 480)
        np.stack([np.stack([sc.render(f, 0), sc.render(f, 1)]) for f in range(n)]).astype(np.uint8).tofile(sys.argv[i + 1])
        print("wrote %d stereo frames 640x480 to %s" % (n, sys.argv[i + 1]))
        sys.exit(0)
    args = [a for a in sys.argv[1:] if a != "--json"]
    record = {}
    for nf in [int(a) for a in args] or [2000, 10000]:
        for name, fused, n in (("fused", True, 56), ("call-by-call", False, 56)):
            ms, ts, feats, fps = observe_image_ms(nf, fused, n_frames=n)
            print("nfeatures %5d %-12s ObserveImage median %.3f ms per call (window full), %6.0f frames/s over %d steady "
                  "frames; first call %.1f ms; features/frame ~%d" %
                  (nf, name, ms, fps, n - 32, 1e3 * ts[0], int(np.median(feats))))
            record["%s_%d" % (name.replace("-", "_"), nf)] = {"observe_image_ms": ms, "frames_per_s": fps,
                                                             "features_per_frame": int(np.median(feats))}
        for name, kw in (("unchanged_caller", dict(pipelined=False, read_every=1, n_frames=160)),
                         ("queued_d32_b32", dict(depth=32, batch_frames=32, n_frames=1632)),
                         ("queued_d128_b64", dict(depth=128, batch_frames=64)), ("queued", dict(repeats=3))):
            runs = queued_fps(nf, **kw)
            fps = sorted(r[0] for r in runs)
            print("nfeatures %5d %-18s %7.0f frames/s (runs: %s), %.3f ms inside ObserveImage (mean)" %
                  (nf, name, fps[len(fps) // 2], " ".join("%.0f" % f for f in fps), runs[0][1]))
            record["%s_%d" % (name, nf)] = {"frames_per_s": fps[len(fps) // 2], "runs": fps, "observe_image_ms_mean": runs[0][1]}
    if "--json" in sys.argv[1:]:
        print(json.dumps({"what": "slam::Frontend::ObserveImage, 640x480, frame_life 10 (tools/time_frontend.py)",
                          "results": record}))
