"""Per-stage GPU time of the ObserveImage queue's batches (vsf_profile_enable around a stream of submits), per frame, beside
the batched step's in-line figures: which stage costs more inside the queue?
    python tools/exp/queue_stages.py [nfeatures] [depth] [batch]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: F401  (before libvsf_hip.so)
from vision_slam_frontend_amd import capi, frontend, synth

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sc = synth.Scene(640, 480)
frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(32)]
calib = frontend.default_calibration().set("fundamental", np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32))
N = 2400
with capi.Context(capi.default_params(640, 480, max_images=2 * batch, nfeatures=nf)) as ctx:
    ctx.observe_configure(depth, 0, 0)
    tickets = []

    def run(n):
        for k in range(n):
            if len(tickets) == depth:
                ctx.observe_collect(tickets.pop(0))
            tickets.append(ctx.observe_submit(*frames[k % 32], calib))
        while tickets:
            ctx.observe_collect(tickets.pop(0))

    run(400)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    run(N)
    dt = time.perf_counter() - t0
    st = ctx.profile_read()
    ctx.profile_enable(False)
    import ctypes as C
    stats = (C.c_int64 * 11)()
    capi.lib().vsf_observe_stats(ctx._h, stats, 11)
print("nfeatures %d, depth %d, <= %d per batch: %.0f frames/s through Python submits/collects; %d batches" % (nf, depth, batch, N / dt, stats[1]))
for k, (ms, launches) in st.items():
    print("  %-20s %8.2f us per frame   (%d launches)" % (k, 1e3 * ms / N, launches))
print("  sum %.2f us per frame" % (1e3 * sum(v[0] for v in st.values()) / N))
