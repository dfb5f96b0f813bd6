"""Where does one stereo frame's time go?  Per-stage device time (hipEvents) and wall time of vsf_stereo_batch_dev for
a batch of ONE frame, at nfeatures 2000 and 10000."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from vision_slam_frontend_amd import capi, synth

NFR = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # stereo frames per call
for NF in (2000, 10000):
    dev = torch.device("cuda", 0)
    ctx = capi.Context(capi.default_params(640, 480, max_images=2 * NFR, nfeatures=NF))
    K = ctx.params.max_keypoints
    frames = synth.stereo_stream(NFR, 640, 480)
    d_img = torch.from_numpy(frames).to(dev)
    kp = torch.zeros((2 * NFR, K, 28), dtype=torch.uint8, device=dev); desc = torch.zeros((2 * NFR, K, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(2 * NFR, dtype=torch.int32, device=dev); m = torch.zeros((NFR, K, 16), dtype=torch.uint8, device=dev)
    nm = torch.zeros(NFR, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    def run():
        ctx.stereo_batch_dev(d_img.data_ptr(), NFR, 640 * 480, 640, kp.data_ptr(), desc.data_ptr(), counts.data_ptr(), m.data_ptr(), nm.data_ptr())
    for _ in range(5): run()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        run(); ctx.sync()
    wall = (time.perf_counter() - t0) / 50
    ctx.profile_enable(True)
    for _ in range(50): run(); ctx.sync()
    st = ctx.profile_read()
    ctx.profile_enable(False)
    print("frames/call %d  " % NFR + "nfeatures %d: wall %.3f ms per call (launch+sync); stages us:" % (NF, wall * 1e3),
          {k: round(v[0] / 50 * 1e3, 1) for k, v in st.items()}, "sum %.3f ms" % (sum(v[0] for v in st.values()) / 50))
    ctx.close()
