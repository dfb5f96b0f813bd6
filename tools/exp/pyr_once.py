import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from vision_slam_frontend_amd import capi, synth
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=2000))
K = ctx.params.max_keypoints
frames = synth.stereo_stream(1, 640, 480)
d_img = torch.from_numpy(frames).to(dev)
kp = torch.zeros((2, K, 28), dtype=torch.uint8, device=dev); desc = torch.zeros((2, K, 32), dtype=torch.uint8, device=dev)
counts = torch.zeros(2, dtype=torch.int32, device=dev); m = torch.zeros((1, K, 16), dtype=torch.uint8, device=dev)
nm = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    ctx.stereo_batch_dev(d_img.data_ptr(), 1, 640 * 480, 640, kp.data_ptr(), desc.data_ptr(), counts.data_ptr(), m.data_ptr(), nm.data_ptr())
    ctx.sync()
    print("----")
