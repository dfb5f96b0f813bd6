"""One JPEG file through both baseline decoders against a saved reference: python tools/exp/jpeg_case.py file.jpg ref.npy"""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from vision_slam_frontend_amd import capi
f = open(sys.argv[1], "rb").read()
ref = np.load(sys.argv[2])
H, W = ref.shape
for serial in (0, 1):
    c = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500))
    c.set_option(capi.OPT_JPEG_SERIAL, serial)
    d = torch.full((H, W), 0x5A, dtype=torch.uint8, device="cuda")
    c.jpeg_decode_gray_batch([f], W, H, d.data_ptr(), W * H, W)
    c.sync()
    g = d.cpu().numpy()
    diff = np.abs(g.astype(int) - ref.astype(int))
    ys, xs = np.nonzero(diff)
    print("serial", serial, "differing pixels", len(ys), "" if len(ys) == 0 else (ys.min(), ys.max(), xs.min(), xs.max(), diff.max()))
    if len(ys):
        y0, x0 = ys.min() // 8 * 8, xs.min() // 8 * 8
        print(g[y0:y0 + 8, x0:x0 + 8]); print(ref[y0:y0 + 8, x0:x0 + 8])
    c.close()
