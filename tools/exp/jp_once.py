import io, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from PIL import Image
from vision_slam_frontend_amd import capi, synth
N=512; W,H=640,480
base = synth.bench_batch(16, W, H, n_scenes=4).reshape(-1, H, W)
files=[]
for i in range(N):
    b=io.BytesIO(); Image.fromarray(base[i%len(base)],"L").save(b,"JPEG",quality=80); files.append(b.getvalue())
ctx = capi.Context(capi.default_params(W, H, max_images=2, nfeatures=2000))
d = torch.zeros((N,H,W),dtype=torch.uint8,device="cuda")
for _ in range(2):
    ctx.jpeg_decode_gray_batch(files, W, H, d.data_ptr(), W*H, W)
    ctx.sync()
