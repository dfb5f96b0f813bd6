"""One-off robustness run: corrupted JPEG files through vsf_jpeg_decode_gray_batch (must return, never fault)."""
import io, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from PIL import Image
from vision_slam_frontend_amd import capi, synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
W, H = 160, 120
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500))
base = []
img = synth.stereo_pair(W, H, 5, n_objects=60)[0]
for kw in (dict(quality=85), dict(quality=40, optimize=True), dict(quality=90, restart_marker_blocks=5)):
    b = io.BytesIO(); Image.fromarray(img, "L").save(b, "JPEG", **kw); base.append(b.getvalue())
rgb = np.stack([img, img[::-1], img[:, ::-1]], 2)
b = io.BytesIO(); Image.fromarray(rgb, "RGB").save(b, "JPEG", quality=75, subsampling=2); base.append(b.getvalue())
counts = {}
d = torch.zeros((8, H, W), dtype=torch.uint8, device=dev)
for it in range(400):
    files = []
    for _ in range(8):
        f = bytearray(base[int(rng.integers(len(base)))])
        kind = int(rng.choice([0, 1, 2, 4])) if len(sys.argv) > 2 else int(rng.integers(5))
        sos = f.find(b"\xFF\xDA")
        if kind == 0:      # bit flips in the entropy-coded data
            for _ in range(int(rng.integers(1, 20))):
                p = int(rng.integers(sos + 12, len(f) - 2)); f[p] ^= 1 << int(rng.integers(8))
        elif kind == 1:    # truncation
            f = f[:int(rng.integers(sos + 12, len(f)))]
        elif kind == 2:    # stray markers / 0xFF bytes
            for _ in range(int(rng.integers(1, 6))):
                p = int(rng.integers(sos + 12, len(f) - 2)); f[p] = 0xFF; f[p + 1] = int(rng.integers(256))
        elif kind == 3:    # header damage
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(2, sos + 12)); f[p] = int(rng.integers(256))
        else:              # random garbage after the headers
            f[sos + 12:] = bytes(rng.integers(0, 256, len(f) - sos - 12, dtype=np.uint8))
        files.append(bytes(f))
    try:
        ctx.jpeg_decode_gray_batch(files, W, H, d.data_ptr(), W * H, W)
        st = ctx.sync(allow_capacity=True)
        counts["ok"] = counts.get("ok", 0) + 1
    except capi.VsfError as e:
        counts[str(e)[:60]] = counts.get(str(e)[:60], 0) + 1
        try:
            ctx.sync(allow_capacity=True)
        except capi.VsfError:
            pass
    if it % 100 == 99:
        print(it + 1, counts, flush=True)
print("done", counts)
