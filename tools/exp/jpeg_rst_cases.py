"""Damaged restart-interval files: device (decoder as chosen, and forced one-wave) against libjpeg; prints the disagreements."""
import io, sys
from pathlib import Path
import numpy as np, torch
from PIL import Image
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import jpeg_ref
from jpeg_mutate import mutate, rst_damage
from vision_slam_frontend_amd import capi, synth
W, H = 160, 120
img = synth.stereo_pair(W, H, 5, n_objects=60)[0]
base = []
for kw in (dict(quality=90, restart_marker_blocks=5), dict(quality=70, restart_marker_blocks=20), dict(quality=80, restart_marker_blocks=1)):
    b = io.BytesIO(); Image.fromarray(img, "L").save(b, "JPEG", **kw); base.append(b.getvalue())
rgb = np.stack([img, img[::-1], img[:, ::-1]], 2)
b = io.BytesIO(); Image.fromarray(rgb, "RGB").save(b, "JPEG", quality=75, subsampling=2, restart_marker_blocks=3); base.append(b.getvalue())
rng = np.random.Generator(np.random.PCG64(8))
ctxs = []
for serial in (0, 1):
    c = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500)); c.debug_jpeg_serial(serial); ctxs.append(c)
tab = {}
for it in range(800):
    k = int(rng.integers(len(base))); kind = int(rng.integers(4))
    f = rst_damage(base[k], rng) if kind == 3 else mutate(base[k], rng, kind)
    st, ref, warn = jpeg_ref.imdecode_gray(f, W, H)
    if st != 0: continue
    res = []
    for c in ctxs:
        d = torch.full((H, W), 0x5A, dtype=torch.uint8, device="cuda")
        c.jpeg_decode_gray_batch([f], W, H, d.data_ptr(), W * H, W); c.sync()
        res.append(bool(np.array_equal(d.cpu().numpy(), ref)))
    key = (k, kind, tuple(res)); tab[key] = tab.get(key, 0) + 1
    if not res[0] and len(sys.argv) > 1:
        Path(sys.argv[1]).mkdir(exist_ok=True); (Path(sys.argv[1]) / ("rst%d_b%d_k%d.jpg" % (it, k, kind))).write_bytes(f)
for key in sorted(tab): print("base %d damage %d (as chosen, one-wave) equal to libjpeg: %s  x %d" % (key + (tab[key],)))
