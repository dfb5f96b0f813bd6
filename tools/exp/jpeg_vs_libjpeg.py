"""How the JPEG ingest's outcome on DAMAGED files compares with the system's libjpeg driven as cv::imdecode drives it
(tests/jpeg_ref.py): per file -- refused / read by either side, and whether the bytes agree when both read it.
    python tools/exp/jpeg_vs_libjpeg.py [n] [seed]"""
import io
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import jpeg_ref  # noqa: E402
from jpeg_mutate import mutate, rst_damage  # noqa: E402
from vision_slam_frontend_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 5))
W, H = 160, 120
img = synth.stereo_pair(W, H, 5, n_objects=60)[0]
base = []
for kw in (dict(quality=85), dict(quality=40, optimize=True), dict(quality=90, restart_marker_blocks=5), dict(quality=70, restart_marker_blocks=1),
           dict(quality=80, progressive=True), dict(quality=60, progressive=True, restart_marker_blocks=7)):
    b = io.BytesIO()
    Image.fromarray(img, "L").save(b, "JPEG", **kw)
    base.append((str(kw), b.getvalue()))
rgb = np.stack([img, img[::-1], img[:, ::-1]], 2)
for kw in (dict(quality=75, subsampling=2), dict(quality=75, subsampling=2, progressive=True)):
    b = io.BytesIO()
    Image.fromarray(rgb, "RGB").save(b, "JPEG", **kw)
    base.append((str(kw), b.getvalue()))


dev = torch.device("cuda", 0)
tab = {}
with capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500)) as ctx:
    for it in range(n):
        k = int(rng.integers(len(base)))
        kind = int(rng.integers(7))
        f = rst_damage(base[k][1], rng) if kind == 6 else mutate(base[k][1], rng, kind)
        st, ref, warn = jpeg_ref.imdecode_gray(f, W, H)
        d = torch.full((H, W), 0x5A, dtype=torch.uint8, device=dev)
        mine = "read"
        try:
            ctx.jpeg_decode_gray_batch([f], W, H, d.data_ptr(), W * H, W)
            ctx.sync(allow_capacity=True)
        except capi.VsfError as e:
            mine = "refused" if e.status == capi.VSF_ERR_INVALID_ARG else "unsupported"
            try:
                ctx.sync(allow_capacity=True)
            except capi.VsfError:
                pass
        theirs = "read" if st == 0 else "refused(%d)" % st
        key = ("damage %d" % kind, theirs + ("+warn" if st == 0 and warn else ""), mine)
        same = None
        if st == 0 and mine == "read":
            same = bool(np.array_equal(d.cpu().numpy(), ref))
        tab.setdefault(key, [0, 0, 0])
        tab[key][0] += 1
        if same is True:
            tab[key][1] += 1
        if same is False:
            tab[key][2] += 1
            if not warn and len(sys.argv) > 3:
                g = d.cpu().numpy()
                diff = np.abs(g.astype(int) - ref.astype(int))
                ys, xs = np.nonzero(diff)
                print("differs without a warning: case %d base %s damage %d: %d pixels, max %d, rows %d..%d cols %d..%d" %
                      (it, base[k][0][:30], kind, len(ys), diff.max(), ys.min(), ys.max(), xs.min(), xs.max()))
                Path(sys.argv[3]).mkdir(exist_ok=True)
                (Path(sys.argv[3]) / ("case%d.jpg" % it)).write_bytes(f)
print("%-42s %-16s %-12s %6s %6s %6s" % ("file kind", "libjpeg", "device", "files", "equal", "differ"))
for key in sorted(tab):
    print("%-42s %-16s %-12s %6d %6d %6d" % (key + tuple(tab[key])))
tot = {}
for (kind, theirs, mine), v in tab.items():
    tot.setdefault((theirs, mine), [0, 0, 0])
    for i in range(3):
        tot[(theirs, mine)][i] += v[i]
print()
for key in sorted(tot):
    print("%-16s %-12s %6d %6d %6d" % (key + tuple(tot[key])))
