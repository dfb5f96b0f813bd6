"""Span statistics of png_inflate_kernel (a -DVSF_PNG_STATS build of k_png.hip): python tools/exp/png_stats.py"""
import ctypes as C, io, sys, zlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np, torch
from PIL import Image
import png_craft as pc
from vision_slam_frontend_amd import capi, synth
W, H, N = 640, 480, 64
root = Path(__file__).resolve().parent.parent.parent / "tests" / "golden" / "real"
photos = [np.asarray(Image.open(f)) for f in sorted(root.glob("*.png"))]
photos = np.stack([a for a in photos if a.shape == (H, W) and a.dtype == np.uint8])
sets = {"synthetic": synth.bench_batch(16, W, H, n_scenes=4).reshape(-1, H, W), "photographs": photos}
lib = capi.lib()
names = ["rounds", "bytes", "bits", "symbols", "matches", "rounds cut short", "span entries", "mended literals", "one-symbol steps"]
for name, base in sets.items():
    for level in (1, 6, "cv"):
        files = []
        for i in range(N):
            k = i % len(base)
            if level == "cv":
                files.append(pc.gray8(base[k], filters=np.full(H, 1), level=1, strategy=zlib.Z_RLE, idat_piece=8192))
            else:
                b = io.BytesIO(); Image.fromarray(base[k], "L").save(b, "PNG", compress_level=level); files.append(b.getvalue())
        ctx = capi.Context(capi.default_params(W, H, max_images=2, nfeatures=2000))
        d = torch.zeros((N, H, W), dtype=torch.uint8, device="cuda")
        lib.vsf_debug_png_stats(None, 1)
        ctx.png_decode_gray_batch(files, W, H, d.data_ptr(), W * H, W)
        ctx.sync()
        out = (C.c_ulonglong * 16)()
        lib.vsf_debug_png_stats(out, 0)
        v = [x / N for x in out]
        print("%s level %s: " % (name, level) + ", ".join("%s %.0f" % (n, x) for n, x in zip(names, v)) +
              " | per round: %.1f symbols, %.1f matches, %.0f bits" % (v[3] / max(v[0], 1), v[4] / max(v[0], 1), v[2] / max(v[0], 1)), flush=True)
        ctx.close()
