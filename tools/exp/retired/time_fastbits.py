"""FAST on bit planes (k_fastbits.hip, VSF_OPT_FAST_BITS) against the march kernel (k_fast.hip): the candidates of every
level of every image compared entry for entry, the final keypoints / descriptors byte for byte, and both forms timed
(per-stage hipEvents, blur in line so that FAST runs by itself).

  python3 tools/time_fastbits.py [width height nfeatures batch [scene]]      scene: bench | sparse | noise
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vision_slam_frontend_amd import capi, synth  # noqa: E402


def main():
    W, H, NF, B = (int(x) for x in (sys.argv[1:5] + ["640", "480", "2000", "64"][len(sys.argv[1:5]):]))
    scene = sys.argv[5] if len(sys.argv) > 5 else "bench"
    dev = torch.device("cuda", 0)
    if scene == "flat":
        yy, xx = np.mgrid[0:H, 0:W]
        imgs = np.broadcast_to(((xx + 2 * yy) // 8 % 256).astype(np.uint8), (B, H, W)).copy()
    elif scene == "noise":
        imgs = np.random.Generator(np.random.PCG64(5)).integers(0, 256, (B, H, W), dtype=np.uint8)
    else:
        fr = synth.bench_batch((B + 1) // 2, W, H, n_objects=(synth.default_object_count(W, H) // 8 if scene == "sparse" else None))
        imgs = fr.reshape(-1, H, W)[:B]
    p = capi.default_params(W, H, max_images=B, nfeatures=NF)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(np.ascontiguousarray(imgs)).to(dev)
        outs = {}
        ctx.set_blur_overlap(False)
        for form in (0, 2):
            ctx.set_option(capi.OPT_FAST_BITS, form)
            d_kp = torch.zeros((B, K, 28), dtype=torch.uint8, device=dev)
            d_desc = torch.zeros((B, K, 32), dtype=torch.uint8, device=dev)
            d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.extract_batch_dev(d_img.data_ptr(), B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr())
            assert ctx.sync() == capi.VSF_OK
            cands = {}
            for im in sorted(set([0, 1, B // 2, B - 1])):
                for lv in range(ctx.nlevels):
                    cands[(im, lv)] = ctx.debug_fast_candidates(im, lv).copy()
            outs[form] = (d_kp.cpu().numpy(), d_desc.cpu().numpy(), d_counts.cpu().numpy(), cands)
            ctx.profile_enable(True)
            for _ in range(3):
                ctx.extract_batch_dev(d_img.data_ptr(), B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr())
            prof = ctx.profile_read()
            ctx.profile_enable(False)
            print("form %d:" % form, {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1]}, flush=True)
        bad = 0
        a, b = outs[0], outs[2]
        for key in a[3]:
            ca, cb = a[3][key], b[3][key]
            if len(ca) != len(cb) or ca.tobytes() != cb.tobytes():
                bad += 1
                if bad <= 8:
                    n = min(len(ca), len(cb))
                    diff = [i for i in range(n) if ca[i].tobytes() != cb[i].tobytes()]
                    print("candidates differ: image %d level %d: %d vs %d entries, first diff at %s" %
                          (key[0], key[1], len(ca), len(cb), diff[:3]))
                    if diff:
                        i = diff[0]
                        print("   march:", ca[i], "\n   bits: ", cb[i])
        print("candidate lists compared: %d, differing: %d" % (len(a[3]), bad))
        print("counts equal:", np.array_equal(a[2], b[2]), " keypoints equal:", a[0].tobytes() == b[0].tobytes(),
              " descriptors equal:", a[1].tobytes() == b[1].tobytes())
        print("candidates per image (level 0 / all levels of image 0):", len(a[3][(0, 0)]), sum(len(a[3][(0, l)]) for l in range(ctx.nlevels)))


if __name__ == "__main__":
    main()
