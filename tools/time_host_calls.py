import sys, time
sys.path.insert(0,'.')
import numpy as np
from vision_slam_frontend_amd import capi, synth
l, r = synth.stereo_pair(640, 480, 0)
for nf in (2000, 10000):
    ctx = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=nf))
    def t(fn, n=10):
        fn(); t0=time.perf_counter()
        for _ in range(n): fn()
        return (time.perf_counter()-t0)/n*1e3
    (kl, dl), (kr, dr) = ctx.extract_pair(l, r)
    print("nf", nf, "kp", len(kl), "extract %.2f ms  extract_pair %.2f ms  get_matches %.2f ms  multi(10) %.2f ms" % (
        t(lambda: ctx.extract(l)), t(lambda: ctx.extract_pair(l, r)), t(lambda: ctx.get_matches(dl, dr)),
        t(lambda: ctx.get_matches_multi([dl]*10, dr))))
    ctx.close()
