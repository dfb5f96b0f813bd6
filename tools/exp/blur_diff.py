"""Where does the GPU blur differ from the oracle?  (debug aid; run on the GPU box from the repo root)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from vision_slam_frontend_amd import capi, synth
from oracle import binding as ob

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
left, right = synth.stereo_pair(w, h, 0)
p = capi.default_params(w, h, max_images=4, nfeatures=2000)
with capi.Context(p) as c:
    c.extract_pair(left, right)
    for im, img in enumerate((left, right)):
        o = ob.Orb(nfeatures=2000)
        o.run(img)
        nbad = 0
        for l in range(50):
            g = c.debug_level_image(im, l, True)
            r = o.level_image(l, True)
            bad = g != r
            if bad.any():
                nbad += 1
                ys, xs = np.nonzero(bad)
                print("image %d level %d %dx%d: %d bad; rows %s cols %s; first (x=%d,y=%d) got %d want %d" % (
                    im, l, g.shape[1], g.shape[0], bad.sum(), np.unique(ys).tolist()[:30], np.unique(xs).tolist()[:40], xs[0], ys[0], g[ys[0], xs[0]], r[ys[0], xs[0]]))
        print("image %d: %d levels differ" % (im, nbad))
