"""ObserveImage at a given nfeatures, pipelined, per depth: python3 tools/time_observe_depth.py 10000 [frames] [depths...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))          # tools/ (time_frontend)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))   # the repository root
import numpy as np
import time_frontend as tf
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
depths = [int(a) for a in sys.argv[3:]] or [3]
for d in depths:
    for rep in range(2):
        ms, ts, feats, fps = tf.observe_image_ms(nf, True, n_frames=n, pipelined=True, in_flight=d)
        print("nf %d, %d frames in flight: %.0f frames/s, host time per call median %.3f ms (p10 %.3f p90 %.3f), features/frame %d"
              % (nf, d, fps, ms, 1e3 * np.percentile(ts[32:], 10), 1e3 * np.percentile(ts[32:], 90), feats[-1]))
ms, ts, feats, fps = tf.observe_image_ms(nf, True, n_frames=64, pipelined=False)
print("nf %d synchronous: %.3f ms per call" % (nf, ms))
