#!/usr/bin/env python3
"""Writes tests/golden/*.npz: seeded synthetic inputs (by sha256) and the CPU oracle's outputs for them.

RESTATEMENT GOLDENS, NOT OPENCV-GENERATED: the reference has no fixtures and OpenCV 3.2.0 is unavailable
(SURVEY.md section 8(c)); these vectors freeze the oracle (so it cannot drift silently) and give the GPU tests a
second, file-based checker.  Regenerate with:  python tools/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import binding as ob  # noqa: E402
from vision_slam_frontend_amd import synth  # noqa: E402

CASES = {
    # name: (width, height, nfeatures, n_objects, frame_idx)
    "stereo_320x240_nf500": (320, 240, 500, 400, 0),
    "stereo_640x480_nf2000": (640, 480, 2000, None, 0),
}


def make(name, w, h, nf, nobj, frame):
    left, right = synth.stereo_pair(w, h, frame, n_objects=nobj)
    out = {"width": w, "height": h, "nfeatures": nf, "n_objects": -1 if nobj is None else nobj, "frame": frame,
           "left_sha256": synth.sha256(left), "right_sha256": synth.sha256(right)}
    descs = []
    for eye, img in (("left", left), ("right", right)):
        o = ob.Orb(nfeatures=nf)
        o.run(img)
        kp, desc = o.result()
        out[eye + "_kp"] = kp
        out[eye + "_desc"] = desc
        out[eye + "_fast_counts"] = np.array([len(o.stage(0, l)) for l in range(o.nlevels)], np.int32)
        out[eye + "_level_counts"] = np.array([len(o.stage(4, l)) for l in range(o.nlevels)], np.int32)
        descs.append(desc)
    idx, dist = ob.knn2_hamming(descs[0], descs[1])
    out["knn_idx"], out["knn_dist"] = idx, dist
    out["matches"] = ob.get_matches(descs[0], descs[1])
    out["fast10_left"] = ob.fast9_16(left, 10, True)
    np.savez_compressed(ROOT / "tests" / "golden" / (name + ".npz"), **out)
    print(name, "kp", len(out["left_kp"]), len(out["right_kp"]), "matches", len(out["matches"]), "fast10",
          len(out["fast10_left"]))


if __name__ == "__main__":
    for name, args in CASES.items():
        make(name, *args)
