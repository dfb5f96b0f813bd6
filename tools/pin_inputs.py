#!/usr/bin/env python3
"""Step 1 of the OpenCV pin kit (tools/pin_with_opencv.cc): writes every input the kit runs through OpenCV into one
directory as .npy files, plus cases.txt (`<case> <nfeatures> <left file> <right file> <full|digest>` per line) and the calibration
matrices of the default FrontendConfig (slam_frontend.cc:565-634 as vision_slam_frontend_amd/frontend.py holds them).

    python3 tools/pin_inputs.py <out dir>

Inputs: the two synthetic fixtures of tools/make_golden.py, the photographs of tests/golden/real (pairs as in its manifest),
and the exact-rounding-tie image of tests/adversarial_images.py.  Needs numpy, and Pillow for the PNG files; no GPU, no
OpenCV, nothing of the product's native code (the calibration numbers come from a JSON snapshot when the host library is
not built)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

SYNTHETIC = {  # as tools/make_golden.py: name -> (width, height, nfeatures, n_objects, frame)
    "stereo_320x240_nf500": (320, 240, 500, 400, 0),
    "stereo_640x480_nf2000": (640, 480, 2000, None, 0),
}
CALIBRATION_SNAPSHOT = ROOT / "tests" / "golden" / "default_calibration.json"


def calibration():
    """{name: float32 array} of the default stereo calibration; refreshes the JSON snapshot when the host library is there."""
    try:
        from vision_slam_frontend_amd import frontend
        c = frontend.default_calibration()
        cal = {k: np.asarray(c.get(k), np.float32).reshape(-1) for k in
               ("projection_left", "projection_right", "camera_matrix_left", "distortion_left")}
        CALIBRATION_SNAPSHOT.write_text(json.dumps({k: [float(x) for x in v] for k, v in cal.items()}, indent=1) + "\n")
        return cal
    except (OSError, ImportError, RuntimeError):
        snap = json.loads(CALIBRATION_SNAPSHOT.read_text())
        return {k: np.asarray(v, np.float32) for k, v in snap.items()}


def cases():
    """[(case, nfeatures, left image, right image)] -- shared with tests/test_pinned_by_opencv.py."""
    from PIL import Image

    import adversarial_images as adv
    from vision_slam_frontend_amd import synth
    out = []
    for name, (w, h, nf, nobj, frame) in SYNTHETIC.items():
        left, right = synth.stereo_pair(w, h, frame, n_objects=nobj)
        out.append((name, nf, left, right))
    real = ROOT / "tests" / "golden" / "real"
    manifest = json.loads((real / "manifest.json").read_text())
    load = lambda n: np.asarray(Image.open(real / manifest["images"][n]["file"]))  # noqa: E731
    photos = [n for n, r in manifest["images"].items() if "file" in r]
    paired = set()
    for key in manifest["pairs"]:
        q, t = key.split("|")
        if q in photos and t in photos:
            out.append(("photo_%s__%s" % (q, t), manifest["images"][q]["nfeatures"], load(q), load(t)))
            paired |= {q, t}
    for n in photos:  # odd sizes: matched against themselves shifted by a few pixels
        if n not in paired:
            img = load(n)
            out.append(("photo_%s" % n, manifest["images"][n]["nfeatures"], img, np.roll(img, (1, 3), (0, 1))))
    ties, _ = adv.blur_ties()
    out.append(("adversarial_blur_ties", 1000, ties, np.ascontiguousarray(ties[:, ::-1])))
    return out


def main():
    if len(sys.argv) != 2:
        print(__doc__)
        return 2
    out = Path(sys.argv[1])
    out.mkdir(parents=True, exist_ok=True)
    cal = calibration()
    np.save(out / "projection_left.npy", cal["projection_left"].reshape(3, 4))
    np.save(out / "projection_right.npy", cal["projection_right"].reshape(3, 4))
    np.save(out / "camera_matrix_left.npy", cal["camera_matrix_left"].reshape(3, 3))
    np.save(out / "distortion_left.npy", cal["distortion_left"].reshape(5, 1))
    lines = ["# case nfeatures left right full|digest   (tools/pin_inputs.py)"]
    for name, nf, left, right in cases():
        np.save(out / (name + "__left.npy"), np.ascontiguousarray(left, np.uint8))
        np.save(out / (name + "__right.npy"), np.ascontiguousarray(right, np.uint8))
        # the small cases keep their whole pyramids in the output (pixel-level diagnosis); the others per-level digests
        detail = "full" if left.size <= 322 * 242 else "digest"
        lines.append("%s %d %s__left.npy %s__right.npy %s" % (name, nf, name, name, detail))
        print("%-48s %4dx%-4d nfeatures %d" % (name, left.shape[1], left.shape[0], nf))
    (out / "cases.txt").write_text("\n".join(lines) + "\n")
    print("wrote %d cases to %s" % (len(lines) - 1, out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
