"""Committed golden vectors (tests/golden/*.npz, made by tools/make_golden.py from the CPU oracle --
restatement goldens, NOT OpenCV-generated; parity is unpinned by the reference).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
CASES = ["stereo_320x240_nf500", "stereo_640x480_nf2000"]


def _load(name):
    from vision_slam_frontend_amd import synth
    g = np.load(GOLD / (name + ".npz"))
    nobj = int(g["n_objects"])
    left, right = synth.stereo_pair(int(g["width"]), int(g["height"]), int(g["frame"]),
                                    n_objects=None if nobj < 0 else nobj)
    assert synth.sha256(left) == str(g["left_sha256"]) and synth.sha256(right) == str(g["right_sha256"])
    return g, left, right


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    g, left, right = _load(name)
    nf = int(g["nfeatures"])
    descs = []
    for eye, img in (("left", left), ("right", right)):
        o = oracle.Orb(nfeatures=nf)
        o.run(img)
        kp, desc = o.result()
        assert kp.tobytes() == g[eye + "_kp"].tobytes()
        np.testing.assert_array_equal(desc, g[eye + "_desc"])
        descs.append(desc)
    assert oracle.get_matches(descs[0], descs[1]).tobytes() == g["matches"].tobytes()
    assert oracle.fast9_16(left, 10, True).tobytes() == g["fast10_left"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(name):
    from vision_slam_frontend_amd import capi
    g, left, right = _load(name)
    p = capi.default_params(int(g["width"]), int(g["height"]), max_images=2, nfeatures=int(g["nfeatures"]))
    with capi.Context(p) as ctx:
        kl, dl = ctx.extract(left)
        kr, dr = ctx.extract(right)
        assert kl.tobytes() == g["left_kp"].tobytes() and kr.tobytes() == g["right_kp"].tobytes()
        np.testing.assert_array_equal(dl, g["left_desc"])
        np.testing.assert_array_equal(dr, g["right_desc"])
        idx, dist = ctx.knn2_hamming(dl, dr)
        np.testing.assert_array_equal(idx, g["knn_idx"])
        np.testing.assert_array_equal(dist, g["knn_dist"])
        assert ctx.get_matches(dl, dr).tobytes() == g["matches"].tobytes()
        assert ctx.fast_detect(left, 10, True, cap=1 << 17).tobytes() == g["fast10_left"].tobytes()
