"""The OpenCV pin kit: the only route from "parity: partial" to "green" is somebody WITH OpenCV
3.2.0 running the reference's own calls on committed inputs -- tools/pin_inputs.py + tools/pin_with_opencv.cc write
tests/golden/opencv/, and test_oracle_equals_opencv below compares the oracle with those files stage by stage
(tests/pin_compare.py), naming the first diverging stage and the INTEGRATION.md section 8 switch to flip.

No OpenCV exists in the build container, so that test SKIPS here until the directory appears.  Everything else of the kit
is tested now, with the ORACLE standing in for OpenCV: the C++ .npy writer / reader against numpy, the loader on a full file
set, and the diagnosis (a perturbed stage is named, not just "something differs")."""
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

import pin_compare  # noqa: E402
import pin_inputs  # noqa: E402

OPENCV_DIR = ROOT / "tests" / "golden" / "opencv"


@pytest.fixture(scope="module")
def small_cases():
    """Three of the kit's cases, small enough for the CPU suite: a synthetic stereo pair, the exact-tie image, a photograph of coins."""
    want = ("stereo_320x240_nf500", "adversarial_blur_ties", "photo_coins_384x303")
    cases = [c for c in pin_inputs.cases() if c[0] in want]
    assert len(cases) == 3
    return cases


@pytest.fixture(scope="module")
def stand_in_dir(tmp_path_factory, small_cases, oracle):
    d = tmp_path_factory.mktemp("pin_stand_in")
    pin_compare.write_from_oracle(d, small_cases)
    return d


def test_level_digest_matches_the_cpp_one(tmp_path):
    """tests/pin_compare.py level_digest == tools/pin_with_opencv.cc level_digest (restated in the C++ test program)."""
    exe = tmp_path / "test_pin_npy"
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-o", str(exe), str(ROOT / "tests" / "cpp" / "test_pin_npy.cc")])
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (61, 47), dtype=np.uint8)
    np.save(tmp_path / "image.npy", img)
    out = subprocess.run([str(exe), "digest", str(tmp_path)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == int(pin_compare.level_digest(img)) and int(out[1]) == int(pin_compare.level_digest(np.zeros(0, np.uint8)))


def test_cpp_npy_files_round_trip_through_numpy(tmp_path):
    """tools/pin_npy.h is what the OpenCV program writes its results with: numpy must read every kind of file it writes
    (cv::KeyPoint / cv::DMatch records as structured arrays, empty arrays included), and it must read what numpy.save writes
    (the input images and calibration matrices of tools/pin_inputs.py)."""
    exe = tmp_path / "test_pin_npy"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++11", "-Wall", "-Werror", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-o", str(exe), str(ROOT / "tests" / "cpp" / "test_pin_npy.cc")])
    subprocess.check_call([str(exe), "write", str(tmp_path)])
    from oracle.binding import DMATCH_DTYPE, KEYPOINT_DTYPE
    kp = np.load(tmp_path / "kp.npy")
    assert kp.dtype == KEYPOINT_DTYPE and kp.shape == (5,)
    assert tuple(kp[2]) == (3.0, 4.5, 93.0, 91.0, np.float32(2e-3), 2, -1)
    empty = np.load(tmp_path / "kp_empty.npy")
    assert empty.dtype == KEYPOINT_DTYPE and empty.shape == (0,)
    m = np.load(tmp_path / "matches.npy")
    assert m.dtype == DMATCH_DTYPE and m.tolist() == [(0, 7, 0, 12.0), (3, 1, 0, 40.0)]
    desc = np.load(tmp_path / "desc.npy")
    assert desc.dtype == np.uint8 and desc.shape == (7, 13) and desc[6, 12] == (90 * 3) % 256
    assert np.array_equal(np.load(tmp_path / "flat.npy"), desc.reshape(-1))
    assert np.load(tmp_path / "idx.npy").tolist() == [[1, -1], [5, 2], [0x7FFFFFFF, 0]]
    p4 = np.load(tmp_path / "p4.npy")
    assert p4.dtype == np.float32 and p4.shape == (2, 4) and p4[1, 2] == np.float32(3e8)
    # the other direction: what tools/pin_inputs.py writes
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    np.save(tmp_path / "image.npy", img)
    P = np.arange(12, dtype=np.float32).reshape(3, 4) * np.float32(1.25)
    np.save(tmp_path / "P.npy", P)
    np.save(tmp_path / "dist.npy", np.zeros((5, 1), np.float32))
    np.save(tmp_path / "kp_np.npy", kp)
    out = subprocess.run([str(exe), "read", str(tmp_path)], capture_output=True, text=True, check=True).stdout.split("\n")
    s = 0
    for b in img.reshape(-1).tolist():
        s = (s * 31 + b) % (1 << 64)
    assert out[0] == "image |u1 2 37 53 %d" % (s % 1000000007)
    assert out[1] == "P <f4 3 4 0 13.75"
    assert out[2] == "dist <f4 5 4" and out[3] == "kp 5 28"


def test_kit_inputs_are_complete(tmp_path):
    """tools/pin_inputs.py: every case has two uint8 images, the calibration is the default FrontendConfig's, cases.txt lists
    what the C++ program parses (`name nfeatures left right full|digest`)."""
    assert subprocess.run([sys.executable, str(ROOT / "tools" / "pin_inputs.py"), str(tmp_path)], capture_output=True).returncode == 0
    lines = [ln.split() for ln in (tmp_path / "cases.txt").read_text().splitlines() if ln and not ln.startswith("#")]
    assert len(lines) >= 11 and {ln[0] for ln in lines} >= {"stereo_640x480_nf2000", "photo_motorcycle_left__motorcycle_right",
                                                             "adversarial_blur_ties"}
    for name, nf, left, right, detail in lines:
        a, b = np.load(tmp_path / left), np.load(tmp_path / right)
        assert a.dtype == np.uint8 and a.ndim == 2 and a.shape == b.shape and int(nf) in (500, 1000, 2000)
        assert detail == ("full" if a.size <= pin_compare.FULL_PIXELS else "digest")
    assert np.load(tmp_path / "projection_right.npy").shape == (3, 4) and np.load(tmp_path / "distortion_left.npy").shape == (5, 1)
    assert abs(float(np.load(tmp_path / "camera_matrix_left.npy")[0, 0]) - 527.873518) < 1e-3  # slam_frontend.cc:565


def test_loader_accepts_a_full_file_set(stand_in_dir, small_cases):
    """The comparison on files in the kit's format (written from the oracle: the stand-in for OpenCV here): every stage of
    every case compares equal, all thirteen files per case are read."""
    results = pin_compare.compare_dir(stand_in_dir, small_cases)
    assert set(results) == {c[0] for c in small_cases}
    for name, result in results.items():
        assert [s for s, _, _ in result] == pin_compare.STAGE_NAMES
        assert pin_compare.first_divergence(result) is None, (name, pin_compare.first_divergence(result))
    ref = pin_compare.load_case(stand_in_dir, "stereo_320x240_nf500")
    assert len(ref["L_kp"]) > 300 and len(ref["matches"]) > 20 and ref["points4d"].shape == (len(ref["matches"]), 4)
    assert ref["L_level_shapes"].shape == (50, 2) and ref["L_pyramid"].size == int(ref["L_level_shapes"].prod(1).sum())
    assert ref["L_pyramid_digest"].dtype == np.uint64 and ref["L_pyramid_digest"].shape == (50,)
    # a larger case carries per-level digests only (a whole set of pyramids would be 60 MB)
    big = pin_compare.load_case(stand_in_dir, "photo_coins_384x303")
    assert "L_pyramid" not in big and big["L_blur_digest"].shape == (50,)


@pytest.mark.parametrize("stage", ["pyramid", "blur", "fast10", "keypoint_positions", "keypoint_response", "keypoint_angle",
                                   "descriptors", "knn", "matches", "triangulate", "undistort"])
def test_loader_names_the_first_diverging_stage(tmp_path, stand_in_dir, small_cases, stage):
    """A file set that differs from the oracle in ONE stage (and, for the image stages, in everything downstream as real
    data would): the diagnosis names that stage, and its hint points at the right row of INTEGRATION.md section 8."""
    name = "stereo_320x240_nf500"
    for f in stand_in_dir.glob(name + "__*.npy"):
        shutil.copy(f, tmp_path / f.name)

    def edit(key, fn):
        a = np.load(tmp_path / ("%s__%s.npy" % (name, key)))
        a = fn(a.copy())
        np.save(tmp_path / ("%s__%s.npy" % (name, key)), a)

    def bump(a, i=1000):
        a.reshape(-1)[i] ^= 1
        return a

    def field(f, fn):
        def g(a):
            a[f][3] = fn(a[f][3])
            return a
        return g

    if stage == "pyramid":
        edit("L_pyramid", lambda a: bump(a, 320 * 240 + 77))  # a pixel of level 1
        edit("L_blur", lambda a: bump(a, 320 * 240 + 77))     # ... and what follows from it
        edit("L_desc", bump)
    elif stage == "blur":
        edit("L_blur_digest", lambda a: bump(a, 7))            # digests alone say which level
        edit("L_desc", bump)
    elif stage == "fast10":
        edit("L_fast10", field("response", lambda v: v + 1))
    elif stage == "keypoint_positions":
        edit("L_kp", field("x", lambda v: v + 1))
        edit("L_desc", bump)
    elif stage == "keypoint_response":
        edit("R_kp", field("response", lambda v: np.nextafter(v, np.float32(1))))
    elif stage == "keypoint_angle":
        edit("L_kp", field("angle", lambda v: np.nextafter(v, np.float32(400))))
    elif stage == "descriptors":
        edit("R_desc", bump)
    elif stage == "knn":
        edit("knn_idx", lambda a: bump(a, 1))
    elif stage == "matches":
        edit("matches", lambda a: a[:-1])
    elif stage == "triangulate":
        def four_row_like(a):
            a[:, 2] *= np.float32(1.001)  # a different DLT system moves the points far beyond rounding
            return a
        edit("points4d", four_row_like)
    elif stage == "undistort":
        edit("undistorted", lambda a: a + np.float32(0.01))
    case = [c for c in small_cases if c[0] == name]
    got = pin_compare.first_divergence(pin_compare.compare_dir(tmp_path, case)[name])
    assert got is not None and got[0] == stage, got
    if stage == "pyramid":
        assert "level 1: 1 of" in got[1]
    if stage == "blur":
        assert "levels [7] differ" in got[1]
    expect_in_hint = {"pyramid": "WITH_IPP", "blur": "blur_sse2", "triangulate": "triangulate_rows", "matches": "ratio",
                      "keypoint_positions": "libstdc++", "descriptors": "cos(angle)"}
    if stage in expect_in_hint:
        assert expect_in_hint[stage] in got[2]
    # homogeneous scale and sign of the triangulated points are NOT a divergence (the SVD's null vector is defined up to both)
    if stage == "triangulate":
        shutil.copy(stand_in_dir / (name + "__points4d.npy"), tmp_path / (name + "__points4d.npy"))
        edit("points4d", lambda a: a * np.float32(-2.0))
        assert pin_compare.first_divergence(pin_compare.compare_dir(tmp_path, case)[name]) is None


def test_missing_files_are_reported(tmp_path, stand_in_dir):
    name = "stereo_320x240_nf500"
    for f in stand_in_dir.glob(name + "__*.npy"):
        if "knn_dist" not in f.name:
            shutil.copy(f, tmp_path / f.name)
    with pytest.raises(FileNotFoundError, match="knn_dist"):
        pin_compare.load_case(tmp_path, name)


@pytest.mark.skipif(not OPENCV_DIR.is_dir(), reason="tests/golden/opencv/ absent: nobody has run tools/pin_with_opencv.cc with a "
                                                    "real OpenCV 3.2.0 yet (INTEGRATION.md section 8) -- parity stays unpinned")
def test_oracle_equals_opencv(oracle):
    """THE pin: every stage of every case of the kit, oracle vs the files a real OpenCV wrote."""
    version = (OPENCV_DIR / "VERSION.txt").read_text()
    assert version.startswith("OpenCV "), "tests/golden/opencv/ was not written by tools/pin_with_opencv.cc: " + version[:80]
    failures = []
    for name, result in pin_compare.compare_dir(OPENCV_DIR, pin_inputs.cases()).items():
        d = pin_compare.first_divergence(result)
        if d is not None:
            failures.append("%s: first divergence at `%s` (%s)\n      -> %s" % (name, d[0], d[1], d[2]))
    assert not failures, "oracle != %s:\n  %s" % (version.splitlines()[0], "\n  ".join(failures))
