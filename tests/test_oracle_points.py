"""SURVEY 8(f) row f2, CPU side: the oracle's restatement of cv::triangulatePoints / cv::undistortPoints as
Calculate3DPoints / UndistortFeaturePoints call them (slam_frontend.cc:117-173, 323-351), checked against statements
written from the DEFINITIONS (numpy float64 SVD of the same DLT system; the forward distortion model), not from the
oracle's code."""
import numpy as np
import pytest


def _reference_calibration():
    # FrontendConfig's literals (slam_frontend.cc:565-611), composed in float64 here
    K1 = np.array([[527.873518, 0, 482.823413], [0, 527.276819, 298.033945], [0, 0, 1]])
    K2 = np.array([[530.158021, 0, 475.540633], [0, 529.682234, 299.995465], [0, 0, 1]])
    A2 = np.array([[0.999593617649873, 0.021411909431148, -0.018818333830411, -0.131707087331978],
                   [-0.021140534893290, 0.999671312094879, 0.014503294761121, 0.003232397463343],
                   [0.019122691705565, -0.014099571235136, 0.999717722536176, -0.001146108483477]])
    P1 = (K1 @ np.eye(3, 4)).astype(np.float32)
    P2 = (K2 @ A2).astype(np.float32)
    dist = np.array([-0.153137, 0.075666, -0.000227, -0.000320, 0], np.float32)
    return K1.astype(np.float32), P1, P2, dist


def _project(P, X):
    x = (P.astype(np.float64) @ X.T).T
    return x[:, :2] / x[:, 2:]


@pytest.mark.parametrize("rows", [6, 4])
def test_triangulation_recovers_projected_points(oracle, rows):
    _, P1, P2, _ = _reference_calibration()
    rng = np.random.default_rng(7)
    X = np.c_[rng.uniform(-3, 3, 200), rng.uniform(-2, 2, 200), rng.uniform(0.5, 30, 200), np.ones(200)]
    x1, x2 = _project(P1, X), _project(P2, X)
    Y = oracle.triangulate_points(P1, P2, x1, x2, rows)
    assert Y.dtype == np.float32 and Y.shape == (200, 4)
    Yn = Y[:, :3].astype(np.float64) / Y[:, 3:].astype(np.float64)
    # float32 pixel coordinates limit the depth accuracy (13 cm baseline): relative depth error ~ z / (f b) * 2^-24 * x
    err = np.abs(Yn - X[:, :3]) / np.maximum(np.abs(X[:, :3]), 1.0)
    assert err.max() < 2e-3 and np.median(err) < 2e-5


@pytest.mark.parametrize("rows", [6, 4])
def test_triangulation_matches_float64_svd_of_the_dlt_system(oracle, rows):
    """Noisy correspondences: the answer is DEFINED as the right singular vector of the smallest singular value of the
    rows x 4 system (x*P2-P0, y*P2-P1[, x*P1-y*P0] per view) built from the float32 inputs widened to double."""
    _, P1, P2, _ = _reference_calibration()
    rng = np.random.default_rng(11)
    n = 300
    X = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(0.5, 30, n), np.ones(n)]
    x1 = (_project(P1, X) + rng.normal(0, 1.0, (n, 2))).astype(np.float32)
    x2 = (_project(P2, X) + rng.normal(0, 1.0, (n, 2))).astype(np.float32)
    Y = oracle.triangulate_points(P1, P2, x1, x2, rows).astype(np.float64)
    P = [P1.astype(np.float64), P2.astype(np.float64)]
    worst = 0.0
    for i in range(n):
        A = []
        for j, x in enumerate((x1[i].astype(np.float64), x2[i].astype(np.float64))):
            A.append(x[0] * P[j][2] - P[j][0])
            A.append(x[1] * P[j][2] - P[j][1])
            if rows == 6:
                A.append(x[0] * P[j][1] - x[1] * P[j][0])
        _, sv, Vt = np.linalg.svd(np.array(A))
        v = Vt[3]
        v = v if np.dot(v, Y[i]) >= 0 else -v
        gap = (sv[2] - sv[3]) / sv[0]
        # both are unit vectors; the oracle's is rounded to float32 (6e-8), conditioning enters through the gap
        worst = max(worst, np.abs(v - Y[i]).max() * min(gap * 1e3, 1.0))
        assert np.abs(v - Y[i]).max() < 1e-6 + 1e-13 / max(gap, 1e-16), (i, gap)
    assert worst < 1e-6


def test_six_and_four_row_systems_differ_on_noisy_points(oracle):
    """The version switch (vsf_calibration.triangulate_rows) is not cosmetic: with noise the two systems weigh the
    equations differently and the points differ far beyond rounding."""
    _, P1, P2, _ = _reference_calibration()
    rng = np.random.default_rng(3)
    X = np.c_[rng.uniform(-2, 2, 50), rng.uniform(-1, 1, 50), rng.uniform(1, 10, 50), np.ones(50)]
    x1 = _project(P1, X) + rng.normal(0, 1.0, (50, 2))
    x2 = _project(P2, X)
    Y6, Y4 = oracle.triangulate_points(P1, P2, x1, x2, 6), oracle.triangulate_points(P1, P2, x1, x2, 4)
    d = np.abs(Y6[:, :3] / Y6[:, 3:] - Y4[:, :3] / Y4[:, 3:]).max()
    assert d > 1e-3


def test_undistort_inverts_the_forward_distortion_model(oracle):
    K, _, _, dist = _reference_calibration()
    k1, k2, p1, p2, k3 = dist.astype(np.float64)
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    rng = np.random.default_rng(5)
    xn = np.c_[rng.uniform(-0.6, 0.6, 500), rng.uniform(-0.45, 0.45, 500)]  # normalised, undistorted
    r2 = (xn ** 2).sum(1)
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = xn[:, 0] * radial + 2 * p1 * xn[:, 0] * xn[:, 1] + p2 * (r2 + 2 * xn[:, 0] ** 2)
    yd = xn[:, 1] * radial + p1 * (r2 + 2 * xn[:, 1] ** 2) + 2 * p2 * xn[:, 0] * xn[:, 1]
    distorted = np.c_[xd * fx + cx, yd * fy + cy].astype(np.float32)
    und = oracle.undistort_points(distorted, K, dist)
    ideal = np.c_[xn[:, 0] * fx + cx, xn[:, 1] * fy + cy]
    # five fixed-point iterations: converged to well below a pixel inside this field of view
    assert np.abs(und - ideal).max() < 0.05
    # against the same five iterations written in vectorised numpy float64 from the published formula
    x = (distorted[:, 0].astype(np.float64) - cx) / 1.0 * (1.0 / fx)
    y = (distorted[:, 1].astype(np.float64) - cy) * (1.0 / fy)
    x0, y0 = x.copy(), y.copy()
    for _ in range(5):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    ref = np.c_[x * fx + cx, y * fy + cy]
    assert np.abs(und.astype(np.float64) - ref).max() <= 2 * np.spacing(np.float32(1000.0))


def test_vision_features_follow_sorted_match_order(oracle):
    """slam_frontend.cc:437-443: points[i] is the i-th SORTED right->left match, attached to keypoint i (quirk Q5)."""
    K, P1, P2, dist = _reference_calibration()
    rng = np.random.default_rng(9)
    n = 40
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    left = np.zeros(n, oracle.KEYPOINT_DTYPE)
    right = np.zeros(n, oracle.KEYPOINT_DTYPE)
    left["x"], left["y"] = rng.uniform(100, 500, n), rng.uniform(100, 400, n)
    right["x"], right["y"] = left["x"] - rng.uniform(2, 40, n), left["y"] + rng.uniform(-1, 1, n)
    rdesc = desc.copy()
    flips = rng.integers(0, 12, n)  # right descriptor = left descriptor with `flips` bits flipped
    for i in range(n):
        for b in rng.choice(256, flips[i], replace=False):
            rdesc[i, b // 8] ^= 1 << (b % 8)
    feats, npts = oracle.vision_features(left, desc, right, rdesc, P1, P2, K, dist)
    m = oracle.sort_and_trim(oracle.get_matches(rdesc, desc), 1.0)
    assert npts == len(m) == n and list(feats["feature_idx"]) == list(range(n))
    X = oracle.triangulate_points(P1, P2, np.c_[left["x"], left["y"]][m["trainIdx"]],
                                  np.c_[right["x"], right["y"]][m["queryIdx"]], 6)
    np.testing.assert_array_equal(feats["point3d"], X[:, :3] / X[:, 3:])
    np.testing.assert_array_equal(feats["pixel"], oracle.undistort_points(np.c_[left["x"], left["y"]], K, dist))
    assert (np.diff(m["distance"]) >= 0).all() and not np.array_equal(m["queryIdx"], np.arange(n))
