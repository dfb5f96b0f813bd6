"""Definition-level known-answer tests that pin the CPU oracle (PARITY UNPINNED: the reference has no golden
vectors and OpenCV 3.2.0 is not available; these KATs come from the published definitions
of the algorithms and from constants derived in SURVEY.md Appendix A / C)."""
import hashlib

import numpy as np
import pytest

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _img(v=100, size=15):
    return np.full((size, size), v, np.uint8)


def test_orb_pattern_hash_and_first_rows(oracle):
    p = oracle.orb_pattern31()
    assert p.shape == (256, 4) and p.dtype == np.int8
    assert hashlib.sha256(p.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    assert p[0].tolist() == [8, -3, 9, 5] and p[1].tolist() == [4, 2, 7, -12] and p[2].tolist() == [-11, 9, -8, 2]
    assert p.min() == -13 and p.max() == 12
    # product and oracle are generated at build time from ONE tracked table (data/orb_pattern31.txt)
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    rows = [l.split() for l in (root / "data" / "orb_pattern31.txt").read_text().splitlines() if l and l[0] != "#"]
    assert np.array_equal(np.array(rows, np.int64).astype(np.int8), p)
    a = (root / "oracle" / "orb_pattern31.inc").read_text()
    b = (root / "vision_slam_frontend_amd" / "csrc" / "orb_pattern31.inc").read_text()
    assert a == b and "GENERATED" in a
    tracked = subprocess.run(["git", "ls-files", "*orb_pattern31.inc"], cwd=root, capture_output=True, text=True)
    assert tracked.returncode != 0 or tracked.stdout.strip() == ""  # the generated copies are not in the history


def test_pyramid_geometry_appendix_c(oracle):
    o = oracle.Orb(nfeatures=2000)
    lv = o.layout(640, 480)
    assert [lv[l][:2] for l in (0, 1, 2, 10, 25, 49)] == [(640, 480), (615, 462), (592, 444), (432, 324), (240, 180),
                                                           (94, 70)]
    assert sum(w * h for w, h, _, _ in lv) == 3992038
    assert np.float32(lv[1][2]) == np.float32(1.04) and abs(lv[49][2] - 6.8333373) < 1e-5
    n2000 = [x[3] for x in lv]
    assert n2000[:5] == [90, 86, 83, 80, 77] and n2000[-3:] == [14, 14, 8] and sum(n2000) == 2000
    n10000 = [x[3] for x in oracle.Orb(nfeatures=10000).layout(640, 480)]
    assert n10000[:5] == [448, 430, 414, 398, 383] and n10000[-3:] == [71, 68, 63] and sum(n10000) == 10000
    n8000 = [x[3] for x in oracle.Orb(nfeatures=8000).layout(1920, 1080)]
    assert n8000[:5] == [358, 344, 331, 318, 306] and n8000[-3:] == [57, 54, 53] and sum(n8000) == 8000
    lv1080 = oracle.Orb(nfeatures=8000).layout(1920, 1080)
    assert lv1080[1][:2] == (1846, 1038) and lv1080[49][:2] == (281, 158)
    assert sum(w * h for w, h, _, _ in lv1080) == 26944950


def test_fast_segment_test_exactly_nine(oracle):
    # 9 contiguous brighter pixels -> corner; 8 -> not. Strictness: p > v + t.
    for start in range(16):
        img = _img()
        for k in range(9):
            dx, dy = CIRCLE[(start + k) % 16]
            img[7 + dy, 7 + dx] = 100 + 21
        def centre_is_corner(im):
            return (7, 7) in {(int(k["x"]), int(k["y"])) for k in oracle.fast9_16(im, 20, nms=False)}
        assert centre_is_corner(img), start
        img8 = img.copy()
        dx, dy = CIRCLE[(start + 8) % 16]
        img8[7 + dy, 7 + dx] = 100
        assert not centre_is_corner(img8)
        # exactly v + t is NOT brighter
        img_eq = img.copy()
        dx, dy = CIRCLE[start]
        img_eq[7 + dy, 7 + dx] = 120
        assert not centre_is_corner(img_eq)


def test_fast_score_is_max_threshold(oracle):
    # score = largest t' for which the pixel is still a corner at threshold t' (cornerScore definition)
    rng = np.random.default_rng(3)
    for _ in range(30):
        img = _img()
        vals = rng.integers(125, 200, 9)
        start = int(rng.integers(0, 16))
        for k in range(9):
            dx, dy = CIRCLE[(start + k) % 16]
            img[7 + dy, 7 + dx] = vals[k]
        s = oracle.fast_corner_score(img, 7, 7, 20)
        assert s == int(vals.min()) - 100 - 1

        def centre_is_corner(t):
            return (7, 7) in {(int(k["x"]), int(k["y"])) for k in oracle.fast9_16(img, t, nms=False)}
        assert centre_is_corner(s) and not centre_is_corner(s + 1)


def test_fast_dark_corner_and_keypoint_fields(oracle):
    img = _img(200)
    for k in range(11):
        dx, dy = CIRCLE[k]
        img[7 + dy, 7 + dx] = 100
    kp = oracle.fast9_16(img, 10, nms=True)
    k = [k for k in kp if (k["x"], k["y"]) == (7, 7)]
    assert len(k) == 1
    k = k[0]
    assert (k["x"], k["y"], k["size"], k["angle"], k["octave"], k["class_id"]) == (7, 7, 7, -1, 0, -1)
    assert k["response"] == 99  # min(v - p) - 1


def test_fast_nms_strictly_greater(oracle):
    # two adjacent corners with equal score suppress each other; a higher one survives alone
    big = np.full((15, 24), 100, np.uint8)
    def put(cx, level):
        for k in range(9):
            dx, dy = CIRCLE[k]
            big[7 + dy, cx + dx] = level
    put(7, 150)
    no_nms = oracle.fast9_16(big, 20, nms=False)
    with_nms = oracle.fast9_16(big, 20, nms=True)
    assert len(with_nms) <= len(no_nms)
    scores = {(int(k["x"]), int(k["y"])): int(k["response"]) for k in with_nms}
    for (x, y), s in scores.items():
        for (x2, y2), s2 in scores.items():
            if (x, y) != (x2, y2):
                assert max(abs(x - x2), abs(y - y2)) > 1  # no two 8-adjacent survivors
    # raster order
    order = [(int(k["y"]), int(k["x"])) for k in no_nms]
    assert order == sorted(order)


def test_fast_rim_is_three_pixels(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    kp = oracle.fast9_16(img, 5, nms=False)
    assert len(kp) > 50
    assert kp["x"].min() >= 3 and kp["x"].max() <= 50 - 4 and kp["y"].min() >= 3 and kp["y"].max() <= 40 - 4


def test_gaussian_kernel_and_constant_image(oracle):
    k = oracle.gaussian_kernel7_fixed()
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18] and k.sum() == 257
    # the 257/256 gain per axis is NOT renormalised: a flat image of v maps to round(v * 257^2 / 65536)
    for v in (0, 1, 100, 200, 255):
        out = oracle.gaussian_blur7(_img(v, 20))
        assert (out == min(255, (v * 257 * 257 + 32768) >> 16)).all(), v


def test_gaussian_blur_matches_direct_2d_sum_and_tie_rule(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (33, 38), dtype=np.uint8)  # width % 4 == 2: last two columns use the scalar rule
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    n = np.zeros(img.shape, np.int64)
    for i in range(7):
        for j in range(7):
            n += k[i] * k[j] * pad[i:i + 33, j:j + 38]
    half_up = np.minimum((n + 32768) >> 16, 255)
    q, rem = n >> 16, n & 0xFFFF
    half_even = np.minimum(q + ((rem > 32768) | ((rem == 32768) & (q & 1 == 1))), 255)
    got = oracle.gaussian_blur7(img, sse2=True)
    np.testing.assert_array_equal(got[:, :36], half_even[:, :36])
    np.testing.assert_array_equal(got[:, 36:], half_up[:, 36:])
    np.testing.assert_array_equal(oracle.gaussian_blur7(img, sse2=False), half_up)


def test_resize_identity_and_tables(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (30, 41), dtype=np.uint8)
    np.testing.assert_array_equal(oracle.resize_linear(img, 41, 30), img)  # scale 1: taps (2048, 0)
    xofs, ia, yofs, ib, xmax = oracle.resize_tables(640, 480, 615, 462)
    assert xofs[0] == 0 and ia[0] + ia[1] == 2048 and xofs[-1] == 638 and xmax == 615
    assert (ia.reshape(-1, 2).sum(1) == 2048).all() and (ib.reshape(-1, 2).sum(1) == 2048).all()
    assert (np.diff(xofs) >= 1).all() and (np.diff(xofs) <= 2).all()
    # constant image stays constant through the fixed-point bilinear
    flat = np.full((480, 640), 137, np.uint8)
    assert (oracle.resize_linear(flat, 615, 462) == 137).all()
    # a horizontal ramp stays within +-1 of the ideal bilinear sample
    ramp = np.tile((np.arange(640) // 3).astype(np.uint8), (480, 1))
    out = oracle.resize_linear(ramp, 615, 462).astype(np.float64)
    sx = (np.arange(615) + 0.5) * (640 / 615) - 0.5
    ideal = np.interp(sx, np.arange(640), ramp[0].astype(np.float64))
    assert np.abs(out[5] - ideal).max() <= 1.0


def test_fast_atan2_quadrants_and_accuracy(oracle):
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    assert abs(oracle.fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -1.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-1.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(4)
    for _ in range(200):
        y, x = (float(v) for v in rng.integers(-50000, 50000, 2))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.02  # 7th-order polynomial: ~0.01 degree


def test_knn2_tie_rule_and_ratio_identity(oracle):
    t = np.zeros((6, 32), np.uint8)
    t[1, 0] = 1          # distance 1
    t[2, 0] = 1          # distance 1 (tie -> lower index first)
    t[3, 0] = 3          # distance 2
    t[4] = 255           # far
    q = np.zeros((1, 32), np.uint8)
    q[0, 1] = 0          # identical to rows 0 and 5 (distance 0, tie on 0 and 5)
    idx, dist = oracle.knn2_hamming(q, t)
    assert idx.tolist() == [[0, 5]] and dist.tolist() == [[0, 0]]
    idx, dist = oracle.knn2_hamming(q, t[1:5])
    assert idx.tolist() == [[0, 1]] and dist.tolist() == [[1, 1]]
    idx, dist = oracle.knn2_hamming(q, t[:1])
    assert idx.tolist() == [[0, -1]] and dist[0, 0] == 0 and dist[0, 1] == np.iinfo(np.int32).max
    # ratio test in double with 0.6f widened: 3 < 0.6f * 5 is TRUE (quirk Q7); exact identity d1*2^24 < 10066330*d2
    r = float(np.float32(0.6))
    assert r * 2 ** 24 == 10066330
    for d1 in range(0, 60):
        for d2 in range(d1, 100):
            assert (d1 < r * d2) == (d1 * 2 ** 24 < 10066330 * d2)
    assert 3 < r * 5


def test_get_matches_semantics(oracle):
    rng = np.random.default_rng(8)
    t = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    q = t[[3, 10, 20]].copy()
    q[1, 0] ^= 0xFF  # 8 bits off: still far closer than random (~128)
    m = oracle.get_matches(q, t)
    assert m["queryIdx"].tolist() == [0, 1, 2] and m["trainIdx"].tolist() == [3, 10, 20]
    assert m["distance"].tolist() == [0, 8, 0] and (m["imgIdx"] == 0).all()
    assert len(oracle.get_matches(q, t[:1])) == 0  # fewer than 2 train rows: no matches (quirk Q6)
    assert len(oracle.get_matches(q[:0], t)) == 0
    # threads do not change the result
    assert oracle.get_matches(q, t, threads=3).tobytes() == m.tobytes()


def test_sort_and_trim(oracle):
    rng = np.random.default_rng(9)
    m = np.zeros(101, oracle.DMATCH_DTYPE)
    m["queryIdx"] = np.arange(101)
    m["trainIdx"] = rng.integers(0, 500, 101)
    m["distance"] = rng.integers(0, 12, 101)
    out = oracle.sort_and_trim(m, float(np.float32(0.3)))
    assert len(out) == int(np.float32(101) * np.float32(0.3)) == 30
    assert (np.diff(out["distance"]) >= 0).all()
    full = oracle.sort_and_trim(m, 1.0)
    assert sorted(full["queryIdx"].tolist()) == list(range(101))
    assert out.tobytes() == full[:30].tobytes()


def test_remove_ambig_stereo(oracle):
    K = oracle.KEYPOINT_DTYPE
    left = np.zeros(4, K)
    right = np.zeros(4, K)
    left["x"], left["y"] = [10, 20, 30, 40], [5, 6, 7, 8]
    right["x"], right["y"] = [8, 17, 26, 35], [5, 6, 9, 8]
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # l^T F r = y_r - y_l (rectified pair)
    m = np.zeros(4, oracle.DMATCH_DTYPE)
    m["queryIdx"] = m["trainIdx"] = np.arange(4)
    keep, res, thr, kept = oracle.remove_ambig_stereo(left, right, m, F, 10000.0)
    assert res.tolist() == [0, 0, 2, 0] and keep.all() and kept == 4
    assert thr == np.float32(2.0 / 4 + 2.0)
    keep, res, thr2, kept = oracle.remove_ambig_stereo(left, right, m, F, 1.0)
    assert keep.tolist() == [True, True, False, True] and kept == 3 and thr2 == thr
    # Quirk Q3 (slam_frontend.cc:392-394), frame by frame: frame A has no stereo match -> the static becomes 0/0 + 2 = NaN;
    # frame B (the frame AFTER the empty one) is filtered against NaN and keeps nothing, but its mean is finite;
    # frame C is filtered normally again.
    _, _, thr_a, kept_a = oracle.remove_ambig_stereo(left, right, m[:0], F, 3.5)
    assert np.isnan(thr_a) and kept_a == 0
    keep_b, _, thr_b, kept_b = oracle.remove_ambig_stereo(left, right, m, F, thr_a)
    assert kept_b == 0 and not keep_b.any() and thr_b == np.float32(2.0 / 4 + 2.0)
    keep_c, _, _, kept_c = oracle.remove_ambig_stereo(left, right, m, F, thr_b)
    assert kept_c == 4 and keep_c.all()


def _residuals_from_definition(left, right, F, eigen33):
    """|l^T F r| in float32, each three-term dot product summed as stated: Eigen 3.3 adds element 0 to (element 1 + element
    2) -- redux_novec_unroller<Func, Derived, 0, 3> splits 1 + 2 --, the alternative is left to right."""
    f32 = np.float32

    def dot3(a, b):
        p = [f32(f32(a[i]) * f32(b[i])) for i in range(3)]
        return f32(p[0] + f32(p[1] + p[2])) if eigen33 else f32(f32(p[0] + p[1]) + p[2])

    out = []
    for lk, rk in zip(left, right):
        l, r = [lk["x"], lk["y"], f32(1)], [rk["x"], rk["y"], f32(1)]
        t = [dot3(l, F[:, j]) for j in range(3)]
        out.append(abs(dot3(t, r)))
    return np.array(out, np.float32)


def test_remove_ambig_stereo_summation_order(oracle):
    """slam_frontend.cc:381-383 with a DENSE fundamental matrix: the two summation orders differ in the last bit for a
    good share of the matches (with the rectified F of the synthetic stream they cannot), the oracle's default is Eigen
    3.3's order and the switch gives the other one."""
    rng = np.random.Generator(np.random.PCG64(7))
    n = 400
    K = oracle.KEYPOINT_DTYPE
    left, right = np.zeros(n, K), np.zeros(n, K)
    left["x"], left["y"] = rng.uniform(0, 640, n).astype(np.float32), rng.uniform(0, 480, n).astype(np.float32)
    right["x"], right["y"] = rng.uniform(0, 640, n).astype(np.float32), rng.uniform(0, 480, n).astype(np.float32)
    # a dense F with entries of the very different magnitudes a calibrated pair gives (the reference computes its F from the
    # camera matrices, slam_frontend.cc:635-644: nine non-zero entries)
    F = np.array([[2.31e-08, -1.17e-05, 3.45e-03], [1.22e-05, 9.8e-08, -0.11], [-4.1e-03, 0.108, 1.0]], np.float32)
    m = np.zeros(n, oracle.DMATCH_DTYPE)
    m["queryIdx"] = m["trainIdx"] = np.arange(n)
    want_e = _residuals_from_definition(left, right, F, True)
    want_s = _residuals_from_definition(left, right, F, False)
    assert (want_e != want_s).sum() > n // 20  # the orders are distinguishable on this input
    try:
        _, res_e, thr_e, _ = oracle.remove_ambig_stereo(left, right, m, F, 10000.0)
        oracle.set_residual_order(1)
        _, res_s, thr_s, _ = oracle.remove_ambig_stereo(left, right, m, F, 10000.0)
    finally:
        oracle.set_residual_order(0)
    assert res_e.tobytes() == want_e.tobytes()
    assert res_s.tobytes() == want_s.tobytes()
    # x86's default float arithmetic keeps sqrt(x * x) == |x| for every residual here (the reference's .norm())
    assert np.array_equal(np.sqrt(res_e * res_e), res_e)


def test_retain_best_keeps_boundary_ties(oracle):
    keys = np.array([5, 9, 7, 7, 7, 1, 7, 3], np.float32)
    r, ids = oracle.retain_best(keys, 3)
    assert sorted(r.tolist(), reverse=True)[:1] == [9.0]
    assert (r >= 7).all() and len(r) >= 3
    r0, _ = oracle.retain_best(keys, 0)
    assert len(r0) == 0
    r9, ids9 = oracle.retain_best(keys, 8)
    assert ids9.tolist() == list(range(8))


def test_orb_end_to_end_invariants(oracle, stereo640):
    o = oracle.Orb(nfeatures=2000)
    n = o.run(stereo640[0])
    kp, desc = o.result()
    assert n == len(kp) == 2000 and desc.shape == (2000, 32)
    assert (np.diff(kp["octave"]) >= 0).all()  # level-major
    for l in (0, 7, 30):
        w, h, s, nf = o.level_info(l)
        k4 = o.stage(4, l)
        assert len(k4) <= nf + 5
        assert (k4["x"] >= 31).all() and (k4["x"] < w - 31).all() and (k4["y"] >= 31).all() and (k4["y"] < h - 31).all()
        assert np.allclose(k4["size"], 31 * s)
        assert ((k4["angle"] >= 0) & (k4["angle"] <= 360)).all()
        # stage 1 is a subset of stage 0 with at least min(len, 2*nf) members
        s0, s1 = o.stage(0, l), o.stage(1, l)
        assert len(s1) >= min(len(s0), 2 * nf)
        set0 = set(zip(s0["x"].tolist(), s0["y"].tolist()))
        assert set(zip(s1["x"].tolist(), s1["y"].tolist())) <= set0
    # descriptor bit 0 of keypoint j follows the definition: I(p0) < I(p1) on the blurred level
    j = 17
    l = int(kp[j]["octave"])
    w, h, s, _ = o.level_info(l)
    blurred = o.level_image(l, True)
    ang = np.float32(kp[j]["angle"]) * np.float32(np.pi / 180.0)
    a, b = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
    cx = int(np.rint(np.float32(kp[j]["x"]) * np.float32(1.0 / np.float32(s))))
    cy = int(np.rint(np.float32(kp[j]["y"]) * np.float32(1.0 / np.float32(s))))
    pat = oracle.orb_pattern31().astype(np.float32)
    bits = []
    for i in range(8):
        x0, y0, x1, y1 = pat[i]
        ix0, iy0 = int(np.rint(x0 * a - y0 * b)), int(np.rint(x0 * b + y0 * a))
        ix1, iy1 = int(np.rint(x1 * a - y1 * b)), int(np.rint(x1 * b + y1 * a))
        bits.append(int(blurred[cy + iy0, cx + ix0] < blurred[cy + iy1, cx + ix1]))
    assert sum(bit << i for i, bit in enumerate(bits)) == int(desc[j, 0])


def _bayer_bg_gray_by_site(m):
    """Independent formulation of BayerBG2BGR + BGR2GRAY: per-site bilinear formulas on the interior, frame replicated
    (the oracle follows OpenCV's loop structure instead)."""
    m = m.astype(np.int32)
    h, w = m.shape
    if w < 3 or h < 3:
        return np.zeros((h, w), np.uint8)
    U, D, L, R, C = m[:-2, 1:-1], m[2:, 1:-1], m[1:-1, :-2], m[1:-1, 2:], m[1:-1, 1:-1]
    diag = (m[:-2, :-2] + m[:-2, 2:] + m[2:, :-2] + m[2:, 2:] + 2) >> 2
    cross, hor, ver = (U + D + L + R + 2) >> 2, (L + R + 1) >> 1, (U + D + 1) >> 1
    yy, xx = np.mgrid[1:h - 1, 1:w - 1]
    ex, ey = xx & 1, yy & 1
    G = np.where(ex == ey, cross, C)
    B = np.where(ey == 1, np.where(ex == 1, C, hor), np.where(ex == 1, ver, diag))
    Rr = np.where(ey == 1, np.where(ex == 1, diag, ver), np.where(ex == 1, hor, C))
    out = np.zeros((h, w), np.int32)
    out[1:-1, 1:-1] = (1868 * B + 9617 * G + 4899 * Rr + 8192) >> 14
    out[1:-1, 0], out[1:-1, -1] = out[1:-1, 1], out[1:-1, -2]
    out[0], out[-1] = out[1], out[-2]
    return out.astype(np.uint8)


def test_bayer_bg_to_gray_known_answers(oracle):
    # a flat mosaic stays flat: (1868 + 9617 + 4899) * v + 8192 >> 14 == v
    flat = np.full((6, 8), 77, np.uint8)
    np.testing.assert_array_equal(oracle.bayer_bg_to_gray(flat), flat)
    # a single bright blue site at (3, 3): B = 255 there, R / G zero -> gray (1868 * 255 + 8192) >> 14 = 29;
    # its green neighbours see it as a horizontal / vertical blue average (128), its red diagonal neighbours as 64
    m = np.zeros((7, 7), np.uint8)
    m[3, 3] = 255
    g = oracle.bayer_bg_to_gray(m)
    assert g[3, 3] == (1868 * 255 + 8192) >> 14
    assert g[3, 2] == g[3, 4] == g[2, 3] == g[4, 3] == (1868 * 128 + 8192) >> 14
    assert g[2, 2] == g[4, 4] == (1868 * 64 + 8192) >> 14
    assert g[0, 0] == g[1, 1] and g[6, 6] == g[5, 5]  # frame copies its inner neighbour


@pytest.mark.parametrize("h,w", [(3, 3), (4, 4), (5, 7), (8, 9), (17, 16), (33, 40), (1, 5), (2, 9), (7, 2), (6, 1),
                                 (480, 640), (101, 203)])
def test_bayer_bg_to_gray_loop_restatement_equals_site_formulas(oracle, h, w):
    m = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w), dtype=np.uint8)
    np.testing.assert_array_equal(oracle.bayer_bg_to_gray(m), _bayer_bg_gray_by_site(m))
