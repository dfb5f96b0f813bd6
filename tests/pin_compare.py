"""Second half of the OpenCV pin kit: compares the files tools/pin_with_opencv.cc wrote (real OpenCV 3.2 outputs) with the
in-repo oracle, stage by stage, and says WHICH stage diverges first and which switch of INTEGRATION.md section 8 that
points at.  Also writes the same file set FROM the oracle (write_from_oracle), which is how the CPU test-suite exercises
this loader without OpenCV (tests/test_pinned_by_opencv.py).

    python3 tests/pin_compare.py tests/golden/opencv        # table of every case and stage; exit code 1 on a divergence
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tools", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

# Stage order = data-flow order: the first stage that differs is the one to look at; later ones differ as a consequence.
# (stage name, what a divergence there means / which INTEGRATION.md section 8 row to read)
STAGES = [
    ("level_shapes", "pyramid level sizes: cvRound(cols / (float)pow(1.04f, l)) -- vsf_level_info / Appendix C of SURVEY.md"),
    ("pyramid", "cv::resize INTER_LINEAR 8u (row `cv::resize 8-bit path`): was OpenCV built with IPP?  VERSION.txt says; "
                "rebuild with -DWITH_IPP=OFF or run the kit without --keep-ipp"),
    ("blur", "cv::GaussianBlur 7x7 sigma 2 (row `GaussianBlur column pass`): flip vsf_params.blur_sse2 / "
             "vsfo_orb_params.blur_sse2 to 0 if only exact ties differ"),
    ("fast10", "FAST-9/16 threshold 10 + NMS (row `FAST score / NMS / raster order`)"),
    ("keypoint_positions", "ORB keypoint set and order (x, y, octave, size): FAST 20 + retainBest + Harris ranking (rows `FAST "
                           "score ...` and `libstdc++ permutations`: nth_element / partition of the reference's libstdc++)"),
    ("keypoint_response", "Harris response floats (row `FAST score / NMS / raster order; Harris float order`)"),
    ("keypoint_angle", "IC angle via fastAtan2 (same row)"),
    ("descriptors", "rBRIEF bits (row `cos(angle) / sin(angle) in computeOrbDescriptors`, or the blur stage above)"),
    ("knn", "BFMatcher::knnMatch(k = 2): (distance, train index) tie rule"),
    ("matches", "ratio test dist1 < (double)0.6f * dist2 (vsf_params.ratio_num / ratio_shift)"),
    ("triangulate", "cv::triangulatePoints (row `6 rows per point`): vsf_calibration.triangulate_rows = 4 for OpenCV >= 3.4.2"),
    ("undistort", "cv::undistortPoints, 5 fixed iterations"),
]
STAGE_NAMES = [s for s, _ in STAGES]
HINT = dict(STAGES)
FILES = ["L_kp", "L_desc", "R_kp", "R_desc", "L_fast10", "L_pyramid_digest", "L_blur_digest", "L_level_shapes", "knn_idx", "knn_dist",
         "matches", "points4d", "undistorted"]
OPTIONAL_FILES = ["L_pyramid", "L_blur"]  # the whole pyramids: written for the small cases only (cases.txt: `full`)
FULL_PIXELS = 322 * 242                   # tools/pin_inputs.py marks images up to this size `full`


def level_digest(level: np.ndarray) -> np.uint64:
    """Position-dependent checksum of a level's bytes, as tools/pin_with_opencv.cc computes it (uint64 wrap-around)."""
    b = np.ascontiguousarray(level, np.uint8).reshape(-1).astype(np.uint64)
    i = np.arange(b.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mult = (i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)) | np.uint64(1)
        return ((b + np.uint64(1)) * mult).sum(dtype=np.uint64)


def _oracle():
    from oracle import binding as ob
    ob.build()
    return ob


def _calibration():
    import pin_inputs
    return pin_inputs.calibration()


def oracle_outputs(nfeatures: int, left: np.ndarray, right: np.ndarray) -> dict:
    """Every array of one case as the ORACLE computes it, keyed like the kit's files."""
    ob = _oracle()
    cal = _calibration()
    out, descs, kps = {}, [], []
    for side, img in (("L", left), ("R", right)):
        o = ob.Orb(nfeatures=nfeatures)
        o.run(np.ascontiguousarray(img))
        kp, desc = o.result()
        out[side + "_kp"], out[side + "_desc"] = kp, desc.reshape(-1, 32)
        kps.append(kp)
        descs.append(desc.reshape(-1, 32))
        if side == "L":
            shapes = [o.level_info(l)[:2][::-1] for l in range(o.nlevels)]  # (rows, cols)
            out["L_level_shapes"] = np.asarray(shapes, np.int32).reshape(-1, 2)
            levels = [(o.level_image(l, False), o.level_image(l, True)) for l in range(o.nlevels)]
            out["L_pyramid"] = np.concatenate([a.reshape(-1) for a, _ in levels])
            out["L_blur"] = np.concatenate([b.reshape(-1) for _, b in levels])
            out["L_pyramid_digest"] = np.asarray([level_digest(a) for a, _ in levels], np.uint64)
            out["L_blur_digest"] = np.asarray([level_digest(b) for _, b in levels], np.uint64)
    out["L_fast10"] = ob.fast9_16(np.ascontiguousarray(left), 10, True)
    if len(descs[0]) and len(descs[1]):
        idx, dist = ob.knn2_hamming(descs[0], descs[1])
        m = ob.get_matches(descs[0], descs[1])
    else:
        idx, dist, m = np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32), np.zeros(0, ob.DMATCH_DTYPE)
    out["knn_idx"], out["knn_dist"], out["matches"] = idx.astype(np.int32), dist.astype(np.int32), m
    lp = np.stack([kps[0]["x"][m["queryIdx"]], kps[0]["y"][m["queryIdx"]]], 1).astype(np.float32).reshape(-1, 2)
    rp = np.stack([kps[1]["x"][m["trainIdx"]], kps[1]["y"][m["trainIdx"]]], 1).astype(np.float32).reshape(-1, 2)
    if len(m):
        out["points4d"] = ob.triangulate_points(cal["projection_left"], cal["projection_right"], lp, rp, rows=6)
        out["undistorted"] = ob.undistort_points(lp, cal["camera_matrix_left"], cal["distortion_left"])
    else:
        out["points4d"], out["undistorted"] = np.zeros((0, 4), np.float32), np.zeros((0, 2), np.float32)
    return out


def write_from_oracle(out_dir, cases) -> None:
    """The kit's file set with the oracle standing in for OpenCV (same names, dtypes and shapes as pin_with_opencv.cc)."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    for name, nf, left, right in cases:
        for key, arr in oracle_outputs(nf, left, right).items():
            if key in OPTIONAL_FILES and left.size > FULL_PIXELS:
                continue
            np.save(out_dir / ("%s__%s.npy" % (name, key)), np.ascontiguousarray(arr))
    (out_dir / "VERSION.txt").write_text("oracle stand-in (tests/pin_compare.py write_from_oracle): NOT OpenCV\ncases %d\n" % len(cases))


def load_case(ref_dir, name: str) -> dict:
    ref_dir = Path(ref_dir)
    missing = [k for k in FILES if not (ref_dir / ("%s__%s.npy" % (name, k))).exists()]
    if missing:
        raise FileNotFoundError("case %s: missing %s in %s" % (name, missing, ref_dir))
    ref = {k: np.load(ref_dir / ("%s__%s.npy" % (name, k))) for k in FILES}
    for k in OPTIONAL_FILES:
        if (ref_dir / ("%s__%s.npy" % (name, k))).exists():
            ref[k] = np.load(ref_dir / ("%s__%s.npy" % (name, k)))
    return ref


def _levels(flat: np.ndarray, shapes: np.ndarray):
    off = 0
    for r, c in shapes:
        yield flat[off:off + int(r) * int(c)].reshape(int(r), int(c))
        off += int(r) * int(c)


def compare_case(ref: dict, mine: dict):
    """[(stage, ok, detail)] in STAGES order; `ref` = the kit's files (OpenCV), `mine` = oracle_outputs of the same inputs."""
    res = []

    def add(stage, ok, detail=""):
        res.append((stage, bool(ok), detail))

    same_shapes = ref["L_level_shapes"].shape == mine["L_level_shapes"].shape and np.array_equal(ref["L_level_shapes"], mine["L_level_shapes"])
    add("level_shapes", same_shapes, "" if same_shapes else "OpenCV %s ... vs oracle %s ..." % (ref["L_level_shapes"][:3].tolist(), mine["L_level_shapes"][:3].tolist()))
    for stage, key in (("pyramid", "L_pyramid"), ("blur", "L_blur")):
        if not same_shapes:
            add(stage, False, "sizes differ")
            continue
        bad = []
        if key in ref and ref[key].shape == mine[key].shape:  # the whole pyramid is there: pixel-level detail
            for l, (a, b) in enumerate(zip(_levels(ref[key], ref["L_level_shapes"]), _levels(mine[key], mine["L_level_shapes"]))):
                n = int((a != b).sum())
                if n:
                    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
                    bad.append("level %d: %d of %d pixels differ (max |diff| %d)" % (l, n, a.size, int(d.max())))
        rd, md = ref[key + "_digest"], mine[key + "_digest"]
        if rd.shape != md.shape:
            bad.append("digest arrays differ in size")
        elif not bad:
            lv = np.flatnonzero(rd != md)
            if len(lv):
                bad.append("levels %s differ (per-level digests; dump those levels with OpenCV for the pixels)" % lv[:8].tolist())
        add(stage, not bad, "; ".join(bad[:4]) + (" ..." if len(bad) > 4 else ""))

    def kp_fields(stage, a, b, fields):
        if len(a) != len(b):
            add(stage, False, "%d keypoints (OpenCV) vs %d (oracle)" % (len(a), len(b)))
            return
        for f in fields:
            ne = np.flatnonzero(a[f].view(np.uint32) != b[f].view(np.uint32)) if a[f].dtype.kind == "f" else np.flatnonzero(a[f] != b[f])
            if len(ne):
                i = int(ne[0])
                add(stage, False, "%d of %d differ in `%s`; first at %d: OpenCV %r vs oracle %r" % (len(ne), len(a), f, i, a[f][i], b[f][i]))
                return
        add(stage, True)

    kp_fields("fast10", ref["L_fast10"], mine["L_fast10"], ("x", "y", "response", "size", "angle", "octave", "class_id"))
    for side in ("L", "R"):
        kp_fields("keypoint_positions", ref[side + "_kp"], mine[side + "_kp"], ("x", "y", "octave", "size", "class_id"))
        kp_fields("keypoint_response", ref[side + "_kp"], mine[side + "_kp"], ("response",))
        kp_fields("keypoint_angle", ref[side + "_kp"], mine[side + "_kp"], ("angle",))
        a, b = ref[side + "_desc"].reshape(-1, 32), mine[side + "_desc"].reshape(-1, 32)
        if a.shape != b.shape:
            add("descriptors", False, "%s vs %s" % (a.shape, b.shape))
        else:
            rows = np.flatnonzero((a != b).any(1))
            bits = int(np.unpackbits(a ^ b).sum())
            add("descriptors", len(rows) == 0, "" if len(rows) == 0 else "%s: %d of %d descriptors differ, %d bits in all; first row %d" % (side, len(rows), len(a), bits, int(rows[0])))
    ok = ref["knn_idx"].shape == mine["knn_idx"].shape and np.array_equal(ref["knn_idx"], mine["knn_idx"]) and np.array_equal(ref["knn_dist"], mine["knn_dist"])
    add("knn", ok, "" if ok else "2-NN tables differ")
    ok = ref["matches"].shape == mine["matches"].shape and ref["matches"].tobytes() == mine["matches"].tobytes()
    add("matches", ok, "" if ok else "%d matches (OpenCV) vs %d (oracle)" % (len(ref["matches"]), len(mine["matches"])))
    a, b = ref["points4d"].astype(np.float64).reshape(-1, 4), mine["points4d"].astype(np.float64).reshape(-1, 4)
    if a.shape != b.shape:
        add("triangulate", False, "%s vs %s" % (a.shape, b.shape))
    else:
        # homogeneous points are defined up to scale (and the SVD's sign): compare the de-homogenised points, 1e-5 relative
        with np.errstate(divide="ignore", invalid="ignore"):
            pa, pb = a[:, :3] / a[:, 3:4], b[:, :3] / b[:, 3:4]
        fin = np.isfinite(pa) & np.isfinite(pb)
        rel = np.abs(pa - pb)[fin] / np.maximum(np.abs(pb[fin]), 1e-30)
        worst = float(rel.max(initial=0.0))
        ok = np.array_equal(np.isfinite(pa), np.isfinite(pb)) and worst <= 1e-5
        add("triangulate", ok, "" if ok else "largest relative difference of (x, y, z) / w: %.3g" % worst)
    a, b = ref["undistorted"].astype(np.float64).reshape(-1, 2), mine["undistorted"].astype(np.float64).reshape(-1, 2)
    ok = a.shape == b.shape and float(np.abs(a - b).max(initial=0.0)) <= 1e-4
    add("undistort", ok, "" if ok else "largest difference %.3g px" % (float(np.abs(a - b).max(initial=0.0)) if a.shape == b.shape else float("nan")))
    # one verdict per stage (a stage listed twice -- left and right image -- passes when both do)
    merged = []
    for s in STAGE_NAMES:
        rows = [r for r in res if r[0] == s]
        merged.append((s, all(r[1] for r in rows), "; ".join(r[2] for r in rows if r[2])))
    return merged


def first_divergence(result):
    """(stage, detail, hint) of the first stage that differs, or None."""
    for stage, ok, detail in result:
        if not ok:
            return stage, detail, HINT[stage]
    return None


def compare_dir(ref_dir, cases):
    """{case: compare_case result}."""
    return {name: compare_case(load_case(ref_dir, name), oracle_outputs(nf, left, right)) for name, nf, left, right in cases}


def main():
    import pin_inputs
    ref_dir = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "tests" / "golden" / "opencv")
    print((Path(ref_dir) / "VERSION.txt").read_text().splitlines()[0] if (Path(ref_dir) / "VERSION.txt").exists() else "(no VERSION.txt)")
    rc = 0
    for name, result in compare_dir(ref_dir, pin_inputs.cases()).items():
        d = first_divergence(result)
        print("%-44s %s" % (name, "identical at every stage" if d is None else "FIRST DIVERGENCE: %s -- %s" % (d[0], d[1])))
        if d is not None:
            print("    -> %s" % d[2])
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
