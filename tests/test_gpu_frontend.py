"""slam::Frontend (C++ host class over the HIP C ABI) against a model of the reference's ObserveImage
(slam_frontend.cc:400-472) assembled from the CPU oracle's pieces: extraction, stereo GetMatches,
RemoveAmbigStereo re-indexing and threshold chain, temporal GetFeatureMatches (std::sort + best 30 %),
Calculate3DPoints / UndistortFeaturePoints (cv::triangulatePoints, cv::undistortPoints restated)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NF = 1000
F_RECT = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # l^T F r = y_r - y_l: rectified synthetic pair


POINT_RTOL, PIXEL_ATOL = 1e-5, 1e-4  # floating point: relative per coordinate of point3d; pixels


def _model(oracle, frames, frame_life, best_percent=np.float32(0.3), calib=None):
    thr = np.float32(10000.0)
    frame_list, factors, kept_frames = [], [], []
    for fid, (left, right) in enumerate(frames):
        ol, orr = oracle.Orb(nfeatures=NF), oracle.Orb(nfeatures=NF)
        ol.run(left)
        orr.run(right)
        kl, dl = ol.result()
        kr, dr = orr.result()
        m = oracle.get_matches(dl, dr)
        keep, _, thr_new, _ = oracle.remove_ambig_stereo(kl, kr, m, F_RECT, float(thr))
        thr = np.float32(thr_new)
        kl2, dl2 = kl[m["queryIdx"][keep]], dl[m["queryIdx"][keep]]
        kr2, dr2 = kr[m["trainIdx"][keep]], dr[m["trainIdx"][keep]]
        for pid, _, pd in frame_list:
            mm = oracle.sort_and_trim(oracle.get_matches(pd, dl2), float(best_percent))
            factors.append((pid, fid, np.stack([mm["queryIdx"], mm["trainIdx"]], 1).astype(np.uint64)))
        if len(frame_list) >= frame_life:
            frame_list.pop(0)
        frame_list.append((fid, kl2, dl2))
        vf = None
        if calib is not None:  # cc:437-443
            vf, _ = oracle.vision_features(kl2, dl2, kr2, dr2, calib.get("projection_left"), calib.get("projection_right"),
                                           calib.get("camera_matrix_left"), calib.get("distortion_left"))
        kept_frames.append((kl2, dl2, float(thr), vf))
    return factors, frame_list, kept_frames


def test_frontend_matches_reference_model(oracle):
    from vision_slam_frontend_amd import frontend, synth
    sc = synth.Scene(640, 480)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(4)]
    fe = frontend.Frontend(640, 480, nfeatures=NF, fundamental=F_RECT, frame_life=3)
    q = np.array([1, 0, 0, 0], np.float32)
    # no odometry yet -> no node
    assert fe.observe_image(*frames[0]) is False
    fe.observe_odometry([0, 0, 0], q, 1.0)
    assert fe.observe_image(*frames[0]) is False  # has not moved since the first odometry sample (quirk Q11 fix)
    thr_seen = []
    for f in range(4):
        fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 10.0 + f)
        assert fe.observe_image(*frames[f], time=99.0) is True
        assert fe.observe_image(*frames[f]) is False  # same pose again: gated by OdomCheck (cc:175-186)
        thr_seen.append(fe.stereo_ambig_constraint)
    assert fe.num_poses == 4
    factors, frame_list, kept = _model(oracle, frames, frame_life=3, calib=frontend.default_calibration())
    # thresholds: mean residual + 2 chain (cc:392-394)
    assert thr_seen == [k[2] for k in kept]
    # vision factors: ids and pairs, in order (cc:424-434)
    got = fe.vision_factors()
    assert len(got) == len(factors) == 0 + 1 + 2 + 3
    for (ga, gb, gp), (ea, eb, ep) in zip(got, factors):
        assert (ga, gb) == (ea, eb)
        np.testing.assert_array_equal(gp, ep)
        assert len(ep) > 3
    # retained frames (after RemoveAmbigStereo re-indexing), sliding window of frame_life
    for i, (fid, kl2, dl2) in enumerate(frame_list):
        gid, gk, gd = fe.frame(i)
        assert gid == fid and gk.tobytes() == kl2.tobytes()
        np.testing.assert_array_equal(gd, dl2)
    # nodes: one feature per surviving left keypoint, pixel = undistorted keypoint position, timestamp = odometry's
    nodes = fe.nodes()
    for f, node in enumerate(nodes):
        kl2 = kept[f][0]
        assert node["node_idx"] == f and node["timestamp"] == 10.0 + f
        feat = node["features"]
        assert len(feat) == len(kl2)
        np.testing.assert_array_equal(feat[:, 0], np.arange(len(kl2), dtype=np.float32))
        want = kept[f][3]
        assert np.abs(feat[:, 1:3].astype(np.float64) - want["pixel"]).max() <= PIXEL_ATOL  # UndistortFeaturePoints
        g, w = feat[:, 3:6].astype(np.float64), want["point3d"].astype(np.float64)     # Calculate3DPoints, (x,y,z)/w
        assert np.array_equal(np.isfinite(g), np.isfinite(w)) and np.isfinite(w).mean() > 0.99
        fin = np.isfinite(w)
        rel = np.abs(g[fin] - w[fin]) / np.maximum(np.abs(w[fin]), 1e-30)
        assert rel.max() <= POINT_RTOL, "frame %d: worst relative point3d error %.3g" % (f, rel.max())
        assert np.abs(w[fin]).max() > 0
        np.testing.assert_allclose(node["pose"], [0.3 * (f + 1), 0, 0, 1, 0, 0, 0], atol=1e-6)
    # odometry factors between consecutive nodes (cc:311-321)
    of = fe.odometry_factors()
    assert [(a, b) for a, b, _ in of] == [(0, 1), (1, 2), (2, 3)]
    for _, _, tq in of:
        np.testing.assert_allclose(tq, [0.3, 0, 0, 1, 0, 0, 0], atol=1e-6)
    # ROS-1 wire bytes of the SLAMProblem (row f3): decode them again and compare with the getters
    import struct
    wire = fe.serialize_problem()
    off = 0

    def rd(fmt):
        nonlocal off
        v = struct.unpack_from("<" + fmt, wire, off)
        off += struct.calcsize("<" + fmt)
        return v

    (n_nodes,) = rd("I")
    assert n_nodes == len(nodes)
    for node in nodes:
        idx, ts = rd("Qd")
        assert idx == node["node_idx"] and ts == node["timestamp"]
        loc_q = rd("7d")  # loc xyz, quaternion xyzw
        np.testing.assert_array_equal(np.float32(loc_q[:3]), node["pose"][:3])
        np.testing.assert_array_equal(np.float32([loc_q[6], loc_q[3], loc_q[4], loc_q[5]]), node["pose"][3:])
        (nf,) = rd("I")
        assert nf == len(node["features"])
        for f in node["features"]:
            fid, px, py, pz, x, y, z = rd("Q6d")
            assert fid == int(f[0]) and pz == 0.0
            np.testing.assert_array_equal(np.float32([px, py, x, y, z]), f[1:6])
    (n_vf,) = rd("I")
    assert n_vf == len(got)
    for a, b, pairs in got:
        pa, pb, npairs = rd("QQI")
        assert (pa, pb, npairs) == (a, b, len(pairs))
        flat = rd("%dQ" % (2 * npairs))
        np.testing.assert_array_equal(np.array(flat, np.uint64).reshape(-1, 2), pairs)
    (n_of,) = rd("I")
    assert n_of == len(of)
    for a, b, tq in of:
        vals = rd("QQ7d")
        assert vals[:2] == (a, b)
        np.testing.assert_array_equal(np.float32(vals[2:5]), tq[:3])
        np.testing.assert_array_equal(np.float32([vals[8], vals[5], vals[6], vals[7]]), tq[3:])
    assert off == len(wire)
    fe.close()


def test_frontend_rotation_gate_and_default_fundamental():
    from vision_slam_frontend_amd import frontend, synth
    left, right = synth.stereo_pair(320, 240, 0, n_objects=300)
    fe = frontend.Frontend(320, 240, nfeatures=300)
    F = fe.fundamental
    assert np.isfinite(F).all() and np.abs(F).max() > 0
    fe.observe_odometry([0, 0, 0], [1, 0, 0, 0], 0.0)
    half = np.deg2rad(11.0) / 2
    fe.observe_odometry([0, 0, 0], [np.cos(half), 0, 0, np.sin(half)], 1.0)  # 11 degrees > 10 degrees
    assert fe.observe_image(left, right) is True
    half = np.deg2rad(15.0) / 2
    fe.observe_odometry([0, 0, 0], [np.cos(half), 0, 0, np.sin(half)], 2.0)  # only 4 degrees more
    assert fe.observe_image(left, right) is False
    assert fe.num_poses == 1
    fe.close()


def test_fused_observe_equals_call_by_call():
    """ObserveImage as ONE submission (vsf_observe_stereo: descriptors of the kept frames resident in HBM, device
    RemoveAmbigStereo / GetFeatureMatches / Calculate3DPoints) against the same class making one C-ABI call per
    reference call with the host steps in between: identical nodes, factors, frames, thresholds and wire bytes --
    including a frame without stereo matches, the NaN threshold after it (quirk Q3) and the window sliding."""
    from vision_slam_frontend_amd import frontend, synth
    sc = synth.Scene(320, 240, n_objects=400)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(7)]
    frames[2] = (frames[2][0], np.full_like(frames[2][1], 128))  # no stereo match in frame 2
    q = np.array([1, 0, 0, 0], np.float32)
    runs = []
    for fused, pipelined in ((True, False), (False, False), (True, True)):
        fe = frontend.Frontend(320, 240, nfeatures=600, fundamental=F_RECT, frame_life=3)
        fe.set_fused(fused)
        fe.set_pipelined(pipelined)
        fe.observe_odometry([0, 0, 0], q, 0.0)
        thr = []
        for f, (l, r) in enumerate(frames):
            fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.0 + f)
            assert fe.observe_image(l, r) is True
            if not pipelined:  # (reading the threshold would drain the pipeline: two frames stay in flight instead)
                thr.append(fe.stereo_ambig_constraint)
            elif f % 3 == 2:   # the odometry gate holds a frame back while others are still on the GPU
                fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.5 + f)
                assert fe.observe_image(l, r) is False
        runs.append(dict(thr=np.float32(thr), factors=fe.vision_factors(), nodes=fe.nodes(),
                         frames=[fe.frame(i) for i in range(3)], wire=fe.serialize_problem(),
                         odo=fe.odometry_factors()))
        fe.close()
    a, b, c = runs
    # pipelined mode (two frames in flight, results booked late with the odometry of their own call): the same problem
    assert c["wire"] == a["wire"] and len(c["nodes"]) == len(a["nodes"]) == 7
    for na, nc in zip(a["nodes"], c["nodes"]):
        assert na["node_idx"] == nc["node_idx"] and na["timestamp"] == nc["timestamp"]
        np.testing.assert_array_equal(na["pose"], nc["pose"])
        assert na["features"].tobytes() == nc["features"].tobytes()
    for (a0, a1, ap), (c0, c1, cp) in zip(a["factors"], c["factors"]):
        assert (a0, a1) == (c0, c1)
        np.testing.assert_array_equal(ap, cp)
    assert len(a["odo"]) == len(c["odo"]) == 6
    for oa, oc in zip(a["odo"], c["odo"]):
        assert oa[:2] == oc[:2]
        np.testing.assert_array_equal(oa[2], oc[2])
    assert a["thr"].tobytes() == b["thr"].tobytes() and np.isnan(a["thr"][2]) and np.isfinite(a["thr"][3])
    assert len(a["factors"]) == len(b["factors"]) == 0 + 1 + 2 + 3 + 3 + 3 + 3
    for (a0, a1, ap), (b0, b1, bp) in zip(a["factors"], b["factors"]):
        assert (a0, a1) == (b0, b1)
        np.testing.assert_array_equal(ap, bp)
    for na, nb in zip(a["nodes"], b["nodes"]):
        assert na["node_idx"] == nb["node_idx"] and na["timestamp"] == nb["timestamp"]
        np.testing.assert_array_equal(na["pose"], nb["pose"])
        assert na["features"].tobytes() == nb["features"].tobytes()  # same arithmetic on host and device
    assert [len(n["features"]) for n in a["nodes"]][2:4] == [0, 0] and len(a["nodes"][4]["features"]) > 20
    for (ia, ka, da), (ib, kb, db) in zip(a["frames"], b["frames"]):
        assert ia == ib and ka.tobytes() == kb.tobytes()
        np.testing.assert_array_equal(da, db)
    assert a["wire"] == b["wire"]


@pytest.mark.parametrize("depth,batch,min_batch", [(1, 1, 0), (4, 4, 0), (32, 32, 0), (32, 8, 4)])
def test_queued_observe_image_books_the_synchronous_problem(depth, batch, min_batch):
    """slam::Frontend with its ObserveImage queue at depths 1, 4 and 32 (frames leave for the GPU in batches, results are
    booked late with the odometry of their own call): the SLAMProblem -- nodes, features, vision and odometry factors, as
    ROS wire bytes -- and the kept frames equal the synchronous mode's, over 40 frames with three frames without stereo
    matches (NaN thresholds behind them), the odometry gate refusing every fifth call, and a read of the problem (which
    drains the queue) in the middle."""
    from vision_slam_frontend_amd import frontend, synth
    sc = synth.Scene(320, 240, n_objects=400)
    frames = [(sc.render(f % 9, 0), sc.render(f % 9, 1)) for f in range(40)]
    for k in (2, 21, 22):
        frames[k] = (frames[k][0], np.full_like(frames[k][1], 128))
    q = np.array([1, 0, 0, 0], np.float32)
    runs = []
    for pipelined in (False, True):
        fe = frontend.Frontend(320, 240, nfeatures=600, fundamental=F_RECT, frame_life=3)
        fe.set_pipelined(pipelined)
        if pipelined:
            fe.set_queue(depth, batch, min_batch)
        fe.observe_odometry([0, 0, 0], q, 0.0)
        mid = None
        for f, (l, r) in enumerate(frames):
            fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.0 + f)
            assert fe.observe_image(l, r) is True
            if f % 5 == 4:  # the odometry gate holds a frame back while others are still in the queue
                fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.5 + f)
                assert fe.observe_image(l, r) is False
            if f == 25:
                mid = (fe.num_poses, fe.serialize_problem())
        runs.append(dict(mid=mid, wire=fe.serialize_problem(), nodes=fe.nodes(), factors=fe.vision_factors(),
                         frames=[fe.frame(i) for i in range(3)], thr=np.float32(fe.stereo_ambig_constraint)))
        fe.close()
    a, c = runs
    assert a["mid"][0] == c["mid"][0] == 26 and a["mid"][1] == c["mid"][1]
    assert a["wire"] == c["wire"] and len(a["nodes"]) == len(c["nodes"]) == 40
    assert a["thr"].tobytes() == c["thr"].tobytes()
    for na, nc in zip(a["nodes"], c["nodes"]):
        assert na["node_idx"] == nc["node_idx"] and na["timestamp"] == nc["timestamp"]
        np.testing.assert_array_equal(na["pose"], nc["pose"])
        assert na["features"].tobytes() == nc["features"].tobytes()
    assert len(a["factors"]) == len(c["factors"]) == 0 + 1 + 2 + 3 * 37
    for (a0, a1, ap), (c0, c1, cp) in zip(a["factors"], c["factors"]):
        assert (a0, a1) == (c0, c1)
        np.testing.assert_array_equal(ap, cp)
    for (ia, ka, da), (ic, kc, dc) in zip(a["frames"], c["frames"]):
        assert ia == ic and ka.tobytes() == kc.tobytes()
        np.testing.assert_array_equal(da, dc)
    assert [len(n["features"]) for n in a["nodes"]][21:24] == [0, 0, 0] and len(a["nodes"][24]["features"]) > 20


@pytest.mark.parametrize("nfeatures", [2000, 10000])
def test_the_benched_queue_books_the_synchronous_problem(nfeatures):
    """What bench.py's `observe_image` leg times, held to the synchronous mode byte for byte: 640x480, the reference's window
    of 10, 2000 features and the reference's own nfeatures = 10000, the class's DEFAULT queue (depth 256, up to 128 frames per
    batch) fed by the C++ driver loop (vsfh_time_sequence: the reference's ObserveOdometry + ObserveImage per frame, no Python
    between the calls, so that batches really fill) -- 420 frames, the whole SLAMProblem as ROS wire bytes."""
    from vision_slam_frontend_amd import frontend, synth
    sc = synth.Scene(640, 480)
    frames = np.stack([np.stack([sc.render(f, 0), sc.render(f, 1)]) for f in range(14)])
    wires = []
    for pipelined in (False, True):
        fe = frontend.Frontend(640, 480, nfeatures=nfeatures, fundamental=F_RECT)
        fe.set_pipelined(pipelined)
        fps, _, _ = fe.time_sequence(frames, 420, warm=0)
        assert fps > 0 and fe.num_poses == 420
        wires.append(fe.serialize_problem())
        fe.close()
    assert len(wires[0]) > 1000000 and wires[0] == wires[1]


def test_a_queued_frontend_is_destroyed_with_frames_in_its_queue():
    """~Frontend with frames staged, on the GPU and uncollected (launcher thread and copy helper alive): the context stops its
    threads, waits for its streams and frees what their kernels write; a Frontend created afterwards works."""
    from vision_slam_frontend_amd import frontend, synth
    sc = synth.Scene(320, 240, n_objects=400)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(12)]
    q = np.array([1, 0, 0, 0], np.float32)
    for depth, batch in ((16, 8), (64, 16), (5, 4)):
        fe = frontend.Frontend(320, 240, nfeatures=600, fundamental=F_RECT, frame_life=3)
        fe.set_pipelined(True)
        fe.set_queue(depth, batch, 0)
        fe.set_queue_threads(launcher=(depth == 64), copy=True)
        fe.observe_odometry([0, 0, 0], q, 0.0)
        for f, (l, r) in enumerate(frames):
            fe.observe_odometry([0.3 * (f + 1), 0, 0], q, 1.0 + f)
            assert fe.observe_image(l, r) is True
        fe.close()  # nothing was read: frames are still in the queue
    fe = frontend.Frontend(320, 240, nfeatures=600, fundamental=F_RECT, frame_life=3)
    fe.observe_odometry([0, 0, 0], q, 0.0)
    fe.observe_odometry([0.3, 0, 0], q, 1.0)
    assert fe.observe_image(*frames[0]) is True and fe.num_poses == 1
    fe.close()
