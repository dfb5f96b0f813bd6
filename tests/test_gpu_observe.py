"""vsf_observe_stereo -- one GPU submission per Frontend::ObserveImage (slam_frontend.cc:400-472) -- against the
reference's sequence assembled from the CPU oracle's pieces, frame by frame over a window that fills, slides and
meets a frame without stereo matches (quirk Q3).  Integer results (keypoints, descriptors, factor pairs, which match
feeds which keypoint) bit for bit; point3d / pixel to the floating-point tolerance of row f2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NF, LIFE = 700, 3
F_RECT = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
POINT_RTOL, PIXEL_ATOL = 1e-5, 1e-4


def _follow_reference_sequence(oracle, frames, width, height, NF, LIFE):
    """Runs `frames` [(left, right)] through vsf_observe_stereo and through the reference's sequence on the oracle, frame by
    frame; returns the context (still open), the calibration and the per-frame feature counts."""
    from vision_slam_frontend_amd import capi, frontend
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    bp = float(np.float32(0.3))
    ctx = capi.Context(capi.default_params(width, height, max_images=2, nfeatures=NF))
    thr = np.float32(10000.0)
    window = []  # filtered left descriptors of the kept frames, oldest first
    sizes = []
    for fid, (left, right) in enumerate(frames):
        got = ctx.observe_stereo(left, right, calib, best_percent=bp, frame_life=LIFE)
        # --- the reference's sequence on the oracle ---
        ol, orr = oracle.Orb(nfeatures=NF), oracle.Orb(nfeatures=NF)
        ol.run(left)
        orr.run(right)
        kl, dl = ol.result()
        kr, dr = orr.result()
        m = oracle.get_matches(dl, dr)
        keep, res, thr_next, kept = oracle.remove_ambig_stereo(kl, kr, m, F_RECT, float(thr))
        kl2, dl2 = kl[m["queryIdx"][keep]], dl[m["queryIdx"][keep]]
        kr2, dr2 = kr[m["trainIdx"][keep]], dr[m["trainIdx"][keep]]
        assert (got["n_left"], got["n_right"], got["n_stereo_matches"]) == (len(kl), len(kr), len(m))
        assert got["threshold"].tobytes() == thr.tobytes() or (np.isnan(got["threshold"]) and np.isnan(thr))
        assert got["threshold_next"].tobytes() == np.float32(thr_next).tobytes() or \
            (np.isnan(got["threshold_next"]) and np.isnan(thr_next))
        assert got["keypoints"].tobytes() == kl2.tobytes()
        np.testing.assert_array_equal(got["descriptors"], dl2)
        assert len(got["factors"]) == len(window)
        for past, fac in zip(window, got["factors"]):  # cc:424-434, oldest kept frame first
            mm = oracle.sort_and_trim(oracle.get_matches(past, dl2), bp)
            np.testing.assert_array_equal(fac["feature_idx_initial"], mm["queryIdx"])
            np.testing.assert_array_equal(fac["feature_idx_current"], mm["trainIdx"])
        rl = oracle.sort_and_trim(oracle.get_matches(dr2, dl2), 1.0)  # cc:129-132
        np.testing.assert_array_equal(got["stereo_pairs"]["feature_idx_initial"], rl["queryIdx"])
        np.testing.assert_array_equal(got["stereo_pairs"]["feature_idx_current"], rl["trainIdx"])
        want, npts = oracle.vision_features(kl2, dl2, kr2, dr2, calib.get("projection_left"), calib.get("projection_right"),
                                            calib.get("camera_matrix_left"), calib.get("distortion_left"))
        f = got["features"]
        assert got["n_points"] == npts and len(f) == len(want)
        np.testing.assert_array_equal(f["feature_idx"], want["feature_idx"])
        assert np.abs(f["pixel"].astype(np.float64) - want["pixel"]).max(initial=0.0) <= PIXEL_ATOL
        g, w = f["point3d"].astype(np.float64), want["point3d"].astype(np.float64)
        fin = np.isfinite(w)
        assert np.array_equal(np.isfinite(g), fin)
        assert (np.abs(g[fin] - w[fin]) / np.maximum(np.abs(w[fin]), 1e-30)).max(initial=0.0) <= POINT_RTOL
        thr = np.float32(thr_next)
        if len(window) >= LIFE:
            window.pop(0)
        window.append(dl2)
        sizes.append(len(kl2))
    return ctx, calib, sizes


def test_observe_stereo_follows_the_reference_sequence(oracle):
    from vision_slam_frontend_amd import synth
    sc = synth.Scene(320, 240, n_objects=400)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(7)]
    frames[3] = (frames[3][0], np.full_like(frames[3][1], 128))  # no stereo match in frame 3
    ctx, calib, sizes = _follow_reference_sequence(oracle, frames, 320, 240, NF, LIFE)
    assert sizes[3] == 0 and sizes[4] == 0 and sizes[5] > 20  # no match; NaN threshold; filtering resumes
    # a reset forgets the window and the threshold
    ctx.observe_reset()
    again = ctx.observe_stereo(*frames[0], calib, best_percent=float(np.float32(0.3)), frame_life=LIFE)
    assert again["threshold"] == np.float32(10000.0) and len(again["factors"]) == 0
    ctx.close()


def test_observe_stereo_on_photographs(oracle):
    """The same frame-by-frame comparison on photographs at 640x480 / 2000 features (tests/golden/real/): a real rectified
    stereo pair (Middlebury motorcycle) that drifts 3 pixels per frame, then a cameraman seen 7 pixels right and 2 pixels
    down by the right camera (a constant epipolar residual of 2: the adaptive threshold of RemoveAmbigStereo settles on
    mean + 2 = 4), a gravel texture with a pure horizontal disparity, and the stereo pair again -- a window of 3 that fills,
    slides and matches across a scene change."""
    from pathlib import Path

    from PIL import Image
    real = Path(__file__).resolve().parent / "golden" / "real"
    img = {n: np.asarray(Image.open(real / (n + ".png"))) for n in ("motorcycle_left", "motorcycle_right", "camera", "camera_shift", "gravel")}
    roll = lambda a, dx: np.ascontiguousarray(np.roll(a, dx, 1))  # noqa: E731
    frames = [(roll(img["motorcycle_left"], 3 * k), roll(img["motorcycle_right"], 3 * k)) for k in range(3)]
    frames += [(roll(img["camera"], 2 * k), roll(img["camera_shift"], 2 * k)) for k in range(2)]
    frames += [(img["gravel"], roll(img["gravel"], 5)), (img["motorcycle_left"], img["motorcycle_right"])]
    ctx, _, sizes = _follow_reference_sequence(oracle, frames, 640, 480, 2000, 3)
    ctx.close()
    assert min(sizes[:3]) > 30 and sizes[3] > 300 and sizes[5] > 100, sizes


def _same_observation(a: dict, b: dict):
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, np.ndarray):
            assert va.tobytes() == vb.tobytes(), k
        elif isinstance(va, list):
            assert len(va) == len(vb) and all(x.tobytes() == y.tobytes() for x, y in zip(va, vb)), k
        elif isinstance(va, (float, np.floating)):
            assert np.float32(va).tobytes() == np.float32(vb).tobytes(), k
        else:
            assert va == vb, k


@pytest.mark.parametrize("slots", [2, 3, 4, 6])
def test_frames_in_flight_equal_the_synchronous_calls(slots):
    """vsf_observe_submit / vsf_observe_collect with two to six frames in the queue (contexts with max_images = 2 x slots:
    the queue's default depth; waiting frames leave as batches) return, frame for frame and byte for byte, what the
    synchronous vsf_observe_stereo returns -- across the window filling and sliding, the frame without stereo matches and
    the NaN threshold after it, whose state travels from tail to tail on the device."""
    from vision_slam_frontend_amd import capi, frontend, synth
    sc = synth.Scene(320, 240, n_objects=400)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(9)]
    frames[3] = (frames[3][0], np.full_like(frames[3][1], 128))
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    bp = float(np.float32(0.3))
    with capi.Context(capi.default_params(320, 240, max_images=2, nfeatures=NF)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, calib, best_percent=bp, frame_life=LIFE) for l, r in frames]
    with capi.Context(capi.default_params(320, 240, max_images=2 * slots, nfeatures=NF)) as ctx:
        got, tickets = [], []
        for l, r in frames:
            if len(tickets) == slots:
                got.append(ctx.observe_collect(tickets.pop(0), frame_life=LIFE))
            tickets.append(ctx.observe_submit(l, r, calib, best_percent=bp, frame_life=LIFE))
        # another frame cannot enter while every slot holds an uncollected frame; a newer frame cannot leave first
        with pytest.raises(capi.VsfError):
            ctx.observe_submit(*frames[0], calib, best_percent=bp, frame_life=LIFE)
        with pytest.raises(capi.VsfError):
            ctx.observe_collect(tickets[1], frame_life=LIFE)
        while tickets:
            got.append(ctx.observe_collect(tickets.pop(0), frame_life=LIFE))
        with pytest.raises(capi.VsfError):
            ctx.observe_collect(0, frame_life=LIFE)  # collected long ago
        # the synchronous call still works on the same context and continues the same sequence
        extra = ctx.observe_stereo(*frames[0], calib, best_percent=bp, frame_life=LIFE)
        assert len(extra["factors"]) == LIFE
    assert len(got) == len(want) == len(frames)
    for g, w in zip(got, want):
        _same_observation(w, g)
    assert sum(len(f["features"]) for f in want) > 100


def _sequence(n, w=320, h=240):
    from vision_slam_frontend_amd import synth
    sc = synth.Scene(w, h, n_objects=400)
    frames = [(sc.render(f % 11, 0), sc.render(f % 11, 1)) for f in range(n)]
    for k in (3, 17, 18):  # no stereo match: a NaN threshold for the frame behind it (quirk Q3), twice in a row at 17, 18
        if k < n:
            frames[k] = (frames[k][0], np.full_like(frames[k][1], 128))
    return frames


@pytest.mark.parametrize("thread", [0, 1])
@pytest.mark.parametrize("depth,batch,min_batch", [(1, 1, 0), (4, 4, 0), (32, 32, 0), (32, 8, 0), (32, 32, 12), (7, 3, 3)])
def test_queue_depths_equal_the_synchronous_calls(depth, batch, min_batch, thread):
    """The queue at depths 1, 4 and 32 (batches of up to `batch` frames; with min_batch, frames wait for company while the
    GPU is busy): 45 frames submitted as fast as the queue takes them -- so that batches of every size form, cut across the
    window filling, the frames without stereo matches and the NaN thresholds behind them -- return byte for byte what the
    synchronous calls return, on their own tickets."""
    from vision_slam_frontend_amd import capi, frontend
    frames = _sequence(45)
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    bp = float(np.float32(0.3))
    with capi.Context(capi.default_params(320, 240, max_images=2, nfeatures=NF)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, calib, best_percent=bp, frame_life=LIFE) for l, r in frames]
    with capi.Context(capi.default_params(320, 240, max_images=2 * batch, nfeatures=NF)) as ctx:
        ctx.set_option(capi.OPT_OBSERVE_THREAD, thread)  # (the launcher thread of queues of depth >= 4: off by default)
        ctx.observe_configure(depth, min_batch, 0)
        got, tickets = [], []
        for l, r in frames:
            if len(tickets) == depth:
                got.append(ctx.observe_collect(tickets.pop(0), frame_life=LIFE))
            tickets.append(ctx.observe_submit(l, r, calib, best_percent=bp, frame_life=LIFE))
        with pytest.raises(capi.VsfError):
            ctx.observe_submit(*frames[0], calib, best_percent=bp, frame_life=LIFE)  # the queue is full
        while tickets:
            got.append(ctx.observe_collect(tickets.pop(0), frame_life=LIFE))
    assert len(got) == len(want) == len(frames)
    for k, (g, w) in enumerate(zip(got, want)):
        _same_observation(w, g)
    assert [len(w["features"]) for w in want][17:20] == [0, 0, 0] and len(want[20]["features"]) > 20
    assert np.isnan(want[18]["threshold"]) and np.isnan(want[19]["threshold"]) and np.isfinite(want[20]["threshold"])


def test_queue_at_a_width_that_is_not_a_multiple_of_64(oracle):
    """328 x 200: the staging pitch (384) differs from the caller's stride, so the rows are staged one by one; the reference's
    sequence on the oracle frame by frame (synchronous calls), then the same frames through a queue of depth 6 in batches."""
    from vision_slam_frontend_amd import capi, frontend, synth
    sc = synth.Scene(328, 200, n_objects=300)
    frames = [(sc.render(f, 0), sc.render(f, 1)) for f in range(7)]
    ctx, calib, sizes = _follow_reference_sequence(oracle, frames, 328, 200, 500, 2)
    ctx.close()
    assert min(sizes) > 10
    bp = float(np.float32(0.3))
    with capi.Context(capi.default_params(328, 200, max_images=2, nfeatures=500)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, calib, best_percent=bp, frame_life=2) for l, r in frames]
    padded = [tuple(np.ascontiguousarray(np.pad(im, ((0, 0), (0, 24))))[:, :328] for im in fr) for fr in frames]  # stride 352
    with capi.Context(capi.default_params(328, 200, max_images=8, nfeatures=500)) as q:
        q.observe_configure(6, 3, 0)
        tickets = [q.observe_submit(l, r, calib, best_percent=bp, frame_life=2) for l, r in padded[:6]]
        got = [q.observe_collect(t, frame_life=2) for t in tickets]
        got.append(q.observe_stereo(*padded[6], calib, best_percent=bp, frame_life=2))
    for g, w in zip(got, want):
        _same_observation(w, g)


def test_a_batch_is_cut_where_the_parameters_change():
    """Frames that wait with another best_percent or another calibration than the frame in front of them never share its
    batch; results equal the synchronous calls with the same per-frame parameters."""
    from vision_slam_frontend_amd import capi, frontend
    frames = _sequence(12)
    F2 = F_RECT.copy()
    F2[2, 2] = 1.5  # (a constant added to every residual: another threshold chain)
    calibs = [frontend.default_calibration().set("fundamental", F_RECT if (k // 3) % 2 == 0 else F2) for k in range(12)]
    bps = [float(np.float32(0.3 if (k // 2) % 2 == 0 else 0.6)) for k in range(12)]
    with capi.Context(capi.default_params(320, 240, max_images=2, nfeatures=NF)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, c, best_percent=b, frame_life=LIFE) for (l, r), c, b in zip(frames, calibs, bps)]
    with capi.Context(capi.default_params(320, 240, max_images=24, nfeatures=NF)) as ctx:
        ctx.observe_configure(12, 12, 0)  # while the GPU is busy, nothing leaves before the parameters change or the collect
        tickets = [ctx.observe_submit(l, r, c, best_percent=b, frame_life=LIFE) for (l, r), c, b in zip(frames, calibs, bps)]
        got = [ctx.observe_collect(t, frame_life=LIFE) for t in tickets]
    for g, w in zip(got, want):
        _same_observation(w, g)
    assert any(len(w["factors"]) == LIFE and len(w["factors"][0]) > 0 for w in want)


def test_capacity_overflow_in_a_batch_stays_on_its_own_ticket():
    """Batches of up to eight frames: every image has a status word of its own, so a frame whose keypoints overflow the
    output capacity gets VSF_ERR_CAPACITY on ITS ticket and its neighbours in the same batch VSF_OK."""
    from vision_slam_frontend_amd import capi, frontend, synth
    w, h = 320, 240
    busy = synth.stereo_pair(w, h, 0, n_objects=400)
    flat = (np.full((h, w), 90, np.uint8), np.full((h, w), 90, np.uint8))
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    p = capi.default_params(w, h, max_images=16, nfeatures=500, max_keypoints=200)
    pattern = [1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1]
    with capi.Context(p) as ctx:
        ctx.observe_configure(20, 6, 0)
        tickets, got = [], []

        def collect(t):
            try:
                ctx.observe_collect(t, frame_life=2)
                got.append(capi.VSF_OK)
            except capi.VsfError as e:
                got.append(e.status)

        for b in pattern:
            if len(tickets) == 20:
                collect(tickets.pop(0))
            tickets.append(ctx.observe_submit(*(busy if b else flat), calib, frame_life=2))
        while tickets:
            collect(tickets.pop(0))
    want = [capi.VSF_ERR_CAPACITY if b else capi.VSF_OK for b in pattern]
    assert got == want, list(zip(pattern, got))


def test_capacity_overflow_is_reported_on_its_own_ticket():
    """Three frames in flight, each on its slot's stream (vsf_observe_submit): a frame whose keypoints overflow the output
    capacity must get VSF_ERR_CAPACITY on ITS ticket, and a frame that does not must get VSF_OK -- whatever its neighbours
    in flight did.  (Round-3 review: all slots shared one device status word; a frame's tail copied and cleared it while
    the next frame's extraction, on another stream, was still setting it -- an overflow could land on the wrong ticket or be
    wiped.)  Busy and flat frames alternate in every order of arrival over 18 frames."""
    from vision_slam_frontend_amd import capi, frontend, synth
    w, h = 320, 240
    busy = synth.stereo_pair(w, h, 0, n_objects=400)          # > 400 keypoints per image at nfeatures 500
    flat = (np.full((h, w), 90, np.uint8), np.full((h, w), 90, np.uint8))
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    p = capi.default_params(w, h, max_images=6, nfeatures=500, max_keypoints=200)   # 6 images: three slots
    pattern = [1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 1]              # 1 = busy frame
    with capi.Context(p) as ctx:
        tickets, got = [], []

        def collect(t):
            try:
                ctx.observe_collect(t, frame_life=2)
                got.append(capi.VSF_OK)
            except capi.VsfError as e:
                got.append(e.status)

        for k, b in enumerate(pattern):
            if len(tickets) == 3:
                collect(tickets.pop(0))
            tickets.append(ctx.observe_submit(*(busy if b else flat), calib, frame_life=2))
        while tickets:
            collect(tickets.pop(0))
    want = [capi.VSF_ERR_CAPACITY if b else capi.VSF_OK for b in pattern]
    assert got == want, list(zip(pattern, got))


def test_other_entry_points_beside_a_queue_with_its_launcher_thread():
    """A queue of depth >= 4 has a launcher thread.  Every other entry point of the context first sends what waits in the
    queue (the thread is idle afterwards), so a host-pointer extraction, an option and a profile read in the middle of a
    stream of submits neither race with it nor change a result; and a context is destroyed with frames waiting, in flight
    and uncollected."""
    from vision_slam_frontend_amd import capi, frontend
    frames = _sequence(30)
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    bp = float(np.float32(0.3))
    with capi.Context(capi.default_params(320, 240, max_images=2, nfeatures=NF)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, calib, best_percent=bp, frame_life=LIFE) for l, r in frames]
        kp_want, desc_want = sync_ctx.extract(frames[5][0])
    with capi.Context(capi.default_params(320, 240, max_images=16, nfeatures=NF)) as ctx:
        ctx.set_option(capi.OPT_OBSERVE_THREAD, 1)
        ctx.observe_configure(30, 8, 0)
        tickets = []
        for k, (l, r) in enumerate(frames):
            tickets.append(ctx.observe_submit(l, r, calib, best_percent=bp, frame_life=LIFE))
            if k == 11:
                kp, desc = ctx.extract(frames[5][0])  # (the same context's extraction buffers and stream)
                assert kp.tobytes() == kp_want.tobytes() and np.array_equal(desc, desc_want)
            if k == 17:
                ctx.set_option(capi.OPT_SELECT_WIDE, 1)
                assert ctx.sync() == capi.VSF_OK
        got = [ctx.observe_collect(t, frame_life=LIFE) for t in tickets]
    for g, w in zip(got, want):
        _same_observation(w, g)
    for _ in range(3):
        ctx = capi.Context(capi.default_params(320, 240, max_images=8, nfeatures=NF))
        ctx.set_option(capi.OPT_OBSERVE_THREAD, 1)
        ctx.observe_configure(12, 0, 0)
        for l, r in frames[:11]:
            ctx.observe_submit(l, r, calib, best_percent=bp, frame_life=LIFE)
        ctx.close()  # frames waiting in staging, in flight and uncollected; the launcher thread possibly in a launch


def test_a_waiting_frame_leaves_by_itself_and_poll_does_not_wait():
    """A queue with a launcher thread: a lone frame nobody collects leaves once no frame has arrived for 100 us (an idle GPU
    takes what waits), and vsf_observe_poll -- which neither waits nor sends anything -- reports it finished a moment later;
    the collect then finds the synchronous call's bytes."""
    import time
    from vision_slam_frontend_amd import capi, frontend
    frames = _sequence(3)
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    bp = float(np.float32(0.3))
    with capi.Context(capi.default_params(320, 240, max_images=2, nfeatures=NF)) as sync_ctx:
        want = [sync_ctx.observe_stereo(l, r, calib, best_percent=bp, frame_life=LIFE) for l, r in frames]
    with capi.Context(capi.default_params(320, 240, max_images=16, nfeatures=NF)) as ctx:
        ctx.set_option(capi.OPT_OBSERVE_THREAD, 1)
        ctx.observe_configure(16, 8, 0)  # (eight frames would have to wait for a busy GPU to take them)
        got = []
        for l, r in frames:
            t = ctx.observe_submit(l, r, calib, best_percent=bp, frame_life=LIFE)
            t0 = time.perf_counter()
            while not ctx.observe_poll(t):
                assert time.perf_counter() - t0 < 5.0, "the frame never left the queue"
                time.sleep(0.0002)
            got.append(ctx.observe_collect(t, frame_life=LIFE))
        with pytest.raises(capi.VsfError):
            ctx.observe_poll(t)  # collected
    for g, w in zip(got, want):
        _same_observation(w, g)


def test_destroy_with_frames_still_in_flight():
    """vsf_destroy waits for every stream the context launched on -- the slots' streams of frames that were submitted and never
    collected included -- before it frees what their kernels write (device buffers, the pinned result and status words);
    and a context created afterwards works (round-3 review: only three of the streams were synchronised)."""
    from vision_slam_frontend_amd import capi, frontend, synth
    w, h = 320, 240
    left, right = synth.stereo_pair(w, h, 0, n_objects=400)
    calib = frontend.default_calibration().set("fundamental", F_RECT)
    for _ in range(3):
        ctx = capi.Context(capi.default_params(w, h, max_images=6, nfeatures=500))
        for _k in range(3):
            ctx.observe_submit(left, right, calib, frame_life=2)
        ctx.close()  # three frames pending
    with capi.Context(capi.default_params(w, h, max_images=6, nfeatures=500)) as ctx:
        r = ctx.observe_stereo(left, right, calib, frame_life=2)
        assert r["n_left"] > 300 and r["n_stereo_matches"] > 20
