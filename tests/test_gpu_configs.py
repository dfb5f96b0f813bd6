"""GPU parity of BASELINE.json's configurations through the batched, device-pointer C ABI
(vsf_extract_batch_dev / vsf_match_batch_dev / vsf_stereo_batch_dev), bit for bit against the CPU oracle:

  configs[1]  640x480 stereo stream, 2000 kp/frame            -> test_stereo_batch_640
  configs[2]  1920x1080 stereo, 8000 kp/frame                 -> test_config3_1080p_8000kp
  configs[3]  frames sharded over ranks (one context per GPU) -> the sharding itself is covered on CPU by
              tests/test_distributed_gloo.py; here: a batch processed as two half-batches gives identical results
  configs[4]  temporal window of 8 frames, N x N multi-query   -> test_config5_temporal_window_multi_query
  plus the size-independent properties of the matcher at full size (idempotence, self-match, symmetry of
  distances) -> test_matcher_properties_full_size
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev_outputs(n_images, K, dev):
    return (torch.zeros((n_images, K, 28), dtype=torch.uint8, device=dev),
            torch.zeros((n_images, K, 32), dtype=torch.uint8, device=dev),
            torch.zeros(n_images, dtype=torch.int32, device=dev))


def _oracle_frame(oracle, left, right, nf):
    a, b = oracle.Orb(nfeatures=nf), oracle.Orb(nfeatures=nf)
    a.run(left)
    b.run(right)
    ka, da = a.result()
    kb, db = b.result()
    return ka, da, kb, db, oracle.get_matches(da, db)


def _run_stereo_batch(capi, frames, nf, lanes=1, pipeline=False, repeats=1, resident=None, tune=False):
    """frames: (B, 2, h, w) uint8 -> per-image keypoints / descriptors and per-frame matches (numpy)."""
    B, _, H, W = frames.shape
    dev = torch.device("cuda", 0)
    p = capi.default_params(W, H, max_images=2 * B, nfeatures=nf)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
        d_kp, d_desc, d_counts = _dev_outputs(2 * B, K, dev)
        d_m = torch.zeros((B, K, 16), dtype=torch.uint8, device=dev)
        d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # torch's fills are done before the context's stream touches the buffers
        ctx.set_lanes(lanes)
        ctx.set_pipeline(pipeline)
        if resident is not None:
            ctx.set_fast_resident(resident)
        if tune:
            g, r = ctx.tune_fast_resident(d_img.data_ptr(), 2 * B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(),
                                          d_counts.data_ptr(), samples=2)
            assert g > 0 and r > 0 and ctx.get_fast_resident() == (3 if r < g else 0)
        torch.cuda.synchronize()
        for _ in range(repeats):
            ctx.stereo_batch_dev(d_img.data_ptr(), B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr(),
                                 d_m.data_ptr(), d_nm.data_ptr())
        assert ctx.sync() == capi.VSF_OK
        kp, desc, counts = d_kp.cpu().numpy(), d_desc.cpu().numpy(), d_counts.cpu().numpy()
        m, nm = d_m.cpu().numpy(), d_nm.cpu().numpy()
    return kp, desc, counts, m, nm


def _check_frame(oracle, frames, f, nf, kp, desc, counts, m, nm):
    ka, da, kb, db, rm = _oracle_frame(oracle, frames[f, 0], frames[f, 1], nf)
    for img, (rk, rd) in ((2 * f, (ka, da)), (2 * f + 1, (kb, db))):
        n = int(counts[img])
        assert n == len(rk), "frame %d image %d: %d vs %d keypoints" % (f, img, n, len(rk))
        assert kp[img, :n].tobytes() == rk.tobytes(), "frame %d image %d keypoints" % (f, img)
        np.testing.assert_array_equal(desc[img, :n], rd, err_msg="frame %d image %d descriptors" % (f, img))
    assert int(nm[f]) == len(rm)
    assert m[f, :len(rm)].tobytes() == rm.tobytes(), "frame %d matches" % f
    return len(rm)


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


def test_stereo_batch_640(capi, oracle):
    """configs[1]: the benchmarked entry point on a batch of distinct frames, every frame bit-exact."""
    from vision_slam_frontend_amd import synth
    frames = synth.bench_batch(6, 640, 480, seed=synth.BASE_SEED + 5, n_scenes=3)
    out = _run_stereo_batch(capi, frames, 2000)
    total = sum(_check_frame(oracle, frames, f, 2000, *out) for f in range(len(frames)))
    assert total > 300


def test_half_batches_equal_full_batch(capi):
    """configs[3]: frames are independent, so sharding a batch (here: two half batches, as two ranks would see them)
    changes nothing."""
    from vision_slam_frontend_amd import synth
    frames = synth.bench_batch(4, 640, 480, seed=synth.BASE_SEED + 9, n_scenes=2)
    full = _run_stereo_batch(capi, frames, 2000)
    lo = _run_stereo_batch(capi, frames[:2], 2000)
    hi = _run_stereo_batch(capi, frames[2:], 2000)
    for name, a, b, c in zip(("kp", "desc", "counts", "matches", "nmatches"), full, lo, hi):
        np.testing.assert_array_equal(a, np.concatenate([b, c]), err_msg=name)
    # ... and so does running the two halves concurrently on the context's two lanes (vsf_set_lanes)
    two = _run_stereo_batch(capi, frames, 2000, lanes=2)
    for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), full, two):
        np.testing.assert_array_equal(a, b, err_msg="lanes=2 " + name)
    # ... and so does cross-call pipelining (vsf_set_pipeline: the next call's pyramid is built in the second pyramid
    # buffer beside the previous call's tail); three back-to-back calls exercise both buffers and the release events
    piped = _run_stereo_batch(capi, frames, 2000, pipeline=True, repeats=3)
    for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), full, piped):
        np.testing.assert_array_equal(a, b, err_msg="pipeline " + name)


@pytest.mark.parametrize("w,h,nf,B", [(640, 480, 2000, 16), (320, 240, 500, 20), (1920, 1080, 8000, 16), (752, 480, 1500, 16)])
def test_resident_fast_equals_grid_fast(capi, oracle, w, h, nf, B):
    """vsf_set_fast_resident: FAST as one resident workgroup per CU that draws cells from a counter (the form a batched
    call uses when its blur would outlast the selection) against FAST as one workgroup per four cells -- same candidates,
    so the same keypoints, descriptors and matches; and whatever vsf_tune_fast_resident chose, as well."""
    from vision_slam_frontend_amd import synth
    frames = synth.bench_batch(B, w, h, seed=synth.BASE_SEED + 33, n_scenes=4)  # >= 32 images: the blur runs beside FAST
    grid = _run_stereo_batch(capi, frames, nf, resident=0)
    for waves in ((3, 2, 4) if w == 640 else (3,)):
        res = _run_stereo_batch(capi, frames, nf, resident=waves, repeats=2)  # (twice: the cell counters are re-armed)
        for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), grid, res):
            np.testing.assert_array_equal(a, b, err_msg="%d waves per SIMD: %s" % (waves, name))
    auto = _run_stereo_batch(capi, frames, nf, repeats=2, tune=True)  # the measured form, whichever it is
    for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), grid, auto):
        np.testing.assert_array_equal(a, b, err_msg="measured choice: " + name)
    for f in ((0, B - 1) if w <= 752 else (B - 1,)):
        _check_frame(oracle, frames, f, nf, *grid)


def test_fast_resident_switch(capi):
    """vsf_set_fast_resident / vsf_get_fast_resident / vsf_tune_fast_resident: argument checking; NO batched call measures or
    waits by itself (round-3 review: the auto-tune blocked inside *_dev entry points) -- every call of a fresh context
    returns while the GPU is still working and runs the grid form; the explicit, blocking tune call returns both medians
    and fixes the form for its batch size only; the outputs are the same in every form."""
    import time

    from vision_slam_frontend_amd import synth
    B, w, h, nf = 192, 320, 240, 500
    frames = synth.bench_batch(B, w, h, seed=synth.BASE_SEED + 41, n_scenes=4)
    dev = torch.device("cuda", 0)
    p = capi.default_params(w, h, max_images=2 * B, nfeatures=nf)
    outs = []
    with capi.Context(p) as ctx:
        for bad in (1, 5, -2):
            with pytest.raises(capi.VsfError):
                ctx.set_fast_resident(bad)
        assert ctx.get_fast_resident() == 0  # nothing measured, nothing set: the grid form
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
        bufs = _dev_outputs(2 * B, K, dev) + (torch.zeros((B, K, 16), dtype=torch.uint8, device=dev),
                                              torch.zeros(B, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()
        # the asynchronous contract: calls 1..4 are queued in far less time than the GPU needs for them
        t0 = time.perf_counter()
        for call in range(4):
            ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
        t_submit = time.perf_counter() - t0
        assert ctx.sync() == capi.VSF_OK
        t_total = time.perf_counter() - t0
        assert t_submit < 0.5 * t_total, "batched calls must not wait for the GPU (%.1f ms of %.1f ms)" % (1e3 * t_submit, 1e3 * t_total)
        assert ctx.get_fast_resident() == 0
        outs.append([t.cpu().numpy().copy() for t in bufs])
        with pytest.raises(capi.VsfError):
            ctx.tune_fast_resident(d_img.data_ptr(), 2 * B, w * h, w, *[t.data_ptr() for t in bufs[:3]], samples=0)
        choices = []
        for _ in range(3):
            g, r = ctx.tune_fast_resident(d_img.data_ptr(), 2 * B, w * h, w, *[t.data_ptr() for t in bufs[:3]], samples=3)
            assert g > 0 and r > 0
            choices.append(ctx.get_fast_resident())
            assert choices[-1] == (3 if r < g else 0)
        ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
        assert ctx.sync() == capi.VSF_OK
        outs.append([t.cpu().numpy().copy() for t in bufs])
        # a batch too small for the blur to run beside FAST has one form: the call says so
        g, r = ctx.tune_fast_resident(d_img.data_ptr(), 16, w * h, w, *[t.data_ptr() for t in bufs[:3]], samples=1)
        assert (g, r) == (0.0, 0.0) and ctx.get_fast_resident() == 0
        ctx.set_fast_resident(3)
        assert ctx.get_fast_resident() == 3
        ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
        assert ctx.sync() == capi.VSF_OK
        outs.append([t.cpu().numpy().copy() for t in bufs])
        ctx.set_fast_resident(0)
        assert ctx.get_fast_resident() == 0
        ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
        assert ctx.sync() == capi.VSF_OK
        ref = [t.cpu().numpy().copy() for t in bufs]
    for call, o in enumerate(outs):
        for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), ref, o):
            np.testing.assert_array_equal(a, b, err_msg="call %d: %s" % (call, name))
    assert int(ref[2].min()) > 100


@pytest.mark.parametrize("w,h,nf,B", [(640, 480, 2000, 96), (320, 240, 500, 100), (1920, 1080, 8000, 96)])
def test_large_batch_equals_small_batches(capi, oracle, w, h, nf, B):
    """A batch that fills the CUs (>= 192 images on an MI355X) builds the one-band levels of its pyramids with the
    image-major LDS kernel (k_pyramid.hip pyramid_image_kernel) instead of per-level launches: identical outputs, and the
    oracle agrees on the first and the last frame."""
    from vision_slam_frontend_amd import synth
    frames = synth.bench_batch(B, w, h, seed=synth.BASE_SEED + 21, n_scenes=4)
    big = _run_stereo_batch(capi, frames, nf)
    for lo in (0, B // 2, B - 2):
        small = _run_stereo_batch(capi, frames[lo:lo + 2], nf)
        for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), big, small):
            sl = slice(2 * lo, 2 * lo + 4) if name in ("kp", "desc", "counts") else slice(lo, lo + 2)
            np.testing.assert_array_equal(a[sl], b, err_msg="%s of frames %d..%d" % (name, lo, lo + 1))
    for f in (0, B - 1):
        _check_frame(oracle, frames, f, nf, *big)


def test_config3_1080p_8000kp(capi, oracle):
    """configs[2]: 1920x1080, 8000 keypoints per image (wide levels: 8 FAST bands, HBM-scratch selection)."""
    from vision_slam_frontend_amd import synth
    left, right = synth.stereo_pair(1920, 1080, 0)
    frames = np.stack([left, right])[None]
    out = _run_stereo_batch(capi, frames, 8000)
    assert int(out[2][0]) > 7000
    n = _check_frame(oracle, frames, 0, 8000, *out)
    assert n > 200


def test_config5_temporal_window_multi_query(capi, oracle):
    """configs[4]: a window of 8 frames; the newest frame's descriptors are the train set of seven query sets (the
    reference's loop over frame_list_, slam_frontend.cc:424-434) in ONE matcher launch."""
    from vision_slam_frontend_amd import synth
    NF, Wn = 2000, 8
    stream = synth.stereo_stream(Wn, 640, 480)
    lefts = np.stack([np.asarray(f[0]) for f in stream])
    dev = torch.device("cuda", 0)
    p = capi.default_params(640, 480, max_images=Wn, nfeatures=NF)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints
        torch.cuda.synchronize()  # torch's fills are done before the context's stream touches the buffers
        d_img = torch.from_numpy(np.ascontiguousarray(lefts)).to(dev)
        d_kp, d_desc, d_counts = _dev_outputs(Wn, K, dev)
        torch.cuda.synchronize()
        ctx.extract_batch_dev(d_img.data_ptr(), Wn, 640 * 480, 640, d_kp.data_ptr(), d_desc.data_ptr(),
                              d_counts.data_ptr())
        npairs = Wn - 1
        q_set = torch.arange(0, npairs, dtype=torch.int32, device=dev)
        t_set = torch.full((npairs,), Wn - 1, dtype=torch.int32, device=dev)
        d_idx = torch.zeros((npairs, K, 2), dtype=torch.int32, device=dev)
        d_dist = torch.zeros((npairs, K, 2), dtype=torch.int32, device=dev)
        d_m = torch.zeros((npairs, K, 16), dtype=torch.uint8, device=dev)
        d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.match_batch_dev(d_desc.data_ptr(), d_counts.data_ptr(), K * 32, q_set.data_ptr(), t_set.data_ptr(), npairs,
                            d_idx.data_ptr(), d_dist.data_ptr(), d_m.data_ptr(), d_nm.data_ptr())
        assert ctx.sync() == capi.VSF_OK
        desc, counts = d_desc.cpu().numpy(), d_counts.cpu().numpy()
        idx, dist, m, nm = d_idx.cpu().numpy(), d_dist.cpu().numpy(), d_m.cpu().numpy(), d_nm.cpu().numpy()
    sets = []
    for i in range(Wn):
        o = oracle.Orb(nfeatures=NF)
        o.run(lefts[i])
        _, rd = o.result()
        n = int(counts[i])
        assert n == len(rd)
        np.testing.assert_array_equal(desc[i, :n], rd)
        sets.append(rd)
    total = 0
    for pr in range(npairs):
        ri, rdist = oracle.knn2_hamming(sets[pr], sets[Wn - 1])
        nq = len(sets[pr])
        np.testing.assert_array_equal(idx[pr, :nq], ri, err_msg="pair %d idx" % pr)
        np.testing.assert_array_equal(dist[pr, :nq], rdist, err_msg="pair %d dist" % pr)
        rm = oracle.get_matches(sets[pr], sets[Wn - 1])
        assert int(nm[pr]) == len(rm)
        assert m[pr, :len(rm)].tobytes() == rm.tobytes(), "pair %d matches" % pr
        total += len(rm)
    assert total > 500


def test_matcher_properties_full_size(capi):
    """Size-independent properties at the bench's full size (2000 x 2000, and 8000 x 8000 for configs[2]):
    a set matched against itself finds itself at distance 0; distances are symmetric; rerunning is idempotent."""
    from vision_slam_frontend_amd import synth
    p = capi.default_params(640, 480, max_images=2, nfeatures=2000)
    with capi.Context(p) as ctx:
        for n in (2000, 8000):
            a = synth.random_descriptors(n, seed=77)
            b = synth.random_descriptors(n, seed=78)
            i_self, d_self = ctx.knn2_hamming(a, a)
            np.testing.assert_array_equal(i_self[:, 0], np.arange(n))
            assert (d_self[:, 0] == 0).all() and (d_self[:, 1] > 0).all()
            iab, dab = ctx.knn2_hamming(a, b)
            iab2, dab2 = ctx.knn2_hamming(a, b)
            np.testing.assert_array_equal(iab, iab2)
            np.testing.assert_array_equal(dab, dab2)
            # the reported distance is the true Hamming distance of the reported pair
            x = np.unpackbits(a ^ b[iab[:, 0]], axis=1).sum(1)
            np.testing.assert_array_equal(x, dab[:, 0])
            # nobody is closer than the reported nearest neighbour (checked on a sample of rows)
            for r in range(0, n, max(1, n // 16)):
                dd = np.unpackbits(a[r][None] ^ b, axis=1).sum(1)
                assert dd.min() == dab[r, 0] and np.sort(dd)[1] == dab[r, 1]
                assert iab[r, 0] == int(np.flatnonzero(dd == dd.min())[0])


def test_options_and_sticky_hip_errors(capi):
    """vsf_set_option / vsf_get_option replace every environment switch (nothing in the library reads the environment);
    and a HIP failure inside an asynchronous call is RETURNED, not dropped (round-3 review: every hipEventRecord /
    hipStreamWaitEvent status was cast to void): whatever a launcher or stream helper notes (vsf_note) comes back from the
    entry point that called it as VSF_ERR_HIP with the runtime's code, and the context works again afterwards."""
    from vision_slam_frontend_amd import synth
    w, h, nf, B = 320, 240, 500, 2
    frames = synth.bench_batch(B, w, h, seed=synth.BASE_SEED + 5, n_scenes=2)
    dev = torch.device("cuda", 0)
    with capi.Context(capi.default_params(w, h, max_images=2 * B, nfeatures=nf)) as ctx:
        defaults = {capi.OPT_FAST_BOTH_MAX: 16, capi.OPT_SELECT_WIDE: 1, capi.OPT_PYRAMID_FEW: 16, capi.OPT_PYRAMID_CHAIN: 8,
                    capi.OPT_PYRAMID_ROWS: 6, capi.OPT_SELECT_BIG_CLASS: 1, capi.OPT_PIPE_AFTER_FAST: 1, capi.OPT_PIPE_PRIORITY: 0,
                    capi.OPT_OBSERVE_THREAD: 0, capi.OPT_PYRAMID_TAIL_MIN: 0, capi.OPT_OBSERVE_COPY_THREAD: 1}
        for opt, want in defaults.items():
            assert ctx.get_option(opt) == want, opt
        for opt, bad in ((99, 0), (-1, 0), (capi.OPT_PYRAMID_ROWS, 0), (capi.OPT_FAST_BOTH_MAX, -1), (capi.OPT_PYRAMID_CHAIN, 65)):
            with pytest.raises(capi.VsfError):
                ctx.set_option(opt, bad)
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
        bufs = _dev_outputs(2 * B, K, dev) + (torch.zeros((B, K, 16), dtype=torch.uint8, device=dev),
                                              torch.zeros(B, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()

        def run():
            ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
            assert ctx.sync() == capi.VSF_OK
            return [t.cpu().numpy().copy() for t in bufs]

        ref = run()
        # every launch choice gives the same bytes
        for opt, val in ((capi.OPT_FAST_BOTH_MAX, 0), (capi.OPT_SELECT_WIDE, 0), (capi.OPT_PYRAMID_CHAIN, 0),
                         (capi.OPT_PYRAMID_FEW, 0), (capi.OPT_PYRAMID_ROWS, 3), (capi.OPT_SELECT_BIG_CLASS, 0),
                         (capi.OPT_PYRAMID_TAIL_MIN, 2)):
            ctx.set_option(opt, val)
            assert ctx.get_option(opt) == val
            got = run()
            for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), ref, got):
                np.testing.assert_array_equal(a, b, err_msg="option %d = %d: %s" % (opt, val, name))
            ctx.set_option(opt, defaults[opt])
        # a failure noted by a launcher / stream helper inside an asynchronous call (injected: the error plumbing, not HIP,
        # is under test) comes back from THAT call as VSF_ERR_HIP with its code, from every launching entry point
        calls = {
            "vsf_stereo_batch_dev": lambda: ctx.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs]),
            "vsf_extract_batch_dev": lambda: ctx.extract_batch_dev(d_img.data_ptr(), 2 * B, w * h, w, *[t.data_ptr() for t in bufs[:3]]),
            "vsf_sync": lambda: ctx.sync(),
        }
        for name, call in calls.items():
            assert capi.lib().vsf_debug_inject_hip_error(ctx._h, 719) == capi.VSF_OK
            with pytest.raises(capi.VsfError) as ei:
                call()
            assert ei.value.status == capi.VSF_ERR_HIP and "719" in str(ei.value), (name, str(ei.value))
            got = run()  # ... and the context works again
            for nm_, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), ref, got):
                np.testing.assert_array_equal(a, b, err_msg="after the failed %s: %s" % (name, nm_))
        # An error stays with the context whose call met it (round-4 advice: the noted error used to live in a host-thread
        # slot, so a call that returned early left it for whichever context the thread drove next): noted during a call of
        # `ctx`, it is not seen by `other`, driven from the same thread in between, and comes back from ctx's next call.
        with capi.Context(capi.default_params(w, h, max_images=2 * B, nfeatures=nf)) as other:
            assert capi.lib().vsf_debug_inject_hip_error(ctx._h, 719) == capi.VSF_OK
            other.stereo_batch_dev(d_img.data_ptr(), B, w * h, w, *[t.data_ptr() for t in bufs])
            assert other.sync() == capi.VSF_OK and capi.lib().vsf_last_hip_error(other._h) == 0
            with pytest.raises(capi.VsfError) as ei:
                ctx.sync()
            assert ei.value.status == capi.VSF_ERR_HIP and "719" in str(ei.value)
            assert capi.lib().vsf_last_hip_error(ctx._h) == 719
            assert ctx.sync() == capi.VSF_OK
        # (a destroyed stream cannot serve as the injected failure: ROCm 7's runtime dereferences stream handles without
        # looking them up -- hipStreamQuery and the launch path both crashed the process on one in round 4)
        ctx.set_stream(None)
        got = run()
        for name, a, b in zip(("kp", "desc", "counts", "matches", "nmatches"), ref, got):
            np.testing.assert_array_equal(a, b, err_msg="after the failed call: " + name)
