"""Edge cases of the hot path on the GPU, bit for bit against the oracle: images without or with very few corners,
odd image sizes whose top pyramid levels are too small for the 31-pixel border (cv::KeyPointsFilter::runByImageBorder
clears them), other ORB parameter sets (the reference's nfeatures = 10000 literal, a classic 8-level / 1.2 pyramid, a
lower FAST threshold), padded row strides on the device API, and the capacity status."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


def _both(capi, oracle, img, nfeatures, **kw):
    h, w = img.shape
    okw = {k: v for k, v in kw.items() if k in ("fast_threshold", "nlevels", "scale_factor")}
    o = oracle.Orb(nfeatures=nfeatures, **okw)
    o.run(img)
    rk, rd = o.result()
    p = capi.default_params(w, h, max_images=1, nfeatures=nfeatures, **kw)
    with capi.Context(p) as ctx:
        for l in range(ctx.nlevels):
            assert ctx.level_info(l) == o.level_info(l)
        kp, desc = ctx.extract(img)
    assert len(kp) == len(rk)
    assert kp.tobytes() == rk.tobytes()
    np.testing.assert_array_equal(desc, rd)
    return kp, desc


def test_flat_image_has_no_keypoints(capi, oracle):
    img = np.full((240, 320), 97, np.uint8)
    kp, desc = _both(capi, oracle, img, 500)
    assert len(kp) == 0
    with capi.Context(capi.default_params(320, 240, max_images=1, nfeatures=500)) as ctx:
        assert len(ctx.get_matches(desc, desc)) == 0
        i2, d2 = ctx.knn2_hamming(np.zeros((3, 32), np.uint8), desc)
        assert (i2 == -1).all()


def test_single_square_few_keypoints(capi, oracle):
    """Far fewer corners than the budget: every level keeps everything it finds (retainBest is a no-op)."""
    img = np.full((240, 320), 60, np.uint8)
    img[100:140, 150:200] = 200
    kp, _ = _both(capi, oracle, img, 500)
    assert 0 < len(kp) < 500


@pytest.mark.parametrize("w,h", [(333, 257), (161, 131), (100, 75), (64, 64)])
def test_odd_and_small_sizes(capi, oracle, w, h):
    """Widths that are no multiple of 4 / 64, and sizes whose upper levels (or all levels) fall under the border."""
    from vision_slam_frontend_amd import synth
    img = synth.stereo_pair(w, h, 0, n_objects=max(40, w * h // 200))[0]
    kp, _ = _both(capi, oracle, img, 700)
    if w >= 161:
        assert len(kp) > 50


def test_reference_literal_nfeatures_10000(capi, oracle, stereo640):
    kp, _ = _both(capi, oracle, stereo640[0], 10000)  # cv::ORB::create(10000, ...) slam_frontend.cc:205
    assert len(kp) > 8000


def test_classic_orb_pyramid_and_low_threshold(capi, oracle, stereo640):
    kp, _ = _both(capi, oracle, stereo640[0], 1000, nlevels=8, scale_factor=1.2, fast_threshold=7)
    assert len(kp) == 1000


def test_single_level(capi, oracle, stereo640):
    kp, _ = _both(capi, oracle, stereo640[0], 300, nlevels=1)
    assert len(kp) >= 300


@pytest.mark.parametrize("thr,nms", [(0, False), (0, True), (1, False), (254, True), (255, True), (255, False)])
def test_fast_detect_extreme_thresholds(capi, oracle, thr, nms):
    """FastFeatureDetector semantics at the ends of the threshold range: at 0 a corner may have score 0 (kept without
    NMS, can never win with NMS); at 255 nothing can differ by more than the threshold."""
    from vision_slam_frontend_amd import synth
    img = synth.stereo_pair(160, 120, 3, n_objects=60)[0]
    img[40:60, 50:90] = 0
    img[45:55, 60:80] = 255  # saturated contrast: |diff| = 255
    r = oracle.fast9_16(img, thr, nms)
    with capi.Context(capi.default_params(160, 120, max_images=1, nfeatures=100)) as ctx:
        g = ctx.fast_detect(img, thr, nms, cap=160 * 120)
    assert len(g) == len(r)
    assert g.tobytes() == r.tobytes()
    if thr == 255:
        assert len(r) == 0
    if thr == 0 and not nms:
        assert len(r) > 1000


def test_padded_row_stride_device_api(capi, oracle, stereo640):
    torch = pytest.importorskip("torch")
    left = stereo640[0]
    h, w = left.shape
    pitch = 704  # rows 64 bytes longer than the image
    buf = np.random.default_rng(3).integers(0, 255, (2, h, pitch), dtype=np.uint8)  # garbage in the padding
    buf[:, :, :w] = left
    dev = torch.device("cuda", 0)
    p = capi.default_params(w, h, max_images=2, nfeatures=2000)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints
        d_img = torch.from_numpy(buf).to(dev)
        d_kp = torch.zeros((2, K, 28), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((2, K, 32), dtype=torch.uint8, device=dev)
        d_counts = torch.zeros(2, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.extract_batch_dev(d_img.data_ptr(), 2, h * pitch, pitch, d_kp.data_ptr(), d_desc.data_ptr(),
                              d_counts.data_ptr())
        assert ctx.sync() == capi.VSF_OK
        o = oracle.Orb(nfeatures=2000)
        o.run(left)
        rk, rd = o.result()
        for i in range(2):
            n = int(d_counts[i])
            assert n == len(rk)
            assert d_kp[i, :n].cpu().numpy().tobytes() == rk.tobytes()
            np.testing.assert_array_equal(d_desc[i, :n].cpu().numpy(), rd)


def test_capacity_status_and_truncation(capi, oracle, stereo640):
    """An output capacity below the keypoint count reports VSF_ERR_CAPACITY and fills exactly the capacity with the
    first keypoints of the level-major order."""
    left = stereo640[0]
    o = oracle.Orb(nfeatures=2000)
    o.run(left)
    rk, rd = o.result()
    p = capi.default_params(640, 480, max_images=1, nfeatures=2000, max_keypoints=777)
    with capi.Context(p) as ctx:
        with pytest.raises(capi.VsfError) as e:
            ctx.extract(left)
        assert e.value.status == capi.VSF_ERR_CAPACITY
        kp = np.zeros(777, capi.KEYPOINT_DTYPE)
        desc = np.zeros((777, 32), np.uint8)
        import ctypes as C
        n = C.c_int()
        st = capi.lib().vsf_extract(ctx._h, left.ctypes.data_as(C.c_void_p), 640, 480, 640, kp.ctypes.data_as(C.c_void_p),
                                    desc.ctypes.data_as(C.c_void_p), 777, C.byref(n))
        assert st == capi.VSF_ERR_CAPACITY and n.value == len(rk)
        assert kp.tobytes() == rk[:777].tobytes()
        np.testing.assert_array_equal(desc, rd[:777])


def test_randomised_sizes_and_parameters():
    """tools/stress_parity.py in the suite: 100 random cases (sizes 80..900 x 70..700, six ORB parameter sets, thresholds,
    scenes incl. pure noise), keypoints, descriptors and stereo matches bit for bit against the oracle."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tools" / "stress_parity.py"), "100", "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches: 0 of 100" in out.stdout


@pytest.mark.parametrize("w,h,n", [(640, 480, 3), (641, 479, 2), (9, 5, 2), (4, 3, 1), (3, 3, 1), (2, 7, 1), (5, 2, 1),
                                   (1031, 67, 2), (257, 19, 2), (513, 33, 1), (258, 7, 1), (256, 17, 1), (5, 3, 1),
                                   (6, 40, 1), (7, 3, 1), (8, 3, 1)])
def test_bayer_bg_to_gray_batch(capi, oracle, w, h, n):
    """Row f4 (behind imdecode): BayerBG2BGR + BGR2GRAY on the device, bit-exact vs the oracle incl. the copied frame,
    ragged row ends, padded strides and degenerate sizes."""
    import torch
    rng = np.random.default_rng(w * 7 + h)
    sp, dp = (w + 3) // 4 * 4 + 8, (w + 3) // 4 * 4 + 4  # padded row strides (multiples of 4)
    src = rng.integers(0, 256, (n, h, sp), dtype=np.uint8)
    dev = torch.device("cuda", 0)
    p = capi.default_params(640, 480, max_images=2, nfeatures=500)
    with capi.Context(p) as ctx:
        d_src = torch.from_numpy(src).to(dev)
        d_dst = torch.full((n, h, dp), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.bayer_bg_to_gray_batch_dev(d_src.data_ptr(), n, w, h, h * sp, sp, d_dst.data_ptr(), h * dp, dp)
        assert ctx.sync() == capi.VSF_OK
        got = d_dst.cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(got[i, :, :w], oracle.bayer_bg_to_gray(src[i, :, :w]), err_msg="image %d" % i)


@pytest.mark.parametrize("w,h,nf", [(3840, 2160, 8000), (4000, 120, 1000), (96, 3000, 500), (2048, 2048, 4000)])
def test_extreme_geometries(oracle, w, h, nf):
    """Image sizes at the ends of what the geometry tables hold (level width <= 4095, 12-bit candidate coordinates): 4K, a
    strip 4000 wide and 120 high (sixteen FAST bands, two strips), a strip 96 wide and 3000 high, a 2048 square --
    keypoints and descriptors of vsf_extract bit for bit against the oracle."""
    from vision_slam_frontend_amd import capi, synth
    left, _ = synth.stereo_pair(w, h, 1, n_objects=max(50, w * h // 4000))
    o = oracle.Orb(nfeatures=nf)
    o.run(left)
    rk, rd = o.result()
    with capi.Context(capi.default_params(w, h, max_images=1, nfeatures=nf)) as ctx:
        kp, desc = ctx.extract(left, cap=len(rk) + 64)
    assert len(kp) == len(rk) > 100 and kp.tobytes() == rk.tobytes()
    np.testing.assert_array_equal(desc, rd)
