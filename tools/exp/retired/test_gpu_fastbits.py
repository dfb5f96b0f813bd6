"""GPU parity of FAST on bit planes (k_fastbits.hip, VSF_OPT_FAST_BITS; cv::FAST_t<16> + cornerScore<16> as ORB calls
them, slam_frontend.cc:274): the candidates of every level against the CPU oracle's FAST stage entry for entry (raster
order, score), the batch's keypoints and descriptors against the march kernel's byte for byte -- on batches that do not
fill an image group, on geometries whose bands take 1..8 lanes per image, on photographs, on noise and on thresholds from
1 to 200."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


def _extract(capi, ctx, imgs, form):
    B, H, W = imgs.shape
    dev = torch.device("cuda", 0)
    K = ctx.params.max_keypoints
    ctx.set_option(capi.OPT_FAST_BITS, form)
    assert ctx.get_option(capi.OPT_FAST_BITS) == form
    Wp = (W + 15) // 16 * 16  # (the device entry points take 16-byte aligned rows)
    padded = np.zeros((B, H, Wp), np.uint8)
    padded[:, :, :W] = imgs
    d_img = torch.from_numpy(padded).to(dev)
    d_kp = torch.zeros((B, K, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, K, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.extract_batch_dev(d_img.data_ptr(), B, Wp * H, Wp, d_kp.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr())
    assert ctx.sync() == capi.VSF_OK
    cands = [[ctx.debug_fast_candidates(i, lv).copy() for lv in range(ctx.nlevels)] for i in range(B)]
    return d_kp.cpu().numpy(), d_desc.cpu().numpy(), d_counts.cpu().numpy(), cands


def _compare_forms(capi, imgs, nfeatures, oracle=None, oracle_images=(), **params):
    B, H, W = imgs.shape
    p = capi.default_params(W, H, max_images=B, nfeatures=nfeatures, **params)
    with capi.Context(p) as ctx:
        march = _extract(capi, ctx, imgs, 0)
        bits = _extract(capi, ctx, imgs, 2)
        nlevels = ctx.nlevels
    total = 0
    for i in range(B):
        for lv in range(nlevels):
            a, b = march[3][i][lv], bits[3][i][lv]
            assert len(a) == len(b), "image %d level %d: %d candidates against %d" % (i, lv, len(b), len(a))
            assert a.tobytes() == b.tobytes(), "image %d level %d" % (i, lv)
            total += len(a)
    np.testing.assert_array_equal(march[2], bits[2])
    assert march[0].tobytes() == bits[0].tobytes(), "keypoints"
    assert march[1].tobytes() == bits[1].tobytes(), "descriptors"
    if oracle is not None:
        for i in oracle_images:
            o = oracle.Orb(oracle.orb_params(nfeatures=nfeatures, fast_threshold=params.get("fast_threshold", 20)))
            o.run(imgs[i])
            for lv in range(nlevels):
                g, r = bits[3][i][lv], o.stage(0, lv)
                assert len(g) == len(r), "image %d level %d against the oracle" % (i, lv)
                for f in ("x", "y", "response"):
                    np.testing.assert_array_equal(g[f], r[f], err_msg="image %d level %d %s" % (i, lv, f))
            rk, rd = o.result()
            n = int(bits[2][i])
            assert n == len(rk) and bits[0][i, :n].tobytes() == rk.tobytes()
            np.testing.assert_array_equal(bits[1][i, :n], rd)
    return total


def test_bits_form_equals_march_and_oracle_640(capi, oracle):
    """BASELINE configs[1] geometry; 11 images = one full group of eight + a group of three."""
    from vision_slam_frontend_amd import synth
    fr = synth.bench_batch(6, 640, 480)
    imgs = fr.reshape(-1, 480, 640)[:11]
    total = _compare_forms(capi, imgs, 2000, oracle, oracle_images=(0, 10))
    assert total > 11 * 50000


@pytest.mark.parametrize("w,h,nf", [(320, 240, 500), (451, 300, 700), (1000, 130, 600), (96, 400, 200), (1920, 1080, 8000)])
def test_bits_form_other_geometries(capi, oracle, w, h, nf):
    """Band widths from a few columns to 248 (1..8 lanes per image, up to 64 images side by side in a wave), levels of one
    strip, odd pitches; 1920x1080 = BASELINE configs[2]."""
    from vision_slam_frontend_amd import synth
    B = 3 if w * h > 1_000_000 else 10
    fr = synth.bench_batch((B + 1) // 2, w, h)
    imgs = fr.reshape(-1, h, w)[:B]
    _compare_forms(capi, imgs, nf, oracle, oracle_images=(B - 1,))


def test_bits_form_noise_and_thresholds(capi, oracle):
    """Uniform noise (a third of the pixels are corners at a low threshold: several compaction rounds per row) and
    thresholds at both ends."""
    rng = np.random.Generator(np.random.PCG64(11))
    imgs = rng.integers(0, 256, (9, 240, 320), dtype=np.uint8)
    imgs[3] = (imgs[3] // 64) * 64 + 31        # plateaus: ties in the suppression
    imgs[4, ::2, ::2] = 255                    # saturated lattice
    imgs[5] = 128 + (imgs[5] % 7)              # almost flat
    for t in (1, 2, 20, 77, 200):
        _compare_forms(capi, imgs, 500, oracle, oracle_images=(0, 3, 4), fast_threshold=t)


def test_bits_form_photographs(capi, oracle):
    """Photographs (tests/golden/real: 0.1 .. 13 % of the pixels are corners, against 18 % in the synthetic scenes)."""
    from pathlib import Path
    from PIL import Image
    root = Path(__file__).resolve().parent / "golden" / "real"
    imgs = [np.asarray(Image.open(f)) for f in sorted(root.glob("*.png"))]
    imgs = np.stack([a for a in imgs if a.dtype == np.uint8 and a.shape == (480, 640)])
    assert len(imgs) >= 8
    _compare_forms(capi, imgs, 2000, oracle, oracle_images=(0, len(imgs) - 1))
