"""GPU parity of vsf_png_decode_gray_batch -- cv::imdecode(msg.data, IMREAD_GRAYSCALE) for PNG payloads
(slam_frontend_main.cc:99-100) -- against the real libpng + zlib (PIL reads every file back; zlib itself says what a damaged
stream is worth): files written by PIL at every compression level and by tests/png_craft.py with every row filter, every
deflate block type (stored, fixed, dynamic, run-length matches of distance 1, long-distance matches), IDAT payloads cut
into pieces, 1 / 2 / 4 / 8 / 16-bit gray and gray + alpha; what libpng refuses is refused (damaged critical chunks, unknown
critical chunks, wrong size), what it only warns about is read (damaged ancillary chunks, a wrong Adler-32, trailing
garbage); damaged compressed data never faults and is refused exactly when the real libpng, driven as cv::imdecode drives
it (tests/png_ref.py), refuses the file."""
import io
import sys
import zlib
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
Image = pytest.importorskip("PIL.Image")

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import png_craft as pc  # noqa: E402


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=500))
    yield c
    c.close()


def pil_gray(png: bytes) -> np.ndarray:
    """What cv::imdecode(IMREAD_GRAYSCALE) returns for a grayscale PNG, computed from libpng's own decode (PIL): 8-bit
    samples as they are, 16-bit samples' high byte (png_set_strip_16), alpha dropped, 1 / 2 / 4-bit replicated."""
    im = Image.open(io.BytesIO(png))
    im.load()
    if im.mode in ("I;16", "I;16B", "I"):
        return (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
    if im.mode in ("LA", "RGBA"):  # (PIL shows 16-bit gray + alpha as 8-bit RGBA: the samples' high bytes)
        return np.asarray(im)[:, :, 0].copy()
    if im.mode == "1":
        return (np.asarray(im).astype(np.uint8) * 255)
    assert im.mode == "L", im.mode
    return np.asarray(im).copy()


def decode(capi, ctx, files, w, h, allow_status=(), pitch=None):
    dev = torch.device("cuda", 0)
    pitch = pitch or (w + 3) // 4 * 4
    n = len(files)
    d = torch.full((n, h, pitch), 0xA5, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = ctx.png_decode_gray_batch(files, w, h, d.data_ptr(), h * pitch, pitch, allow_status=allow_status)
    if st != capi.VSF_OK:
        return st, None, None
    sync = ctx.sync(allow_capacity=False) if not allow_status else None
    return st, d.cpu().numpy()[:, :, :w], sync


def scene(w, h, seed=0, smooth=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    base = (xx * 3 + yy * 2) % 256 if smooth else 0
    return ((base + rng.integers(0, 24 if smooth else 256, (h, w))) % 256).astype(np.uint8)


def test_pil_written_files_every_level(capi, ctx):
    from vision_slam_frontend_amd import synth
    left, right = synth.stereo_pair(640, 480, 0)
    files, want = [], []
    for img in (left, right, scene(640, 480, 1), np.zeros((480, 640), np.uint8), scene(640, 480, 2, smooth=False)):
        for level in (0, 1, 6, 9):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="PNG", compress_level=level)
            files.append(buf.getvalue())
            want.append(img)
    st, got, sync = decode(capi, ctx, files, 640, 480)
    assert st == capi.VSF_OK and sync == capi.VSF_OK
    for i, (g, wv, f) in enumerate(zip(got, want, files)):
        np.testing.assert_array_equal(pil_gray(f), wv)
        np.testing.assert_array_equal(g, wv, err_msg="file %d" % i)


@pytest.mark.parametrize("strategy,level", [(zlib.Z_DEFAULT_STRATEGY, 6), (zlib.Z_FIXED, 6), (zlib.Z_RLE, 6),
                                             (zlib.Z_HUFFMAN_ONLY, 6), (zlib.Z_FILTERED, 9), (zlib.Z_DEFAULT_STRATEGY, 0),
                                             (zlib.Z_DEFAULT_STRATEGY, 1)])
def test_crafted_filters_and_block_types(capi, ctx, strategy, level):
    """Every row filter (one type per row, all five in turn and at random), every deflate block type, IDAT in pieces."""
    rng = np.random.Generator(np.random.PCG64(42 + strategy * 16 + level))
    cases = []
    for w, h in ((640, 480), (67, 41), (1, 1), (5, 300), (300, 3), (64, 64), (65, 129)):
        img = scene(w, h, int(rng.integers(1 << 30)), smooth=bool(rng.integers(2)))
        for filters in (np.arange(h) % 5, rng.integers(0, 5, h), np.full(h, 4), np.full(h, 3)):
            piece = [None, 1 if w * h < 10000 else 4096, 7 if w * h < 10000 else 9000, 8192][int(rng.integers(4))]
            cases.append((w, h, img, pc.gray8(img, filters=filters, level=level, strategy=strategy, idat_piece=piece)))
    by_size = {}
    for w, h, img, f in cases:
        by_size.setdefault((w, h), []).append((img, f))
    for (w, h), group in by_size.items():
        with capi.Context(capi.default_params(max(w, 64), max(h, 64), max_images=2, nfeatures=100)) as c:
            st, got, sync = decode(capi, c, [f for _, f in group], w, h)
            assert st == capi.VSF_OK and sync == capi.VSF_OK, (w, h)
            for i, (img, f) in enumerate(group):
                np.testing.assert_array_equal(pil_gray(f), img, err_msg="PIL disagrees with the crafted file")
                np.testing.assert_array_equal(got[i], img, err_msg="%dx%d file %d" % (w, h, i))


def test_long_matches_far_distances_and_overlaps(capi, ctx):
    """Periodic content: matches of length 258 at distances 1 .. 32768, self-overlapping copies, a window that wraps."""
    w, h = 640, 480
    files, want = [], []
    for period in (1, 2, 3, 5, 63, 64, 65, 257, 258, 259, 640, 641 * 3, 32768 // 641 * 641, 32767):
        flat = (np.arange(w * h) % period * 37 % 251).astype(np.uint8)
        img = flat.reshape(h, w)
        files.append(pc.gray8(img, level=9))
        want.append(img)
    rng = np.random.Generator(np.random.PCG64(3))
    tile = rng.integers(0, 256, 32768, dtype=np.uint8)   # incompressible, then repeated: matches at distance exactly 32768
    img = np.resize(tile, w * h).reshape(h, w)
    files.append(pc.gray8(img, level=9))
    want.append(img)
    # the same at distances a little short of the window, each repetition with a few bytes changed: far matches with literals
    # and short matches behind them -- bytes whose places in a 32 KiB ring are the far match's source
    for period, flips in ((32700, 0.02), (32767, 0.005), (32768 - 258, 0.05), (32768 - 64, 0.2), (20000, 0.01)):
        flat = np.resize(rng.integers(0, 256, period, dtype=np.uint8), w * h)
        hit = rng.random(w * h) < flips
        hit[:period] = False
        flat[hit] = rng.integers(0, 256, int(hit.sum()), dtype=np.uint8)
        img = flat.reshape(h, w)
        for level in (1, 9):
            files.append(pc.gray8(img, level=level))
            want.append(img)
    st, got, sync = decode(capi, ctx, files, w, h)
    assert st == capi.VSF_OK and sync == capi.VSF_OK
    for i in range(len(files)):
        np.testing.assert_array_equal(pil_gray(files[i]), want[i])
        np.testing.assert_array_equal(got[i], want[i], err_msg="file %d" % i)


def test_bit_depths_and_alpha(capi):
    rng = np.random.Generator(np.random.PCG64(9))
    for w, h in ((640, 480), (37, 29), (8, 8), (9, 5)):
        with capi.Context(capi.default_params(max(w, 64), max(h, 64), max_images=2, nfeatures=100)) as c:
            files, want = [], []
            filters = rng.integers(0, 5, h)
            for depth in (1, 2, 4):
                v = rng.integers(0, 1 << depth, (h, w)).astype(np.uint8)
                files.append(pc.write_png(pc.pack_samples(v, depth), w, h, depth, 0, filters=filters))
                want.append((v * (255 // ((1 << depth) - 1))).astype(np.uint8))
            v16 = rng.integers(0, 65536, (h, w)).astype(np.uint16)
            files.append(pc.write_png(pc.pack_samples(v16, 16), w, h, 16, 0, filters=filters))
            want.append((v16 >> 8).astype(np.uint8))
            ga = rng.integers(0, 256, (h, w, 2)).astype(np.uint8)
            files.append(pc.write_png(pc.pack_samples(ga, 8), w, h, 8, 4, filters=filters))
            want.append(ga[:, :, 0])
            ga16 = rng.integers(0, 65536, (h, w, 2)).astype(np.uint16)
            files.append(pc.write_png(pc.pack_samples(ga16, 16), w, h, 16, 4, filters=filters))
            want.append((ga16[:, :, 0] >> 8).astype(np.uint8))
            # gray with a tRNS chunk: png_set_tRNS_to_alpha, then the alpha is stripped again
            v8 = rng.integers(0, 256, (h, w)).astype(np.uint8)
            files.append(pc.write_png(pc.pack_samples(v8, 8), w, h, 8, 0, filters=filters, extra_before=[pc.chunk(b"tRNS", b"\x00\x07")]))
            want.append(v8)
            st, got, sync = decode(capi, c, files, w, h)
            assert st == capi.VSF_OK and sync == capi.VSF_OK
            for i in range(len(files)):
                np.testing.assert_array_equal(pil_gray(files[i]), want[i], err_msg="PIL, %dx%d file %d" % (w, h, i))
                np.testing.assert_array_equal(got[i], want[i], err_msg="%dx%d file %d" % (w, h, i))


def test_what_libpng_refuses_is_refused_and_what_it_warns_about_is_read(capi, ctx):
    w, h = 96, 64
    img = scene(w, h, 5)
    good = pc.gray8(img, filters=np.arange(h) % 5)
    with capi.Context(capi.default_params(w, h, max_images=2, nfeatures=100)) as c:
        def status(f, ww=w, hh=h):
            return c.png_decode_gray_batch([f], ww, hh, torch.zeros((hh, (ww + 3) // 4 * 4), dtype=torch.uint8, device="cuda").data_ptr(),
                                           hh * ((ww + 3) // 4 * 4), (ww + 3) // 4 * 4,
                                           allow_status=(capi.VSF_ERR_INVALID_ARG, capi.VSF_ERR_UNSUPPORTED))
        assert status(good) == capi.VSF_OK and c.sync() == capi.VSF_OK
        assert status(good, w + 1, h) == capi.VSF_ERR_INVALID_ARG          # another size
        assert status(pc.gray8(img, bad_idat_crc=True)) == capi.VSF_ERR_INVALID_ARG
        assert status(good[:-12]) == capi.VSF_ERR_INVALID_ARG               # no IEND
        assert status(good[:40]) == capi.VSF_ERR_INVALID_ARG
        assert status(b"\x89PNG\r\n\x1a\n") == capi.VSF_ERR_INVALID_ARG
        assert status(pc.gray8(img, extra_before=[pc.chunk(b"ABCD", b"xyz")])) == capi.VSF_ERR_INVALID_ARG  # unknown critical chunk
        rgb_rows = np.dstack([img, img // 2, 255 - img]).reshape(h, -1)
        iccp = pc.chunk(b"iCCP", b"x\x00\x00" + zlib.compress(b"not a profile"))
        assert status(pc.write_png(rgb_rows, w, h, 8, 2, extra_before=[iccp])) == capi.VSF_ERR_UNSUPPORTED  # colour + iCCP
        assert status(pc.write_png(np.zeros((h, w), np.uint8), w, h, 8, 3)) == capi.VSF_ERR_INVALID_ARG    # "Missing PLTE before IDAT"
        # a wrong Adler-32 in the piece of input that also holds the image's last byte: zlib checks it in the call that
        # delivers the last row ("incorrect data check"), png_read_IDAT_data raises png_error
        s = bytearray(pc.idat_stream(good))
        s[-1] ^= 0xFF
        wrong_adler = bytes(s)
        assert status(pc.replace_idat(good, wrong_adler)) == capi.VSF_OK
        with pytest.raises(capi.VsfError) as ei:
            c.sync()
        assert ei.value.status == capi.VSF_ERR_INVALID_ARG
        assert c.sync() == capi.VSF_OK
        # warnings only: a damaged ancillary chunk, an ancillary chunk nobody knows, bytes behind the end of the zlib stream
        # ("Extra compressed data"), and the same wrong Adler-32 when it sits in a chunk of its own -- libpng then reads it
        # with no row left to fill (png_read_finish_IDAT), where a zlib error is a warning
        readable = [
            pc.gray8(img, extra_before=[pc.chunk(b"tEXt", b"Comment\x00hello", bad_crc=True)]),
            pc.gray8(img, extra_before=[pc.chunk(b"prVt", b"\x01\x02\x03")], extra_after=[pc.chunk(b"tIME", bytes(7))]),
            pc.replace_idat(good, pc.idat_stream(good) + b"trailing garbage"),
            pc.write_png(None, w, h, 8, 0, stream=wrong_adler, idat_piece=len(wrong_adler) - 4),
        ]
        st, got, sync = decode(capi, c, readable, w, h)
        assert st == capi.VSF_OK and sync == capi.VSF_OK
        for i, f in enumerate(readable):
            np.testing.assert_array_equal(got[i], img, err_msg="file %d" % i)
        # (PIL's own chunk reader is stricter than libpng here -- it refuses a damaged ancillary chunk -- so these are held
        # against the real libpng, driven as cv::imdecode drives it)
        import png_ref
        if png_ref.available():
            for i, f in enumerate(readable):
                ref_status, ref_img, _ = png_ref.imdecode_gray(f, w, h)
                assert ref_status == 0, (i, png_ref.last_error())
                np.testing.assert_array_equal(ref_img, img)
            for f in (pc.gray8(img, bad_idat_crc=True), good[:-12], good[:40], b"\x89PNG\r\n\x1a\n",
                      pc.gray8(img, extra_before=[pc.chunk(b"ABCD", b"xyz")]), pc.replace_idat(good, wrong_adler)):
                assert png_ref.imdecode_gray(f, w, h)[0] != 0
            # a stream cut inside its last bytes: every pixel is there, png_read_end still fails ("Not enough image data")
            for cut in (1, 3, 4, 5, 9):
                f = pc.replace_idat(good, pc.idat_stream(good)[:-cut])
                assert png_ref.imdecode_gray(f, w, h)[0] == 2 and png_ref.last_error() == "Not enough image data"
                assert status(f) == capi.VSF_OK
                with pytest.raises(capi.VsfError) as ei:
                    c.sync()
                assert ei.value.status == capi.VSF_ERR_INVALID_ARG
                assert c.sync() == capi.VSF_OK


def test_damaged_compressed_data_never_faults_and_is_refused_as_libpng_refuses_it(capi):
    """400 files whose compressed data is damaged (bit flips, cuts, zeroed runs, insertions -- anywhere, or only near the
    end), continues behind the image's last byte (further blocks, garbage) or both, with the IDAT payload cut into chunks
    of various sizes.  The reference is the real libpng driven as cv::imdecode drives it (tests/png_ref.py: png_read_image
    and png_read_end): it either delivers an image -- then the decode must be those bytes -- or refuses the file
    (cv::imdecode returns an empty Mat): then the flag must be set.  That includes everything zlib still reads behind
    the last byte in the call that delivers it (the Adler-32, block headers, code tables) and the drain of png_read_end:
    a stream whose IDAT data runs out before it has ended is refused although every pixel was there."""
    import png_ref
    if not png_ref.available():
        pytest.skip("no libpng16.so.16 to build tests/cpp/png_ref.c against")
    w, h = 160, 120
    rng = np.random.Generator(np.random.PCG64(2026))
    flagged = clean = 0
    with capi.Context(capi.default_params(w, h, max_images=2, nfeatures=100)) as c:
        dev = torch.device("cuda", 0)
        for it in range(400):
            k = it % 6
            img = scene(w, h, 100 + it, smooth=k % 2 == 0)
            strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED][k]
            rows = pc.filter_rows(img, 1, rng.integers(0, 5, h))
            extra = [0, 0, 1, 700, 40000][int(rng.integers(5))]   # the stream goes on behind the image ("Too much image data")
            tail_bytes = bytes(rng.integers(0, 256, extra, dtype=np.uint8))
            stream = pc.deflate(rows + tail_bytes, [6, 6, 6, 6, 0, 9][k], strategy)
            how = int(rng.integers(5))
            if how == 0:
                stream = pc.mutate_stream(stream, rng)
            elif how == 1:  # damage near the end only
                cut = max(2, len(stream) - int(rng.integers(1, 300)))
                stream = stream[:cut] + pc.mutate_stream(b"xx" + stream[cut:], rng)[2:]
            elif how == 2:
                stream = stream + bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
            elif how == 3:  # cut inside the last bytes: the end-of-block code, the check value
                stream = stream[:max(2, len(stream) - int(rng.integers(1, 12)))]
            if len(stream) < 8:
                continue
            piece = [None, len(stream) - 4, len(stream) - 7, 100, 5000, 8192, 9000][int(rng.integers(7))]
            f = pc.write_png(None, w, h, 8, 0, stream=stream, idat_piece=piece)
            ref_status, ref_img, _ = png_ref.imdecode_gray(f, w, h)
            d = torch.zeros((h, w), dtype=torch.uint8, device=dev)
            st = c.png_decode_gray_batch([f], w, h, d.data_ptr(), w * h, w, allow_status=(capi.VSF_ERR_INVALID_ARG,))
            if st != capi.VSF_OK:   # the host refused it (a zlib header that no longer passes its checks)
                assert ref_status != 0, "file %d: refused by the host, read by libpng" % it
                continue
            try:
                sync = c.sync()
            except capi.VsfError as e:
                sync = e.status
            if ref_status != 0:
                assert sync == capi.VSF_ERR_INVALID_ARG, "file %d: not flagged; libpng: %s" % (it, png_ref.last_error())
                flagged += 1
            else:
                assert sync == capi.VSF_OK, "file %d: flagged although libpng delivers the image" % it
                np.testing.assert_array_equal(d.cpu().numpy(), ref_img, err_msg="file %d" % it)
                clean += 1
    assert flagged > 60 and clean > 60, (flagged, clean)


def pil_unfilter(rows, w, h):
    """Reconstructs 8-bit gray scanlines (filter byte + w bytes per row) with PIL: the rows go back into a stored-block zlib
    stream inside a fresh PNG."""
    stream = pc.deflate(rows.tobytes(), 0)
    return pil_gray(pc.write_png(None, w, h, 8, 0, stream=stream))


def test_committed_fixtures_pinned_by_libpng(capi):
    """tests/golden/png (tools/make_png_golden.py): files written by PIL's libpng and by png_craft, with what libpng read out
    of them when the fixtures were made (the gray ones through PIL, the colour and palette ones through libpng driven as
    cv::imdecode drives it).  No PIL or libpng at test time: the vectors are the reference."""
    gold = HERE / "golden" / "png"
    expected = dict(np.load(gold / "expected_gray.npz"))
    expected.update(np.load(gold / "expected_gray_colour.npz"))   # colour and palette files, read by libpng's rgb_to_gray
    by_size = {}
    for f in sorted(gold.glob("*.png")):
        by_size.setdefault(expected[f.stem].shape, []).append(f)
    assert sum(len(v) for v in by_size.values()) >= 35
    for (h, w), files in by_size.items():
        with capi.Context(capi.default_params(max(w, 64), max(h, 64), max_images=2, nfeatures=100)) as c:
            st, got, sync = decode(capi, c, [f.read_bytes() for f in files], w, h)
            assert st == capi.VSF_OK and sync == capi.VSF_OK
            for i, f in enumerate(files):
                np.testing.assert_array_equal(got[i], expected[f.stem], err_msg=f.name)


def test_ingest_chain_png_bayer_extract(capi, oracle):
    """DecodeImage as a whole (slam_frontend_main.cc:98-109) followed by ExtractFeatures, for a PNG payload: a Bayer mosaic
    stored as a gray PNG -> imdecode -> BayerBG2BGR -> BGR2GRAY -> ORB, against the same chain on the oracle, with the
    extraction's pyramid pipelined behind the decode (the call is handed the images by the ingest's own event)."""
    from vision_slam_frontend_amd import synth
    dev = torch.device("cuda", 0)
    mosaics = [synth.stereo_pair(640, 480, 5 + k)[k & 1] for k in range(4)]
    files = []
    for k, m in enumerate(mosaics):
        b = io.BytesIO()
        Image.fromarray(m, "L").save(b, "PNG", compress_level=[1, 6, 9, 0][k])
        files.append(b.getvalue())
    with capi.Context(capi.default_params(640, 480, max_images=4, nfeatures=500)) as ctx:
        ctx.set_pipeline(True)
        K = ctx.params.max_keypoints
        d_mosaic = torch.zeros((4, 480, 640), dtype=torch.uint8, device=dev)
        d_gray = torch.zeros((4, 480, 640), dtype=torch.uint8, device=dev)
        d_kp = torch.zeros((4, K, 28), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((4, K, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for rep in range(2):   # (twice: the second call's pyramid is the pipelined one)
            ctx.png_decode_gray_batch(files, 640, 480, d_mosaic.data_ptr(), 640 * 480, 640)
            ctx.bayer_bg_to_gray_batch_dev(d_mosaic.data_ptr(), 4, 640, 480, 640 * 480, 640, d_gray.data_ptr(), 640 * 480, 640)
            ctx.extract_batch_dev(d_gray.data_ptr(), 4, 640 * 480, 640, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr())
        assert ctx.sync() == capi.VSF_OK
        np.testing.assert_array_equal(d_mosaic.cpu().numpy(), np.stack(mosaics))
        for k in range(4):
            gray = oracle.bayer_bg_to_gray(mosaics[k])
            np.testing.assert_array_equal(d_gray.cpu().numpy()[k], gray)
            o = oracle.Orb(nfeatures=500)
            o.run(gray)
            rk, rd = o.result()
            n = int(d_n.cpu()[k])
            assert n == len(rk) > 300
            assert d_kp.cpu().numpy()[k, :n].tobytes() == rk.tobytes()
            np.testing.assert_array_equal(d_desc.cpu().numpy()[k, :n], rd)


def test_imdecode_tells_jpeg_from_png(capi, oracle):
    """vsf_imdecode_gray_batch: a batch of mixed payloads, as DecodeImage meets them (slam_frontend_main.cc:99-100): runs of
    JPEG and PNG files in one call, each image at its own index; a BMP is refused as unsupported."""
    from vision_slam_frontend_amd import synth
    dev = torch.device("cuda", 0)
    w, h = 320, 240
    imgs = [synth.stereo_pair(w, h, 40 + k, n_objects=200)[k & 1] for k in range(7)]
    files, want = [], []
    for k, img in enumerate(imgs):
        b = io.BytesIO()
        if k == 4:    # a colour PNG among them: libpng's rgb_to_gray (the integer sum, truncated)
            rgb = np.dstack([img, np.roll(img, 3, 0), 255 - img])
            Image.fromarray(rgb, "RGB").save(b, "PNG")
            files.append(b.getvalue())
            r, g, bl = (rgb[..., i].astype(np.uint32) for i in range(3))
            want.append(((9797 * r + 19234 * g + 3737 * bl) >> 15).astype(np.uint8))
        elif k in (0, 1, 6):
            Image.fromarray(img, "L").save(b, "PNG", compress_level=k % 3 * 3)
            files.append(b.getvalue())
            want.append(img)
        else:
            Image.fromarray(img, "L").save(b, "JPEG", quality=85)
            files.append(b.getvalue())
            want.append(oracle.jpeg_decode_gray(b.getvalue()))
    with capi.Context(capi.default_params(w, h, max_images=2, nfeatures=100)) as c:
        d = torch.zeros((7, h, w), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert c.imdecode_gray_batch(files, w, h, d.data_ptr(), w * h, w) == capi.VSF_OK
        assert c.sync() == capi.VSF_OK
        got = d.cpu().numpy()
        for k in range(7):
            np.testing.assert_array_equal(got[k], want[k], err_msg="file %d" % k)
        bmp = io.BytesIO()
        Image.fromarray(imgs[0], "L").save(bmp, "BMP")
        st = c.imdecode_gray_batch([files[0], bmp.getvalue()], w, h, d.data_ptr(), w * h, w, allow_status=(capi.VSF_ERR_UNSUPPORTED,))
        assert st == capi.VSF_ERR_UNSUPPORTED
        assert c.sync() == capi.VSF_OK


def test_colour_and_palette_files_as_libpngs_rgb_to_gray(capi):
    """Colour types 2, 3 and 6 -- cv::imdecode(IMREAD_GRAYSCALE) hands them to libpng's rgb_to_gray(0.299, 0.587) -- against the
    real libpng driven the same way (tests/png_ref.py): 8- and 16-bit RGB / RGBA, palettes of 1, 2, 4 and 8 bits (short
    palettes, indices behind their end, tRNS), every row filter, no gamma chunk / gAMA at several values / sRGB (the linearised
    sum through the two tables), cHRM and bKGD beside them; what the restatement leaves out is UNSUPPORTED, never wrong."""
    import struct
    import png_ref
    if not png_ref.available():
        pytest.skip("no libpng16.so.16 to build tests/cpp/png_ref.c against")
    rng = np.random.Generator(np.random.PCG64(99))
    cases = []   # (w, h, file, libpng must read it)
    srgb_chrm = pc.chunk(b"cHRM", struct.pack(">8I", 31270, 32900, 64000, 33000, 30000, 60000, 15000, 6000))
    gamma_sets = [[], [pc.chunk(b"gAMA", struct.pack(">I", 45455))], [pc.chunk(b"gAMA", struct.pack(">I", 100000))],
                  [pc.chunk(b"gAMA", struct.pack(">I", 96000))], [pc.chunk(b"gAMA", struct.pack(">I", 50000))],
                  [pc.chunk(b"gAMA", struct.pack(">I", 220000))], [pc.chunk(b"gAMA", struct.pack(">I", 94999))],
                  [pc.chunk(b"sRGB", b"\x01")], [pc.chunk(b"cHRM", bytes(32))],
                  [srgb_chrm, pc.chunk(b"gAMA", struct.pack(">I", 45455))], [pc.chunk(b"gAMA", struct.pack(">I", 60000)), srgb_chrm],
                  [pc.chunk(b"sRGB", b"\x00"), pc.chunk(b"gAMA", struct.pack(">I", 45455)), srgb_chrm],
                  [pc.chunk(b"gAMA", struct.pack(">I", 100000)), pc.chunk(b"sRGB", b"\x03")],
                  [pc.chunk(b"sRGB", b"\x02"), pc.chunk(b"gAMA", struct.pack(">I", 30000))],
                  [pc.chunk(b"gAMA", struct.pack(">I", 45455), bad_crc=True)], [pc.chunk(b"bKGD", bytes(6)), pc.chunk(b"gAMA", struct.pack(">I", 31250))]]
    for w, h in ((96, 64), (67, 41), (1, 1), (5, 130), (200, 3)):
        for gi, gamma in enumerate(gamma_sets):
            smooth = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
            for ctype, channels in ((2, 3), (6, 4)):
                px = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
                if gi % 2:
                    px[..., 0] = smooth
                    px[..., 1] = smooth // 2 + px[..., 1] // 8
                if gi % 3 == 0:
                    px[::2, ::3, :3] = px[::2, ::3, :1]      # r = g = b pixels take the other branch
                cases.append((w, h, pc.write_png(px.reshape(h, -1), w, h, 8, ctype, filters=rng.integers(0, 5, h), extra_before=gamma,
                                                 level=int(rng.choice([1, 6])))))
                if gi < 4 or gi == 8:  # 16-bit colour: without a gamma that matters
                    px16 = rng.integers(0, 65536, (h, w, channels)).astype(">u2")
                    cases.append((w, h, pc.write_png(px16.view(np.uint8).reshape(h, -1), w, h, 16, ctype, filters=rng.integers(0, 5, h),
                                                     extra_before=gamma if gi != 1 else [])))
            for depth in (1, 2, 4, 8):
                entries = int(rng.integers(1, (1 << depth) + 1))
                pal = rng.integers(0, 256, (entries, 3), dtype=np.uint8)
                if entries > 2:
                    pal[1] = pal[1, 0]
                idx = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)   # (some indices lie behind the palette's end)
                extra = list(gamma) + [pc.chunk(b"PLTE", pal.tobytes())] + ([pc.chunk(b"tRNS", bytes(rng.integers(0, 256, entries, dtype=np.uint8)))] if gi % 2 else [])
                cases.append((w, h, pc.write_png(pc.pack_samples(idx, depth), w, h, depth, 3, filters=rng.integers(0, 5, h), extra_before=extra)))
    by_size = {}
    for w, h, f in cases:
        by_size.setdefault((w, h), []).append(f)
    checked = 0
    for (w, h), files in by_size.items():
        refs = [png_ref.imdecode_gray(f, w, h) for f in files]
        assert all(r[0] == 0 for r in refs), "libpng refuses a file this test wrote"
        with capi.Context(capi.default_params(max(w, 64), max(h, 64), max_images=2, nfeatures=100)) as c:
            st, got, sync = decode(capi, c, files, w, h)
            assert st == capi.VSF_OK and sync == capi.VSF_OK, (w, h, st, sync)
            for i, r in enumerate(refs):
                np.testing.assert_array_equal(got[i], r[1], err_msg="%dx%d file %d" % (w, h, i))
                checked += 1
    assert checked == len(cases) and checked > 300
    # left out on purpose: refused, not guessed
    w, h = 96, 64
    px = rng.integers(0, 256, (h, w * 3), dtype=np.uint8)
    px16 = rng.integers(0, 256, (h, w * 6), dtype=np.uint8)
    g = pc.chunk(b"gAMA", struct.pack(">I", 45455))
    with capi.Context(capi.default_params(w, h, max_images=2, nfeatures=100)) as c:
        def status(f):
            return c.png_decode_gray_batch([f], w, h, torch.zeros((h, w), dtype=torch.uint8, device="cuda").data_ptr(), h * w, w,
                                           allow_status=(capi.VSF_ERR_INVALID_ARG, capi.VSF_ERR_UNSUPPORTED))
        for extra in ([g, g], [pc.chunk(b"cHRM", bytes(32)), g], [pc.chunk(b"sRGB", b"\x00"), pc.chunk(b"cHRM", bytes(32))],
                      [pc.chunk(b"gAMA", struct.pack(">I", 5))], [pc.chunk(b"sRGB", b"\x09")]):
            assert status(pc.write_png(px, w, h, 8, 2, extra_before=extra)) == capi.VSF_ERR_UNSUPPORTED
        assert status(pc.write_png(px16, w, h, 16, 2, extra_before=[g])) == capi.VSF_ERR_UNSUPPORTED
        assert status(pc.write_png(px16, w, h, 16, 2, extra_before=[pc.chunk(b"gAMA", struct.pack(">I", 100000))])) == capi.VSF_OK
        assert c.sync() == capi.VSF_OK


def test_interlaced_files(capi):
    """Adam7: seven passes, each filtered as an image of its own, for every colour type and bit depth and sizes on which passes
    are empty (1 x 1 ... 9 x 9) -- against the real libpng (png_set_interlace_handling + png_read_image)."""
    import struct
    import png_ref
    if not png_ref.available():
        pytest.skip("no libpng16.so.16 to build tests/cpp/png_ref.c against")
    rng = np.random.Generator(np.random.PCG64(7))
    by_size = {}
    for w, h in [(a, b) for a in range(1, 10) for b in (1, 2, 3, 5, 8, 9)] + [(96, 64), (67, 41), (640, 480), (5, 300), (300, 3)]:
        kinds = [(8, 0), (16, 0), (1, 0), (2, 0), (4, 0), (8, 4), (16, 4), (8, 2), (8, 6), (16, 2), (16, 6), (8, 3), (4, 3), (2, 3), (1, 3)]
        if w * h > 100000:
            kinds = [(8, 0), (8, 2), (4, 3)]
        for depth, ctype in kinds:
            channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
            shape = (h, w) + ((channels,) if channels > 1 else ())
            samples = rng.integers(0, 1 << depth, shape)
            extra = []
            if ctype == 3:
                extra = [pc.chunk(b"PLTE", bytes(rng.integers(0, 256, 3 * int(rng.integers(1, (1 << depth) + 1)), dtype=np.uint8)))]
            elif ctype in (2, 6) and depth == 8 and rng.random() < 0.5:
                extra = [pc.chunk(b"gAMA", struct.pack(">I", 45455))]
            f = pc.write_png_adam7(samples, depth, ctype, rng, level=int(rng.choice([1, 6])), extra_before=extra,
                                   idat_piece=[None, 100, 8192][int(rng.integers(3))])
            by_size.setdefault((w, h), []).append(f)
    n = 0
    for (w, h), files in by_size.items():
        refs = [png_ref.imdecode_gray(f, w, h) for f in files]
        assert all(r[0] == 0 and r[2][2] == 1 for r in refs), (w, h)
        with capi.Context(capi.default_params(max(w, 64), max(h, 64), max_images=2, nfeatures=100)) as c:
            st, got, sync = decode(capi, c, files, w, h)
            assert st == capi.VSF_OK and sync == capi.VSF_OK, (w, h, st, sync)
            for i, r in enumerate(refs):
                np.testing.assert_array_equal(got[i], r[1], err_msg="%dx%d file %d" % (w, h, i))
                n += 1
    assert n > 800
