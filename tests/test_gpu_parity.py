"""GPU parity: every HIP kernel, through the C ABI (libvsf_hip.so), against the CPU oracle on the same seeded
synthetic inputs.  Bit-exact for everything (integer / byte / index work; the few float outputs -- Harris
response, angle, scaled coordinates -- are compared as raw bits as well)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="module")
def ctx640(capi):
    p = capi.default_params(640, 480, max_images=4, nfeatures=2000)
    c = capi.Context(p)
    yield c
    c.close()


@pytest.fixture(scope="module")
def run640(ctx640, oracle, stereo640):
    left, right = stereo640
    o = oracle.Orb(nfeatures=2000)
    o.run(left)
    kp, desc = ctx640.extract(left)
    return o, kp, desc


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_geometry_matches_oracle(ctx640, run640):
    o = run640[0]
    for l in range(50):
        assert ctx640.level_info(l) == o.level_info(l)


def test_pyramid_levels_bit_exact(ctx640, run640):
    o = run640[0]
    for l in range(50):
        np.testing.assert_array_equal(ctx640.debug_level_image(0, l, False), o.level_image(l, False), err_msg="level %d" % l)


def test_blurred_levels_bit_exact(ctx640, run640):
    o = run640[0]
    for l in range(50):
        np.testing.assert_array_equal(ctx640.debug_level_image(0, l, True), o.level_image(l, True), err_msg="level %d" % l)


def test_fast_candidates_raster_order(ctx640, run640):
    o = run640[0]
    total = 0
    for l in range(50):
        g = ctx640.debug_fast_candidates(0, l)
        r = o.stage(0, l)
        assert len(g) == len(r), "level %d" % l
        for f in ("x", "y", "response"):
            np.testing.assert_array_equal(g[f], r[f], err_msg="level %d %s" % (l, f))
        total += len(r)
    assert total > 10000


def test_level_keypoints_order_response_angle(ctx640, run640):
    o = run640[0]
    for l in range(50):
        g = ctx640.debug_level_keypoints(0, l)
        r = o.stage(4, l)
        assert len(g) == len(r), "level %d" % l
        np.testing.assert_array_equal(g["x"], r["x"], err_msg="level %d x" % l)
        np.testing.assert_array_equal(g["y"], r["y"], err_msg="level %d y" % l)
        np.testing.assert_array_equal(_bits(g["response"]), _bits(r["response"]), err_msg="level %d harris" % l)
        np.testing.assert_array_equal(_bits(g["angle"]), _bits(r["angle"]), err_msg="level %d angle" % l)


def test_extract_keypoints_and_descriptors_bit_exact(run640):
    o, kp, desc = run640
    rk, rd = o.result()
    assert len(kp) == len(rk) == 2000
    assert kp.tobytes() == rk.tobytes()
    np.testing.assert_array_equal(desc, rd)


def test_knn2_and_matches(ctx640, oracle, stereo640, run640):
    _, _, desc_l = run640
    _, desc_r = ctx640.extract(stereo640[1])
    gi, gd = ctx640.knn2_hamming(desc_l, desc_r)
    ri, rd = oracle.knn2_hamming(desc_l, desc_r)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gd, rd)
    gm = ctx640.get_matches(desc_l, desc_r)
    rm = oracle.get_matches(desc_l, desc_r)
    assert len(rm) > 50
    assert gm.tobytes() == rm.tobytes()


def test_extract_pair_and_matches_multi_equal_single_calls(ctx640, oracle, stereo640, run640):
    """The batched host-pointer entry points (one stereo frame's two extractions; one train set against many query
    sets) return exactly what the one-at-a-time calls return."""
    from vision_slam_frontend_amd import synth
    _, kp_l, desc_l = run640
    kp_r, desc_r = ctx640.extract(stereo640[1])
    (pk_l, pd_l), (pk_r, pd_r) = ctx640.extract_pair(stereo640[0], stereo640[1])
    assert pk_l.tobytes() == kp_l.tobytes() and pk_r.tobytes() == kp_r.tobytes()
    np.testing.assert_array_equal(pd_l, desc_l)
    np.testing.assert_array_equal(pd_r, desc_r)
    q_sets = [desc_l, synth.random_descriptors(700, seed=3), desc_r[:5], desc_r[:0], synth.adversarial_descriptors(900, seed=4)]
    got = ctx640.get_matches_multi(q_sets, desc_r)
    assert len(got) == len(q_sets)
    for q, g in zip(q_sets, got):
        assert g.tobytes() == oracle.get_matches(q, desc_r).tobytes()
    assert len(got[0]) > 50 and len(got[3]) == 0


def test_knn2_tie_rule_adversarial(ctx640, oracle):
    from vision_slam_frontend_amd import synth
    q = synth.adversarial_descriptors(700, seed=1)
    t = synth.adversarial_descriptors(900, seed=2)
    gi, gd = ctx640.knn2_hamming(q, t)
    ri, rd = oracle.knn2_hamming(q, t)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gd, rd)
    assert ctx640.get_matches(q, t).tobytes() == oracle.get_matches(q, t).tobytes()


# sizes around the matcher's tiles: 32 train rows per MFMA tile, 64 queries per wave / 256 per workgroup, 128-row split
# chunks, 8192 train rows per 13-bit key range (k_match.hip)
@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (5, 1), (1, 2), (3, 3), (257, 255), (256, 513), (64, 31), (64, 32),
                                   (65, 33), (63, 64), (129, 65), (33, 127), (33, 129), (10, 8191), (10, 8192),
                                   (70, 8193), (300, 20000)])
def test_matcher_edge_sizes(ctx640, oracle, nq, nt):
    from vision_slam_frontend_amd import synth
    q = synth.random_descriptors(max(nq, 1), seed=10)[:nq]
    t = synth.random_descriptors(max(nt, 1), seed=11)[:nt]
    gi, gd = ctx640.knn2_hamming(q, t)
    ri, rd = oracle.knn2_hamming(q, t)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gd, rd)
    assert ctx640.get_matches(q, t).tobytes() == oracle.get_matches(q, t).tobytes()


def test_knn2_ties_across_key_ranges(ctx640, oracle):
    """Equal distances in different 32-row tiles, split chunks and 8192-row key ranges resolve to the lower train index."""
    from vision_slam_frontend_amd import synth
    q = synth.adversarial_descriptors(300, seed=3)
    t = np.concatenate([synth.adversarial_descriptors(4500, seed=4)] * 4)[:17000]  # every train row occurs 3-4 times
    gi, gd = ctx640.knn2_hamming(q, t)
    ri, rd = oracle.knn2_hamming(q, t)
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gd, rd)


def test_fast_detect_standalone(ctx640, oracle, stereo640):
    left = stereo640[0]
    for thr in (10, 20, 40):
        g = ctx640.fast_detect(left, thr, True, cap=1 << 17)
        r = oracle.fast9_16(left, thr, True)
        assert len(r) > 100
        assert g.tobytes() == r.tobytes(), "threshold %d" % thr


def test_fast_detect_without_nms(ctx640, oracle, stereo640):
    left = stereo640[0]
    g = ctx640.fast_detect(left, 30, False, cap=1 << 18)
    r = oracle.fast9_16(left, 30, False)
    assert g.tobytes() == r.tobytes()


def _selection_inputs():
    rng = np.random.default_rng(99)
    cases = []
    for n in (1, 2, 3, 5, 48, 49, 50, 100, 257, 1000, 4096, 4097, 8525, 20000, 70000):
        cases.append(("uniform", rng.uniform(-1, 1, n).astype(np.float32)))
        cases.append(("ties", rng.integers(20, 60, n).astype(np.float32)))
        cases.append(("heavy_ties", rng.integers(20, 23, n).astype(np.float32)))
    cases.append(("sorted", np.arange(5000, dtype=np.float32)))
    cases.append(("reverse", np.arange(5000, dtype=np.float32)[::-1].copy()))
    cases.append(("equal", np.full(3000, 7, np.float32)))
    cases.append(("organ", np.minimum(np.arange(6000), 5999 - np.arange(6000)).astype(np.float32)))
    return cases


@pytest.mark.parametrize("use_lds", [False, True])
def test_parallel_retain_best_matches_libstdcxx(ctx640, oracle, use_lds):
    """The GPU's parallel Hoare passes must leave exactly libstdc++'s nth_element + partition permutation."""
    checked = 0
    for name, keys in _selection_inputs():
        n = len(keys)
        for k in sorted({0, 1, 2, n // 50, n // 7, n // 2, n - 1, n, n + 5}):
            if k < 0:
                continue
            rk, rid = oracle.retain_best(keys, k)
            gk, gid = ctx640.debug_retain_best(keys, k, use_lds=use_lds, mode=0)
            assert len(gid) == len(rid), (name, n, k)
            np.testing.assert_array_equal(gid, rid, err_msg="%s n=%d k=%d" % (name, n, k))
            np.testing.assert_array_equal(gk, rk.view(np.uint32))
            checked += 1
    assert checked > 200


def test_parallel_retain_best_packed_scores(ctx640, oracle):
    rng = np.random.default_rng(5)
    for n in (300, 5000, 30000):
        score = rng.integers(20, 90, n)
        packed = ((score.astype(np.uint32) << 24) | rng.integers(0, 1 << 24, n).astype(np.uint32))
        for k in (10, n // 20, n // 3):
            rk, rid = oracle.retain_best(score.astype(np.float32), k)
            gk, gid = ctx640.debug_retain_best(packed.view(np.float32), k, mode=1)
            np.testing.assert_array_equal(gid, rid, err_msg="n=%d k=%d" % (n, k))


@pytest.mark.parametrize("size", [(97, 61), (333, 217), (752, 480), (768, 432), (800, 600), (64, 64)])
def test_pyramid_of_small_batches_any_size(capi, oracle, size):
    """A batch of up to 16 images builds its pyramid with pyramid_slab_kernel (chains of levels per launch, the last
    level cut into slabs that recompute their border rows; widths above 768 keep the per-level launches): every level of
    every image, for the reference's 50-level / 1.04 pyramid and for a classic 8-level / 1.2 one, against the oracle."""
    import torch
    from vision_slam_frontend_amd import synth
    w, h = size
    dev = torch.device("cuda", 0)
    for kw in (dict(), dict(scale_factor=1.2, nlevels=8)):
        p = capi.default_params(w, h, max_images=9, nfeatures=300)
        for k, v in kw.items():
            setattr(p, k, v)
        with capi.Context(p) as ctx:
            K = ctx.params.max_keypoints
            for n in (1, 2, 3, 4, 9):
                imgs = np.stack([synth.stereo_pair(w, h, 300 + 7 * n + i, n_objects=40)[i & 1] for i in range(n)])
                pitch = (w + 15) // 16 * 16  # (device-pointer calls take rows at a multiple of four bytes)
                padded = np.zeros((n, h, pitch), np.uint8)
                padded[:, :, :w] = imgs
                d = torch.from_numpy(padded).to(dev)
                kp = torch.zeros((n, K, 28), dtype=torch.uint8, device=dev)
                de = torch.zeros((n, K, 32), dtype=torch.uint8, device=dev)
                cn = torch.zeros(n, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                ctx.extract_batch_dev(d.data_ptr(), n, pitch * h, pitch, kp.data_ptr(), de.data_ptr(), cn.data_ptr())
                ctx.sync(allow_capacity=True)
                for i in range(n):
                    o = oracle.Orb(nfeatures=300, **kw)
                    o.run(imgs[i])
                    for l in range(ctx.nlevels):
                        np.testing.assert_array_equal(ctx.debug_level_image(i, l, False), o.level_image(l, False),
                                                      err_msg="%dx%d n=%d image %d level %d %s" % (w, h, n, i, l, kw))


def test_knn2_extreme_distances(oracle):
    """The matcher's key comes out of a matrix instruction (FP4: 8192 (d + 1) + row as an exact float): the ends of the
    distance range -- 0 (a row against itself), 256 (against its complement),
    every train row at 256 (the second best is then 256 too), and rows that differ in one bit only -- and 10 300 train rows
    (three 4096-row key chunks, a last tile of 12 rows) must give the oracle's (index, distance) pairs."""
    from vision_slam_frontend_amd import capi, synth
    t = synth.random_descriptors(10300, seed=5)
    q = np.concatenate([t[[0, 31, 32, 4095, 4096, 8191, 8192, 10299]],          # distance 0, second best random
                        ~t[[5, 4097]],                                           # distance 256 to one row
                        synth.random_descriptors(40, seed=6)])
    one_bit = t[[100, 5000]].copy()
    one_bit[0, 0] ^= 1
    one_bit[1, 31] ^= 0x80
    q = np.ascontiguousarray(np.concatenate([q, one_bit]))
    all_far_t = np.ascontiguousarray(np.repeat(~q[:1], 70, axis=0))           # every train row at distance 256 from q[0]
    want = oracle.knn2_hamming(q, t)
    want_far = oracle.knn2_hamming(q[:1], all_far_t)
    assert want[1][0, 0] == 0 and want[1][8, 0] < 256 and want_far[1].tolist() == [[256, 256]] and want_far[0].tolist() == [[0, 1]]
    with capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=2000)) as ctx:
        gi, gd = ctx.knn2_hamming(q, t)
        np.testing.assert_array_equal(gi, want[0])
        np.testing.assert_array_equal(gd, want[1])
        fi, fd = ctx.knn2_hamming(q[:1], all_far_t)
        np.testing.assert_array_equal(fi, want_far[0])
        np.testing.assert_array_equal(fd, want_far[1])
        assert ctx.get_matches(q, t).tobytes() == oracle.get_matches(q, t).tobytes()
