"""The batched matcher's launch forms (k_match.hip): a launch of many pairs runs unsplit (one workgroup per 128 queries walks
the whole train set), a launch of few pairs splits the train sets over workgroups and merges packed keys.  Same pairs through
both and through the oracle's batchDistance restatement: indices, distances and ratio-test matches byte for byte -- on ragged
sets (empty, one row, tile edges +-1, beyond one 4096-row key range) full of duplicates and near-duplicates, where only the
rule "smaller distance first, ties to the lower train index" (core/stat.cpp batchDistance, reached from
slam_frontend.cc:525-527) decides.  (Written for round 6's two-query-tiles-per-wave kernel, tools/exp/match_wide.patch,
which passed it and was not faster.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def capi():
    from vision_slam_frontend_amd import capi
    capi.lib()
    return capi


def _sets(K, counts, seed):
    from vision_slam_frontend_amd import synth
    rng = np.random.default_rng(seed)
    desc = np.zeros((len(counts), K, 32), np.uint8)
    for i, n in enumerate(counts):
        if n == 0:
            continue
        if i % 3 == 0:
            d = synth.adversarial_descriptors(n, seed=seed + i, n_unique=max(5, n // 40))
        else:
            d = synth.random_descriptors(n, seed=seed + i)
            dup = rng.integers(0, n, max(1, n // 8))  # copies of rows of the same set and of its neighbour
            d[dup] = d[rng.integers(0, n, len(dup))]
            if i > 0 and counts[i - 1] > 0:
                take = rng.integers(0, counts[i - 1], max(1, n // 6))
                d[rng.integers(0, n, len(take))] = desc[i - 1, take]
        desc[i, :n] = d
    return desc


@pytest.mark.parametrize("nf,counts,n_pairs,group", [
    (2000, [2000, 1999, 0, 1, 31, 32, 33, 127, 128, 129, 255, 257, 1000, 1024, 1500, 1984, 2000, 2000, 640, 64], 144, 6),
    (6000, [6000, 4096, 4097, 4095, 5000, 0, 1, 33, 4128, 6000], 40, 3),
])
def test_unsplit_matcher_equals_split_matcher_and_oracle(capi, oracle, nf, counts, n_pairs, group):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(nf + n_pairs)
    n_sets = len(counts)
    q = rng.integers(0, n_sets, n_pairs).astype(np.int32)
    t = rng.integers(0, n_sets, n_pairs).astype(np.int32)
    q[:n_sets], t[:n_sets] = np.arange(n_sets), (np.arange(n_sets) + 1) % n_sets  # every set is a query set once
    q[n_sets:2 * n_sets], t[n_sets:2 * n_sets] = np.arange(n_sets), np.arange(n_sets)  # ... and once matches itself
    p = capi.default_params(640, 480, max_images=2, nfeatures=nf)
    with capi.Context(p) as ctx:
        K = ctx.params.max_keypoints  # the sets' capacity: nfeatures and some room
        assert K >= max(counts)
        desc = _sets(K, counts, seed=nf)
        d_desc = torch.from_numpy(desc).to(dev)
        d_counts = torch.from_numpy(np.asarray(counts, np.int32)).to(dev)
        d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)

        def outputs():
            return (torch.full((n_pairs, K, 2), -7, dtype=torch.int32, device=dev),
                    torch.full((n_pairs, K, 2), -7, dtype=torch.int32, device=dev),
                    torch.zeros((n_pairs, K, 16), dtype=torch.uint8, device=dev),
                    torch.full((n_pairs,), -7, dtype=torch.int32, device=dev))

        wide, narrow = outputs(), outputs()
        torch.cuda.synchronize()
        # one launch of all pairs: qtiles x pairs fills the chip -> unsplit
        ctx.match_batch_dev(d_desc.data_ptr(), d_counts.data_ptr(), K * 32, d_q.data_ptr(), d_t.data_ptr(), n_pairs,
                            *(x.data_ptr() for x in wide))
        # the same pairs a few at a time: small launches -> train sets split over workgroups, merged by 64-bit CAS
        for g0 in range(0, n_pairs, group):
            n = min(group, n_pairs - g0)
            ctx.match_batch_dev(d_desc.data_ptr(), d_counts.data_ptr(), K * 32, d_q.data_ptr() + 4 * g0,
                                d_t.data_ptr() + 4 * g0, n, narrow[0].data_ptr() + g0 * K * 8,
                                narrow[1].data_ptr() + g0 * K * 8, narrow[2].data_ptr() + g0 * K * 16,
                                narrow[3].data_ptr() + g0 * 4)
        assert ctx.sync() == capi.VSF_OK
        wide = [x.cpu().numpy() for x in wide]
        narrow = [x.cpu().numpy() for x in narrow]
    total = 0
    for pr in range(n_pairs):
        nq = counts[q[pr]]
        np.testing.assert_array_equal(wide[0][pr, :nq], narrow[0][pr, :nq], err_msg="pair %d idx" % pr)
        np.testing.assert_array_equal(wide[1][pr, :nq], narrow[1][pr, :nq], err_msg="pair %d dist" % pr)
        assert (wide[0][pr, nq:] == -7).all() and (wide[1][pr, nq:] == -7).all(), "pair %d: rows past the set written" % pr
        assert wide[3][pr] == narrow[3][pr], "pair %d nmatches" % pr
        nm = int(wide[3][pr])
        assert wide[2][pr, :nm].tobytes() == narrow[2][pr, :nm].tobytes(), "pair %d matches" % pr
        total += nm
    assert total > 0
    # the oracle on a sample of pairs: the first 2 n_sets cover every set as query and the self-matches
    for pr in list(range(0, 2 * n_sets, 3)) + [n_pairs - 1]:
        qs, ts = desc[q[pr], :counts[q[pr]]], desc[t[pr], :counts[t[pr]]]
        if len(qs) == 0:
            assert wide[3][pr] == 0
            continue
        ri, rd = oracle.knn2_hamming(qs, ts)
        np.testing.assert_array_equal(wide[0][pr, :len(qs)], ri, err_msg="pair %d idx vs oracle" % pr)
        np.testing.assert_array_equal(wide[1][pr, :len(qs)], rd, err_msg="pair %d dist vs oracle" % pr)
        rm = oracle.get_matches(qs, ts)
        assert int(wide[3][pr]) == len(rm) and wide[2][pr, :len(rm)].tobytes() == rm.tobytes(), "pair %d matches vs oracle" % pr
