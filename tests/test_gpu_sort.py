"""The order-exact std::sort of Frontend::GetFeatureMatches (slam_frontend.cc:289-291) on the device, data-parallel
(k_frontend.hip sort_trim_par_kernel: libstdc++'s introsort restated on a workgroup) and one-lane, against the host
libstdc++'s std::sort (the oracle's vsfo_sort_and_trim) on inputs chosen to reach every part of the algorithm:
tie-heavy keys (257 possible distances), sorted / reversed / organ-pipe runs, constant arrays, median-of-3 killer
sequences (they exhaust the depth limit: heap-sort fallback), sizes around 16 (the insertion-sort threshold), 64 / 256
(the single-wave passes) and beyond (workgroup passes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _median3_killer(n):
    """Musser's median-of-3 killer permutation (even n): drives introsort into its heap-sort fallback."""
    n -= n % 2
    k = n // 2
    a = np.zeros(n, np.int64)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def _cases(rng, n):
    yield "uniform", rng.integers(0, 257, n)
    yield "few values", rng.integers(20, 24, n)
    yield "constant", np.full(n, 77)
    yield "sorted", np.sort(rng.integers(0, 257, n))
    yield "reversed", np.sort(rng.integers(0, 257, n))[::-1]
    half = np.sort(rng.integers(0, 257, (n + 1) // 2))
    yield "organ pipe", np.concatenate([half, half[::-1]])[:n]
    yield "distinct shuffled", rng.permutation(n) % 65000
    yield "distinct sorted", np.arange(n)
    yield "median-of-3 killer", np.resize(_median3_killer(n + 1), n) if n >= 2 else np.zeros(n, np.int64)
    yield "sawtooth", np.arange(n) % 17


@pytest.fixture(scope="module")
def ctx():
    from vision_slam_frontend_amd import capi
    capi.lib()
    c = capi.Context(capi.default_params(640, 480, max_images=2, nfeatures=9000))  # rows up to 9256
    yield c
    c.close()


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 33, 63, 64, 65, 200, 255, 256, 257, 300, 511, 777, 1024, 2500, 8000, 9256])
def test_parallel_sort_equals_std_sort(ctx, oracle, n):
    rng = np.random.Generator(np.random.PCG64(1000 + n))
    names, lists = [], []
    for name, d in _cases(rng, n):
        m = np.zeros(n, oracle.DMATCH_DTYPE)
        m["distance"] = np.asarray(d[:n], np.float32)
        m["queryIdx"] = np.arange(n)
        m["trainIdx"] = (np.arange(n) * 7 + 3) % 60000
        names.append(name)
        lists.append(m)
    batch = np.stack(lists) if n else np.zeros((len(lists), 0), oracle.DMATCH_DTYPE)
    for bp in (1.0, 0.3):
        want = [oracle.sort_and_trim(m, bp) for m in lists]
        for serial in (False, True):
            got = ctx.debug_sort_trim(batch, best_percent=bp, serial=serial)
            for name, g, w in zip(names, got, want):
                assert len(g) == len(w), (name, n, bp, serial)
                np.testing.assert_array_equal(g[:, 0], w["queryIdx"], err_msg="%s n=%d bp=%g serial=%s" % (name, n, bp, serial))
                np.testing.assert_array_equal(g[:, 1], w["trainIdx"])
