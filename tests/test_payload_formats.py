"""CPU-side checks of the two wire layouts the host decodes (include/vsf.h): the compact gather payload of
vsf_pack_outputs_dev and the result of vsf_observe_stereo, built here byte by byte from the header's description."""
import numpy as np
import pytest

from vision_slam_frontend_amd import capi


def _features(n, seed):
    rng = np.random.default_rng(seed)
    f = np.zeros(n, capi.VISION_FEATURE_DTYPE)
    f["feature_idx"] = np.arange(n)
    f["pixel"] = rng.uniform(0, 640, (n, 2))
    f["point3d"] = rng.normal(0, 3, (n, 3))
    return f


def _matches(n, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, capi.FEATURE_MATCH_DTYPE)
    m["feature_idx_initial"] = rng.integers(0, 2000, n)
    m["feature_idx_current"] = rng.integers(0, 2000, n)
    return m


def test_unpack_outputs_round_trip_and_errors():
    feats = [_features(n, n) for n in (5, 0, 17)]
    pairs = [_matches(n, 100 + n) for n in (3, 0, 0, 9)]
    body = b"".join(f.tobytes() for f in feats) + b"".join(m.tobytes() for m in pairs)
    counts = np.array([len(f) for f in feats] + [len(m) for m in pairs], np.uint32)
    total = 16 + 4 * len(counts) + len(body)
    hdr = np.array([capi.PAYLOAD_MAGIC, len(feats), len(pairs), total], np.uint32)
    raw = np.frombuffer(hdr.tobytes() + counts.tobytes() + body + b"\xAB" * 32, np.uint8)  # (slack after the payload)
    assert total == 16 + 4 * 7 + 28 * 22 + 16 * 12
    got_f, got_m = capi.unpack_outputs(raw)
    assert [x.tobytes() for x in got_f] == [x.tobytes() for x in feats]
    assert [x.tobytes() for x in got_m] == [x.tobytes() for x in pairs]
    with pytest.raises(ValueError):
        capi.unpack_outputs(raw[:total - 1])  # truncated
    bad = raw.copy()
    bad[0] ^= 1
    with pytest.raises(ValueError):
        capi.unpack_outputs(bad)


def test_decode_observation_layout():
    nfeat, lists = 6, [_matches(2, 1), _matches(0, 2), _matches(5, 3)]  # two kept frames + the right->left list
    feats = _features(nfeat, 9)
    kp = np.zeros(nfeat, capi.KEYPOINT_DTYPE)
    kp["x"], kp["octave"] = np.arange(nfeat), 3
    desc = np.random.default_rng(4).integers(0, 256, (nfeat, 32), dtype=np.uint8)
    npairs = np.zeros(4, np.uint32)  # padded to a multiple of 4 words
    npairs[:3] = [len(m) for m in lists]
    body = feats.tobytes() + b"".join(m.tobytes() for m in lists) + kp.tobytes() + desc.tobytes()
    total = 64 + 16 + len(body)
    hdr = np.zeros(16, np.uint32)
    hdr[:8] = [0x4F465356, 3, nfeat, total, 1990, 1985, 240, 5]
    hdr[8:11] = np.array([1.25, 3.25, 3.25], np.float32).view(np.uint32)
    raw = np.frombuffer(hdr.tobytes() + npairs.tobytes() + body, np.uint8)
    d = capi.decode_observation(raw)
    assert (d["n_left"], d["n_right"], d["n_stereo_matches"], d["n_points"]) == (1990, 1985, 240, 5)
    assert d["mean"] == np.float32(1.25) and d["threshold"] == d["threshold_next"] == np.float32(3.25)
    assert d["features"].tobytes() == feats.tobytes() and d["keypoints"].tobytes() == kp.tobytes()
    assert np.array_equal(d["descriptors"], desc)
    assert [m.tobytes() for m in d["factors"]] == [m.tobytes() for m in lists[:2]]
    assert d["stereo_pairs"].tobytes() == lists[2].tobytes()
