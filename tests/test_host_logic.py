"""CPU-side checks of the product's host logic: the order-exact selection restatement against libstdc++, the
C-ABI library (loads, exports every symbol include/vsf.h declares, parameter helpers), the synthetic generator."""
import ctypes
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_select_restatement_equals_libstdcxx(tmp_path):
    exe = tmp_path / "test_select"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "cpp" / "test_select.cc")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "failures=0" in out.stdout
    assert int(re.search(r"heap_fallbacks_on_killers=(\d+)", out.stdout).group(1)) > 0


@pytest.fixture(scope="module")
def capi():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g
    from vision_slam_frontend_amd import capi
    if not capi.LIB_PATH.exists():
        g.build()
    return capi


def test_describe_sincos_matches_libm(tmp_path):
    """k_describe.hip's sincos_2pi (restated for the host, same arithmetic) against libm's cos/sin rounded to float
    on a strided subset of all float angles in [0, 360]; stride 1 (every float, ~60 s) was run by hand: 0 mismatches
    over 1 135 869 953 values."""
    exe = tmp_path / "test_sincos"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe),
                           str(ROOT / "tests" / "cpp" / "test_sincos_exhaustive.c"), "-lm"])
    out = subprocess.run([str(exe), "257"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout


def test_ros_wire_format_known_answer(tmp_path):
    """host/slam_to_ros.h against bytes assembled field by field from msg/*.msg."""
    exe = tmp_path / "test_ros_wire"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "cpp" / "test_ros_wire.cc")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_library_exports_every_declared_symbol(capi):
    header = (ROOT / "include" / "vsf.h").read_text()
    declared = set(re.findall(r"\b(vsf_[a-z0-9_]+)\s*\(", header))
    declared -= {"vsf_status"}
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = ctypes.CDLL(str(capi.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_no_torch_types_and_no_oracle_in_product():
    # the boundary is a plain C ABI; the product never touches the oracle
    header = (ROOT / "include" / "vsf.h").read_text()
    assert "torch" not in header and "at::" not in header
    for path in (ROOT / "vision_slam_frontend_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".h", ".cc", ".cpp"}:
            text = path.read_text()
            assert "oracle" not in text.lower(), path


def test_params_default_and_ratio(capi):
    p = capi.default_params(640, 480, max_images=4)
    assert (p.nfeatures, p.nlevels, p.edge_threshold, p.first_level, p.wta_k, p.score_type, p.patch_size,
            p.fast_threshold) == (10000, 50, 31, 0, 2, 0, 31, 20)  # slam_frontend.cc:205-213
    assert np.float32(p.scale_factor) == np.float32(1.04)
    assert (p.fast_detector_threshold, p.fast_detector_nms) == (10, 1)  # slam_frontend.cc:191
    # 0.6f widened to double (slam_frontend.cc:555) == 10066330 / 2^24, stored in lowest terms
    assert p.ratio_num * 2 ** (24 - p.ratio_shift) == 10066330 and p.ratio_num / 2 ** p.ratio_shift == float(np.float32(0.6))
    q = capi.default_params(640, 480, nn_match_ratio=0.75)
    assert (q.ratio_num, q.ratio_shift) == (3, 2)
    assert capi.lib().vsf_params_set_ratio(ctypes.byref(q), ctypes.c_float(0.0)) == capi.VSF_ERR_INVALID_ARG
    assert capi.lib().vsf_status_string(capi.VSF_ERR_CAPACITY).decode().startswith("capacity")
    assert capi.lib().vsf_stage_name(1).decode() == "fast_score_nms"


def test_create_without_gpu_fails_loudly(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.VsfError):
        capi.Context(capi.default_params(640, 480))
    bad = capi.default_params(640, 480, wta_k=3)
    h = ctypes.c_void_p()
    assert capi.lib().vsf_create(ctypes.byref(bad), 0, ctypes.byref(h)) == capi.VSF_ERR_UNSUPPORTED


def test_synthetic_generator_is_deterministic():
    from vision_slam_frontend_amd import synth
    left, right = synth.stereo_pair(640, 480, 0)
    assert left.shape == (480, 640) and left.dtype == np.uint8 and left.flags.c_contiguous
    assert synth.sha256(left).startswith("0427f20208354c21") and synth.sha256(right).startswith("cb1a0507eddcfc6a")
    b = synth.bench_batch(10, 320, 240, n_scenes=4, n_objects=100)
    assert b.shape == (10, 2, 240, 320)
    assert len({synth.sha256(b[i, 0]) for i in range(10)}) == 10  # every frame is a distinct image
    d = synth.random_descriptors(100)
    assert d.shape == (100, 32) and synth.sha256(d) == synth.sha256(synth.random_descriptors(100))
    a = synth.adversarial_descriptors(500)
    assert len(np.unique(a, axis=0)) < 450  # many exact duplicates


def test_counter_files_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """profiles/traffic*.json and profiles/r04/valu_ceiling.json carry a digest of csrc's CODE (comments and white space do
    not count); bench.py publishes their numbers only while the sources it runs still hash to it (round-3 review: nothing
    tied the committed counters to the kernels that ran)."""
    import json
    import shutil
    import sys

    from vision_slam_frontend_amd import buildinfo
    h = buildinfo.kernel_source_hash()
    assert len(h) == 64
    # (whether the committed files are fresh is a fact about the last profiling run, not a test: a kernel edit without a
    # counter refresh must only make bench.py say so)
    committed = json.loads((ROOT / "profiles" / "traffic.json").read_text())["source_hash"]
    # a copy of csrc: a reworded comment keeps the digest, a changed token does not
    copy = tmp_path / "csrc"
    shutil.copytree(ROOT / "vision_slam_frontend_amd" / "csrc", copy, ignore=shutil.ignore_patterns("*.o", "*.inc"))
    monkeypatch.setattr(buildinfo, "CSRC", copy)
    assert buildinfo.kernel_source_hash() == h
    f = copy / "k_fast.hip"
    f.write_text("// a new remark\n" + f.read_text().replace("FAST-9/16 segment test", "FAST 9 of 16"))
    assert buildinfo.kernel_source_hash() == h
    f.write_text(f.read_text().replace("constexpr int SR = VSF_FAST_STRIP_ROWS;", "constexpr int SR = VSF_FAST_STRIP_ROWS + 0;"))
    assert buildinfo.kernel_source_hash() != h
    sys.argv = ["bench.py"]
    sys.path.insert(0, str(ROOT))
    import bench
    stages, stale = bench.committed_counters(640, 480, 2000, 256)
    assert stale and stages  # (bench.py then prints roofline.traffic: null and traffic_stale: true)
    monkeypatch.setattr(buildinfo, "CSRC", ROOT / "vision_slam_frontend_amd" / "csrc")
    stages, stale = bench.committed_counters(640, 480, 2000, 256)
    assert stale == (committed != h) and stages["fast_score_nms"]["valu_wave_insts_per_step"] > 1e9


def test_comm_entry_points_check_their_arguments(capi):
    """The multi-GPU exchange of include/vsf.h without a GPU: every entry point refuses null / out-of-range arguments with a
    status (never a crash), and vsf_comm_destroy(NULL) is a no-op."""
    L = capi.lib()
    h = ctypes.c_void_p()
    buf = (ctypes.c_uint8 * 128)()
    assert L.vsf_comm_unique_id(None) == capi.VSF_ERR_INVALID_ARG
    assert L.vsf_comm_create(None, buf, 0, 1, ctypes.byref(h)) == capi.VSF_ERR_INVALID_ARG
    assert L.vsf_comm_info(None, None, None, None) == capi.VSF_ERR_INVALID_ARG
    assert L.vsf_allgather_dev(None, None, None, None, 4) == capi.VSF_ERR_INVALID_ARG
    assert L.vsf_gather_payload_dev(None, None, None, 16, None, 16, 0) == capi.VSF_ERR_INVALID_ARG
    L.vsf_comm_destroy(None)
    assert L.vsf_set_option(None, 0, 0) == capi.VSF_ERR_INVALID_ARG and L.vsf_get_option(None, 0, None) == capi.VSF_ERR_INVALID_ARG
    g, r = ctypes.c_float(), ctypes.c_float()
    assert L.vsf_tune_fast_resident(None, None, 64, 0, 0, None, None, None, 3, ctypes.byref(g), ctypes.byref(r)) == capi.VSF_ERR_INVALID_ARG
    assert L.vsf_debug_inject_hip_error(None, 1) == capi.VSF_ERR_INVALID_ARG
