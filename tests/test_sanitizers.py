"""the CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer.
  * the oracle (oracle/vsf_oracle.cc, vsf_oracle_jpeg.cc; `make -C oracle asan`): its known-answer, definition-level and
    JPEG tests re-run in a child pytest with the sanitizer runtime preloaded and VSF_ORACLE_LIB pointing at the
    instrumented build;
  * the host-side restatements that ship in the product and are plain C++: the order-exact selection / sort header
    (csrc/vsf_select.h through tests/cpp/test_select.cc, 38 655 cases incl. adversarial inputs) and the ROS-1 wire encoder
    (host/slam_to_ros.h through tests/cpp/test_ros_wire.cc), compiled with -fsanitize=address,undefined and run;
  * the JPEG marker / table parser that reads untrusted bytes has its own fuzz test (tests/test_jpeg_host_asan.py).
The GPU kernels themselves cannot run under a sanitizer on this pool (no GPU ASan / XNACK)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _runtimes():
    rt = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        if not p or not Path(p).exists():
            pytest.skip("no %s in this toolchain" % name)
        rt.append(p)
    return rt


def test_oracle_known_answer_tests_under_asan_ubsan():
    rt = _runtimes()
    r = subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    lib = ROOT / "oracle" / "libvsf_oracle_asan.so"
    assert r.returncode == 0 and lib.exists(), r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=" ".join(rt), VSF_ORACLE_LIB=str(lib),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["tests/test_oracle_kat.py", "tests/test_jpeg_oracle.py", "tests/test_oracle_points.py",
             "tests/test_oracle_definitions.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests,
                       cwd=str(ROOT), capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert " passed" in p.stdout and "failed" not in p.stdout


@pytest.mark.parametrize("src,extra", [("test_select.cc", []), ("test_ros_wire.cc", [])])
def test_host_cpp_under_asan_ubsan(tmp_path, src, extra):
    _runtimes()
    exe = tmp_path / src.replace(".cc", "")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-o", str(exe), str(ROOT / "tests" / "cpp" / src)] + extra)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
