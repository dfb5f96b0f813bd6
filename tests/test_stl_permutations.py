"""tools/pin/stl_permutations.cc: the permutations libstdc++ leaves in cv::KeyPointsFilter::retainBest and in the
std::sort of Frontend::GetFeatureMatches for 800 deterministic key sets.  Built here with the local compiler (the one the
oracle is built with): the program is deterministic and agrees with the in-repo restatement's host tests; and when the pin kit
has produced GCC 7's file (tests/golden/opencv/stl_permutations_gcc7.txt, tools/pin/Dockerfile), the two must be equal --
which settles the GCC 7-vs-11 risk named in DESIGN.md section 2."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GCC7 = ROOT / "tests" / "golden" / "opencv" / "stl_permutations_gcc7.txt"


@pytest.fixture(scope="module")
def local_output(tmp_path_factory):
    exe = tmp_path_factory.mktemp("stl") / "stl_permutations"
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-o", str(exe), str(ROOT / "tools" / "pin" / "stl_permutations.cc")])
    a = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    b = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert a == b  # deterministic
    return a


def test_the_dump_is_well_formed(local_output):
    lines = [l for l in local_output.splitlines() if not l.startswith("#")]
    assert len(lines) == 800 and sum(l.startswith("retainBest ") for l in lines) == 400
    for l in lines:
        head, ids = l.split(":")
        kind, _case, n, keep = head.split()
        ids = [int(x) for x in ids.split()]
        assert len(set(ids)) == len(ids) and all(0 <= i < int(n) for i in ids)
        if kind == "sort":
            assert len(ids) == int(keep)
        else:
            assert len(ids) >= min(int(keep), int(n))  # ties with the boundary response are all kept


@pytest.mark.skipif(not GCC7.exists(), reason="tests/golden/opencv/stl_permutations_gcc7.txt absent: the pin kit "
                                               "(tools/pin/Dockerfile) has not been run")
def test_gcc7_libstdcxx_permutes_like_the_local_one(local_output):
    strip = lambda t: [l for l in t.splitlines() if not l.startswith("#")]  # noqa: E731
    assert strip(GCC7.read_text()) == strip(local_output)
