"""SURVEY section 5 (sanitizers) for the one place where the product parses UNTRUSTED bytes on the host: the JPEG marker /
table parser and upload planner (csrc/vsf_jpeg_host.cc; reference counterpart: cv::imdecode at slam_frontend_main.cc:98-100).
`make -C vision_slam_frontend_amd/csrc asan` builds that translation unit alone with -fsanitize=address,undefined (plain
g++, no GPU code); a child process preloads the sanitizer runtime, runs the 24 fixture files (baseline and progressive), hand-made files whose DHT
counts oversubscribe the code space and 2000 seeded mutations (bit flips, truncation, stray markers, header damage, garbage,
DHT counts rewritten with consistent lengths) through vsf_jpeg_plan + vsf_jpeg_fill, and must exit
cleanly: any out-of-bounds access, overflow or misaligned access aborts it with a report.  No GPU involved."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "vision_slam_frontend_amd" / "csrc"
LIB = ROOT / "vision_slam_frontend_amd" / "libvsf_jpeg_host_asan.so"
GOLD = Path(__file__).resolve().parent / "golden" / "jpeg"

DRIVER = r'''
import ctypes as C, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, sys.argv[3])
from jpeg_mutate import mutate, rewrite_dht, drop_last_scans
import jpeg_craft
lib = C.CDLL(sys.argv[1])
lib.vsf_jpeg_host_check.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
gold = Path(sys.argv[2])
expected = np.load(gold / "expected_gray.npz")
files = {p.stem: p.read_bytes() for p in sorted(gold.glob("*.jpg"))}

def check(batch, w, h, serial=0):
    n = len(batch)
    ptrs = (C.c_char_p * n)(*batch)
    sizes = (C.c_size_t * n)(*[len(b) for b in batch])
    total, csum = C.c_uint64(), C.c_uint32()
    return lib.vsf_jpeg_host_check(ptrs, sizes, n, w, h, serial, C.byref(total), C.byref(csum)), total.value

ok = bad = 0
for name, data in files.items():
    h, w = expected[name].shape
    if name.startswith("prog"):  # without its last scan the file stops short of full precision: refused as unsupported
        st, _ = check([drop_last_scans(data, 1)], w, h)
        assert st == 4, ("an incomplete progression must be refused as unsupported", name, st)
    for serial in (0, 1):
        st, total = check([data], w, h, serial)
        assert st == 0 and total > len(data) // 2, (name, st, total)
    st, _ = check([data], w + 1, h)
    assert st == 1, ("wrong size must be refused", name, st)
# regression (round-3 review): a DHT whose counts oversubscribe the code space -- 200 codes of length 1 -- with CONSISTENT
# segment lengths used to index ~100 KB past the lookup tables in build_dev_huff; it must be refused (as libjpeg does:
# JERR_BAD_HUFF_TABLE), for the DC and for the AC table, for every length the first-level table covers and beyond
rng = np.random.Generator(np.random.PCG64(7))
small = jpeg_craft.write_gray_jpeg(jpeg_craft.random_coefficients(rng, 1, 1), [16] * 64,
                                   (jpeg_craft.STD_DC_BITS, jpeg_craft.STD_DC_VALS), jpeg_craft.flat_ac_table())
assert check([small], 8, 8)[0] == 0
for which in (0, 1):
    for length in range(1, 17):
        for count in (200, 255, (1 << min(length, 7)) + 1):
            counts = [0] * 16
            counts[length - 1] = count
            st, _ = check([rewrite_dht(small, rng, counts, which, max_value=15 if which == 0 else 255)], 8, 8)
            assert st == (1 if count >= (1 << length) else 0), (which, length, count, st)
full = [0] * 16
full[7] = 255                       # 255 codes of length 8 leave the all-ones code free: a legal table
assert check([rewrite_dht(small, rng, full, 0, max_value=15)], 8, 8)[0] == 0
# round-4 advice: a DC table whose symbols exceed 15 (libjpeg: JERR_BAD_HUFF_TABLE when the scan sets it up) is refused --
# the same counts, one value raised; the AC table may carry any byte
def with_dc_value(data, value, which=0):
    pos = data.find(b"\xFF\xC4")
    assert pos > 0 and data[pos + 4] >> 4 == which
    b = bytearray(data)
    b[pos + 4 + 17] = value      # first symbol of the first table of the first DHT segment
    return bytes(b)
assert check([with_dc_value(small, 11)], 8, 8)[0] == 0
for v in (16, 17, 64, 255):
    assert check([with_dc_value(small, v)], 8, 8)[0] == 1, v
for name in ("prog_gray_320x240_q85", "gray_320x240_q80", "ycc420_71x53_q75", "prog_ycc420_200x136_q75"):
    if name in files:
        hh, ww = expected[name].shape
        assert check([with_dc_value(files[name], 200)], ww, hh)[0] == 1, name
# ... and an interleaved scan of more than 10 blocks per MCU (jdinput.c per_scan_setup: JERR_BAD_MCU_SIZE): sampling factors
# 4x2 + 2x1 + 2x1 = 12
def with_sampling(data, hv):
    pos = data.find(b"\xFF\xC0")
    if pos < 0:
        pos = data.find(b"\xFF\xC2")
    b = bytearray(data)
    assert b[pos + 9] == 3
    for c, x in enumerate(hv):
        b[pos + 11 + 3 * c] = x
    return bytes(b)
for name in sorted(files):
    if name.startswith(("ycc", "prog_ycc")):
        hh, ww = expected[name].shape
        st, _ = check([with_sampling(files[name], (0x42, 0x21, 0x21))], ww, hh)
        assert st == 1, (name, st)
        break
# crafted scan scripts (progressive, and sequential files in several scans): planned, and their mutations survive the parser
crafted = [(d, w, h) for _, d, w, h in jpeg_craft.progressive_cases()[:5] + jpeg_craft.multiscan_sequential_cases()]
for data, w, h in crafted:
    st, total = check([data], w, h)
    assert st == 0 and total > len(data) // 2, (st, total)
rng = np.random.Generator(np.random.PCG64(99))
for it in range(300):
    data, w, h = crafted[int(rng.integers(len(crafted)))]
    st, _ = check([mutate(data, rng)], w, h)
    assert st in (0, 1, 4), st
rng = np.random.Generator(np.random.PCG64(20261004))
names = sorted(files)
for it in range(2000):
    name = names[int(rng.integers(len(names)))]
    h, w = expected[name].shape
    batch = [mutate(files[name], rng) for _ in range(int(rng.integers(1, 4)))]
    if it % 7 == 0:
        batch.append(files[name])  # a good file next to damaged ones
    st, _ = check(batch, w, h, int(rng.integers(2)))
    assert st in (0, 1, 4), st   # VSF_OK, VSF_ERR_INVALID_ARG, VSF_ERR_UNSUPPORTED -- never anything else, never a crash
    ok += st == 0
    bad += st != 0
# The same damage, one file at a time, against the system's libjpeg driven as cv::imdecode drives it (tests/jpeg_ref.py): what
# libjpeg reads without a warning the parser accepts; what libjpeg gives up on while decoding or in jpeg_finish_decompress (a
# marker code it does not know behind the scan's data, a second frame ...) the parser refuses -- the marker walk of
# vsf_jpeg_host.cc (baseline_tail_ok), here under the sanitizers.
import jpeg_ref
agree = gave_up = refused_too = 0
if jpeg_ref.available():
    for it in range(1500):
        name = names[int(rng.integers(len(names)))]
        h, w = expected[name].shape
        f = mutate(files[name], rng)
        st, _ = check([f], w, h)
        ref, _, warn = jpeg_ref.imdecode_gray(f, w, h)
        if ref == 0 and warn == 0:
            assert st == 0, (it, name, st, "libjpeg reads it without a warning")
            agree += 1
        elif ref == 2:
            gave_up += 1
            refused_too += st != 0
    assert agree > 50 and gave_up > 200 and refused_too >= 0.97 * gave_up, (agree, gave_up, refused_too)
    print("libjpeg: %d silent reads accepted, %d of %d give-ups refused" % (agree, refused_too, gave_up))
for junk in (b"", b"\xff", b"\xff\xd8", b"\xff\xd8\xff", b"\xff\xd8\xff\xda\x00", bytes(100), b"\xff\xd8" + b"\xff\xc0" * 50):
    st, _ = check([junk], 8, 8)
    assert st in (1, 4), (junk[:8], st)
print("done ok=%d refused=%d" % (ok, bad))
'''


def test_jpeg_host_parser_under_asan_and_ubsan(tmp_path):
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (asan_rt and Path(asan_rt).exists() and ubsan_rt and Path(ubsan_rt).exists()):
        pytest.skip("no sanitizer runtime in this toolchain")
    r = subprocess.run(["make", "-s", "-C", str(CSRC), "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and LIB.exists(), r.stderr[-2000:]
    script = tmp_path / "drive.py"
    script.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD="%s %s" % (asan_rt, ubsan_rt),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, str(script), str(LIB), str(GOLD), str(Path(__file__).resolve().parent)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-4000:])
    assert p.stdout.strip().splitlines()[-1].startswith("done") and "ok=" in p.stdout
    ok = int(p.stdout.split("ok=")[1].split()[0])
    refused = int(p.stdout.split("refused=")[1].split()[0])
    assert ok > 300 and refused > 100  # the mutations reached both the accepting and the refusing paths
    import jpeg_ref
    if jpeg_ref.available():  # (then the child has asked libjpeg, too)
        assert "give-ups refused" in p.stdout, p.stdout[-500:]
    # the instrumentation is live: a deliberate one-byte heap overflow (VSF_ASAN_SELFTEST) aborts the same child
    q = subprocess.run([sys.executable, str(script), str(LIB), str(GOLD), str(Path(__file__).resolve().parent)],
                       capture_output=True, text=True, env=dict(env, VSF_ASAN_SELFTEST="1"), timeout=600)
    assert q.returncode != 0 and "heap-buffer-overflow" in q.stderr, (q.returncode, q.stderr[-1500:])
