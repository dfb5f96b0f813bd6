#!/usr/bin/env python3
"""Writes tests/golden/jpeg/: small baseline and progressive JPEG files and what libjpeg-turbo (through Pillow, present in this image)
decodes them to with out_color_space = JCS_GRAYSCALE -- the decode cv::imdecode(..., IMREAD_GRAYSCALE) performs
(slam_frontend_main.cc:99-100; OpenCV's grfmt_jpeg.cpp sets JCS_GRAYSCALE for a gray read, default JDCT_ISLOW).

These ARE third-party-generated vectors (unlike tests/golden/*.npz): libjpeg's ISLOW inverse DCT is exact integer
arithmetic and libjpeg-turbo's SIMD version is bit-identical to it, so any libjpeg-family decoder -- the one inside the
reference's OpenCV included -- produces these bytes.  Regenerate with:  python tools/make_jpeg_golden.py
"""
import io
import sys
from pathlib import Path

import numpy as np
from PIL import Image, features

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vision_slam_frontend_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden" / "jpeg"


def scene(w, h, seed):
    return synth.stereo_pair(w, h, seed, n_objects=max(10, w * h // 300))[0]


def cases():
    rng = np.random.default_rng(2024)
    g320 = scene(320, 240, 1)
    yield "gray_320x240_q80", g320, "L", dict(quality=80)
    yield "gray_64x48_noise_q80", rng.integers(0, 256, (48, 64), dtype=np.uint8), "L", dict(quality=80)
    yield "gray_33x17_q95", scene(33, 17, 2), "L", dict(quality=95)
    yield "gray_8x8_q90", scene(8, 8, 3), "L", dict(quality=90)
    yield "gray_1x1_q75", np.array([[201]], np.uint8), "L", dict(quality=75)
    yield "gray_200x120_q30", scene(200, 120, 4), "L", dict(quality=30)
    yield "gray_96x80_q100", scene(96, 80, 5), "L", dict(quality=100)
    yield "gray_160x120_optimized", scene(160, 120, 6), "L", dict(quality=85, optimize=True)
    yield "gray_160x120_restart4", scene(160, 120, 7), "L", dict(quality=80, restart_marker_blocks=4)
    yield "gray_100x60_restart_rows", scene(100, 60, 8), "L", dict(quality=70, restart_marker_rows=1)
    yield "gray_saturated_64x64", np.where(rng.random((64, 64)) < 0.5, 0, 255).astype(np.uint8), "L", dict(quality=60)
    rgb = np.stack([scene(71, 53, 9), scene(71, 53, 10), scene(71, 53, 11)], 2)
    yield "ycc420_71x53_q75", rgb, "RGB", dict(quality=75, subsampling=2)
    yield "ycc422_71x53_q75", rgb, "RGB", dict(quality=75, subsampling=1)
    yield "ycc444_40x40_q90", rgb[:40, :40], "RGB", dict(quality=90, subsampling=0)
    yield ("ycc420_restart_64x64", np.stack([scene(64, 64, 12)] * 3, 2), "RGB",
           dict(quality=80, subsampling=2, restart_marker_blocks=3))
    # progressive files (SOF2): libjpeg's default scan script -- spectral selection AND successive approximation, end-of-band
    # runs, AC refinement with correction bits; for colour files the DC scans interleave all three components
    yield "progressive_64x48", g320[:48, :64], "L", dict(quality=80, progressive=True)
    yield "prog_gray_320x240_q85", g320, "L", dict(quality=85, progressive=True)
    yield "prog_gray_33x17_q30", scene(33, 17, 13), "L", dict(quality=30, progressive=True)
    yield "prog_gray_160x120_restart5", scene(160, 120, 14), "L", dict(quality=80, progressive=True, restart_marker_blocks=5)
    yield "prog_noise_64x48_q95", rng.integers(0, 256, (48, 64), dtype=np.uint8), "L", dict(quality=95, progressive=True)
    rgb2 = np.stack([scene(200, 136, 15), scene(200, 136, 16), scene(200, 136, 17)], 2)
    yield "prog_ycc420_200x136_q75", rgb2, "RGB", dict(quality=75, subsampling=2, progressive=True)
    yield "prog_ycc422_71x53_q60", rgb, "RGB", dict(quality=60, subsampling=1, progressive=True)
    yield "prog_ycc444_40x40_q92", rgb[:40, :40], "RGB", dict(quality=92, subsampling=0, progressive=True)
    yield ("prog_ycc420_restart_rows_100x60", rgb2[:60, :100], "RGB",
           dict(quality=70, subsampling=2, progressive=True, restart_marker_rows=1))


def main():
    assert features.check_feature("libjpeg_turbo"), "expected a libjpeg-turbo backed Pillow"
    OUT.mkdir(parents=True, exist_ok=True)
    expected = {}
    for name, img, mode, kw in cases():
        b = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(img), mode).save(b, "JPEG", **kw)
        data = b.getvalue()
        (OUT / (name + ".jpg")).write_bytes(data)
        im = Image.open(io.BytesIO(data))
        im.draft("L", im.size)  # -> cinfo.out_color_space = JCS_GRAYSCALE
        dec = np.asarray(im.convert("L") if im.mode != "L" else im)
        assert dec.shape == img.shape[:2] and dec.dtype == np.uint8
        expected[name] = dec
        print("%-28s %6d bytes  %dx%d" % (name, len(data), dec.shape[1], dec.shape[0]))
    np.savez_compressed(OUT / "expected_gray.npz", **expected)
    print("libjpeg-turbo via Pillow %s; %d cases" % (Image.__version__, len(expected)))


if __name__ == "__main__":
    main()
