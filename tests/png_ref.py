"""The real libpng as cv::imdecode(IMREAD_GRAYSCALE) drives it (tests/cpp/png_ref.c, built on demand against the system's
libpng16.so.16): the reference of the PNG parity tests.  `available()` is False where the library is missing."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    out = HERE / "cpp" / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libpng_ref.so"
    src = HERE / "cpp" / "png_ref.c"
    try:
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", str(src), "-o", str(so), "-l:libpng16.so.16"],
                                  stderr=subprocess.DEVNULL)
        lib = C.CDLL(str(so))
    except (OSError, subprocess.CalledProcessError):
        return None
    lib.png_ref_version.restype = C.c_char_p
    lib.png_ref_last_error.restype = C.c_char_p
    lib.png_ref_last_warning.restype = C.c_char_p
    lib.png_ref_gray.restype = C.c_int
    lib.png_ref_gray.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = lib
    return lib


def available() -> bool:
    return _load() is not None


def version() -> str:
    return _load().png_ref_version().decode()


def last_error() -> str:
    """libpng's message for the last refusal (png_error), '' if none."""
    return _load().png_ref_last_error().decode(errors="replace")


def last_warning() -> str:
    return _load().png_ref_last_warning().decode(errors="replace")


def imdecode_gray(png: bytes, w: int = 0, h: int = 0, max_side: int = 8192):
    """-> (status, image or None, info): status 0 decoded, 1 header refused, 2 data refused (imdecode returns an empty Mat),
    3 another size than (w, h).  info = (bit depth, colour type, interlace, row bytes after the transformations)."""
    lib = _load()
    ww, hh = C.c_int(0), C.c_int(0)
    info = (C.c_int * 4)()
    if w > 0 and h > 0:
        pitch, rows = w * 8 + 64, h
    else:
        pitch, rows = max_side * 8, max_side
    buf = np.zeros((rows, pitch), np.uint8)
    st = lib.png_ref_gray(png, len(png), w, h, buf.ctypes.data, pitch, C.byref(ww), C.byref(hh), info)
    if st != 0:
        return st, None, tuple(info)
    return 0, buf[:hh.value, :ww.value].copy(), tuple(info)
