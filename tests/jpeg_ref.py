"""The system's libjpeg as cv::imdecode(IMREAD_GRAYSCALE) drives it (tests/cpp/jpeg_ref.c, built on demand against
libjpeg.so.8): a second reference for the JPEG ingest beside PIL's bundled libjpeg-turbo.  `available()` is False where the
library is missing or its layout check fails."""
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
    # libjpeg-turbo's SIMD routines multiply in 16 bits where jidctint.c / jdhuff.c use the machine's int: on coefficients no
    # encoder writes (damaged files) they give other pixels than the C code every libjpeg shares.  The reference is the C code.
    import os
    os.environ["JSIMD_FORCENONE"] = "1"
    out = HERE / "cpp" / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libjpeg_ref.so"
    src = HERE / "cpp" / "jpeg_ref.c"
    try:
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", str(src), "-o", str(so), "-l:libjpeg.so.8"],
                                  stderr=subprocess.DEVNULL)
        lib = C.CDLL(str(so))
    except (OSError, subprocess.CalledProcessError):
        return None
    lib.jpeg_ref_gray.restype = C.c_int
    lib.jpeg_ref_gray.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.jpeg_ref_warnings.restype = C.c_long
    _lib = lib
    return lib


def available() -> bool:
    return _load() is not None


def imdecode_gray(jpeg: bytes, w: int = 0, h: int = 0, max_side: int = 8192):
    """-> (status, image or None, warnings): status 0 decoded, 1 header refused, 2 data refused, 3 another size, 4 CMYK."""
    lib = _load()
    ww, hh = C.c_int(0), C.c_int(0)
    pitch, rows = (w + 64, h) if w > 0 and h > 0 else (max_side, max_side)
    buf = np.zeros((rows, pitch), np.uint8)
    st = lib.jpeg_ref_gray(jpeg, len(jpeg), w, h, buf.ctypes.data, pitch, C.byref(ww), C.byref(hh))
    if st != 0:
        return st, None, lib.jpeg_ref_warnings()
    return 0, buf[:hh.value, :ww.value].copy(), lib.jpeg_ref_warnings()
