"""Hand-made PNG files for the parity tests of vsf_png_decode_gray_batch (cv::imdecode(IMREAD_GRAYSCALE) for PNG,
slam_frontend_main.cc:99-100): chunk framing and CRCs written here, the row filters applied here with a chosen filter type
per row, the compression done by the real zlib with a chosen strategy (stored blocks, fixed Huffman codes, run-length
matches of distance 1, Huffman only, the default), the IDAT payload cut into pieces of a chosen size.  What a decoder must
return for such a file is what libpng returns: the tests read every file back with PIL (libpng + zlib) as the reference.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = b"\x89PNG\r\n\x1a\n"


def chunk(kind: bytes, data: bytes, bad_crc: bool = False) -> bytes:
    crc = zlib.crc32(kind + data) & 0xFFFFFFFF
    if bad_crc:
        crc ^= 0x5A5A5A5A
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", crc)


def ihdr(w: int, h: int, depth: int, ctype: int, interlace: int = 0) -> bytes:
    return chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))


def paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def filter_rows(rows: np.ndarray, bpp: int, types) -> bytes:
    """rows: (h, row_bytes) uint8 raw scanlines; types: per-row filter type 0..4 -> the filtered scanlines."""
    h, rb = rows.shape
    out = bytearray()
    prev = np.zeros(rb, np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if rb > bpp else np.zeros(rb, np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if rb > bpp else np.zeros(rb, np.int32)
        t = int(types[y])
        if t == 0:
            f = cur
        elif t == 1:
            f = cur - left
        elif t == 2:
            f = cur - prev
        elif t == 3:
            f = cur - ((left + prev) >> 1)
        else:
            f = cur - paeth(left, prev, upleft)
        out.append(t)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def deflate(data: bytes, level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, wbits: int = 15) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return c.compress(data) + c.flush()


def pack_samples(img: np.ndarray, depth: int) -> np.ndarray:
    """img: (h, w) sample values < 2**depth (depth 1, 2, 4, 8) or (h, w[, 2]) for 16-bit / gray+alpha -> raw scanlines."""
    if depth == 8:
        return np.ascontiguousarray(img.reshape(img.shape[0], -1).astype(np.uint8))
    if depth == 16:
        be = img.astype(">u2")
        return np.ascontiguousarray(be.view(np.uint8).reshape(img.shape[0], -1))
    h, w = img.shape
    ppb = 8 // depth
    wp = (w + ppb - 1) // ppb * ppb
    padded = np.zeros((h, wp), np.uint8)
    padded[:, :w] = img
    out = np.zeros((h, wp // ppb), np.uint8)
    for p in range(ppb):
        out |= (padded[:, p::ppb] << (8 - depth * (p + 1))).astype(np.uint8)
    return out


def write_png(raw_rows: np.ndarray, w: int, h: int, depth: int, ctype: int, *, filters=None, level: int = 6,
              strategy: int = zlib.Z_DEFAULT_STRATEGY, idat_piece: int | None = None, extra_before=(), extra_after=(),
              bad_idat_crc: bool = False, interlace: int = 0, wbits: int = 15, stream: bytes | None = None) -> bytes:
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, depth * channels // 8)
    if stream is None:
        if filters is None:
            filters = np.zeros(h, np.int32)
        stream = deflate(filter_rows(raw_rows, bpp, filters), level, strategy, wbits)
    out = SIGNATURE + ihdr(w, h, depth, ctype, interlace)
    for c in extra_before:
        out += c
    if idat_piece is None:
        out += chunk(b"IDAT", stream, bad_idat_crc)
    else:
        for i in range(0, len(stream), idat_piece):
            out += chunk(b"IDAT", stream[i:i + idat_piece], bad_idat_crc and i == 0)
    for c in extra_after:
        out += c
    return out + chunk(b"IEND", b"")


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))   # x0, y0, dx, dy


def adam7_filtered(samples: np.ndarray, depth: int, ctype: int, rng: np.random.Generator) -> bytes:
    """The filtered scanlines of the seven passes (a random filter per row), one after the other: what an interlaced file's
    zlib stream holds.  samples: (h, w) or (h, w, channels) values of `depth` bits."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, depth * channels // 8)
    data = b""
    for x0, y0, dx, dy in ADAM7:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = pack_samples(sub.astype(np.uint16 if depth == 16 else np.uint8), depth)
        data += filter_rows(rows, bpp, rng.integers(0, 5, sub.shape[0]))
    return data


def write_png_adam7(samples: np.ndarray, depth: int, ctype: int, rng: np.random.Generator, *, level: int = 6,
                    strategy: int = zlib.Z_DEFAULT_STRATEGY, idat_piece: int | None = None, extra_before=()) -> bytes:
    """An interlaced file (PNG specification 8.2)."""
    h, w = samples.shape[:2]
    return write_png(None, w, h, depth, ctype, stream=deflate(adam7_filtered(samples, depth, ctype, rng), level, strategy),
                     idat_piece=idat_piece, interlace=1, extra_before=extra_before)


def gray8(img: np.ndarray, **kw) -> bytes:
    h, w = img.shape
    return write_png(pack_samples(img, 8), w, h, 8, 0, **kw)


def idat_stream(png: bytes) -> bytes:
    """The concatenated IDAT payloads of a file (chunk walk without checks)."""
    pos, out = 8, b""
    while pos + 12 <= len(png):
        n = struct.unpack(">I", png[pos:pos + 4])[0]
        if png[pos + 4:pos + 8] == b"IDAT":
            out += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    return out


def replace_idat(png: bytes, stream: bytes) -> bytes:
    """The same file with another IDAT payload (one chunk, correct CRC)."""
    pos, out, done = 8, png[:8], False
    while pos + 12 <= len(png):
        n = struct.unpack(">I", png[pos:pos + 4])[0]
        kind = png[pos + 4:pos + 8]
        if kind == b"IDAT":
            if not done:
                out += chunk(b"IDAT", stream)
                done = True
        else:
            out += png[pos:pos + 12 + n]
        pos += 12 + n
    return out


def mutate_stream(stream: bytes, rng: np.random.Generator) -> bytes:
    """Damage inside the compressed data (behind the 2-byte zlib header): bit flips, a cut, a zeroed run."""
    b = bytearray(stream)
    kind = int(rng.integers(4))
    if len(b) < 8:
        return bytes(b)
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(2, len(b)))
            b[i] ^= 1 << int(rng.integers(8))
    elif kind == 1:
        del b[int(rng.integers(2, len(b))):]
    elif kind == 2:
        i = int(rng.integers(2, len(b)))
        n = int(rng.integers(1, 32))
        b[i:i + n] = bytes(min(n, len(b) - i))
    else:
        i = int(rng.integers(2, len(b)))
        b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    return bytes(b)


def idat_pieces(png: bytes):
    """The IDAT payloads of a file, one bytes object per chunk."""
    pos, out = 8, []
    while pos + 12 <= len(png):
        n = struct.unpack(">I", png[pos:pos + 4])[0]
        if png[pos + 4:pos + 8] == b"IDAT" and n:
            out.append(png[pos + 8:pos + 8 + n])
        pos += 12 + n
    return out


def zlib_reference(pieces, expected: int, read_size: int = 8192):
    """What zlib makes of a file's IDAT payloads when libpng reads `expected` bytes of image out of them and cv::imdecode then
    calls png_read_end -- a restatement in terms of zlib calls, kept beside the real libpng (tests/png_ref.py) as the readable
    statement of the rule.  libpng hands zlib at most `read_size` (PNG_IDAT_READ_SIZE) bytes of ONE chunk at a time.  While
    rows are wanted, an error or the end of the data is fatal -- and zlib sees, behind the last byte, the rest of the piece it
    was working on.  Then png_read_finish_IDAT drains what is left into a 1024-byte scratch buffer: errors are warnings, the
    end of the stream is fine, one piece that yields nothing ends the loop -- but a refill that finds no IDAT data left is
    png_error("Not enough image data").  Returns the image's bytes, or None when cv::imdecode would return nothing."""
    if isinstance(pieces, (bytes, bytearray)):
        pieces = [bytes(pieces)]
    feed = [c[i:i + read_size] for c in pieces for i in range(0, len(c), read_size)]
    d = zlib.decompressobj()
    out = b""
    k = 0
    pending = b""
    try:
        while len(out) < expected:
            if not pending:
                if k == len(feed) or d.eof:
                    return None          # "Not enough image data"
                pending = feed[k]
                k += 1
            out += d.decompress(pending, expected - len(out))
            pending = d.unconsumed_tail
    except zlib.error:
        return None
    extra = 0
    try:
        while not d.eof:
            if not pending:
                if k == len(feed):
                    return None          # the drain's refill: "Not enough image data"
                pending = feed[k]
                k += 1
            extra += len(d.decompress(pending, 1024))
            pending = d.unconsumed_tail
            if extra == 0 and not pending:
                break                    # (a piece that gave nothing: libpng's loop ends)
    except zlib.error:
        pass                             # a warning
    return out
