"""Test helper: writes baseline gray JPEG files from COEFFICIENTS with a Huffman table of the caller's choosing
(ITU-T T.81 Annex B syntax, Annex C code assignment, F.1.2 entropy coding) -- what no encoder library lets one do.
Used to reach table shapes libjpeg's encoder never emits (scores of codes longer than 9 bits), which the device's
parallel decoder must hand to the one-wave decoder.  The expected pixels come from libjpeg-turbo (Pillow) reading the
same file, never from this module."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                   14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39,
                   46, 53, 60, 61, 54, 47, 55, 62, 63])


def canonical_codes(bits, vals):
    """T.81 Annex C: symbol -> (code, length) for BITS[1..16] / HUFFVAL."""
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


def flat_ac_table(short=(0x00, 0x01), long_len=10):
    """An AC table whose few `short` symbols get 2-bit codes and every other run/size symbol a `long_len`-bit code."""
    syms = [s for s in short]
    rest = [(r << 4) | z for r in range(16) for z in range(1, 11) if ((r << 4) | z) not in short] + \
           [s for s in (0x00, 0xF0) if s not in short]
    bits = [0] * 16
    bits[1] = len(short)
    bits[long_len - 1] = len(rest)
    return bits, syms + rest


STD_DC_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
STD_DC_VALS = list(range(12))


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, length):
        self.acc = (self.acc << length) | code
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _size_bits(v):
    s = int(abs(v)).bit_length()
    return s, (v if v >= 0 else v + (1 << s) - 1)


def write_gray_jpeg(coef, qt, dc_table, ac_table):
    """coef: (blocks_y, blocks_x, 64) ints in NATURAL order (quantised values); qt: 64 ints, natural order."""
    by, bx, _ = coef.shape
    h, w = by * 8, bx * 8
    seg = lambda marker, body: bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + bytes(body)
    out = bytearray(b"\xFF\xD8")
    out += seg(0xDB, [0] + [int(qt[z]) for z in ZIGZAG])
    out += seg(0xC0, [8, h >> 8, h & 255, w >> 8, w & 255, 1, 1, 0x11, 0])
    for cls, (bits, vals) in ((0, dc_table), (1, ac_table)):
        out += seg(0xC4, [cls << 4] + list(bits) + list(vals))
    out += seg(0xDA, [1, 1, 0x00, 0, 63, 0])
    dc, ac = canonical_codes(*dc_table), canonical_codes(*ac_table)
    w_, pred = _Bits(), 0
    for y in range(by):
        for x in range(bx):
            zz = coef[y, x][ZIGZAG]
            s, extra = _size_bits(int(zz[0]) - pred)
            pred = int(zz[0])
            w_.put(*dc[s])
            if s:
                w_.put(extra, s)
            run = 0
            last = max([i for i in range(1, 64) if zz[i]] or [0])
            for i in range(1, last + 1):
                if zz[i] == 0:
                    run += 1
                    continue
                while run > 15:
                    w_.put(*ac[0xF0])
                    run -= 16
                s, extra = _size_bits(int(zz[i]))
                w_.put(*ac[(run << 4) | s])
                w_.put(extra, s)
                run = 0
            if last < 63:
                w_.put(*ac[0x00])
    w_.flush()
    return bytes(out + w_.out + b"\xFF\xD9")


def random_coefficients(rng, by, bx, density=0.3, amplitude=40):
    c = np.zeros((by, bx, 64), np.int64)
    c[..., 0] = rng.integers(-60, 60, (by, bx))
    mask = rng.random((by, bx, 63)) < density * np.linspace(1.0, 0.1, 63)
    c[..., 1:] = np.where(mask, rng.integers(-amplitude, amplitude + 1, (by, bx, 63)), 0)
    return c
