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


# ---- progressive files (T.81 Annex G) with a scan script of the caller's choosing ------------------------------------
def skewed_table(freq):
    """A legal Huffman table for the symbols in `freq` (symbol -> count): the six most frequent get codes of 1..6 bits, the
    others share one longer length (so short AND long codes occur; the all-ones code stays free)."""
    syms = sorted(freq, key=lambda s: (-freq[s], s))
    bits = [0] * 16
    head = min(len(syms), 6)
    for i in range(head):
        bits[i] = 1
    rest = len(syms) - head
    if rest:
        extra = int(rest).bit_length()  # 2^extra >= rest + 1
        bits[6 + extra - 1] += rest
    return bits, syms


class _Scan:
    """Collects one scan as ('s', symbol) / ('b', value, nbits) / ('r',) items; tables are made from the symbol counts."""

    def __init__(self):
        self.items, self.freq = [], {}
        self.eobrun, self.pending = 0, []   # end-of-band run in progress; correction bits waiting behind it

    def sym(self, s):
        self.items.append(("s", s))
        self.freq[s] = self.freq.get(s, 0) + 1

    def bits(self, v, n):
        if n:
            self.items.append(("b", int(v) & ((1 << n) - 1), n))

    def flush_eobrun(self):
        if self.eobrun:
            n = self.eobrun.bit_length() - 1
            self.sym(n << 4)
            self.bits(self.eobrun, n)
            self.eobrun = 0
        for b in self.pending:
            self.bits(b, 1)
        self.pending = []

    def restart(self):
        self.flush_eobrun()
        self.items.append(("r",))


def write_progressive_jpeg(width, height, comps, script, restart_interval=0):
    """comps: [(h, v, qt[64 natural], coef[(blocks_y, blocks_x, 64) natural order, over the PADDED block grid])], component 0
    = luminance; script: [(component indices, Ss, Se, Ah, Al)].  Every scan gets Huffman tables of its own.  The expected
    pixels come from libjpeg-turbo reading the file, never from this module."""
    seg = lambda marker, body: bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + bytes(body)
    hmax, vmax = max(c[0] for c in comps), max(c[1] for c in comps)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xFF\xD8")
    for i, (_, _, qt, _) in enumerate(comps):
        out += seg(0xDB, [i] + [int(qt[z]) for z in ZIGZAG])
    sof = [8, height >> 8, height & 255, width >> 8, width & 255, len(comps)]
    for i, (h, v, _, _) in enumerate(comps):
        sof += [i + 1, (h << 4) | v, i]
    out += seg(0xC2, sof)
    if restart_interval:
        out += seg(0xDD, [restart_interval >> 8, restart_interval & 255])
    shift = lambda v, al: (abs(int(v)) >> al) * (1 if v >= 0 else -1)  # AC point transform: towards zero
    for cidx, Ss, Se, Ah, Al in script:
        sc = _Scan()
        pred = [0] * len(cidx)
        units = []  # per MCU: [(slot, block)]
        if len(cidx) == 1:
            h, v, _, coef = comps[cidx[0]]
            bw, bh = -(-(-(-width * h // hmax)) // 8), -(-(-(-height * v // vmax)) // 8)
            units = [[(0, coef[y, x])] for y in range(bh) for x in range(bw)]
        else:
            for my in range(mcus_y):
                for mx in range(mcus_x):
                    u = []
                    for slot, ci in enumerate(cidx):
                        h, v, _, coef = comps[ci]
                        u += [(slot, coef[my * v + y, mx * h + x]) for y in range(v) for x in range(h)]
                    units.append(u)
        for n, unit in enumerate(units):
            if restart_interval and n and n % restart_interval == 0:
                sc.restart()
                pred = [0] * len(cidx)
            for slot, blk in unit:
                zz = [int(t) for t in blk[ZIGZAG]]
                if Ss == 0 and Ah == 0:
                    v = zz[0] >> Al                      # DC point transform: arithmetic shift
                    s, extra = _size_bits(v - pred[slot])
                    pred[slot] = v
                    sc.sym(slot * 256 + s)               # (the slot picks the table; removed when the codes are looked up)
                    sc.bits(extra, s)
                elif Ss == 0:
                    sc.bits((zz[0] >> Al) & 1, 1)
                elif Ah == 0:
                    r = 0
                    for k in range(Ss, Se + 1):
                        a = shift(zz[k], Al)
                        if a == 0:
                            r += 1
                            continue
                        sc.flush_eobrun()
                        while r > 15:
                            sc.sym(0xF0)
                            r -= 16
                        s, extra = _size_bits(a)
                        sc.sym((r << 4) | s)
                        sc.bits(extra, s)
                        r = 0
                    if r:
                        sc.eobrun += 1
                        if sc.eobrun == 0x7FFF:
                            sc.flush_eobrun()
                else:
                    a = [abs(zz[k]) >> Al for k in range(64)]
                    eob = max([k for k in range(Ss, Se + 1) if a[k] == 1] or [-1])
                    r, corr = 0, []
                    for k in range(Ss, Se + 1):
                        if a[k] == 0:
                            r += 1
                            continue
                        while r > 15 and k <= eob:
                            sc.flush_eobrun()
                            sc.sym(0xF0)
                            r -= 16
                            for b in corr:
                                sc.bits(b, 1)
                            corr = []
                        if a[k] > 1:
                            corr.append(a[k] & 1)
                            continue
                        sc.flush_eobrun()
                        sc.sym((r << 4) | 1)
                        sc.bits(0 if zz[k] < 0 else 1, 1)
                        for b in corr:
                            sc.bits(b, 1)
                        corr, r = [], 0
                    if r or corr:
                        sc.eobrun += 1
                        sc.pending += corr
                        if sc.eobrun == 0x7FFF or len(sc.pending) > 900:
                            sc.flush_eobrun()
        sc.flush_eobrun()
        # tables of this scan: one per scan component for a DC-first scan, one for an AC scan, none for DC refinement
        codes = {}
        if not (Ss == 0 and Ah):
            slots = range(len(cidx)) if Ss == 0 else [0]
            for slot in slots:
                freq = {s & 255: n for s, n in sc.freq.items() if (s >> 8) == slot} if Ss == 0 else dict(sc.freq)
                freq = freq or {0: 1}
                bits, vals = skewed_table(freq)
                out += seg(0xC4, [((0 if Ss == 0 else 1) << 4) | slot] + bits + vals)
                codes[slot] = canonical_codes(bits, vals)
        sos = [len(cidx)]
        for slot, ci in enumerate(cidx):
            sos += [ci + 1, (slot << 4) if Ss == 0 else 0]
        out += seg(0xDA, sos + [Ss, Se, (Ah << 4) | Al])
        w_, rst = _Bits(), 0
        for it in sc.items:
            if it[0] == "s":
                w_.put(*codes[it[1] >> 8 if Ss == 0 else 0][it[1] & 255])
            elif it[0] == "b":
                w_.put(it[1], it[2])
            else:
                w_.flush()
                w_.out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
        w_.flush()
        out += w_.out
    return bytes(out + b"\xFF\xD9")


def progressive_cases():
    """Scan scripts libjpeg's encoder does not write: spectral selection alone; four refinement passes per coefficient;
    DC scans per component and AC bands cut in odd places, chroma first; an interleaved DC scan of two of the three
    components; restart intervals with long end-of-band runs; an end-of-band run of more than 32767 blocks.
    -> [(name, bytes, width, height)]"""
    rng = np.random.default_rng(1234)
    # (small steps: far outside 0..255 libjpeg's C code wraps modulo 1024 where its SIMD code saturates -- no encoder gets there)
    qt = lambda: rng.integers(1, 5, 64)
    out = []

    def comp(h, v, by, bx, density=0.3, amplitude=40):
        return (h, v, qt(), random_coefficients(rng, by, bx, density, amplitude))

    out.append(("spectral_selection_only", write_progressive_jpeg(
        75, 43, [comp(1, 1, 6, 10)], [((0,), 0, 0, 0, 0), ((0,), 1, 5, 0, 0), ((0,), 6, 63, 0, 0)]), 75, 43))
    deep = [((0,), 0, 0, 0, 3), ((0,), 1, 63, 0, 3)]
    for al in (2, 1, 0):
        deep += [((0,), 0, 0, al + 1, al), ((0,), 1, 63, al + 1, al)]
    out.append(("four_refinement_passes", write_progressive_jpeg(64, 64, [comp(1, 1, 8, 8, 0.4, 60)], deep), 64, 64))
    c420 = [comp(2, 2, 16, 20), comp(1, 1, 8, 10), comp(1, 1, 8, 10)]
    odd = [((0,), 0, 0, 0, 1), ((1,), 0, 0, 0, 1), ((2,), 0, 0, 0, 1), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0),
           ((0,), 1, 1, 0, 2), ((0,), 2, 9, 0, 1), ((0,), 10, 63, 0, 0), ((0,), 1, 1, 2, 1), ((0,), 0, 0, 1, 0),
           ((1,), 0, 0, 1, 0), ((0,), 1, 1, 1, 0), ((2,), 0, 0, 1, 0), ((0,), 2, 9, 1, 0)]
    out.append(("dc_per_component_odd_bands", write_progressive_jpeg(150, 120, c420, odd), 150, 120))
    c444 = [comp(1, 1, 5, 7), comp(1, 1, 5, 7), comp(1, 1, 5, 7)]
    two = [((0, 1), 0, 0, 0, 0), ((2,), 0, 0, 0, 0), ((0,), 1, 63, 0, 1), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0),
           ((0,), 1, 63, 1, 0)]
    out.append(("dc_of_two_components_interleaved", write_progressive_jpeg(50, 37, c444, two), 50, 37))
    c422 = [comp(2, 1, 6, 10, 0.02, 9), comp(1, 1, 6, 5, 0.02, 9), comp(1, 1, 6, 5, 0.02, 9)]
    std = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2),
           ((0,), 1, 63, 2, 1), ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0)]
    out.append(("restarts_and_long_eob_runs", write_progressive_jpeg(77, 45, c422, std, restart_interval=3), 77, 45))
    sparse = np.zeros((176, 200, 64), np.int64)
    sparse[..., 0] = rng.integers(-20, 20, (176, 200))
    for y, x in ((0, 0), (3, 7), (100, 150), (175, 199)):
        sparse[y, x, rng.integers(1, 64, 6)] = rng.integers(-30, 31, 6)
    big = [((0,), 0, 0, 0, 0), ((0,), 1, 63, 0, 1), ((0,), 1, 63, 1, 0)]
    out.append(("eob_run_beyond_32767_blocks", write_progressive_jpeg(1600, 1408, [(1, 1, qt(), sparse)], big), 1600, 1408))
    return out


def write_multiscan_sequential_jpeg(width, height, comps, scans, restart_interval=0):
    """A SEQUENTIAL (SOF0) file whose components come in several scans -- e.g. [(0,), (1, 2)] or [(0, 1), (2,)] -- which
    libjpeg's encoder writes only on request (cjpeg -scans).  comps as for write_progressive_jpeg."""
    seg = lambda marker, body: bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + bytes(body)
    hmax, vmax = max(c[0] for c in comps), max(c[1] for c in comps)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xFF\xD8")
    for i, (_, _, qt, _) in enumerate(comps):
        out += seg(0xDB, [i] + [int(qt[z]) for z in ZIGZAG])
    sof = [8, height >> 8, height & 255, width >> 8, width & 255, len(comps)]
    for i, (h, v, _, _) in enumerate(comps):
        sof += [i + 1, (h << 4) | v, i]
    out += seg(0xC0, sof)
    if restart_interval:
        out += seg(0xDD, [restart_interval >> 8, restart_interval & 255])
    for cidx in scans:
        items, freq = [], {}   # ('s', cls, slot, symbol) / ('b', value, nbits) / ('r',)

        def sym(cls, slot, s):
            items.append(("s", cls, slot, s))
            freq.setdefault((cls, slot), {})
            freq[(cls, slot)][s] = freq[(cls, slot)].get(s, 0) + 1

        if len(cidx) == 1:
            h, v, _, coef = comps[cidx[0]]
            bw, bh = -(-(-(-width * h // hmax)) // 8), -(-(-(-height * v // vmax)) // 8)
            units = [[(0, coef[y, x])] for y in range(bh) for x in range(bw)]
        else:
            units = []
            for my in range(mcus_y):
                for mx in range(mcus_x):
                    u = []
                    for slot, ci in enumerate(cidx):
                        h, v, _, coef = comps[ci]
                        u += [(slot, coef[my * v + y, mx * h + x]) for y in range(v) for x in range(h)]
                    units.append(u)
        pred = [0] * len(cidx)
        for n, unit in enumerate(units):
            if restart_interval and n and n % restart_interval == 0:
                items.append(("r",))
                pred = [0] * len(cidx)
            for slot, blk in unit:
                zz = [int(t) for t in blk[ZIGZAG]]
                s, extra = _size_bits(zz[0] - pred[slot])
                pred[slot] = zz[0]
                sym(0, slot, s)
                if s:
                    items.append(("b", extra & ((1 << s) - 1), s))
                run = 0
                last = max([i for i in range(1, 64) if zz[i]] or [0])
                for i in range(1, last + 1):
                    if zz[i] == 0:
                        run += 1
                        continue
                    while run > 15:
                        sym(1, slot, 0xF0)
                        run -= 16
                    s, extra = _size_bits(zz[i])
                    sym(1, slot, (run << 4) | s)
                    items.append(("b", extra & ((1 << s) - 1), s))
                    run = 0
                if last < 63:
                    sym(1, slot, 0x00)
        codes = {}
        for (cls, slot), f in sorted(freq.items()):
            bits, vals = skewed_table(f)
            out += seg(0xC4, [(cls << 4) | slot] + bits + vals)
            codes[(cls, slot)] = canonical_codes(bits, vals)
        sos = [len(cidx)]
        for slot, ci in enumerate(cidx):
            sos += [ci + 1, (slot << 4) | slot]
        out += seg(0xDA, sos + [0, 63, 0])
        w_, rst = _Bits(), 0
        for it in items:
            if it[0] == "s":
                w_.put(*codes[(it[1], it[2])][it[3]])
            elif it[0] == "b":
                w_.put(it[1], it[2])
            else:
                w_.flush()
                w_.out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
        w_.flush()
        out += w_.out
    return bytes(out + b"\xFF\xD9")


def multiscan_sequential_cases():
    """-> [(name, bytes, width, height)]: the luminance in a scan of its own (before and after the chroma scans), the
    luminance interleaved with one chroma component, with restart intervals."""
    rng = np.random.default_rng(4321)
    qt = lambda: rng.integers(1, 5, 64)
    comp = lambda h, v, by, bx: (h, v, qt(), random_coefficients(rng, by, bx))
    out = []
    c420 = [comp(2, 2, 8, 10), comp(1, 1, 4, 5), comp(1, 1, 4, 5)]
    out.append(("y_then_cb_then_cr", write_multiscan_sequential_jpeg(75, 61, c420, [(0,), (1,), (2,)]), 75, 61))
    out.append(("chroma_first_restarts", write_multiscan_sequential_jpeg(75, 61, c420, [(1, 2), (0,)], restart_interval=4), 75, 61))
    c444 = [comp(1, 1, 5, 7), comp(1, 1, 5, 7), comp(1, 1, 5, 7)]
    out.append(("y_with_cb_then_cr", write_multiscan_sequential_jpeg(50, 37, c444, [(0, 1), (2,)]), 50, 37))
    c422 = [comp(2, 1, 6, 10), comp(1, 1, 6, 5), comp(1, 1, 6, 5)]
    out.append(("cr_then_y_with_cb_422", write_multiscan_sequential_jpeg(77, 45, c422, [(2,), (0, 1)], restart_interval=2), 77, 45))
    return out
