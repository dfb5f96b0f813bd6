"""Seeded mutators for JPEG robustness tests (the five of round 2's one-off fuzz script): bit flips, truncation and stray
markers in the entropy-coded data, damage to the headers, garbage behind the headers -- and, since round 4, a sixth that
rewrites the code-length counts of a DHT segment while keeping every segment length CONSISTENT (plain header damage leaves
the lengths inconsistent, so the parser refuses the file before it ever builds a table from it)."""
import numpy as np


def segments(data: bytes):
    """(marker, offset of the marker's 0xFF, total length incl. the two marker bytes) for the header segments up to SOS."""
    out, pos = [], 2
    while pos + 4 <= len(data) and data[pos] == 0xFF:
        m = data[pos + 1]
        ln = (data[pos + 2] << 8) | data[pos + 3]
        out.append((m, pos, ln + 2))
        if m == 0xDA:
            break
        pos += ln + 2
    return out


def all_segments(data: bytes):
    """Like segments(), but through every scan of a multi-scan file (the entropy-coded data behind an SOS is stepped over:
    it ends at the first 0xFF followed by anything but 0x00, RSTn or 0xFF)."""
    out, pos = [], 2
    while pos + 4 <= len(data) and data[pos] == 0xFF:
        m = data[pos + 1]
        if m == 0xD9:
            break
        ln = (data[pos + 2] << 8) | data[pos + 3]
        out.append((m, pos, ln + 2))
        pos += ln + 2
        if m == 0xDA:
            while pos + 1 < len(data) and not (data[pos] == 0xFF and data[pos + 1] not in (0x00, 0xFF) and
                                               not 0xD0 <= data[pos + 1] <= 0xD7):
                pos += 1
    return out


def drop_last_scans(data: bytes, n: int) -> bytes:
    """A multi-scan (progressive) file without its last `n` scans (and the tables defined for them), EOI appended."""
    segs = all_segments(data)
    sos = [i for i, (m, _, _) in enumerate(segs) if m == 0xDA]
    assert len(sos) > n, (len(sos), n)
    cut = sos[len(sos) - n]
    while cut > 0 and segs[cut - 1][0] in (0xC4, 0xDD, 0xDB):  # the DHT / DRI / DQT segments in front of that scan
        cut -= 1
    return data[:segs[cut][1]] + b"\xff\xd9"


def rewrite_dht(data: bytes, rng: np.random.Generator, counts=None, which=None, max_value=255) -> bytes:
    """Replaces the counts (and as many values as they call for, drawn from 0..max_value) of one table of one DHT segment;
    the segment's length field is recomputed, so the file stays well-formed up to the meaning of the counts themselves.
    (A DC table's values must stay <= 15 for the table to be legal: libjpeg refuses larger ones, and so does the parser.)"""
    dht = [(pos, ln) for m, pos, ln in segments(data) if m == 0xC4]
    if not dht:
        return data
    pos, ln = dht[int(rng.integers(len(dht))) if which is None else which]
    body = bytearray(data[pos + 4:pos + ln])
    tables, i = [], 0
    while i + 17 <= len(body):
        total = sum(body[i + 1:i + 17])
        tables.append((i, 17 + total))
        i += 17 + total
    if not tables:
        return data
    t0, tl = tables[int(rng.integers(len(tables)))]
    if counts is None:
        counts = [0] * 16
        budget = int(rng.integers(1, 257))
        while budget > 0:  # a few lengths get large counts: most draws oversubscribe the code space
            l = int(rng.integers(16))
            c = min(budget, int(rng.integers(1, 256)), 255 - counts[l])
            counts[l] += c
            budget -= max(c, 1)
    total = sum(counts)
    new_table = bytes([body[t0]]) + bytes(counts) + bytes(rng.integers(0, max_value + 1, total, dtype=np.uint8))
    body[t0:t0 + tl] = new_table
    seg = b"\xFF\xC4" + (len(body) + 2).to_bytes(2, "big") + bytes(body)
    return data[:pos] + seg + data[pos + ln:]


def mutate(data: bytes, rng: np.random.Generator, kind: int | None = None) -> bytes:
    f = bytearray(data)
    kind = int(rng.integers(6)) if kind is None else kind
    if kind == 5:      # DHT counts rewritten, lengths kept consistent
        return rewrite_dht(data, rng)
    sos = f.find(b"\xFF\xDA")
    body = max(sos, 0) + 12
    if len(f) <= body + 4:  # nothing behind the scan header: fall back to header damage
        kind = 3
    if kind == 0:      # bit flips in the entropy-coded data
        for _ in range(int(rng.integers(1, 20))):
            p = int(rng.integers(body, len(f) - 2))
            f[p] ^= 1 << int(rng.integers(8))
    elif kind == 1:    # truncation
        f = f[:int(rng.integers(body, len(f)))]
    elif kind == 2:    # stray markers / 0xFF bytes
        for _ in range(int(rng.integers(1, 6))):
            p = int(rng.integers(body, len(f) - 2))
            f[p] = 0xFF
            f[p + 1] = int(rng.integers(256))
    elif kind == 3:    # header damage (lengths, table classes, sampling factors, dimensions ...)
        hi = min(max(body, 4), len(f))
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(2, hi)) if hi > 2 else 0
            f[p] = int(rng.integers(256))
    else:              # random garbage after the headers
        f[body:] = bytes(rng.integers(0, 256, len(f) - body, dtype=np.uint8))
    return bytes(f)


def rst_damage(data: bytes, rng: np.random.Generator) -> bytes:
    """Restart markers renumbered, destroyed, turned into invalid codes or into stuffed bytes (a file without any: bit flips)."""
    b = bytearray(data)
    sos = b.find(b"\xFF\xDA")
    idx = [i for i in range(max(sos, 0), len(b) - 1) if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7]
    if not idx:
        return mutate(data, rng, 0)
    for _ in range(int(rng.integers(1, 4))):
        i = idx[int(rng.integers(len(idx)))]
        how = int(rng.integers(4))
        if how == 0:
            b[i + 1] = 0xD0 + int(rng.integers(8))
        elif how == 1:
            b[i] = int(rng.integers(255))
        elif how == 2:
            b[i + 1] = int(rng.integers(1, 0xC0))
        else:
            b[i + 1] = 0x00
    return bytes(b)
