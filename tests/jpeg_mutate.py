"""Seeded mutators for JPEG robustness tests (the five of round 2's one-off fuzz script): bit flips, truncation and stray
markers in the entropy-coded data, damage to the headers, garbage behind the headers."""
import numpy as np


def mutate(data: bytes, rng: np.random.Generator, kind: int | None = None) -> bytes:
    f = bytearray(data)
    kind = int(rng.integers(5)) if kind is None else kind
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
