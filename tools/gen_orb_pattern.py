#!/usr/bin/env python3
"""Build step: data/orb_pattern31.txt (the 256x4 rBRIEF sampling pattern, OpenCV's ``bit_pattern_31_``; third-party
DATA from Rublee et al., "ORB: an efficient alternative to SIFT or SURF", ICCV 2011 -- OpenCV is not vendored by the
reference and not installed here; an identical copy ships with scikit-image 0.18.3) -> a C initialiser list.

    python tools/gen_orb_pattern.py <out.inc>

Called by vision_slam_frontend_amd/csrc/Makefile and oracle/Makefile, so the product and the oracle are generated from
ONE tracked file and cannot drift apart; the sha256 of the table is checked on every build.
Row i = (x0, y0, x1, y1): descriptor bit i is  I(x0,y0) < I(x1,y1)  after rotation."""
import hashlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "data" / "orb_pattern31.txt"
SHA = "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def table():
    rows = [tuple(int(v) for v in l.split()) for l in SRC.read_text().splitlines() if l.strip() and not l.startswith("#")]
    assert len(rows) == 256 and all(len(r) == 4 and all(-128 <= v <= 127 for v in r) for r in rows)
    raw = bytes((v + 256) % 256 for r in rows for v in r)
    digest = hashlib.sha256(raw).hexdigest()
    assert digest == SHA, "data/orb_pattern31.txt does not hash to the pinned table: %s" % digest
    return rows


def main() -> int:
    rows = table()
    lines = ["/* 256 x {x0,y0,x1,y1} int8 -- GENERATED at build time by tools/gen_orb_pattern.py from",
             " * data/orb_pattern31.txt; sha256(int8 row-major) = %s */" % SHA]
    for i in range(0, 256, 4):
        lines.append("  " + " ".join("%d,%d,%d,%d," % rows[j] for j in range(i, i + 4)))
    text = "\n".join(lines) + "\n"
    for o in sys.argv[1:]:
        p = Path(o)
        if not p.exists() or p.read_text() != text:
            p.write_text(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
