"""What ties committed counter files to the code they were taken from: a digest of the kernel sources and build flags."""
import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"


def kernel_source_hash() -> str:
    """sha256 over csrc's sources, headers and Makefile (sorted by name; generated and built files excluded)."""
    h = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir() if p.suffix in {".hip", ".h", ".cc"} or p.name == "Makefile")
    for p in files:
        h.update(p.name.encode() + b"\0")
        h.update(p.read_bytes())
    return h.hexdigest()
