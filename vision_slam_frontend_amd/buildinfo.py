"""What ties committed counter files to the code they were taken from: a digest of the kernel sources and build flags."""
import hashlib
import re
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
_COMMENT = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)


def _code_only(text: str) -> bytes:
    """The text without comments and with runs of white space collapsed: rewording a comment does not change the kernels
    (string literals in these files contain neither `//` nor `/*`)."""
    return " ".join(_COMMENT.sub(" ", text).split()).encode()


def kernel_source_hash() -> str:
    """sha256 over the CODE of csrc's sources and headers and over its Makefile (sorted by name; generated and built files
    excluded; comments and white space do not count)."""
    h = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir() if p.suffix in {".hip", ".h", ".cc"} or p.name == "Makefile")
    for p in files:
        h.update(p.name.encode() + b"\0")
        h.update(p.read_bytes() if p.name == "Makefile" else _code_only(p.read_text()))
    return h.hexdigest()
