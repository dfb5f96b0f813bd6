"""Deterministic synthetic stereo streams (SURVEY.md section 8(d) "Synthetic inputs").

All arithmetic is integer (no libm), so the same bytes come out on every machine with the
same numpy bit-generator: smooth periodic background + rotated rectangles / "L" corners with
random contrast and a private cell texture + uniform +-3 noise.  The right image re-renders every
object shifted by its own disparity (2..40 px, smooth over the image) with an independent noise seed, so real stereo matches exist; the
temporal stream shifts the whole scene by (3, 1) px per frame and reseeds the noise.

There is no dataset and no network in this project: every test, the smoke run and bench.py
use these images (``data: "synthetic"``).
"""
from __future__ import annotations

import hashlib

import numpy as np

BASE_SEED = 0xC0FFEE

# (cos, sin) * 1024 for k * 22.5 degrees, k = 0..15 (integers, hard-coded: no libm involved).
_ROT_Q10 = [(1024, 0), (946, 392), (724, 724), (392, 946), (0, 1024), (-392, 946), (-724, 724), (-946, 392),
            (-1024, 0), (-946, -392), (-724, -724), (-392, -946), (0, -1024), (392, -946), (724, -724),
            (946, -392)]


def _wave(phase: np.ndarray) -> np.ndarray:
    """Smooth periodic integer wave, period 1024, range [-4096, 4096] (piecewise parabola)."""
    p = np.mod(phase, 1024)
    t = np.mod(p, 512)
    v = (t * (512 - t)) // 16
    return np.where(p < 512, v, -v)


def default_object_count(width: int, height: int) -> int:
    # 1500 objects at 640x480, ~10000 at 1920x1080.
    return int(round(1500 * (width * height) / (640 * 480) * (10000 / 10125)))


class Scene:
    """A fixed set of objects; frames and eyes are rendered from it."""

    def __init__(self, width: int, height: int, n_objects: int | None = None, seed: int = BASE_SEED):
        self.w, self.h = int(width), int(height)
        self.seed = int(seed)
        n = default_object_count(width, height) if n_objects is None else int(n_objects)
        rng = np.random.Generator(np.random.PCG64(self.seed))
        self.span_x, self.span_y = self.w + 96, self.h + 96  # objects wrap inside a padded canvas
        self.cx = rng.integers(0, self.span_x, n)
        self.cy = rng.integers(0, self.span_y, n)
        self.ha = rng.integers(3, 21, n)  # half sizes -> 6..40 px
        self.hb = rng.integers(3, 21, n)
        self.rot = rng.integers(0, 16, n)
        self.kind = rng.integers(0, 3, n)  # 0,1 rectangle; 2 "L"
        mag = rng.integers(30, 121, n)
        sgn = rng.integers(0, 2, n) * 2 - 1
        self.delta = mag * sgn
        # disparity 2..40 px, smooth in the image (ground-plane like: nearer towards the bottom) with a +-1 px
        # per-object jitter; fully independent disparities would shuffle the (heavily overlapping) objects
        # inside every 31-px patch and leave no true stereo correspondences.
        self.disp = 3 + (self.cy * 36) // self.span_y + rng.integers(-1, 2, n)
        # per-object 8x8 cell texture (cells of 3..6 px in object coordinates): makes descriptors distinctive,
        # otherwise every rectangle corner looks alike and the 0.6 ratio test rejects nearly everything.
        self.tex = rng.integers(-45, 46, (n, 8, 8))
        self.cell = rng.integers(3, 7, n)
        self.bg = [(int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(0, 1024)),
                    int(rng.integers(4, 10))) for _ in range(6)]

    def _background(self, xoff: int, yoff: int) -> np.ndarray:
        x = np.arange(self.w, dtype=np.int64)[None, :] + xoff
        y = np.arange(self.h, dtype=np.int64)[:, None] + yoff
        acc = np.zeros((self.h, self.w), dtype=np.int64)
        for fx, fy, ph, amp in self.bg:
            acc += amp * _wave(fx * x + fy * y + ph)
        return 110 + acc // 4096

    def render(self, frame_idx: int = 0, eye: int = 0) -> np.ndarray:
        """eye 0 = left, 1 = right.  Returns a C-contiguous (h, w) uint8 image (stride == w)."""
        img = self.render_clean(frame_idx, eye)
        return self.add_noise(img, frame_idx, eye)

    def add_noise(self, clean: np.ndarray, frame_idx: int, eye: int, salt: int = 0) -> np.ndarray:
        nrng = np.random.Generator(np.random.PCG64([self.seed, int(frame_idx), int(eye), 0x5EED + int(salt)]))
        img = clean + nrng.integers(-3, 4, size=clean.shape)
        return np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8))

    def render_clean(self, frame_idx: int = 0, eye: int = 0) -> np.ndarray:
        """Noise-free int64 rendering (values may leave [0,255]; add_noise clips)."""
        f = int(frame_idx)
        sx, sy = 3 * f, 1 * f
        base = self._background(sx + (2 if eye else 0), sy)
        img = base.copy()
        for i in range(len(self.cx)):
            cx = (int(self.cx[i]) - sx - (int(self.disp[i]) if eye else 0)) % self.span_x - 48
            cy = (int(self.cy[i]) - sy) % self.span_y - 48
            a, b = int(self.ha[i]), int(self.hb[i])
            r = a + b + 1
            x0, x1 = max(cx - r, 0), min(cx + r + 1, self.w)
            y0, y1 = max(cy - r, 0), min(cy + r + 1, self.h)
            if x0 >= x1 or y0 >= y1:
                continue
            c, s = _ROT_Q10[int(self.rot[i])]
            dx = np.arange(x0, x1, dtype=np.int64)[None, :] - cx
            dy = np.arange(y0, y1, dtype=np.int64)[:, None] - cy
            u = dx * c + dy * s
            v = dy * c - dx * s
            m = (np.abs(u) <= a * 1024) & (np.abs(v) <= b * 1024)
            if self.kind[i] == 2:
                m &= ~((u > 0) & (v > 0))
            cs = int(self.cell[i]) * 1024
            tex = self.tex[i][np.mod(u // cs, 8), np.mod(v // cs, 8)]
            win = img[y0:y1, x0:x1]
            win[m] = base[y0:y1, x0:x1][m] + int(self.delta[i]) + tex[m]
        return img


def stereo_pair(width: int = 640, height: int = 480, frame_idx: int = 0, seed: int = BASE_SEED,
                n_objects: int | None = None):
    sc = Scene(width, height, n_objects, seed)
    return sc.render(frame_idx, 0), sc.render(frame_idx, 1)


def stereo_stream(n_frames: int, width: int = 640, height: int = 480, seed: int = BASE_SEED,
                  n_objects: int | None = None, distinct_scenes: bool = False):
    """(n_frames, 2, h, w) uint8.  distinct_scenes=True gives every frame its own object set (seed + idx),
    which is what the throughput bench uses; False renders one scene moving (3,1) px/frame."""
    out = np.empty((n_frames, 2, height, width), dtype=np.uint8)
    sc = None if distinct_scenes else Scene(width, height, n_objects, seed)
    for f in range(n_frames):
        s = Scene(width, height, n_objects, seed + f) if distinct_scenes else sc
        out[f, 0] = s.render(0 if distinct_scenes else f, 0)
        out[f, 1] = s.render(0 if distinct_scenes else f, 1)
    return out


def bench_batch(n_frames: int, width: int = 640, height: int = 480, seed: int = BASE_SEED, n_scenes: int = 8,
                n_objects: int | None = None) -> np.ndarray:
    """(n_frames, 2, h, w) uint8 for the throughput bench: `n_scenes` independently rendered stereo scenes,
    each reused with a different circular shift and fresh noise (every frame is a distinct image with the
    keypoint statistics of a rendered one, at a fraction of the rendering cost)."""
    n_scenes = max(1, min(n_scenes, n_frames))
    scenes = [Scene(width, height, n_objects, seed + 7919 * i) for i in range(n_scenes)]
    clean = [(sc.render_clean(0, 0), sc.render_clean(0, 1)) for sc in scenes]
    out = np.empty((n_frames, 2, height, width), dtype=np.uint8)
    for f in range(n_frames):
        s, k = f % n_scenes, f // n_scenes
        dx, dy = 17 * k, 5 * k
        for eye in (0, 1):
            img = np.roll(clean[s][eye], (dy, dx), axis=(0, 1)) if k else clean[s][eye]
            out[f, eye] = scenes[s].add_noise(img, 0, eye, salt=k)
    return out


def random_descriptors(n: int, seed: int = 1234) -> np.ndarray:
    """n x 32 uint8, i.i.d. uniform (matcher micro-benchmark input)."""
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, (n, 32), dtype=np.uint8)


def adversarial_descriptors(n: int, seed: int = 4321, n_unique: int = 37) -> np.ndarray:
    """Many duplicate rows and near-duplicates: stresses the (distance, index) tie rule."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pool = rng.integers(0, 256, (n_unique, 32), dtype=np.uint8)
    d = pool[rng.integers(0, n_unique, n)].copy()
    flip = rng.integers(0, 4, n)  # 0..3 single-bit flips
    for i in range(n):
        for _ in range(int(flip[i])):
            d[i, rng.integers(0, 32)] ^= np.uint8(1 << int(rng.integers(0, 8)))
    return d


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
