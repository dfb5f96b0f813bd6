"""Deterministic input images outside the synthetic-scene family of vision_slam_frontend_amd.synth (round-3 review: every
extraction test image came from one generator).  Each targets a property of a kernel:

  plateaus        >= 64x64 blocks of 255 and of 0 (and a few mid-gray ones) with texture between them: the matrix-core blur's
                  column pass holds the row sums as f16 denormals and relies on N < 2^24.01 and on saturation above 256
  checker<p>      0 / 255 checkerboard of period p (1, 2, 7): maximal contrast at every scale of the pyramid, every FAST arc
                  either all-bright or all-dark
  blur_ties       vertical and horizontal step edges whose 7x7 fixed-point Gaussian sum N = sum k_i k_j p_ij lands EXACTLY on
                  N mod 65536 = 32768, where OpenCV's SSE2 column pass (half-to-even, columns [0, w - w%4)) and its scalar
                  tail (half-up) round differently -- on a width that is not a multiple of 4, so both rules are in use
  line_art        1-pixel lines (horizontal, vertical, diagonal, a grid) on black
  dense_texture   uniform noise of full contrast: more than 12 288 FAST candidates on level 0 at 640x480 (the largest LDS
                  selection class of k_select.hip holds 12 288, the common one 3 072)

No random state outside numpy's seeded PCG64."""
import numpy as np

GAUSS = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)  # OpenCV's 8-bit fixed-point Gaussian(7, sigma 2), sum 257


def plateaus(w=640, h=480):
    rng = np.random.Generator(np.random.PCG64(101))
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    # smooth the texture a little so that FAST finds corners, not only noise
    img = ((img.astype(np.uint16) + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, (1, 1), (0, 1))) // 4).astype(np.uint8)
    blocks = [(16, 16, 96, 128, 255), (140, 40, 64, 64, 0), (260, 200, 128, 96, 255), (40, 300, 80, 160, 0),
              (400, 60, 200, 70, 255), (420, 160, 70, 200, 0), (300, 380, 64, 64, 128), (520, 400, 100, 64, 255),
              (200, 20, 64, 64, 254), (560, 260, 64, 128, 1)]
    for x, y, bw, bh, v in blocks:
        if x + bw <= w and y + bh <= h:
            img[y:y + bh, x:x + bw] = v
    return img


def checkerboard(period, w=640, h=480):
    yy, xx = np.mgrid[0:h, 0:w]
    return (((xx // period + yy // period) & 1) * 255).astype(np.uint8)


def _tie_pairs():
    """(A, B, s): a step from value A to value B whose row sum with s of the 257 weight units on the A side is exactly 32768
    (N = 257 * 32768 = 128.5 * 65536: an exact half after the column pass over constant columns)."""
    out = []
    cum = np.cumsum(GAUSS)  # weight on the left side when the step sits after tap j
    for s in cum[:-1]:
        for a in range(256):
            rest = 32768 - int(s) * a
            if rest >= 0 and rest % (257 - int(s)) == 0 and rest // (257 - int(s)) <= 255:
                out.append((a, rest // (257 - int(s)), int(s)))
    return out


def blur_ties(w=322, h=242):
    """Vertical stripes (constant along y) whose step positions put exact rounding ties on whole columns, then the same
    pattern transposed in the lower half (ties along rows).  Returns (image, number of level-0 pixels whose exact N is a tie)."""
    pairs = _tie_pairs()
    assert pairs, "no (A, B, s) with A s + B (257 - s) = 32768"
    img = np.zeros((h, w), np.uint8)
    x = 0
    k = 0
    row = np.zeros(w, np.uint8)
    while x + 16 <= w:  # 8 columns of A, 8 columns of B: every tap position of the step occurs at some output column
        a, b, _ = pairs[k % len(pairs)]
        row[x:x + 8] = a
        row[x + 8:x + 16] = b
        x += 16
        k += 1
    row[x:] = row[x - 1] if x > 0 else 0
    img[:h // 2] = row[None, :]
    col = np.zeros(h - h // 2, np.uint8)
    y = 0
    k = 3
    while y + 16 <= len(col):
        a, b, _ = pairs[k % len(pairs)]
        col[y:y + 8] = a
        col[y + 8:y + 16] = b
        y += 16
        k += 1
    col[y:] = col[y - 1] if y > 0 else 0
    img[h // 2:] = col[:, None]
    return img, count_blur_ties(img)


def count_blur_ties(img):
    """Pixels of `img` whose exact 2-D fixed-point Gaussian sum (BORDER_REFLECT_101) is a rounding tie."""
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    h, w = img.shape
    rows = sum(GAUSS[j] * p[:, j:j + w] for j in range(7))
    n = sum(GAUSS[i] * rows[i:i + h] for i in range(7))
    return int(((n % 65536) == 32768).sum())


def line_art(w=640, h=480):
    img = np.zeros((h, w), np.uint8)
    for y in range(40, h - 40, 37):
        img[y, 35:w - 35] = 255
    for x in range(50, w - 50, 41):
        img[35:h - 35, x] = 255
    for d in range(0, min(w, h) - 80):
        img[40 + d, 40 + d] = 255
        img[h - 41 - d, 40 + d + (w - h)] = 200
    img[100:380:6, 500:600] = 180  # a comb of 1-px lines, 6 px apart
    return img


def dense_texture(w=640, h=480):
    return np.random.Generator(np.random.PCG64(7)).integers(0, 256, (h, w), dtype=np.uint8)


def all_640x480():
    """name -> image, every one 640x480 (one context geometry: they also form the batches of the batched entry points)."""
    return {"plateaus": plateaus(), "checker1": checkerboard(1), "checker2": checkerboard(2), "checker7": checkerboard(7),
            "line_art": line_art(), "dense_texture": dense_texture()}
