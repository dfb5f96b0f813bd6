#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/prof.sh pmc ...) of
`bench.py --steps 3 --warmup 1`:  python tools/make_traffic.py <fetch_dir> <write_dir> <batch>"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
STAGE_OF = {"fast_march": "fast_score_nms", "blur_march": "gauss_blur7", "orb_orient_describe": "orb_describe",
            "knn2": "hamming_knn2", "ratio_compact": "ratio_compact", "orb_select": "select_harris_angle",
            "resize_march": "pyramid_resize", "resize_strip": "pyramid_resize", "pyramid_image": "pyramid_resize"}
STEPS = 4  # 1 warm-up + 3 timed steps in each pmc run


def per_stage(d):
    out = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, st in STAGE_OF.items():
                if key in r["Kernel_Name"]:
                    out[st] += float(r["Counter_Value"]) / STEPS
                    break
    return out


fetch, write, batch = per_stage(sys.argv[1]), per_stage(sys.argv[2]), int(sys.argv[3])
out = {
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) of `python bench.py --steps 3 "
            "--warmup 1`, summed over the stage's kernels, per step (= per launch for single-launch stages). gfx950 FETCH_SIZE "
            "reports half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM); calibrated here on the blur kernel, whose "
            "4-byte-per-lane reads of ~1.12x its algorithmic bytes read 0.54x: hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE.",
    "config": {"width": 640, "height": 480, "nfeatures": 2000, "batch": batch},
    "stages": {st: {"FETCH_SIZE_KiB": round(fetch[st], 1), "WRITE_SIZE_KiB": round(write[st], 1),
                    "hbm_bytes_per_step": int((2 * fetch[st] + write[st]) * 1024)} for st in sorted(fetch)},
}
(ROOT / "profiles" / "traffic.json").write_text(json.dumps(out, indent=1))
print({k: v["hbm_bytes_per_step"] for k, v in out["stages"].items()})
