#!/usr/bin/env python3
"""profiles/traffic.json (or traffic_<name>.json) from three rocprofv3 --pmc passes (counters only, one counter set per
pass: FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -- tools/prof_round.sh) of
`bench.py --steps 3 --warmup 1 --blur-inline --no-pipeline` (in line: a kernel's counters do not depend on what runs beside it, and the
run then has exactly 4 steps):
    python tools/make_traffic.py <fetch_dir> <write_dir> <valu_dir> <batch> [width height nfeatures [name]]"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vision_slam_frontend_amd.buildinfo import kernel_source_hash  # noqa: E402
STAGE_OF = {"fast_march": "fast_score_nms", "blur_march": "gauss_blur7", "blur_mma": "gauss_blur7",
            "pyramid_slab": "pyramid_resize", "orb_orient_describe": "orb_describe",
            "knn2": "hamming_knn2", "ratio_compact": "ratio_compact", "orb_select": "select_harris_angle",
            "resize_march": "pyramid_resize", "resize_strip": "pyramid_resize", "pyramid_image": "pyramid_resize",
            "stereo_residual": "frontend_tail", "stereo_thresholds": "frontend_tail", "stereo_filter": "frontend_tail",
            "sort_trim": "frontend_tail", "vision_features": "frontend_tail", "pack_offsets": "frontend_tail",
            "pack_copy": "frontend_tail"}
STEPS = 4  # 1 warm-up + 3 timed steps in each pmc run


def per_stage(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, st in STAGE_OF.items():
                if key in r["Kernel_Name"]:
                    out[r["Counter_Name"]][st] += float(r["Counter_Value"]) / STEPS
                    break
    return out


fetch, write, valu = per_stage(sys.argv[1])["FETCH_SIZE"], per_stage(sys.argv[2])["WRITE_SIZE"], per_stage(sys.argv[3])
batch = int(sys.argv[4])
w, h, nf = (int(v) for v in sys.argv[5:8]) if len(sys.argv) >= 8 else (640, 480, 2000)
name = sys.argv[8] if len(sys.argv) >= 9 else ""
out = {
    "note": "rocprofv3 --pmc passes (separate runs, counters only) of `python bench.py --steps 3 --warmup 1 --blur-inline --no-pipeline`, summed over the "
            "stage's kernels, per step (= per launch for single-launch stages).  gfx950 FETCH_SIZE reports half the bytes of a "
            "coalesced stream (MI355X_MICROARCH.md, HBM); calibrated on the blur kernel, whose 4-byte-per-lane reads of ~1.12x "
            "its algorithmic bytes read 0.54x: hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB units).  valu_wave_insts = "
            "SQ_INSTS_VALU: wave64 VALU instructions issued (SQ_ACTIVE_INST_VALU, in quad-cycles, equals it); what one costs a "
            "SIMD, per opcode, is measured in profiles/r04/valu_issue_table.json and priced per kernel in valu_ceiling.json.  mfma_busy_cycles = "
            "SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs (a 32x32 MFMA of 8 passes holds its SIMD's matrix pipe 32 cycles).",
    # the kernels these counters belong to: bench.py publishes them only while csrc/ still hashes to this
    "source_hash": kernel_source_hash(),
    "config": {"width": w, "height": h, "nfeatures": nf, "batch": batch},
    "stages": {st: {"FETCH_SIZE_KiB": round(fetch[st], 1), "WRITE_SIZE_KiB": round(write[st], 1),
                    "hbm_bytes_per_step": int((2 * fetch[st] + write[st]) * 1024),
                    "valu_wave_insts_per_step": int(valu["SQ_INSTS_VALU"].get(st, 0)),
                    "valu_active_quad_cycles_per_step": int(valu["SQ_ACTIVE_INST_VALU"].get(st, 0)),
                    "mfma_busy_cycles_per_step": int(valu["SQ_VALU_MFMA_BUSY_CYCLES"].get(st, 0)),
                    "waves_per_step": int(valu["SQ_WAVES"].get(st, 0))} for st in sorted(fetch)},
}
(ROOT / "profiles" / ("traffic_%s.json" % name if name else "traffic.json")).write_text(json.dumps(out, indent=1))
print({k: (v["hbm_bytes_per_step"], v["valu_wave_insts_per_step"]) for k, v in out["stages"].items()})
