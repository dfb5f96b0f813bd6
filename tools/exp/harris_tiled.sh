#!/bin/bash
# Round-6 experiment: what would a TILED unblurred pyramid save the keypoint selection?  The Harris patches
# are fetched at the addresses a tiled level would give them (k_select.hip -DVSF_EXP_HARRIS_TILED reads the tiled BLURRED
# level: wrong values, the right access pattern -- an upper bound of the saving with no writer to pay for), against the
# shipped row-major fetch: stage time in line (bench.py) and FETCH_SIZE of the orb_select kernels (rocprofv3 --pmc).
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/harris_tiled
mkdir -p $out
for v in normal exp; do
  cp tools/exp/_so/libvsf_hip_$v.so vision_slam_frontend_amd/libvsf_hip.so
  python3 bench.py --leg --steps 10 --warmup 2 > $out/bench_$v.json 2> $out/bench_$v.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$v -o pmc -- python3 bench.py --leg --blur-inline --no-pipeline --steps 3 --warmup 1 > $out/pmc_$v.json 2> $out/pmc_$v.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$v -o pmc -- python3 bench.py --leg --blur-inline --no-pipeline --steps 3 --warmup 1 > $out/pmcw_$v.json 2> $out/pmcw_$v.err
done
cp tools/exp/_so/libvsf_hip_normal.so vision_slam_frontend_amd/libvsf_hip.so
python3 - $out <<'PY'
import sys, glob, csv, json
out = sys.argv[1]
for v in ("normal", "exp"):
    j = json.loads(open(out + "/bench_%s.json" % v).read().strip().splitlines()[-1])
    tot = {}
    for kind in ("fetch", "write"):
        t = 0.0
        for f in glob.glob(out + "/%s_%s/**/*counter_collection.csv" % (kind, v), recursive=True):
            for r in csv.DictReader(open(f)):
                if "orb_select" in r["Kernel_Name"]:
                    t += float(r["Counter_Value"])
        tot[kind] = t / 8  # per step: 1 warm-up + 3 steps, then the in-line pass (1 + 3)
    print("%-6s value %.0f frames/s, ms/step %.3f, select in line %.3f ms, orb_select FETCH_SIZE %.0f KiB/step (x2 = %.2f GB), WRITE_SIZE %.0f KiB/step" % (
        v, j["value"], j["ms_per_step"], j["stages_ms_per_step_in_line"]["select_harris_angle"], tot["fetch"], 2 * tot["fetch"] * 1024 / 1e9, tot["write"]))
PY
