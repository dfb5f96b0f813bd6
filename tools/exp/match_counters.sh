#!/bin/bash
# Counter passes over the matcher alone (tools/time_match.py): LDS and wait counters of knn2_fp4_kernel, per launch
export TMPDIR=/tmp
out=$PWD/gpurun_out/match_pmc; mkdir -p $out
i=0
for counters in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $out/p$i -o pmc -- python3 tools/time_match.py > $out/p$i.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "knn2_fp4" in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:60], r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in acc.items():
        print(key, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
