# per-kernel times of bench.py --ingest <fmt> under rocprofv3: bash tools/exp/ingest_kernels.sh jpeg|png
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ingest_$1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ingest_$1 -o b -- python3 $R/bench.py --ingest $1 --no-cpu-baseline --no-observe --no-sustained > $R/gpurun_out/prof_ingest_$1.json 2> $R/gpurun_out/prof_ingest_$1.err
cd $R
find gpurun_out/prof_ingest_$1 -name "*kernel_stats.csv" | xargs head -24 | cut -c1-200
