# per-kernel times of tools/time_png.py under rocprofv3 (kernel trace): bash tools/exp/png_kernels.sh [n_files]
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_png
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_png -o png -- python3 $GRAFT_REPO_ROOT/tools/time_png.py ${1:-512} > $GRAFT_REPO_ROOT/gpurun_out/prof_png.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "^(synthetic|photographs)" gpurun_out/prof_png.log
find gpurun_out/prof_png -name "*kernel_stats.csv" | xargs head -4
