# counters of the PNG kernels (three passes): bash tools/exp/png_pmc.sh [n_files]  -> gpurun_out/png_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_png_*
i=0
for counters in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $R/gpurun_out/pmc_png_$i -o pmc -- python3 $R/tools/time_png.py ${1:-512} > $R/gpurun_out/pmc_png_$i.log 2>&1 || exit 1
done
cd $R
for i in 1 2 3 4; do python3 tools/pmc_summary.py gpurun_out/pmc_png_$i | grep png_; done > gpurun_out/png_pmc.txt
cat gpurun_out/png_pmc.txt
