#!/bin/bash
# counters of one kernel (separate --pmc passes, counters only)   usage: tools/exp/pmc_kernel.sh <outdir> <kernel pattern> [bench args]
export TMPDIR=/tmp
out=$PWD/gpurun_out/$1; pat=$2; shift; shift
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --no-observe --blur-inline --no-pipeline --steps 2 --warmup 1 "$@" > $out/p$i.json 2> $out/p$i.err
  python3 tools/exp/pmc_sum.py $out/p$i $pat
done
