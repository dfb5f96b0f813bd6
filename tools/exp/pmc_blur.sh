#!/bin/bash
# counters of the blur kernel (separate --pmc passes, counters only)   usage: tools/exp/pmc_blur.sh <outdir> [kernel pattern]
export TMPDIR=/tmp
out=$PWD/gpurun_out/$1; pat=${2:-blur}
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/p$i.json 2> $out/p$i.err
  python3 tools/exp/pmc_sum.py $out/p$i $pat
done
