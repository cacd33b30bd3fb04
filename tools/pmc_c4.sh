#!/bin/bash
# PMC passes for the C4 (MFMA stress) config; counters collected in their own runs (gpurun rule).
set -e
OUT=${1:-/root/repo/gpurun_out/pmc_c4}
CFG=${2:-c4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/sq -- python3 /root/repo/bench.py --config $CFG --steps 6 --warmup 2 --kernel-steps 0 --no-cpu-baseline --no-extras > $OUT/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum \
  --kernel-trace --output-format csv -d $OUT/tcc -- python3 /root/repo/bench.py --config $CFG --steps 6 --warmup 2 --kernel-steps 0 --no-cpu-baseline --no-extras > $OUT/tcc.log 2>&1
echo done
