#!/bin/bash
# Round-2 profiles: rocprofv3 --kernel-trace --stats of the bench workloads (C2 headline, C4, c2conv), then the PMC passes
# (separate runs, no trace domains besides the kernel trace: gpurun rule).  Outputs under gpurun_out/prof_r02/.
set -e
OUT=/root/repo/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c4 c2conv; do
  case $cfg in c2) ST="--steps 640 --warmup 100";; c4) ST="--steps 96 --warmup 16";; c2conv) ST="--steps 320 --warmup 32";; esac
  python3 /root/repo/bench.py --config $cfg $ST --repeats 1 --kernel-steps 20 --no-cpu-baseline --no-extras > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 /root/repo/bench.py --config $cfg $ST --repeats 1 --kernel-steps 0 --no-cpu-baseline --no-extras > $OUT/trace_$cfg.log 2>&1
  echo "trace $cfg done"
done
bash /root/repo/tools/pmc_traffic.sh c2 $OUT/pmc_traffic_c2 40 > $OUT/pmc_traffic_c2.log 2>&1; echo "traffic c2 done"
bash /root/repo/tools/pmc_c4.sh $OUT/pmc_c4 c4 > $OUT/pmc_c4.log 2>&1; echo "pmc c4 done"
find $OUT -name "*.csv" | head -40
