#!/bin/bash
# Round-3 profiles: rocprofv3 --kernel-trace --stats of the bench workloads (C2 headline, C4, c2conv), then the PMC passes
# (separate runs, no trace domains besides the kernel trace: gpurun rule).  Outputs under gpurun_out/prof_r03/.
# usage: profile_r03.sh [configs...]   (default: c2 c4 c2conv)
OUT=/root/repo/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CFGS=${@:-c2 c4 c2conv}
for cfg in $CFGS; do
  case $cfg in c2) ST="--steps 640 --warmup 100";; c4) ST="--steps 96 --warmup 16";; c2conv) ST="--steps 320 --warmup 32";; esac
  python3 /root/repo/bench.py --config $cfg $ST --repeats 3 --kernel-steps 40 --no-cpu-baseline --no-extras > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  rm -rf $OUT/trace_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 /root/repo/bench.py --config $cfg $ST --repeats 1 --kernel-steps 0 --no-cpu-baseline --no-extras > $OUT/trace_$cfg.log 2>&1
  TR=$(find $OUT/trace_$cfg -name "*kernel_trace.csv" | head -1); KS=$(find $OUT/trace_$cfg -name "*kernel_stats.csv" | head -1)
  cp $KS $OUT/r03_${cfg}_kernel_stats.csv
  python3 /root/repo/tools/per_launch.py $TR $OUT/bench_$cfg.json $OUT/trace_$cfg.log > $OUT/r03_${cfg}_per_launch.json 2> $OUT/per_launch_$cfg.err
  rm -rf $OUT/pmc_traffic_$cfg
  bash /root/repo/tools/pmc_traffic.sh $cfg $OUT/pmc_traffic_$cfg 40 > $OUT/pmc_traffic_$cfg.log 2>&1 && cp $OUT/pmc_traffic_$cfg/traffic.json $OUT/r03_${cfg}_traffic.json
  echo "profiled $cfg"; tail -3 $OUT/pmc_traffic_$cfg.log
  rm -rf $OUT/trace_$cfg/*/*.db 2>/dev/null
done
rm -rf $OUT/pmc_c4; bash /root/repo/tools/pmc_c4.sh $OUT/pmc_c4 c4 > $OUT/pmc_c4.log 2>&1 && python3 /root/repo/tools/pmc_summary.py $OUT/pmc_c4 $OUT/r03_c4_pmc_summary.json > $OUT/pmc_summary.log 2>&1; echo "pmc c4 done"
# keep the merged scratch small: drop the raw traces (the summaries above are what profiles/ takes)
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
