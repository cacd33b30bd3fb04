#!/bin/bash
# HBM traffic of the bench workload per kernel launch, as /opt/skills/guides/MI355X_MICROARCH.md
# ("HBM") prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots; FETCH_SIZE
# takes 3, WRITE_SIZE 2), units of KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide
# coalesced reads (doubled by tools/pmc_traffic.py).  No sys/hip trace is combined with --pmc.
# --single-step: every step is submitted with its own staging launch, so launches can be told apart by position
# (same kernels and arguments as inside the 16-step replays).
set -e
CFG=${1:-c2}
OUT=${2:-/root/repo/gpurun_out/pmc_traffic_$CFG}
STEPS=${3:-40}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 /root/repo/bench.py --config $CFG --steps 32 --warmup 8 --repeats 1 --kernel-steps 4 --no-cpu-baseline --no-extras > $OUT/order.json 2> $OUT/order.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --config $CFG --steps $STEPS --warmup 10 --repeats 1 --kernel-steps 0 --no-cpu-baseline --no-extras --single-step > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --config $CFG --steps $STEPS --warmup 10 --repeats 1 --kernel-steps 0 --no-cpu-baseline --no-extras --single-step > $OUT/write.log 2>&1
python3 /root/repo/tools/pmc_traffic.py $OUT $CFG $OUT/order.json
