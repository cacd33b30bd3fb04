#!/bin/bash
# the hipIpc all-reduce's workgroup count (AVAE_IPC_BLOCKS) on the N = 2 one-GPU rehearsal: us per all-reduce (eager events) and ms per step
cd /root/repo
for NB in 32 64 128 256; do
  AVAE_IPC_BLOCKS=$NB AVAE_BENCH_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2962$((NB/64 % 10)) \
    bench.py --gpus 2 --steps 320 --warmup 32 --repeats 3 --comm ipc --comm-buckets 1 --wire fp32 --kernel-steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['collective']
print('blocks $NB: ms/step', d['ms_per_step'], 'allreduce us', c.get('allreduce_us'), 'one-rank pipeline', c.get('pipeline_one_rank_ms_per_step'), 'exposed', c.get('exposed_us_per_step'))"
done
