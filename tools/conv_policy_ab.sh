#!/bin/bash
# A/B of the implicit-GEMM policy of the conv branch (AVAE_IMPL_POLICY, avae_host.hip::plan_memory) on one box: ms/step and per-launch us
cd /root/repo
for P in "E:fw,H:fwb,D1:fwb,DT:b" "E:fw,H:fwb,D1:fwb,DT:wb" "E:fw,H:fwb,D1:fwb,DT:" "E:fwb,H:fwb,D1:fwb,DT:fwb" "E:w,H:fwb,D1:fwb,DT:b" "E:f,H:fwb,D1:fwb,DT:b"; do
  AVAE_IMPL_POLICY="$P" python bench.py --config c2conv --steps 320 --warmup 32 --repeats 3 --no-cpu-baseline --kernel-steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$P', d['ms_per_step'], len(d['launch_order']), 'launches')
print('   ', ' '.join('%s=%.1f' % (k, v) for k, v in d['kernels_us'].items() if k not in ('_null_kernel',)))
"
done
AVAE_NO_IMPLICIT=1 python bench.py --config c2conv --steps 320 --warmup 32 --repeats 3 --no-cpu-baseline --kernel-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AVAE_NO_IMPLICIT', d['ms_per_step'])"
