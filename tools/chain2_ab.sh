#!/bin/bash
# A/B of the two-link chain experiment (AVAE_CHAIN2, k_chain2) on one box: alternating bench runs, ms per step and the launches' us
cd /root/repo
for i in 1 2 3; do
  for ON in 0 1; do
    if [ $ON = 1 ]; then export AVAE_CHAIN2=1; else unset AVAE_CHAIN2; fi
    python bench.py --no-extras --no-cpu-baseline --kernel-steps 100 --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_us']
print('chain2=$ON  ms/step', d['ms_per_step'], ' '.join('%s=%.2f' % (n, k[n]) for n in k if n.startswith('fwd_enc')))"
  done
done
