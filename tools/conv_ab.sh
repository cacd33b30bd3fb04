#!/bin/bash
# A/B of an environment switch on the conv/deconv image branch (c2conv), alternating on ONE box:  tools/conv_ab.sh VAR VALUE [pairs]
VAR=$1; VAL=$2; PAIRS=${3:-2}
cd /root/repo
for p in $(seq $PAIRS); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $VAR=$VAL; else unset $VAR; fi
    python bench.py --config c2conv --steps 320 --warmup 32 --repeats 3 --kernel-steps 40 --no-extras --no-cpu-baseline > gpurun_out/conv_ab.json 2> gpurun_out/conv_ab.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/conv_ab.json").read().strip().splitlines()[-1])
k = d.get("kernels_us", {})
print("$VAR=%s" % ("$VAL" if $on else "-"), "ms/step", d["ms_per_step"], "launches", len(d["launch_order"]),
      " ".join("%s %.1f" % (n, v) for n, v in k.items() if "direct" in n or "sums" in n or "im2col" in n))
PY
  done
done
