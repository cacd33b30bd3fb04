#!/bin/bash
# c2conv with the direct stage's fast kernels at different workgroups-per-image (forward, dX, dF), ONE box
cd /root/repo
for sp in 1,1,1 2,2,2 4,4,4 2,4,2 1,4,1 1,1,1; do
  export AVAE_THIN_FAST_SPLIT=$sp
  python bench.py --config c2conv --steps 320 --warmup 32 --repeats 3 --kernel-steps 40 --no-extras --no-cpu-baseline > gpurun_out/conv_ab.json 2> gpurun_out/conv_ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/conv_ab.json").read().strip().splitlines()[-1])
k = d.get("kernels_us", {})
print("split $sp", "ms/step", d["ms_per_step"], " ".join("%s %.1f" % (n, v) for n, v in k.items() if "direct" in n or "sums" in n))
PY
done
