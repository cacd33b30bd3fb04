#!/bin/bash
# A/B of an environment switch on the C4 step, alternating on ONE box:  tools/c4_ab.sh VAR VALUE [pairs]
# prints ms/step, the GEMM launches' MFMA fraction and the hidden / wgrad launch times of every run
VAR=$1; VAL=$2; PAIRS=${3:-2}
cd /root/repo
for p in $(seq $PAIRS); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $VAR=$VAL; else unset $VAR; fi
    python bench.py --config c4 --steps 60 --warmup 20 --repeats 3 --kernel-steps 40 --no-extras --no-cpu-baseline > gpurun_out/c4_ab.json 2> gpurun_out/c4_ab.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/c4_ab.json").read().strip().splitlines()[-1])
k = d.get("kernels_us", {})
print("$VAR=%s" % ("$VAL" if $on else "-"), "ms/step", d["ms_per_step"], "gemm_frac", d.get("gemm_launches", {}).get("mfma_frac"),
      " ".join("%s %.1f" % (n, k[n]) for n in ("fwd_enc2", "bwd_dec3", "fwd_enc1", "bwd_out", "fwd_out_loss", "wgrad") if n in k))
PY
  done
done
