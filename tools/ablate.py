#!/usr/bin/env python3
"""Diagnostic (not part of the product): builds libavae variants with parts of the GEMM main loop
removed (-DAVAE_ABL_NO_MFMA / -DAVAE_ABL_NO_DMA) into gpurun_out/abl_*/ and times the launches of
one config with each, to see which resource the loop is waiting for.  Results of the ablated
builds are numerically meaningless by construction."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
for tag, defs in (("full", []), ("no_mfma", ["-DAVAE_ABL_NO_MFMA"]), ("no_dma", ["-DAVAE_ABL_NO_DMA"])):
    out = os.path.join(ROOT, "gpurun_out", "abl_" + tag)
    shutil.rmtree(out, ignore_errors=True)
    shutil.copytree(ROOT, out, ignore=shutil.ignore_patterns("gpurun_out", ".git", "*.so", "__pycache__", "profiles", "tests"))
    src = [os.path.join(out, "vae_assoc_amd", "csrc", f) for f in ("avae_kernels.hip", "avae_host.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + defs + src
                   + ["-o", os.path.join(out, "vae_assoc_amd", "libavae.so")], check=True)
    r = subprocess.run([sys.executable, os.path.join(out, "bench.py"), "--config", cfg, "--steps", "100", "--warmup", "10",
                        "--kernel-steps", "30", "--no-cpu-baseline"], capture_output=True, text=True, cwd=out)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stderr[-500:])
        continue
    j = json.loads(line[-1])
    k = j["kernels_us"]
    print("%-8s ms/step %.3f  fwd_enc2 %.1f  fwd_dec2 %.1f  bwd_dec2 %.1f  wgrad1 %.1f" % (
        tag, j["ms_per_step"], k.get("fwd_enc2", 0), k.get("fwd_dec2", 0), k.get("bwd_dec2", 0), k.get("wgrad1", k.get("wgrad", 0))))
