#!/usr/bin/env python3
"""One submission per step (avae_train_step) vs the data-parallel seam (avae_step_backward + avae_step_apply) on one GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
archs, B, dtype, label = bench.CONFIGS[cfg]
model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **bench.HYPER)
rng = np.random.default_rng(1)
img, jnt = bench.synth(rng, 4 * B)
data = torch.as_tensor(np.concatenate([img, jnt], axis=1)).cuda()
batches = [[data[i * B:(i + 1) * B, :784], data[i * B:(i + 1) * B, 784:]] for i in range(4)]
n = 200 if cfg == "c4" else 2000
for mode in ("step", "seam", "step", "seam"):
    for i in range(20):
        if mode == "step":
            model.partial_fit(batches[i % 4], return_cost=False)
        else:
            model._backward(batches[i % 4]); model._apply(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if mode == "step":
            model.partial_fit(batches[i % 4], return_cost=False)
        else:
            model._backward(batches[i % 4]); model._apply(False)
    torch.cuda.synchronize()
    print(mode, "%.2f us/step" % ((time.perf_counter() - t0) / n * 1e6), flush=True)
