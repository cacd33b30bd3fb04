#!/usr/bin/env python3
"""Is the C2 step loop host-bound?  Times enqueue-only vs enqueue+drain, and profiles the host side."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder

archs, B, dtype, label = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, use_graph=os.environ.get("NOGRAPH") is None, **bench.HYPER)
rng = np.random.default_rng(1)
img, jnt = bench.synth(rng, 16 * B)
data = torch.as_tensor(np.concatenate([img, jnt], axis=1)).cuda()
batches = [[data[i * B:(i + 1) * B, :784], data[i * B:(i + 1) * B, 784:]] for i in range(16)]
for i in range(200):
    model.partial_fit(batches[i % 16], return_cost=False)
torch.cuda.synchronize()
for n in (200, 1000, 4000):
    t0 = time.perf_counter()
    for i in range(n):
        model.partial_fit(batches[i % 16], return_cost=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d enqueue %.2f us/step, total %.2f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(2000):
    model.partial_fit(batches[i % 16], return_cost=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
