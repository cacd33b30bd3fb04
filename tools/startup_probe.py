#!/usr/bin/env python3
"""Diagnostic: what a run of n steps costs from an IDLE GPU (the timed region of `bench.py --steps 20`): host time of the
partial_fit_steps call and time to completion, n = 4 .. 64.  Measured: completion = ~20-30 us + n x 62.4 us (C2), i.e. the driver's
20-step repeats carry ~1.3 us per step of start-up (Python marshalling, the staging node's SetParams, the first packets of the
graph launch, the closing synchronise) that the default 640-step repeats amortise."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
archs, B, dtype, _ = bench.CONFIGS["c2"]
hp = bench.hyper_for(archs)
rng = np.random.default_rng(0)
mat, edges = bench.synth_for(rng, archs, 64 * B)
data = torch.as_tensor(mat).cuda()
X = [data[:, edges[k]:edges[k + 1]] for k in range(2)]
m = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **hp)
for n in (16, 4, 20, 64): m.partial_fit_steps([x[:n * B] for x in X], n, return_cost=False)
torch.cuda.synchronize()
for n in (4, 16, 20, 32, 64):
    hs, ts = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.partial_fit_steps([x[:n * B] for x in X], n, return_cost=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print("n=%3d  host call %7.1f us   to completion %8.1f us   per step %6.2f us" % (n, np.median(hs) * 1e6, np.median(ts) * 1e6, np.median(ts) * 1e6 / n))
