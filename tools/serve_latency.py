#!/usr/bin/env python3
"""Latency / throughput of the rows-agnostic decoder (`generate`, SURVEY.md 8(f) rank 4: the CEM / GUI callers of the
reference decode 1 live row padded to batch_size, 10-50 times per iteration; here all rollouts go in one call of any
row count).  C2 nets; device tensors in and out (no host copies) and NumPy in / out."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
g.build()
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder

import ctypes as C
archs, B, dtype, label = bench.CONFIGS["c2"]
model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **bench.HYPER)
rng = np.random.default_rng(0)
for rows in (1, 64, 1024, 16384):
    z_np = rng.standard_normal((rows, 20)).astype(np.float32)
    z_dev = torch.as_tensor(z_np).cuda()
    for name, z in (("device tensors", z_dev), ("numpy in/out", z_np)):
        n = 300 if rows <= 1024 else 50
        for _ in range(10):
            model.generate(z)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            out = model.generate(z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print("rows %6d  %-15s %8.1f us/call  %10.0f rows/s" % (rows, name, dt * 1e6, rows / dt))

# the C ABI alone: avae_generate on preallocated outputs (what a C / C++ host pays per call; the Python rows above add two tensor
# allocations and the ctypes marshalling)
for rows in (1, 64):
    z = torch.as_tensor(rng.standard_normal((rows, 20)).astype(np.float32)).cuda()
    outs = [torch.empty((rows, int(na["n_input"])), dtype=torch.float32, device="cuda") for na in archs]
    ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    L, h, st, zp = model._L, model._h, model._stream(), z.data_ptr()
    for _ in range(20):
        L.avae_generate(h, zp, rows, ptrs, st)
    torch.cuda.synchronize()
    n = 2000
    t = time.perf_counter()
    for _ in range(n):
        L.avae_generate(h, zp, rows, ptrs, st)
    t_host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("rows %6d  C ABI, preallocated outputs: %6.1f us/call back to back (host side of a call: %.1f us)" % (rows, dt * 1e6, t_host * 1e6))
