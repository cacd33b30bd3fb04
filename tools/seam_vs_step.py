#!/usr/bin/env python3
"""One submission per step (avae_train_step) vs the data-parallel seam (avae_step_backward + avae_step_apply), per step
and as runs of 16 batches staged together (avae_stage_batches + avae_step_backward_staged), on one GPU (no all-reduce)."""
import ctypes as C
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
model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, use_graph=os.environ.get("NOGRAPH") is None, **bench.HYPER)
rng = np.random.default_rng(1)
img, jnt = bench.synth(rng, 4 * B)
data = torch.as_tensor(np.concatenate([img, jnt], axis=1)).cuda()
batches = [[data[i * B:(i + 1) * B, :784], data[i * B:(i + 1) * B, 784:]] for i in range(4)]
n = 200 if cfg == "c4" else 2000
whole = [data[:, :784], data[:, 784:]]
ptrs = (C.c_void_p * 2)(whole[0].data_ptr(), whole[1].data_ptr())
lds = (C.c_int32 * 2)(931, 931)


def staged_run(nb):
    L, h, st = model._L, model._h, model._stream()
    assert L.avae_stage_batches(h, nb, ptrs, lds, None, st) == 0
    for j in range(nb):
        assert L.avae_step_backward_staged(h, j, st) == 0
        model._apply(False)


for mode in ("step", "seam", "staged", "step", "seam", "staged"):
    for i in range(20):
        if mode == "step":
            model.partial_fit(batches[i % 4], return_cost=False)
        elif mode == "seam":
            model._backward(batches[i % 4]); model._apply(False)
        else:
            staged_run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i = 0
    while i < n:
        if mode == "step":
            model.partial_fit(batches[i % 4], return_cost=False); i += 1
        elif mode == "seam":
            model._backward(batches[i % 4]); model._apply(False); i += 1
        else:
            staged_run(4); i += 4
    torch.cuda.synchronize()
    print(mode, "%.2f us/step" % ((time.perf_counter() - t0) / n * 1e6), flush=True)
