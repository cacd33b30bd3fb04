#!/usr/bin/env python3
"""Sanity check of the collective the data-parallel path issues, with the RCCL backend and one rank (all this pool
offers): SUM all-reduce of the float32 view into the library's workspace, between backward and Adam, for a run of
steps -- must leave the weights bitwise equal to the single-replica run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
archs, B, dtype, _ = bench.CONFIGS["c2"]
hp = bench.hyper_for(archs)
rng = np.random.default_rng(0)
mat, edges = bench.synth_for(rng, archs, 8 * B)
data = torch.as_tensor(mat).cuda()
X = [data[:, edges[k]:edges[k + 1]] for k in range(2)]
a = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, data_parallel=True, **hp)
b = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **hp)
g = a._grad_view
print("grad view:", g.dtype, g.shape, "offset %% 256 = %d" % (g.data_ptr() % 256), "contiguous", g.is_contiguous())
L, h, st = a._L, a._h, a._stream()
for i0 in range(0, 8, 4):          # what partial_fit_steps does under data parallelism, with the collective forced
    ts, ptrs, lds, e = a._batch_args([x[i0 * B:(i0 + 4) * B] for x in X], None, 4)
    assert L.avae_stage_batches(h, 4, ptrs, lds, None, st) == 0
    for j in range(4):
        assert L.avae_step_backward_staged(h, j, st) == 0
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        a._apply(False)
b.partial_fit_steps(X, 8, return_cost=False)
torch.cuda.synchronize()
same = np.array_equal(a.get_params(), b.get_params())
print("weights equal to the single-replica run:", same, " last costs", a.cost_history(1)[0], b.cost_history(1)[0])
dist.destroy_process_group()
sys.exit(0 if same else 1)
