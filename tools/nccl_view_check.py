#!/usr/bin/env python3
"""Sanity check of the gradient collective with the RCCL backend and ONE rank (all a one-GPU box offers), in a process of its own:
  a) comm='library': libavae's own communicator -- ncclUniqueId from rank 0 handed round
     by torch.distributed, ncclCommInitRank inside avae_create, per step: backward part -> ncclAllReduce of the bucket's ranges on
     the library's comm stream -> Adam per bucket (avae_host.hip::dp_step), driven by avae_train_step / avae_train_steps;
  b) comm='torch': the same buckets through torch.distributed.all_reduce on views of the library's workspace
     (avae_dp_backward / avae_dp_apply), collective forced although world_size is 1.
Both must leave the weights bitwise equal to the plain single-replica run (the sum over one rank is the identity)."""
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
mat, edges = bench.synth_for(rng, archs, 21 * B)
data = torch.as_tensor(mat).cuda()
X = [data[:, edges[k]:edges[k + 1]] for k in range(2)]
kw = dict(transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **hp)
ref = AssocVariationalAutoEncoder(archs, **kw)
ref.partial_fit_steps(X, 21, return_cost=False)
ref.partial_fit([x[:B] for x in X], return_cost=False)
ok = True

a = AssocVariationalAutoEncoder(archs, data_parallel=True, comm="library", **kw)
assert a._comm_lib and a._cfg.use_comm == 1 and len(a._buckets) == 2
print("buckets (float ranges of the gradient buffer):", a._buckets)
a.partial_fit_steps(X, 21, return_cost=False)                                # 16 + 5 staged batches, bucketed pipeline per step
a.partial_fit([x[:B] for x in X], return_cost=False)
torch.cuda.synchronize()
same = np.array_equal(a.get_params(), ref.get_params()) and np.array_equal(a.cost_history(22), ref.cost_history(22))
print("library-owned RCCL communicator: weights and costs equal to the single-replica run:", same)
ok = ok and same

b = AssocVariationalAutoEncoder(archs, data_parallel=True, comm="torch", **kw)
g = b._grad_view
print("grad view:", g.dtype, tuple(g.shape), "offset %% 256 = %d" % (g.data_ptr() % 256), "contiguous", g.is_contiguous())
L, h, st = b._L, b._h, b._stream()
for i0 in range(0, 21, 16):          # what partial_fit_steps does with a host-owned collective, the collective forced at world_size 1
    n = min(16, 21 - i0)
    ts, ptrs, lds, e = b._batch_args([x[i0 * B:(i0 + n) * B] for x in X], None, n)
    assert L.avae_stage_batches(h, n, ptrs, lds, None, st) == 0
    for j in range(n):
        b._staged_j = j
        for k, ranges in enumerate(b._buckets):
            b._backward_bucket(k)
            for off, cnt in ranges:
                dist.all_reduce(g[off:off + cnt], op=dist.ReduceOp.SUM)
        for k in range(len(b._buckets)):
            b._apply_bucket(k, False)
b._stage([x[:B] for x in X])
for k, ranges in enumerate(b._buckets):
    b._backward_bucket(k)
    for off, cnt in ranges:
        dist.all_reduce(g[off:off + cnt], op=dist.ReduceOp.SUM)
for k in range(len(b._buckets)):
    b._apply_bucket(k, False)
torch.cuda.synchronize()
same = np.array_equal(b.get_params(), ref.get_params()) and np.array_equal(b.cost_history(22), ref.cost_history(22))
print("torch-owned collective over the same buckets: weights and costs equal to the single-replica run:", same)
ok = ok and same
dist.destroy_process_group()
sys.exit(0 if ok else 1)
