#!/usr/bin/env python3
"""The hipIpc all-reduce kernel alone: N processes sharing one MI355X (gloo bootstrap), every rank launching only the gradient
collective (avae_comm_allreduce) back to back: microseconds per all-reduce round and a check of the sum.

  python tools/ipc_allreduce_bench.py [c2|c4] [N ...]        (N <= 6: the box's cap on processes per GPU)

On one GPU every "link" is local HBM and the ranks share its CUs, so this prices the kernel's own passes and hand-shakes -- not
xGMI -- and exercises the shard / chunk / flag indexing.  (Replicas inside ONE process do not work for this: their spinning
kernels were not run concurrently by the runtime -- rank 0 timed out waiting for rank 1, whose kernel started when rank 0's ended.)"""
import ctypes as C
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, cfg_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AVAE_IPC_TIMEOUT_MS="5000")
    os.environ.setdefault("AVAE_IPC_BLOCKS", str(max(16, 256 // world)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    g.build()
    import bench
    from vae_assoc_amd import _capi
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    archs, B, dtype, label = bench.CONFIGS[cfg_name]
    hp = bench.hyper_for(archs)
    L = _capi.lib()
    for wire in ("fp32", "bf16"):
        for nb in (1, 2):
            m = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=32, compute_dtype=dtype, seed=0, device=0, data_parallel=True,
                                            comm="ipc", comm_buckets=nb, wire_dtype=wire, **hp)
            assert m._comm == "ipc"
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            n_el = m._grad_view.numel()
            srcs = [np.random.default_rng(r).standard_normal(n_el).astype(np.float32) for r in range(world)]      # every rank can make every rank's vector
            gsrc = torch.as_tensor(srcs[rank]).cuda()
            want = torch.as_tensor(np.sum([s.astype(np.float64) for s in srcs], axis=0)).cuda()

            def round_():
                for b in range(nb):
                    _capi.check(m._h, L.avae_comm_allreduce(m._h, b, st), "avae_comm_allreduce")
            m._grad_view.copy_(gsrc)
            torch.cuda.synchronize()
            dist.barrier()
            round_()
            m.synchronize()
            d = (m._grad_view.double() - want).abs()
            err = float(d.max() / want.abs().max())
            if err > (1e-6 if wire == "fp32" else 8e-3):
                bad = torch.nonzero(d > 1e-3 * want.abs().max()).flatten()
                print("rank", rank, "N", world, wire, nb, "err", err, "bad entries", bad.numel(), "first", bad[:8].tolist(), "last", bad[-8:].tolist(),
                      "of", d.numel(), "values", m._grad_view[bad[:4]].tolist(), want[bad[:4]].tolist(), flush=True)
                raise SystemExit(1)
            iters = 300
            for _ in range(30):
                round_()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                round_()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / iters * 1e6
            m.synchronize()
            t = torch.tensor([us], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if rank == 0:
                print("%s N=%d %s wire, %d bucket(s): %8.2f us per round (%.2f MB of gradient per rank, %d workgroups per rank; sum rel err %.1e)"
                      % (cfg_name, world, wire, nb, float(t.item()), m._grad_view.numel() * 4 / 1e6, max(16, 256 // world), err), flush=True)
            del m
            dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 6]
    for N in worlds:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(worker, args=(N, port, cfg_name), nprocs=N, join=True)
