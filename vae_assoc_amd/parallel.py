"""Sample-sharded data parallelism for the assoc-VAE train step.

The reference is single-process (one tf.InteractiveSession, vae_assoc.py:66); data parallelism
is something this build adds.  Every term of the cost is a sum over batch rows (vae_assoc.py:
319-371), so the step shards by sample with no change of numerics:

  * rank r holds rows [r*B_loc, (r+1)*B_loc) of every modality and THE SAME rows of eps
    (eps is per row and shared across modalities, :90, not across rows);
  * each replica scales its mean terms (Bernoulli recon, KL) by 1/B_global and its sum terms
    (Gaussian recon :327-328, association :355-365) by 1, so the SUM over ranks of the local
    gradients equals the single-process global-batch gradient;
  * one SUM all-reduce per step of one flat fp32 buffer: the gradient with the local cost
    piggy-backed in its last element (RCCL over xGMI on GPUs; gloo in the CPU tests);
  * parameters and Adam state are replicated and stay bit-identical because every rank applies
    the same reduced gradient.

`dp_train_step` is the whole protocol; a replica only has to provide
``_backward(X, eps)``, ``_grad_tensor()`` and ``_apply()`` -- the HIP model does
(vae_assoc.AssocVariationalAutoEncoder), and so does the oracle-backed stand-in the CPU
tests use to exercise this file with world_size 2 under gloo.
"""
import torch
import torch.distributed as dist


class GradSync(object):
    def __init__(self, process_group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("data_parallel=True needs an initialised torch.distributed process group "
                               "(backend 'nccl' = RCCL on ROCm, or 'gloo' on CPU)")
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)

    @property
    def backend(self):
        return dist.get_backend(self.group)

    def _boot(self, t, device):
        """bootstrap tensors live where the backend wants them: the MODEL's device under nccl (not torch's current one)"""
        return t.to(device) if (self.backend == "nccl" and device is not None and str(device) != "cpu") else t

    def broadcast_bytes(self, data, n, src=0, device=None):
        """rank `src`'s n bytes on every rank (bootstrap of the library's RCCL communicator: the ncclUniqueId)"""
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == src:
            t = torch.tensor(list(bytes(data)), dtype=torch.uint8)
        if self.world_size > 1:
            t = self._boot(t, device)
            dist.broadcast(t, src=src, group=self.group)
        return bytes(t.cpu().tolist())

    def all_gather_bytes(self, data, n, device=None):
        """every rank's n bytes, concatenated in rank order (bootstrap of the hipIpc exchange: the exported block handles)"""
        mine = self._boot(torch.tensor(list(bytes(data)), dtype=torch.uint8), device)
        if self.world_size == 1:
            return bytes(data)
        out = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(out, mine, group=self.group)
        return b"".join(bytes(t.cpu().tolist()) for t in out)

    def all_reduce_ranges_(self, flat, ranges, async_op=False):
        """In-place SUM all-reduce of the (offset, count) float ranges of one bucket of the flat gradient buffer."""
        works = []
        if self.world_size > 1:
            for off, cnt in ranges:
                w = dist.all_reduce(flat[off:off + cnt], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                if async_op:
                    works.append(w)
        return works

    def all_reduce_(self, flat):
        """In-place SUM all-reduce of the flat gradient(+cost) buffer."""
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def sum_scalar(self, value, device):
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        self.all_reduce_(t)
        return float(t.item())

    def broadcast_int(self, value, src=0, device=None):
        """rank `src`'s integer on every rank (train(): one NumPy seed for the shuffles of all ranks)"""
        t = torch.tensor([int(value)], dtype=torch.int64)
        if self.world_size > 1:
            t = self._boot(t, device)
            dist.broadcast(t, src=src, group=self.group)
        return int(t.item())

    def all_agree(self, ok, device):
        """True on every rank iff `ok` on every rank (so that all ranks raise -- or none does)"""
        return self.sum_scalar(1.0 if ok else 0.0, device) >= self.world_size

    def local_rows(self, batch_local):
        """Row range of this rank inside the global batch."""
        return self.rank * batch_local, (self.rank + 1) * batch_local

    def shard(self, X_global, batch_local):
        lo, hi = self.local_rows(batch_local)
        return [x[lo:hi] for x in X_global]


def dp_train_step(replica, sync, X_local, eps_local=None):
    """One data-parallel step: local fwd+bwd -> SUM all-reduce of grad(+cost) -> identical Adam."""
    replica._backward(X_local, eps_local)
    sync.all_reduce_(replica._grad_tensor())
    return replica._apply()


def dp_train_step_bucketed(replica, sync, buckets, X_local, eps_local=None):
    """The same step cut into gradient buckets (a list of range lists, in the order their gradients become available):
    backward part b -> all-reduce of bucket b's ranges, started as soon as they exist and left running beside backward part
    b+1 -> Adam per bucket once its ranges have arrived.  The replica provides ``_stage(X, eps)``, ``_backward_bucket(b)``,
    ``_grad_tensor()`` and ``_apply_bucket(b, want_cost)``; the library-owned RCCL pipeline (avae_host.hip::dp_step) is this
    schedule on two HIP streams."""
    replica._stage(X_local, eps_local)
    g = replica._grad_tensor()
    pending = []
    for b, ranges in enumerate(buckets):
        replica._backward_bucket(b)
        pending.append(sync.all_reduce_ranges_(g, ranges, async_op=True))
    cost = None
    for b in range(len(buckets)):
        for w in pending[b]:
            w.wait()
        cost = replica._apply_bucket(b, b == len(buckets) - 1)
    return cost
