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

    def all_reduce_(self, flat):
        """In-place SUM all-reduce of the flat gradient(+cost) buffer."""
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def sum_scalar(self, value, device):
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        self.all_reduce_(t)
        return float(t.item())

    def broadcast_int(self, value, src=0):
        """rank `src`'s integer on every rank (train(): one NumPy seed for the shuffles of all ranks)"""
        t = torch.tensor([int(value)], dtype=torch.int64)
        if self.world_size > 1:
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=src, group=self.group)
        return int(t.item())

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
