"""Data-parallel host path on CPU: world_size 2 over gloo.

The seam tested here -- shard rows and eps by rank, scale mean terms by 1/B_global, ONE SUM
all-reduce of the flat gradient(+cost) buffer, identical Adam on every rank -- is exactly
vae_assoc_amd/parallel.py, the code the HIP replicas use.  On this GPU-less box the replica
behind the protocol is an oracle-backed stand-in (tests may use the oracle; the product does not)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import make_arch, synth_batch
from oracle import vae_assoc_oracle as O
from vae_assoc_amd.parallel import GradSync, dp_train_step

ARCHS = [make_arch("image", 60, 20, 16, 5), make_arch("joint", 21, 12, 10, 5)]
BIN, W, LAM, LR = [True, False], [50.0, 1.0], 8.0, 1e-3
B_GLOBAL, WORLD = 12, 2


class OracleReplica(object):
    """Implements the replica protocol of parallel.dp_train_step on the CPU oracle."""

    def __init__(self, batch_local, batch_global, params):
        self.model = O.OracleAssocVAE(ARCHS, BIN, "relu", W, LAM, LR, batch_local, params_flat=params)
        self.batch_global = batch_global
        self.flat = torch.zeros(O.param_count(ARCHS) + 1, dtype=torch.float64)

    def _backward(self, X, eps):
        c, g, _ = self.model.cost_and_grads(X, eps, batch_global=self.batch_global)
        self.flat[:-1] = torch.from_numpy(g)
        self.flat[-1] = c

    def _grad_tensor(self):
        return self.flat

    def _apply(self):
        self.model.apply_gradients(self.flat[:-1].numpy())
        return float(self.flat[-1])


def _data():
    rng = np.random.default_rng(31)
    X = synth_batch(rng, B_GLOBAL, [a["n_input"] for a in ARCHS], BIN)
    eps = rng.standard_normal((3, B_GLOBAL, 5))
    p0 = O.flatten_params(ARCHS, O.init_params(ARCHS, np.random.default_rng(0))) + 0.01 * rng.standard_normal(O.param_count(ARCHS))
    return X, eps, p0


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        X, eps, p0 = _data()
        sync = GradSync()
        assert (sync.rank, sync.world_size) == (rank, WORLD)
        bl = B_GLOBAL // WORLD
        lo, hi = sync.local_rows(bl)
        rep = OracleReplica(bl, B_GLOBAL, p0)
        costs = [dp_train_step(rep, sync, sync.shard(X, bl), eps[s][lo:hi]) for s in range(3)]
        tot = sync.sum_scalar(float(rank + 1), "cpu")
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), costs=np.array(costs), params=rep.model.get_params(), tot=tot)
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_global_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    X, eps, p0 = _data()
    ref = O.OracleAssocVAE(ARCHS, BIN, "relu", W, LAM, LR, B_GLOBAL, params_flat=p0)
    ref_costs = [ref.partial_fit(X, eps[s]) for s in range(3)]
    r = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(WORLD)]
    assert np.array_equal(r[0]["params"], r[1]["params"])               # replicas stay bit-identical
    assert np.allclose(r[0]["costs"], ref_costs, rtol=1e-12)            # summed cost == global-batch cost
    assert np.array_equal(r[0]["costs"], r[1]["costs"])
    assert np.abs(r[0]["params"] - ref.get_params()).max() < 1e-12
    assert r[0]["tot"] == 3.0


def test_gradsync_requires_process_group():
    import pytest
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    with pytest.raises(RuntimeError):
        GradSync()
