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
from vae_assoc_amd.parallel import GradSync, dp_train_step, dp_train_step_bucketed

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


def oracle_buckets():
    """[decoder-side ranges, encoder-side ranges] in the ORACLE's flat layout (the library's own ranges, in its internal layout,
    come from avae_dp_plan): per modality [enc..., heads | dec..., out], the cost slot after the last range of bucket 0."""
    b0, b1, off = [], [], 0
    for na in ARCHS:
        n_enc = sum(int(np.prod(s)) for n, s in O.layer_shapes(na) if n.startswith("enc_"))
        n_dec = sum(int(np.prod(s)) for n, s in O.layer_shapes(na) if n.startswith("dec_"))
        b1.append((off, n_enc))
        b0.append((off + n_enc, n_dec))
        off += n_enc + n_dec
    b0[-1] = (b0[-1][0], b0[-1][1] + 1)
    return [b0, b1]


class BucketedReplica(OracleReplica):
    """the bucketed seam of parallel.dp_train_step_bucketed on the CPU oracle: a bucket's ranges are published by its backward part"""

    def _stage(self, X, eps):
        self._X, self._eps = X, eps
        self.flat.zero_()

    def _backward_bucket(self, b):
        c, g, _ = self.model.cost_and_grads(self._X, self._eps, batch_global=self.batch_global)
        full = np.concatenate([g, [c]])
        for off, cnt in oracle_buckets()[b]:
            self.flat[off:off + cnt] = torch.from_numpy(full[off:off + cnt])

    def _apply_bucket(self, b, want_cost):
        if b == 1:                      # Adam is element-wise: applying per bucket = applying once all buckets have arrived
            self.model.apply_gradients(self.flat[:-1].numpy())
        return float(self.flat[-1]) if want_cost else None


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
        # the same steps through the bucketed schedule (two all-reduces of ranges, started as the ranges appear)
        rep2 = BucketedReplica(bl, B_GLOBAL, p0)
        costs2 = [dp_train_step_bucketed(rep2, sync, oracle_buckets(), sync.shard(X, bl), eps[s][lo:hi]) for s in range(3)]
        assert costs2 == costs and np.array_equal(rep2.model.get_params(), rep.model.get_params())
        # the ncclUniqueId bootstrap: rank 0's bytes on every rank
        got = sync.broadcast_bytes(bytes(range(128)) if rank == 0 else b"", 128)
        assert got == bytes(range(128))
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


# ----------------------------------------------------------------------------- train() under data parallelism
N_ROWS, EPOCHS = 64, 3          # -> 51 training rows after the 80/10/10 split: 4 global batches of 12 per epoch, reshuffles on wrap


def _train_data():
    rng = np.random.default_rng(77)
    data = np.concatenate(synth_batch(rng, N_ROWS, [a["n_input"] for a in ARCHS], BIN), axis=1)
    eps = rng.standard_normal((64, B_GLOBAL, 5))
    p0 = O.flatten_params(ARCHS, O.init_params(ARCHS, np.random.default_rng(1)))
    return data, eps, p0


class TrainReplica(OracleReplica):
    """partial_fit / evaluate_cost as vae_assoc.AssocVariationalAutoEncoder offers them under data_parallel=True: local rows
    in, global-batch cost out; eps of step s = rows [lo, hi) of the global eps of that step."""

    def __init__(self, sync, eps_all, p0):
        bl = B_GLOBAL // sync.world_size
        OracleReplica.__init__(self, bl, B_GLOBAL, p0)
        self._sync, self.eps_all, self.step = sync, eps_all, 0
        self.lo, self.hi = sync.local_rows(bl)

    def partial_fit(self, X):
        c = dp_train_step(self, self._sync, X, self.eps_all[self.step][self.lo:self.hi])
        self.step += 1
        return c

    def evaluate_cost(self, X):
        c, _, _ = self.model.cost_and_grads(X, self.eps_all[-1][self.lo:self.hi], batch_global=self.batch_global)
        return self._sync.sum_scalar(c, "cpu")


def _train_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from vae_assoc_amd import dataset
        from vae_assoc_amd.vae_assoc import train_loop
        data, eps, p0 = _train_data()
        np.random.seed(5)                                   # same split on every rank
        ds = dataset.construct_datasets(data.copy())
        np.random.seed(1000 + rank)                         # ranks arrive with DIFFERENT RNG states: train_loop must align them
        sync = GradSync()
        rep = TrainReplica(sync, eps, p0)
        _m, hist = train_loop(rep, ds, ARCHS, B_GLOBAL // WORLD, training_epochs=EPOCHS, display_step=100, early_stop=2, sync=sync)
        seed_after = int(np.random.randint(0, 2 ** 31 - 1))
        np.savez(os.path.join(out_dir, "t%d.npz" % rank), hist=np.array(hist), params=rep.model.get_params(), steps=rep.step,
                 seed_after=seed_after)
        # ranks that hold different data are refused
        bad = dataset.construct_datasets(data.copy() + (0.5 if rank else 0.0), shuffle=False)
        try:
            train_loop(TrainReplica(sync, eps, p0), bad, ARCHS, B_GLOBAL // WORLD, training_epochs=1, sync=sync)
            raised = False
        except ValueError:
            raised = True
        assert raised
    finally:
        dist.destroy_process_group()


def test_train_loop_shards_global_batches(tmp_path):
    """train() under data parallelism must walk ONE shuffled data set in global batches of world*B rows, rank r
    taking rows [r*B, (r+1)*B) of each -- the run then equals the single-process run with batch_size = world*B: same number of
    steps, same avg_cost_hist (global-batch cost, weight B_global/n_samples), same weights."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_train_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r = [np.load(os.path.join(str(tmp_path), "t%d.npz" % k)) for k in range(WORLD)]
    assert np.array_equal(r[0]["params"], r[1]["params"]) and np.array_equal(r[0]["hist"], r[1]["hist"])
    assert int(r[0]["seed_after"]) == int(r[1]["seed_after"])            # the ranks' shuffles stayed aligned
    # single process, global batch: the oracle's own restatement of the reference loop (vae_assoc.py:498-583), seeded like rank 0
    from vae_assoc_amd import dataset
    data, eps, p0 = _train_data()
    np.random.seed(5)
    ds = dataset.construct_datasets(data.copy())
    np.random.seed(1000)
    np.random.seed(int(np.random.randint(0, 2 ** 31 - 1)))               # what train_loop broadcasts from rank 0
    ref, ref_hist = O.train(ds, ARCHS, binary=BIN, weights=W, assoc_lambda=LAM, learning_rate=LR, batch_size=B_GLOBAL,
                            training_epochs=EPOCHS, early_stop=2, params_flat=p0, eps_fn=lambda st: eps[st])
    assert int(r[0]["steps"]) == len(ref_hist) == EPOCHS * (ds.train._data.shape[0] // B_GLOBAL)
    assert np.allclose(r[0]["hist"], ref_hist, rtol=1e-11)
    assert np.abs(r[0]["params"] - ref.get_params()).max() < 1e-11


def test_gradsync_requires_process_group():
    import pytest
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    with pytest.raises(RuntimeError):
        GradSync()
