"""Data-parallel path of the HIP replicas on the GPU box: two ranks (both on cuda:0, gloo backend -- RCCL
refuses two ranks on one device) run vae_assoc_amd's own partial_fit with data_parallel=True, i.e.
avae_stage_batches / avae_dp_backward -> SUM all-reduce of the bucket's range of the gradient(+cost) view -> avae_dp_apply, and must
reproduce the single-replica global-batch run: same costs, same parameters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import make_arch, synth_batch

pytestmark = pytest.mark.gpu

ARCHS = [make_arch("image", 784, 64, 48, 20), make_arch("joint", 147, 40, 32, 20)]
# the conv/deconv image branch next to the MLP joint branch: direct, adjoint-frame and patch-matrix stages, split reductions
ARCHS_CONV = [dict(make_arch("image", 784, 8, 24, 20), hidden_conv=True, n_hidden_gener_1=24, n_hidden_gener_2=8),
              make_arch("joint", 147, 40, 32, 20)]
KW = dict(binary=[True, False], transfer_fct="relu", weights=[50.0, 1.0], assoc_lambda=8.0, seed=3)
B_LOC, WORLD, STEPS = 32, 2, 3


def _data():
    rng = np.random.default_rng(17)
    X = synth_batch(rng, B_LOC * WORLD, [784, 147], [True, False])
    eps = rng.standard_normal((STEPS, B_LOC * WORLD, 20)).astype(np.float32)
    return X, eps


def _worker(rank, port, out_dir, dtype, conv=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import __graft_entry__ as g
        g.build()
        from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
        X, eps = _data()
        ARCHS = ARCHS_CONV if conv else globals()["ARCHS"]
        m = AssocVariationalAutoEncoder(ARCHS, batch_size=B_LOC, compute_dtype=dtype, device=0, data_parallel=True, **KW)
        assert m._cfg.batch_global == B_LOC * WORLD and m._cfg.row_offset == rank * B_LOC
        lo, hi = rank * B_LOC, (rank + 1) * B_LOC
        costs = [m.partial_fit([x[lo:hi] for x in X], eps[s][lo:hi]) for s in range(STEPS)]
        ev = m.evaluate_cost([x[lo:hi] for x in X], eps[0][lo:hi])          # summed over ranks inside
        # the same steps as one run (avae_stage_batches(n) + avae_dp_backward(j, b)): must be bitwise the same
        m2 = AssocVariationalAutoEncoder(ARCHS, batch_size=B_LOC, compute_dtype=dtype, device=0, data_parallel=True, **KW)
        run_x = [np.concatenate([x[lo:hi]] * STEPS) for x in X]
        run_eps = np.concatenate([eps[s][lo:hi] for s in range(STEPS)])
        last = m2.partial_fit_steps(run_x, STEPS, run_eps)
        assert last == costs[-1] and np.array_equal(m2.get_params(), m.get_params())
        assert np.array_equal(m2.cost_history(STEPS), np.array(costs, dtype=np.float32))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), costs=np.array(costs), params=m.get_params(), ev=ev)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_hip_replicas_match_single_replica_global_batch(tmp_path, dtype):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path), dtype), nprocs=WORLD, join=True)
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    X, eps = _data()
    full = AssocVariationalAutoEncoder(ARCHS, batch_size=B_LOC * WORLD, compute_dtype=dtype, device=0, **KW)
    ref_costs = [full.partial_fit(X, eps[s]) for s in range(STEPS)]
    ref_ev = full.evaluate_cost(X, eps[0])
    r = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(WORLD)]
    assert np.array_equal(r[0]["params"], r[1]["params"])                 # replicas stay bit-identical
    assert np.array_equal(r[0]["costs"], r[1]["costs"])
    tol = 1e-5 if dtype == "fp32" else 3e-4                                # accumulation order: K=64 vs 32+32
    assert np.allclose(r[0]["costs"], ref_costs, rtol=tol)
    assert abs(float(r[0]["ev"]) - ref_ev) <= tol * abs(ref_ev)
    # fp32: weights track the global-batch run up to Adam's amplification of rounding-level gradient differences
    assert np.abs(r[0]["params"] - full.get_params()).max() <= (2e-4 if dtype == "fp32" else 7.5e-3)


def test_two_hip_replicas_with_conv_modality(tmp_path):
    """The same check with the conv/deconv image branch (its helper launches -- direct stage, adjoint-frame gradients, row / column
    sums, split-K reductions -- all finish before the all-reduce reads the gradient buffer)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path), "fp32", True), nprocs=WORLD, join=True)
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    X, eps = _data()
    full = AssocVariationalAutoEncoder(ARCHS_CONV, batch_size=B_LOC * WORLD, compute_dtype="fp32", device=0, **KW)
    ref_costs = [full.partial_fit(X, eps[s]) for s in range(STEPS)]
    r = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(WORLD)]
    assert np.array_equal(r[0]["params"], r[1]["params"])
    assert np.allclose(r[0]["costs"], ref_costs, rtol=1e-5)
    assert np.abs(r[0]["params"] - full.get_params()).max() <= 5e-4


def _train_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import __graft_entry__ as g
        g.build()
        from vae_assoc_amd import dataset, vae_assoc as V
        rng = np.random.default_rng(23)
        data = np.concatenate(synth_batch(rng, 400, [784, 147], [True, False]), axis=1)
        np.random.seed(5)
        ds = dataset.to_device(dataset.construct_datasets(data)) if rank == 0 else dataset.construct_datasets(data)   # device- and host-resident feeders
        np.random.seed(1000 + rank)
        m, hist = V.train(ds, ARCHS, binary=KW["binary"], weights=KW["weights"], assoc_lambda=KW["assoc_lambda"], batch_size=B_LOC,
                          training_epochs=3, display_step=10, early_stop=2, compute_dtype="fp32", seed=3, device=0, data_parallel=True)
        np.savez(os.path.join(out_dir, "t%d.npz" % rank), hist=np.array(hist), params=m.get_params())
    finally:
        dist.destroy_process_group()


def test_train_under_data_parallel_equals_global_batch_run(tmp_path):
    """train(..., data_parallel=True) on two replicas (per-rank batch 32) == train(batch_size=64) in one process: the ranks walk
    one shuffled data set in global batches and each takes its rows; eps comes from the in-kernel generator keyed by GLOBAL row."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_train_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd import dataset, vae_assoc as V
    rng = np.random.default_rng(23)
    data = np.concatenate(synth_batch(rng, 400, [784, 147], [True, False]), axis=1)
    np.random.seed(5)
    ds = dataset.construct_datasets(data)
    np.random.seed(1000)
    np.random.seed(int(np.random.randint(0, 2 ** 31 - 1)))               # what train_loop broadcasts from rank 0
    full, ref_hist = V.train(ds, ARCHS, binary=KW["binary"], weights=KW["weights"], assoc_lambda=KW["assoc_lambda"],
                             batch_size=B_LOC * WORLD, training_epochs=3, display_step=10, early_stop=2, compute_dtype="fp32", seed=3, device=0)
    r = [np.load(os.path.join(str(tmp_path), "t%d.npz" % k)) for k in range(WORLD)]
    assert np.array_equal(r[0]["params"], r[1]["params"]) and np.array_equal(r[0]["hist"], r[1]["hist"])
    assert len(ref_hist) == len(r[0]["hist"]) == 3 * (320 // (B_LOC * WORLD))
    assert np.allclose(r[0]["hist"], ref_hist, rtol=2e-5)
    assert np.abs(r[0]["params"] - full.get_params()).max() <= 5e-4


def test_rccl_backend_collective_on_the_gradient_view():
    """The gradient collective on the RCCL backend (one rank is all a one-GPU box offers), tools/nccl_view_check.py in a process
    of its own: (a) the library-owned communicator (ncclCommInitRank inside avae_create, ncclAllReduce per bucket on the
    library's comm stream, Adam per bucket) and (b) torch.distributed.all_reduce over the same buckets both leave weights and
    costs bitwise equal to the single-replica run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_view_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "library-owned RCCL communicator: weights and costs equal to the single-replica run: True" in r.stdout
    assert "torch-owned collective over the same buckets: weights and costs equal to the single-replica run: True" in r.stdout
