"""BASELINE C3 and C5 at their GLOBAL batch (2048 rows per step) on the one MI355X of the GPU box.

The 8-GPU configurations cannot run here as 8 devices; what can run is everything but the xGMI wires: the batch of 2048 cut into
8 shards of 256 rows, each shard on its own replica built through the C ABI with batch_global = 2048 and row_offset = r * 256
(vae_assoc.py:319-371 decides which terms carry 1/B: Bernoulli recon and KL are means over the GLOBAL batch, Gaussian recon and the
association penalty are sums), the 8 gradient buffers summed, the sum applied by every replica:

  * sequentially in ONE process (8 handles; avae_stage_batches / avae_dp_backward / avae_grad_buffer / avae_dp_apply), against the
    committed oracle fixture of the B = 2048 run (tests/golden/c3_b2048.npz, c5_b2048.npz: cost, sampled gradient entries of every
    tensor, three Adam steps) and against the single-replica HIP run at batch 2048;
  * as 4 processes x 512 rows through partial_fit / train() (the box allows 6 processes on the card, so 8 x 256 cannot be
    8 processes), with the torch.distributed collective and with the library's own hipIpc all-reduce.
They stay "unmeasured on 8 GPUs"; they are no longer untested at their size."""
import ctypes as C
import importlib.util
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, concat_masks, hip_relu_masks
from oracle import vae_assoc_oracle as O

pytestmark = pytest.mark.gpu
R, B_LOC = 8, 256


def _dp():
    spec = importlib.util.spec_from_file_location("make_golden_dp", os.path.join(GOLDEN, "make_golden_dp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def V():
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd import vae_assoc
    return vae_assoc


def _replica(V, c, lam, dtype, r, world, b_loc):
    """replica r of `world`: b_loc rows of the global batch, mean terms over world * b_loc rows"""
    from vae_assoc_amd import _capi
    m = V.AssocVariationalAutoEncoder(c["archs"], binary=c["binary"], transfer_fct=c["act"], weights=c["weights"], assoc_lambda=lam,
                                      learning_rate=c["lr"], batch_size=b_loc, compute_dtype=dtype, seed=1)
    m._L.avae_destroy(m._h)
    m._cfg.row_offset, m._cfg.batch_global = b_loc * r, b_loc * world
    h = C.c_void_p()
    _capi.check(None, m._L.avae_create(C.byref(m._cfg), C.byref(h)), "avae_create")
    m._h = h
    return m


def _check_sampled(G, tag, flat, what, tol):
    ptr, idx, bad = G["sample_ptr"], G["sample_idx"], []
    for t, name in enumerate(G["names"]):
        sl = slice(int(ptr[t]), int(ptr[t + 1]))
        mx = max(float(G["gmax_" + tag][t]), 1e-30) if what == "g" else 1.0
        err = float(np.abs(flat[idx[sl]] - G[what + "sample_" + tag][sl]).max()) / mx
        if err > tol:
            bad.append((str(name), err))
    return bad


def _eight_replicas(V, name, tag, lam, dtype, ctol, gtol, ptol):
    dp = _dp()
    G = np.load(os.path.join(GOLDEN, "%s_b2048.npz" % name), allow_pickle=False)
    c, X, eps, p0 = dp.inputs(name)
    chk = [float(x.astype(np.float64).sum()) for x in X] + [float(eps.astype(np.float64).sum()), float(p0.astype(np.float64).sum())]
    assert np.allclose(chk, G["checksum"], rtol=1e-12), "the seeded inputs are not the ones the fixture was made from"
    Xd = [torch.as_tensor(x).cuda() for x in X]
    ed = torch.as_tensor(eps).cuda()
    reps = [_replica(V, c, lam, dtype, r, R, B_LOC) for r in range(R)]
    for m in reps:
        m.set_params(p0)
    nb = len(reps[0]._buckets)
    costs = []
    for s in range(dp.STEPS):
        masks = []
        for r, m in enumerate(reps):                       # local backward of every shard (both buckets)
            m._backward([x[r * B_LOC:(r + 1) * B_LOC] for x in Xd], ed[s, r * B_LOC:(r + 1) * B_LOC])
            if s == 0:
                masks.append(hip_relu_masks(m, c["archs"]))
        gsum = torch.zeros_like(reps[0]._grad_tensor())
        for m in reps:                                     # SUM over the ranks in rank order: what the collective delivers
            gsum += m._grad_tensor()
        for m in reps:
            m._grad_tensor().copy_(gsum)
        if s == 0:
            g_hip = reps[0].get_grads().astype(np.float64)
            # (a) the committed fixture: relu pre-activations within rounding of 0 get opposite decisions from two arithmetic types
            # (a handful of the 2048 x 2800 hidden units; each flips one sample's contribution to a few tensors), so the sampled
            # entries are held to a bound that catches a wrong term or scale, not a flipped unit ...
            bad = _check_sampled(G, tag, g_hip, "g", 2e-2)
            assert not bad, (name, tag, dtype, bad)
            # (b) ... and EVERY entry of every tensor is held to the plain tolerance against the oracle run that is handed the
            # kernels' relu decisions (same inputs, same arithmetic model as the fixture's run)
            ora = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], lam, c["lr"], c["B"], params_flat=p0.astype(np.float64),
                                   quant="bf16" if dtype == "bf16" else None)
            c_or, g_or, _ = ora.cost_and_grads(X, eps[0], masks=concat_masks(masks))
            assert abs(c_or - float(G["cost_" + tag])) <= 1e-9 * abs(c_or)          # the decisions do not enter the cost
            off, worst = 0, []
            for na_i, na in enumerate(c["archs"]):
                for nm, shp in O.layer_shapes(na):
                    n = int(np.prod(shp))
                    err = np.abs(g_hip[off:off + n] - g_or[off:off + n]).max() / max(np.abs(g_or[off:off + n]).max(), 1e-30)
                    if err > gtol:
                        worst.append(("m%d.%s" % (na_i, nm), float(err)))
                    off += n
            assert not worst, (name, tag, dtype, worst)
        step_costs = [m._apply() for m in reps]
        assert len(set(step_costs)) == 1
        costs.append(step_costs[0])
    assert nb == 2
    want = G["costs_" + tag]
    assert abs(costs[0] - want[0]) <= ctol * abs(want[0]), (costs, want)
    assert np.allclose(costs, want, rtol=max(ctol, 3e-4 if dtype == "bf16" else 2e-5)), (costs, want)
    ps = [m.get_params() for m in reps]
    for r in range(1, R):
        assert np.array_equal(ps[0], ps[r]), "replica %d drifted from replica 0" % r
    # Weights after three Adam steps against the fixture.  Adam normalises the step, so the few units whose relu decision differs
    # between the two arithmetic types (checked away above by handing the decisions over) move some weights of their rows by up to
    # lr per step: at least 99.5 % of every tensor's sampled entries within ptol, none further than the three steps can take it.
    ptr, idx = G["sample_ptr"], G["sample_idx"]
    for t, nm in enumerate(G["names"]):
        sl = slice(int(ptr[t]), int(ptr[t + 1]))
        d = np.abs(ps[0].astype(np.float64)[idx[sl]] - G["p3sample_" + tag][sl])
        assert (d <= ptol).mean() >= 0.995 and d.max() <= 3.2 * c["lr"], (name, tag, str(nm), float(d.max()), float((d <= ptol).mean()))
    # the single-replica HIP run at the global batch
    full = V.AssocVariationalAutoEncoder(c["archs"], binary=c["binary"], transfer_fct=c["act"], weights=c["weights"], assoc_lambda=lam,
                                         learning_rate=c["lr"], batch_size=c["B"], compute_dtype=dtype, seed=1)
    full.set_params(p0)
    fc = [full.partial_fit(Xd, ed[s]) for s in range(dp.STEPS)]
    assert np.allclose(costs, fc, rtol=1e-5 if dtype == "fp32" else 3e-4), (costs, fc)
    assert np.abs(ps[0] - full.get_params()).max() <= (2e-4 if dtype == "fp32" else 7.5e-3)
    return costs


@pytest.mark.parametrize("dtype,tag", [("fp32", "f64"), ("bf16", "bf16")])
def test_c3_global_batch_2048_as_eight_replicas(V, dtype, tag):
    """C3: 8 x 256 rows.  fp32 operands against the fp64 fixture (cost 1e-5, gradients 1e-4 of the tensor maximum); bf16 operands --
    the benchmarked arithmetic -- against the fixture of the oracle that rounds where the kernels round (5e-5 / 3e-3) and, for the
    cost, against the fp64 run at north_star's 1e-3."""
    fp32 = dtype == "fp32"
    costs = _eight_replicas(V, "c3", tag, 8.0, dtype, 1e-5 if fp32 else 5e-5, 1e-4 if fp32 else 3e-3, 2e-4 if fp32 else 7.5e-3)
    if not fp32:
        G = np.load(os.path.join(GOLDEN, "c3_b2048.npz"), allow_pickle=False)
        assert np.allclose(costs, G["costs_f64"], rtol=1e-3), (costs, G["costs_f64"])


@pytest.mark.parametrize("li", range(6))
def test_c5_global_batch_2048_lambda_sweep(V, li):
    """C5: three modalities (img + jnt + 256-d aux), fp32 operands, 8 x 256 rows, every lambda of BASELINE's sweep."""
    lam = _dp().LAMBDAS[li]
    _eight_replicas(V, "c5", "f64_lam%d" % li, lam, "fp32", 1e-5, 1e-4, 2e-4)


# ----------------------------------------------------------------------------- the same global batch across processes
def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, out_dir, name, comms):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AVAE_IPC_TIMEOUT_MS="20000", AVAE_IPC_BLOCKS="64")    # (several ranks share ONE GPU here: their spinning exchange kernels must all be resident)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        g.build()
        from vae_assoc_amd import dataset, vae_assoc as V
        dp = _dp()
        c, X, eps, p0 = dp.inputs(name)
        b = c["B"] // world
        for comm, wire in comms:
            m = V.AssocVariationalAutoEncoder(c["archs"], binary=c["binary"], transfer_fct=c["act"], weights=c["weights"],
                                              assoc_lambda=c["assoc_lambda"], learning_rate=c["lr"], batch_size=b, compute_dtype="fp32",
                                              device=0, data_parallel=True, comm=comm, wire_dtype=wire, comm_buckets=1 if comm == "ipc" else 2)
            assert m._comm == comm and m._cfg.batch_global == c["B"]
            m.set_params(p0)
            costs = [m.partial_fit([x[rank * b:(rank + 1) * b] for x in X], eps[s][rank * b:(rank + 1) * b]) for s in range(dp.STEPS)]
            np.savez(os.path.join(out_dir, "%s_%s_r%d.npz" % (comm, wire, rank)), costs=np.array(costs), params=m.get_params())
            del m
            dist.barrier()
        # train() at the global batch: every rank walks the same shuffled set, takes its 512 rows of each 2048-row batch
        data = np.concatenate(X, axis=1)
        np.random.seed(5)
        ds = dataset.construct_datasets(np.concatenate([data, data[::-1]]))          # 4096 rows -> 3276 train -> one global batch / epoch
        np.random.seed(1000 + rank)
        m, hist = V.train(ds, c["archs"], binary=c["binary"], weights=c["weights"], assoc_lambda=c["assoc_lambda"], batch_size=b,
                          training_epochs=2, display_step=10, compute_dtype="fp32", seed=3, device=0, data_parallel=True,
                          comm="ipc" if world == 2 else "torch", comm_buckets=1)
        np.savez(os.path.join(out_dir, "train_r%d.npz" % rank), hist=np.array(hist), params=m.get_params())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 2])
def test_c3_global_batch_2048_across_processes(V, tmp_path, world):
    """The C3 global batch through partial_fit / train() with real processes (the card's process cap is 6, so 8 x 256 rows cannot be
    8 processes): 4 ranks x 512 rows with the gradient summed by torch.distributed (gloo); 2 ranks x 1024 rows also with the library's
    own hipIpc all-reduce, fp32 and bf16 wire.  (The hipIpc training runs stay at two processes: with four of them time-sliced on ONE
    GPU, ranks spinning in the exchange kernel have starved their peers past a 20 s timeout -- an artefact of sharing the device that
    separate GPUs do not have; the exchange kernel alone is tested with four.)"""
    comms = [("torch", "fp32")] + ([("ipc", "fp32"), ("ipc", "bf16")] if world == 2 else [])
    mp.spawn(_rank_worker, args=(world, _port(), str(tmp_path), "c3", comms), nprocs=world, join=True)
    G = np.load(os.path.join(GOLDEN, "c3_b2048.npz"), allow_pickle=False)
    for comm, wire in comms:
        r = [np.load(os.path.join(str(tmp_path), "%s_%s_r%d.npz" % (comm, wire, k))) for k in range(world)]
        for k in range(1, world):
            assert np.array_equal(r[0]["params"], r[k]["params"]) and np.array_equal(r[0]["costs"], r[k]["costs"]), (comm, wire, k)
        tol = 1e-5 if wire == "fp32" else 1e-3
        assert np.allclose(r[0]["costs"], G["costs_f64"], rtol=tol), (comm, wire, r[0]["costs"], G["costs_f64"])
        assert r[0]["costs"][0] == pytest.approx(float(G["cost_f64"]), rel=1e-5)       # step 0: untouched by the wire format
        bad = _check_sampled(G, "f64", r[0]["params"].astype(np.float64), "p3", 2e-4 if wire == "fp32" else 4e-3)
        assert not bad, (comm, wire, bad)
    # train(): equal on every rank, and equal to the single-process run with batch_size 2048
    t = [np.load(os.path.join(str(tmp_path), "train_r%d.npz" % k)) for k in range(world)]
    for k in range(1, world):
        assert np.array_equal(t[0]["params"], t[k]["params"]) and np.array_equal(t[0]["hist"], t[k]["hist"])
    from vae_assoc_amd import dataset
    dp = _dp()
    c, X, eps, p0 = dp.inputs("c3")
    data = np.concatenate(X, axis=1)
    np.random.seed(5)
    ds = dataset.construct_datasets(np.concatenate([data, data[::-1]]))
    np.random.seed(1000)
    np.random.seed(int(np.random.randint(0, 2 ** 31 - 1)))               # what train_loop broadcasts from rank 0
    full, ref_hist = V.train(ds, c["archs"], binary=c["binary"], weights=c["weights"], assoc_lambda=c["assoc_lambda"], batch_size=c["B"],
                             training_epochs=2, display_step=10, compute_dtype="fp32", seed=3, device=0)
    assert len(ref_hist) == len(t[0]["hist"]) == 2
    assert np.allclose(t[0]["hist"], ref_hist, rtol=2e-5)
    assert np.abs(t[0]["params"] - full.get_params()).max() <= 5e-4
