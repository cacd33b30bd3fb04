"""The library-owned gradient collectives on the GPU box (the reference has none, vae_assoc.py:66).

AVAE_COMM_IPC -- the hand-written one-shot all-reduce over hipIpc peers (avae_comm.hip) -- is exercised for real with 2 processes
(training) and 2 / 4 processes (the exchange kernel alone) sharing the one MI355X (hipIpc works between processes on one device; RCCL refuses two ranks on one device): exported
exchange blocks, flag hand-shake, shard indexing, fp32 and bf16 wire, one and two buckets, single steps and captured runs of 16,
against (a) the torch.distributed (gloo) collective over the same buckets -- bitwise for two ranks on the fp32 wire -- and (b) the
single-replica run at the global batch.  What a one-GPU box cannot show is xGMI itself: peers on other devices."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import make_arch, synth_batch

pytestmark = pytest.mark.gpu

ARCHS = [make_arch("image", 784, 500, 500, 20), make_arch("joint", 147, 200, 200, 20)]       # the C2 / C3 nets
KW = dict(binary=[True, False], transfer_fct="relu", weights=[50.0, 1.0], assoc_lambda=8.0, seed=3)
B_LOC, STEPS, RUN = 64, 3, 20


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(world):
    rng = np.random.default_rng(29)
    n = B_LOC * world
    X = synth_batch(rng, n * (STEPS + RUN), [784, 147], [True, False])
    eps = rng.standard_normal((STEPS + RUN, n, 20)).astype(np.float32)
    return X, eps


def _shard(X, eps, world, rank, s):
    """rows of `rank` inside global batch s"""
    n = B_LOC * world
    lo, hi = s * n + rank * B_LOC, s * n + (rank + 1) * B_LOC
    return [x[lo:hi] for x in X], eps[s][rank * B_LOC:(rank + 1) * B_LOC]


def _worker(rank, world, port, out_dir, variants):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AVAE_IPC_TIMEOUT_MS="20000", AVAE_IPC_BLOCKS="64")    # (several ranks share ONE GPU here: their spinning exchange kernels must all be resident)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        g.build()
        from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
        X, eps = _data(world)
        for name, kw in variants:
            m = AssocVariationalAutoEncoder(ARCHS, batch_size=B_LOC, device=0, data_parallel=True, **dict(KW, **kw))
            assert m._comm == kw.get("comm", "torch"), (name, m._comm)          # no silent fallback to torch.distributed
            assert m._cfg.batch_global == B_LOC * world and m._cfg.row_offset == rank * B_LOC
            costs = []
            for s in range(STEPS):
                xs, es = _shard(X, eps, world, rank, s)
                costs.append(m.partial_fit(xs, es))
            run_x = [np.concatenate([_shard(X, eps, world, rank, STEPS + i)[0][k] for i in range(RUN)]) for k in range(2)]
            run_e = np.concatenate([_shard(X, eps, world, rank, STEPS + i)[1] for i in range(RUN)])
            m.partial_fit_steps(run_x, RUN, run_e, return_cost=False)            # 16 in one captured graph + 4
            hist = m.cost_history(STEPS + RUN)
            assert np.array_equal(hist[:STEPS], np.array(costs, dtype=np.float32))
            np.savez(os.path.join(out_dir, "%s_r%d.npz" % (name, rank)), hist=hist, params=m.get_params())
            del m
            dist.barrier()
    finally:
        dist.destroy_process_group()


def _single(world, dtype):
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    X, eps = _data(world)
    n = B_LOC * world
    full = AssocVariationalAutoEncoder(ARCHS, batch_size=n, compute_dtype=dtype, device=0, **KW)
    full.partial_fit_steps(X, STEPS + RUN, eps.reshape(-1, 20), return_cost=False)
    return full.cost_history(STEPS + RUN), full.get_params()


def _load(out_dir, name, world):
    return [np.load(os.path.join(out_dir, "%s_r%d.npz" % (name, k))) for k in range(world)]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ipc_allreduce_two_processes(tmp_path, dtype):
    """N = 2 on one MI355X: the IPC all-reduce (two buckets, and as ONE all-reduce of the whole buffer) is bitwise the gloo
    collective on the fp32 wire (a two-term sum has one order); on the bf16 wire the replicas stay bit-identical and the costs stay
    within north_star's 1e-3 of the fp32-wire run."""
    variants = [("torch", dict(comm="torch", compute_dtype=dtype)),
                ("ipc2", dict(comm="ipc", compute_dtype=dtype)),
                ("ipc1", dict(comm="ipc", comm_buckets=1, compute_dtype=dtype)),
                ("ipc2_bf16", dict(comm="ipc", wire_dtype="bf16", compute_dtype=dtype))]
    mp.spawn(_worker, args=(2, _port(), str(tmp_path), variants), nprocs=2, join=True)
    ref_hist, ref_params = _single(2, dtype)
    t = _load(str(tmp_path), "torch", 2)
    for name in ("ipc2", "ipc1"):
        r = _load(str(tmp_path), name, 2)
        assert np.array_equal(r[0]["params"], r[1]["params"]) and np.array_equal(r[0]["hist"], r[1]["hist"]), name
        assert np.array_equal(r[0]["params"], t[0]["params"]) and np.array_equal(r[0]["hist"], t[0]["hist"]), name
    tol = 1e-5 if dtype == "fp32" else 3e-4
    assert np.allclose(t[0]["hist"], ref_hist, rtol=tol)
    assert np.abs(t[0]["params"] - ref_params).max() <= (5e-4 if dtype == "fp32" else 2.5e-2)
    w = _load(str(tmp_path), "ipc2_bf16", 2)
    assert np.array_equal(w[0]["params"], w[1]["params"]) and np.array_equal(w[0]["hist"], w[1]["hist"])
    rel = np.abs(w[0]["hist"] - t[0]["hist"]) / np.abs(t[0]["hist"])
    assert rel.max() <= 1e-3, "bf16 wire: cost drift %.2e over %d steps" % (rel.max(), STEPS + RUN)
    assert rel[0] == 0.0                                  # the first step's cost is the pre-update forward pass: untouched by the wire


@pytest.mark.parametrize("comm", ["ipc", "library"])
@pytest.mark.parametrize("kw", [dict(), dict(comm_buckets=1)])
def test_one_rank_collective_is_the_plain_step(comm, kw):
    """comm='ipc' / 'library' with no torch.distributed at all: a one-rank exchange (the sum over one rank is the identity) -- the C
    ABI's collective has no dependency on torch; 3 single steps + a run of 18 are bitwise the plain run's, with two buckets and with
    the single all-reduce."""
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    rng = np.random.default_rng(5)
    archs = [make_arch("image", 784, 64, 48, 20), make_arch("joint", 147, 40, 32, 20)]
    X = synth_batch(rng, 18 * 32, [784, 147], [True, False])
    res = []
    for c in (None, comm):
        m = AssocVariationalAutoEncoder(archs, batch_size=32, compute_dtype="bf16", device=0, comm=c, **dict(KW, **(kw if c else {})))
        assert m._comm_lib == (c is not None) and len(m._buckets) == (2 if not (c and kw) else 1)
        costs = [m.partial_fit([x[i * 32:(i + 1) * 32] for x in X]) for i in range(3)]
        m.partial_fit_steps(X, 18, return_cost=False)
        res.append((costs, m.cost_history(21).copy(), m.get_params()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_bf16_wire_on_the_rccl_path_one_rank():
    """wire_dtype='bf16' with the RCCL backend (one rank): pack -> ncclAllReduce(bf16) + the cost in fp32 -> unpack.  The reduced
    gradient is the local one rounded to bf16: the first cost is exact, the weights move as Adam moves them on a gradient with
    2^-9 relative noise."""
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    rng = np.random.default_rng(6)
    X = synth_batch(rng, 4 * 64, [784, 147], [True, False])
    ms = [AssocVariationalAutoEncoder(ARCHS, batch_size=64, compute_dtype="fp32", device=0, comm=c, wire_dtype=w, **KW)
          for c, w in ((None, "fp32"), ("library", "bf16"))]
    hist = []
    for m in ms:
        m.partial_fit_steps(X, 4, return_cost=False)
        hist.append(m.cost_history(4).copy())
    g0, g1 = ms[0].get_grads().astype(np.float64), ms[1].get_grads().astype(np.float64)
    assert hist[0][0] == hist[1][0]
    assert np.abs(hist[0] - hist[1]).max() <= 1e-3 * np.abs(hist[0]).max()
    # the last step's gradients: bf16-rounded on the wire (and computed from slightly different weights)
    assert np.abs(g1 - g0).max() <= 2e-2 * np.abs(g0).max()
    as_bf16 = torch.as_tensor(g1.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(as_bf16, g1.astype(np.float32)), "what Adam consumed is exactly representable in bf16"


def _vector_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AVAE_IPC_TIMEOUT_MS="20000", AVAE_IPC_BLOCKS=str(256 // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import __graft_entry__ as g
        g.build()
        from vae_assoc_amd import _capi
        from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
        L = _capi.lib()
        worst = {}
        # C2's nets (6.13 MB), a net whose whole buffer is a few granules (most workgroups have an empty chunk of every shard and
        # still have to signal), and three modalities of odd widths
        shapes = {"c2": ARCHS,
                  "tiny": [make_arch("a", 5, 3, 2, 1), make_arch("b", 3, 2, 2, 1)],
                  "odd": [make_arch("a", 333, 0, 0, 7, n_hidden=[129, 65, 31]), make_arch("b", 17, 0, 0, 7, n_hidden=[9]), make_arch("c", 250, 0, 0, 7, n_hidden=[77, 200])]}
        for shape, wire, nb in [(sh, wi, n) for sh in shapes for wi in ("fp32", "bf16") for n in (1, 2)]:
            if True:
                kw = dict(KW)
                if len(shapes[shape]) != 2:
                    kw.update(binary=[True, False, False], weights=[1.0, 1.0, 1.0])
                m = AssocVariationalAutoEncoder(shapes[shape], batch_size=32, compute_dtype="bf16", device=0, data_parallel=True, comm="ipc", comm_buckets=nb,
                                                wire_dtype=wire, **kw)
                n_el = m._grad_view.numel()
                srcs = [np.random.default_rng(100 + r).standard_normal(n_el).astype(np.float32) for r in range(world)]
                want = np.sum([s.astype(np.float64) for s in srcs], axis=0)
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for rep in range(3):                       # the sequence numbers / flags / slots are reused call after call
                    m._grad_view.copy_(torch.as_tensor(srcs[rank]))
                    torch.cuda.synchronize()
                    dist.barrier()
                    for b in range(nb):
                        assert L.avae_comm_allreduce(m._h, b, st) == 0
                    m.synchronize()
                    got = m._grad_view.cpu().numpy().astype(np.float64)
                    d = np.abs(got - want)
                    tol = 1e-6 if wire == "fp32" else 8e-3          # bf16: 2^-9 per term of the sum, once more on the rounded result
                    assert d[-1] <= 1e-6 * max(1.0, abs(want[-1])), "the cost slot always travels as fp32"
                    assert d.max() <= tol * np.abs(want).max(), (world, shape, wire, nb, rep, float(d.max()), int(np.argmax(d)))
                    worst["%s_%s_%d" % (shape, wire, nb)] = got
                del m
                dist.barrier()
        np.savez(os.path.join(out_dir, "vec_r%d.npz" % rank), **worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_allreduce_kernel_on_random_vectors(tmp_path, world):
    """The exchange kernel alone (avae_comm_allreduce) on dense random vectors, EVERY entry of the buffer checked against the fp64
    sum, three calls in a row, fp32 and bf16 wire, one and two buckets, 128 / 64 workgroups per rank.  (A real gradient buffer has
    zero pad columns; round 3's first kernel staged loads through inline asm whose data the compiler copied before it had landed --
    2 % of the granules, invisible to the training parity tests until the loops around it changed.)  The replicas' results are
    bit-identical."""
    mp.spawn(_vector_worker, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "vec_r%d.npz" % k)) for k in range(world)]
    for key in r[0].files:
        for k in range(1, world):
            assert np.array_equal(r[0][key], r[k][key]), (key, k)
