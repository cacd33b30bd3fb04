#!/usr/bin/env python3
"""Headline benchmark: paired-samples/sec of the img+jnt assoc-VAE train step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c2conv|c4|c5] [--no-cpu-baseline]

A "step" is one pass of the hot path (input staging, forward, fused losses, backward,
[gradient all-reduce], Adam + shadow refresh) over one batch of synthetic paired samples that is
already resident in HBM.  Default workload = BASELINE.json configs[1] ("C2": 784-500-500 /
147-200-200, n_z=20, batch 256 per GPU, bf16 operands); under torchrun every rank runs the same
per-GPU batch (weak scaling) with one RCCL SUM all-reduce of the flat gradient per step.
Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant
kernel, measured with HIP events on the launch stream in a separate eager pass right after the
timed region) and `cpu_baseline` (the NumPy oracle timed on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16
MFMA_F32_PEAK_TF = 157.3


def arch(scope, n_in, hs, n_z):
    return dict(scope=scope, hidden_conv=False, n_hidden_recog_1=hs[0], n_hidden_recog_2=hs[min(1, len(hs) - 1)],
                n_hidden_gener_1=hs[0], n_hidden_gener_2=hs[min(1, len(hs) - 1)], n_input=n_in, n_z=n_z,
                n_hidden=list(hs))


CONFIGS = {
    # name: (archs, per-GPU batch, compute dtype, label)
    "c1": ([arch("image", 784, [500, 500], 20), arch("joint", 147, [200, 200], 20)], 100, "bf16",
           "C1 img+jnt assoc-VAE 784-500-500/147-200-200 n_z=20 batch=100"),
    "c2": ([arch("image", 784, [500, 500], 20), arch("joint", 147, [200, 200], 20)], 256, "bf16",
           "C2 img+jnt assoc-VAE 784-500-500/147-200-200 n_z=20 batch=256/GPU bf16"),
    "c4": ([arch("image", 784, [1024] * 4, 64), arch("joint", 147, [1024] * 4, 64)], 4096, "bf16",
           "C4 4x1024 MLP enc/dec n_z=64 batch=4096 bf16 (MFMA stress)"),
    # conv encoder / deconv decoder image branch (vae_assoc_ujichar_img_jnt.py:72-80, commented-out configuration) + MLP joint branch
    "c2conv": ([dict(scope="image", hidden_conv=True, n_hidden_recog_1=16, n_hidden_recog_2=64, n_hidden_gener_1=64, n_hidden_gener_2=16,
                     n_input=784, n_z=20), arch("joint", 147, [200, 200], 20)], 256, "bf16",
               "C2 with the conv/deconv image branch (16/64, 64/16), n_z=20 batch=256/GPU bf16"),
    "c5": ([arch("image", 784, [500, 500], 20), arch("joint", 147, [200, 200], 20), arch("aux", 256, [200, 200], 20)], 256, "fp32",
           "C5 img+jnt+aux(256) assoc-VAE n_z=20 batch=256/GPU fp32"),
}
HYPER = dict(binary=[True, False], weights=[50.0, 1.0], assoc_lambda=8.0, learning_rate=1e-3)   # script values


def hyper_for(archs):
    """Script hyper-parameters, extended with Gaussian weight-1 entries for modalities beyond image + joint (C5's aux)."""
    extra = len(archs) - 2
    return dict(HYPER, binary=HYPER["binary"] + [False] * extra, weights=HYPER["weights"] + [1.0] * extra)


def synth_for(rng, archs, rows):
    cols = synth(rng, rows, n_aux=archs[2]["n_input"] if len(archs) > 2 else 0)
    return np.concatenate(cols, axis=1), np.concatenate([[0], np.cumsum([a["n_input"] for a in archs])])


def synth(rng, rows, n_aux=0):
    """SURVEY.md 8d synthetic inputs: stroke-like images in [0,1] (70 % dark), z-scored joint features (and a z-scored
    aux modality for C5)."""
    img = (np.clip(rng.beta(0.25, 1.5, size=(rows, 784)), 0, 1) * (rng.random((rows, 784)) >= 0.7)).astype(np.float32)
    jnt = rng.standard_normal((rows, 147)).astype(np.float32)
    if n_aux:
        return img, jnt, rng.standard_normal((rows, n_aux)).astype(np.float32)
    return img, jnt


def dense_layers(na):
    hs, n_in, nz = na["n_hidden"], na["n_input"], na["n_z"]
    enc, prev = [], n_in
    for h in hs:
        enc.append((prev, h)); prev = h
    head = (prev, 2 * nz)
    dec, prev = [], nz
    for h in hs:
        dec.append((prev, h)); prev = h
    return enc, head, dec, (prev, n_in)


def launch_work(archs, B, es):
    """Algorithmic HBM bytes and FLOPs of every launch of one step (SURVEY.md 8d accounting:
    operands read once, results written once, compute-dtype activations, fp32 Adam state).
    Launch names mirror avae_host.hip::build_training_plan."""
    out = {}
    if any(na.get("hidden_conv") for na in archs):      # conv branch: see conv_launch_work (priced separately)
        return conv_launch_work(archs, B, es)

    def add(name, by, fl):
        b0, f0 = out.get(name, (0, 0))
        out[name] = (b0 + by, f0 + fl)
    P = 0
    wg = []                                   # (bytes, flops, params) per weight-gradient item, in launch order
    # big nets run the latent item (KL + association terms) as a launch of its own (avae_host.hip::build_training_plan)
    cd = lambda a, b: -(-a // b)
    latent_alone = (sum(cd(B, 256) * cd(na["n_input"], 64) for na in archs) >= 192 and sum(cd(B, 128) * cd(na["n_input"], 128) for na in archs) >= 192
                    and not any(na["n_input"] <= 64 for na in archs) and B > 64)
    for na in archs:
        enc, head, dec, outl = dense_layers(na)
        L = len(enc)
        nz, n_in = na["n_z"], na["n_input"]
        for k, (i, o) in enumerate(enc):
            add("fwd_enc%d" % (k + 1), (B * i + (i + 1) * o + B * o) * es, 2 * B * (i + 1) * o)
        i, o = head
        add("fwd_head", (B * i + (i + 1) * o) * es + B * o * 4 + B * nz * (4 + es), 2 * B * (i + 1) * o)
        for k, (i, o) in enumerate(dec):
            add("fwd_dec%d" % (k + 1), (B * i + (i + 1) * o + B * o) * es, 2 * B * (i + 1) * o)
        add("latent" if latent_alone else "fwd_out_loss", B * (2 + 3) * nz * 4, 0)   # latent item: mulv in, static grads out
        i, o = outl
        add("fwd_out_loss", (B * i + (i + 1) * o + B * o) * es + B * o * 4, 2 * B * (i + 1) * o)

        def dgrad(name, i, o):                                                   # dA, W, Y_prev in; dA_prev out
            add(name, (B * o + i * o + 2 * B * i) * es, 2 * B * i * o)

        def wgrad(i, o):                                                         # X, dA in (+ optimiser traffic below)
            wg.append(((B * i + B * o) * es, 2 * B * (i + 1) * o, (i + 1) * o, o))
        dgrad("bwd_out", *outl)
        for k in range(L - 1, 0, -1):
            dgrad("bwd_dec%d" % (k + 1), *dec[k])
        dgrad("bwd_dec1_latent", *dec[0])
        dgrad("bwd_head", *head)
        for k in range(L - 1, 0, -1):
            dgrad("bwd_enc%d" % (k + 1), *enc[k])
        wgrad(*outl)
        for k in range(L - 1, 0, -1):
            wgrad(*dec[k])
        wgrad(*dec[0])
        wgrad(*head)
        for k in range(L - 1, 0, -1):
            wgrad(*enc[k])
        wgrad(*enc[0])
        add("prep", B * n_in * (4 + 4 + es), 0)
        P += sum((i + 1) * o for i, o in enc + dec + [head, outl])
    # the host's launch chunking (build_training_plan): big nets keep their narrow products in launches of their own
    narrow = lambda it: it[2] // it[3] <= 64 or it[3] <= 64                       # M = in+1, N = out
    wide128 = sum(-(-(it[2] // it[3]) // 128) * -(-it[3] // 128) for it in wg if not narrow(it))
    groups = [[it for it in wg if not narrow(it)], [it for it in wg if narrow(it)]] if wide128 >= 192 else [wg]
    chunk = 32                                                                   # kMaxTnItems
    chunks = [g[c0:c0 + chunk] for g in groups for c0 in range(0, len(g), chunk)]
    for c, items in enumerate(chunks):
        suffix = str(c + 1) if len(chunks) > 1 else ""
        for by, fl, p, _ in items:
            add("wgrad" + suffix, by + p * 4, fl)                                # fp32 gradient out
    add("adam", 7 * P * 4 + P * es, 0)
    return out, P


def conv_stage_shapes(na):
    """The nine stages of a hidden_conv modality (reference vae_assoc.py:169-199 encoder, :249-291 decoder, deconv.py:107-127) as
    (name, kind, input pixels, output pixels, k, Cin, Cout): a conv multiplies per OUTPUT pixel k*k*Cin x Cout, a transposed conv
    per INPUT pixel Cin x k*k*Cout (its algorithmic work: the zero-dilated form the patch-matrix route multiplies is 4x that for
    stride 2), a dense layer is a 1-pixel conv.  SURVEY.md 8 rows A3/A4 give the same GEMM-equivalent shapes."""
    r1, r2 = int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])
    g1, g2 = int(na["n_hidden_gener_1"]), int(na["n_hidden_gener_2"])
    nz, n_in = int(na["n_z"]), int(na["n_input"])
    return [("enc1", "conv", 784, 196, 5, 1, r1), ("enc2", "conv", 196, 49, 5, r1, 2 * r1), ("enc3", "conv", 49, 9, 5, 2 * r1, r2),
            ("head", "dense", 1, 1, 1, 9 * r2, 2 * nz),
            ("dec1", "tconv", 1, 9, 3, nz, g1), ("dec2", "tconv", 9, 49, 5, g1, g1 // 2), ("dec3", "tconv", 49, 196, 5, g1 // 2, g2),
            ("dec4", "tconv", 196, 784, 5, g2, 1), ("out", "dense", 1, 1, 1, n_in, n_in)]


def conv_launch_work(archs, B, es):
    """Algorithmic FLOPs and HBM bytes of one step of a model with a conv/deconv modality, per GEMM-like launch of
    avae_host.hip::build_training_plan: MACs as conv_stage_shapes counts them (x2; forward, input gradient -- none for the first
    conv, its input is data -- and filter gradient), bytes = the stage's input and output maps and its filter once each in the
    compute type (fp32 for filter gradients and optimiser state).  The helper launches of the patch-matrix routes (im2col, col2im,
    overlap-add, split-K reductions, adjoint filter shadows, permutes, row sums) do no algorithmic work: their time counts
    against the step, their bytes are overhead."""
    out = {}

    def add(name, by, fl):
        b0, f0 = out.get(name, (0, 0))
        out[name] = (b0 + by, f0 + fl)
    P = 0
    for na in archs:
        if not na.get("hidden_conv"):
            w, p = launch_work([na], B, es)
            for k, (by, fl) in w.items():
                if k != "adam":
                    add(k, by, fl)
            P += p
            continue
        for name, kind, pin, pout, k, ci, co in conv_stage_shapes(na):
            macs = (pout * k * k * ci * co) if kind == "conv" else (pin * k * k * ci * co)      # dense: pin = pout = k = 1
            wts = k * k * ci * co + (0 if kind == "conv" else co)                                # convs carry no bias (vae_assoc.py:480-489)
            P += wts
            a_in, a_out = B * pin * ci, B * pout * co
            # launch names of the implicit-GEMM plan (round 3), then of round 2's routes (AVAE_NO_IMPLICIT=1): a step has one or the other
            fwd = {"enc1": ["conv_enc1"], "enc2": ["conv_enc2"], "enc3": ["conv_enc3"], "head": ["fwd_head"], "dec1": ["conv_dec1"],
                   "dec2": ["conv_dec2", "conv_dec2_scatter"], "dec3": ["conv_dec3", "conv_dec3_scatter"], "dec4": ["conv_dec4_direct"], "out": ["fwd_out_loss"]}[name]
            bwd = {"enc1": [], "enc2": ["conv_bwd_enc2"], "enc3": ["conv_bwd_enc3"], "head": ["bwd_head"], "dec1": ["bwd_dec1_latent", "conv_dec1_latent"],
                   "dec2": ["conv_bwd_dec2", "conv_bwd_dec2_adj"], "dec3": ["conv_bwd_dec3", "conv_bwd_dec3_adj"], "dec4": ["conv_bwd_dec4_direct"], "out": ["bwd_out"]}[name]
            wgr = "conv_dec4_wgrad_direct" if name == "dec4" else "wgrad"
            # (both names are priced: a plan holds one of them per stage -- which one is the planner's per-stage policy -- and only the
            # launches that exist in the measured step enter any sum)
            for nm in fwd:
                add(nm, (a_in + wts + a_out) * es + (a_out * 4 if name in ("head", "out") else 0), 2 * B * macs)
            for nm in bwd:
                add(nm, (a_out + wts + 2 * a_in) * es, 2 * B * macs)
            add(wgr, (a_in + a_out) * es + wts * 4, 2 * B * macs)
        add("fwd_out_loss", B * (2 + 3) * int(na["n_z"]) * 4, 0)
        add("prep", B * int(na["n_input"]) * (4 + 4 + es), 0)
    add("adam", 0, 0)
    out["adam"] = (7 * P * 4 + P * es, 0)
    return out, P


def cpu_baseline(archs, B, budget_s=10.0):
    """The CPU oracle (NumPy fp32 restatement of vae_assoc.py, kind "port") timed on this box's
    host cores on a bounded sample of the same workload: once with the BLAS pool capped at the box's CPU share for
    one GPU (16 threads: the headline `value`), once with one thread per visible core (`all_cores`)."""
    from oracle import vae_assoc_oracle as O
    rng = np.random.default_rng(20260104)
    mat, edges = synth_for(rng, archs, B)
    X = [mat[:, edges[k]:edges[k + 1]] for k in range(len(archs))]
    eps = rng.standard_normal((B, archs[0]["n_z"])).astype(np.float32)
    hp = hyper_for(archs)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def timed(threads, budget):
        m = O.OracleAssocVAE(archs, hp["binary"], "relu", hp["weights"], hp["assoc_lambda"], hp["learning_rate"], B,
                             dtype=np.float32, seed=0)
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=threads)
        except ImportError:
            limiter = None
        try:
            m.partial_fit(X, eps)                # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                m.partial_fit(X, eps)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= budget or n >= 2000:
                    break
        finally:
            if limiter is not None:
                limiter.restore_original_limits()
        return n, dt

    # with one thread per visible core (256 on the GPU box) the small GEMMs of this workload spend their time in thread
    # hand-offs and the oracle runs 3-4x slower than with the 16 threads of one GPU's CPU share: both are reported
    cores = min(visible, 16)
    n, dt = timed(cores, budget_s)
    out = {"value": round(n * B / dt, 1), "unit": "paired-samples/s", "cores": cores, "kind": "port", "impl": "numpy-openblas-fp32",
           "cores_visible": visible,
           "sample": "%d train steps of batch %d (NumPy/OpenBLAS fp32 oracle of vae_assoc.py, %d BLAS threads, %.1f s)" % (n, B, cores, dt)}
    if visible > cores:
        n2, dt2 = timed(visible, budget_s / 2)
        out["all_cores"] = {"value": round(n2 * B / dt2, 1), "cores": visible,
                            "sample": "%d steps, %d BLAS threads, %.1f s" % (n2, visible, dt2)}
    return out


GEMM_LAUNCH_PREFIXES = ("fwd_enc", "fwd_dec", "fwd_head", "fwd_out_loss", "bwd_", "wgrad", "conv_enc", "conv_dec", "conv_bwd")


def measure(name, args, world, rank, local_rank, steps, warmup, repeats, kernel_steps, dtype=None, comm=None, wire=None, local_only=False, buckets=None):
    """Builds the model of config `name`, times `repeats` x `steps` train steps (each repeat bracketed by barrier +
    synchronise; MAX over ranks per repeat), then one eager pass with per-launch HIP events.  Returns a dict."""
    import torch
    import torch.distributed as dist
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    archs, B, cfg_dtype, label = CONFIGS[name]
    dtype = dtype or cfg_dtype
    es = 2 if dtype == "bf16" else 4
    comm = comm if comm is not None else ((args.comm if args.comm != "auto" else "ipc") if (args.force_comm and world == 1) else None)
    model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, device=local_rank,
                                        seed=0, use_graph=not args.no_graph, data_parallel=world > 1 and not local_only,
                                        comm=comm, comm_buckets=buckets or args.comm_buckets or 2, wire_dtype=wire or (args.wire if args.wire != "auto" else "fp32"), **hyper_for(archs))
    if world > 1 and not local_only:
        assert model._comm == comm, "the %s collective did not come up (fell back to %s)" % (comm, model._comm)
    # resident synthetic data: 16 batches per rank (rank r owns global rows [r*B, (r+1)*B) of each global batch)
    nb = 16
    rng = np.random.default_rng(20260104 + rank)
    mat, edges = synth_for(rng, archs, nb * B)
    data = torch.as_tensor(mat).cuda()                                           # [nb*B, sum n_input], split by pointer + stride
    n_mod = len(archs)
    batches = [[data[i * B:(i + 1) * B, edges[k]:edges[k + 1]] for k in range(n_mod)] for i in range(nb)]
    whole = [data[:, edges[k]:edges[k + 1]] for k in range(n_mod)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    if local_only:                       # a one-rank measurement inside a multi-rank job: no cross-rank barriers, no MAX
        world = 1

    def run(n):
        """n train steps over the resident batches in order (eps: in-kernel Philox stream).  A single replica
        hands runs of consecutive batches over in one submission, as train() does between two reshuffles;
        under data parallelism the batches of a run are staged together and every step is backward -> all-reduce -> Adam."""
        i = 0
        while i < n:
            k = i % nb
            m = 1 if args.single_step else min(nb - k, n - i)
            if m == 1:
                model.partial_fit(batches[k], return_cost=False)
            else:
                model.partial_fit_steps([w[k * B:(k + m) * B] for w in whole], m, return_cost=False)
            i += m

    run(warmup)
    dts = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        run(steps)                       # EXACTLY `steps` steps per timed repeat
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        dts.append(dt)
    dt = float(np.median(dts))
    last_cost = float(model.cost_history(1)[0])

    pcie = None
    if args.host_input and world == 1:
        # the boundary as the reference's callers use it: numpy batches on the host, copied over PCIe every step
        hb = [[t.cpu().pin_memory() for t in b] for b in batches]
        n = max(1, steps // 4)
        for i in range(20):
            model.partial_fit(hb[i % nb], return_cost=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n):
            model.partial_fit(hb[i % nb], return_cost=False)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        pcie = {"value": round(B * n / dt1, 1), "unit": "paired-samples/s", "ms_per_step": round(dt1 / n * 1e3, 5),
                "note": "pinned host batches copied H2D inside every step; not the headline value"}

    # ---- per-kernel device time: eager launches, each stamped by hipExtLaunchKernel start/stop events on the launch stream
    kern = {}
    if kernel_steps > 0:               # EVERY rank takes the steps (under data parallelism each step carries a collective); rank 0 reports
        L, h = model._L, model._h
        L.avae_timing_enable(h, 1)
        for i in range(kernel_steps):
            model.partial_fit(batches[i % nb], return_cost=False)
        buf = C.create_string_buffer(1 << 16)
        L.avae_timing_report(h, buf, len(buf))
        L.avae_timing_enable(h, 0)
        for line in (buf.value.decode().splitlines() if rank == 0 else []):
            nm, calls, avg_ms, min_ms = line.split()
            base = nm.split(".")[0]          # large problems run one launch per modality: "<name>", "<name>.1", ...
            c0, a0, m0 = kern.get(base, (0, 0.0, 0.0))
            kern[base] = (max(c0, int(calls)), a0 + float(avg_ms), m0 + float(min_ms))
    if world > 1:
        barrier()
    model.synchronize()                  # (raises if a bounded wait of the hipIpc all-reduce gave up)
    res = {"name": name, "label": label, "B": B, "dtype": dtype, "es": es, "archs": archs, "dt": dt, "dts": dts, "steps": steps,
           "kern": kern, "last_cost": last_cost, "pcie": pcie, "n_params": int(model.n_params), "comm": model._comm,
           "buckets": len(model._buckets), "wire": wire or (args.wire if args.wire != "auto" else "fp32")}
    del model, data, batches, whole
    torch.cuda.empty_cache()
    return res


def choose_comm(args, world, rank, local_rank):
    """N > 1: the fastest collective that reproduces torch.distributed's result on THIS node.  Every candidate trains the bench
    model for a few steps (single steps and a captured run) from the same weights, on the same batches and eps stream as a
    torch.distributed reference, and must land on the same weights; the ranks agree on the verdict, so they switch together."""
    import torch
    import torch.distributed as dist
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    if args.comm != "auto":
        return args.comm, ["--comm %s" % args.comm], args.wire == "bf16", "not checked (--comm given)"
    archs, B, cfg_dtype, _ = CONFIGS[args.config]
    dtype = args.dtype or cfg_dtype
    rng = np.random.default_rng(7 + rank)
    mat, edges = synth_for(rng, archs, 20 * B)
    data = torch.as_tensor(mat).cuda()
    whole = [data[:, edges[k]:edges[k + 1]] for k in range(len(archs))]
    os.environ.setdefault("AVAE_IPC_TIMEOUT_MS", "10000")

    def trial(comm, wire="fp32"):
        m = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, device=local_rank, seed=0,
                                        use_graph=not args.no_graph, data_parallel=True, comm=comm, comm_buckets=args.comm_buckets or 2,
                                        wire_dtype=wire, **hyper_for(archs))
        if m._comm != comm:
            raise RuntimeError("did not come up")
        for i in range(2):
            m.partial_fit([w[i * B:(i + 1) * B] for w in whole], return_cost=False)
        m.partial_fit_steps([w[2 * B:] for w in whole], 18, return_cost=False)
        m.synchronize()
        return m.get_params(), m.cost_history(20).copy()

    def agree(ok):
        t = torch.tensor([1.0 if ok else 0.0], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    log = []
    ref = trial("torch")
    for cand in ("ipc", "library"):
        ok, why = True, "matches torch.distributed"
        try:
            p, c = trial(cand)
            dp, dc = float(np.abs(p - ref[0]).max()), float((np.abs(c - ref[1]) / np.abs(ref[1])).max())
            # (only the order of the N-term fp32 sum differs: rounding-level gradient differences, amplified by Adam on the
            # ill-conditioned elements and, with bf16 operands, by one-ulp flips of the weight shadows; a broken collective is O(1) off)
            ok = bool(np.isfinite(p).all()) and dp <= (5e-4 if dtype == "fp32" else 5e-3) and dc <= (1e-4 if dtype == "fp32" else 1e-3)
            if not ok:
                why = "differs from torch.distributed: max|dtheta| %.2e, cost rel %.2e" % (dp, dc)
        except Exception as e:
            ok, why = False, repr(e)[:160]
        all_ok = agree(ok)
        log.append("%s: %s%s" % (cand, why, "" if all_ok or not ok else " (another rank failed)"))
        if all_ok:
            # bf16 on the wire with this backend: the costs of the same 20 steps within north_star's 1e-3 of the reference's
            wok, note = False, "fp32 operands: fp32 wire"
            if dtype == "bf16" and args.wire in ("auto", "bf16"):
                try:
                    pw, cw = trial(cand, "bf16")
                    drift = float((np.abs(cw - ref[1]) / np.abs(ref[1])).max())
                    wok = bool(np.isfinite(pw).all()) and drift <= 1e-3
                    note = "cost drift over 20 steps vs the fp32-wire torch.distributed run: %.2e (bound 1e-3)" % drift
                except Exception as e:
                    note = repr(e)[:160]
                wok = agree(wok)
            return cand, log, wok, note
    return "torch", log, False, "torch.distributed collective: fp32 wire"


def price(res):
    """roofline of the dominant launch + step-level fractions + the GEMM launches' MFMA fraction, from measure()'s result"""
    archs, B, es, kern, dt, steps = res["archs"], res["B"], res["es"], res["kern"], res["dt"], res["steps"]
    kern = dict(kern)
    for whole in ("wgrad", "adam"):       # data-parallel pipeline: the launch runs once per gradient bucket
        parts = [n for n in kern if n.startswith(whole + "_")]
        if parts and whole not in kern:
            kern[whole] = (max(kern[n][0] for n in parts), sum(kern[n][1] for n in parts), sum(kern[n][2] for n in parts))
    peak_tf = MFMA_BF16_PEAK_TF if es == 2 else MFMA_F32_PEAK_TF
    work, P = launch_work(archs, B, es)
    unfused = {}
    for n in list(kern):                  # "a+b": launch b rides in launch a (tail product, avae_host.hip::fuse_tail; Adam in the wgrad epilogue)
        parts = n.split("+")
        if len(parts) > 1 and all(p in work for p in parts):
            vals = [work.pop(p) for p in parts]
            by = sum(v[0] for v in vals)
            unfused[n] = by               # the sum of the parts: what the launches moved BEFORE they were fused
            if parts == ["wgrad", "adam"]:
                by -= 4 * P               # the fused launch stores the gradient but never re-reads it: a launch is priced on what it must move
            work[n] = (by, sum(v[1] for v in vals))
    names = [n for n in kern if n in work]
    step_us = dt / steps * 1e6
    dom = max(names, key=lambda n: kern[n][1]) if names else None
    roof = None
    if dom:
        by, fl = work[dom]
        avg_s = kern[dom][1] * 1e-3
        t_hbm, t_mfma = by / (HBM_PEAK_GBS * 1e9), fl / (peak_tf * 1e12)
        if t_mfma > t_hbm:
            ach = fl / avg_s / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4)}
        else:
            ach = by / avg_s / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        if dom in unfused and unfused[dom] != by:
            roof["frac_unfused_accounting"] = round(unfused[dom] / avg_s / 1e9 / HBM_PEAK_GBS, 4)      # round 2's pricing (sum of the parts)
        traffic, src = None, None        # HBM bytes per launch from committed rocprofv3 --pmc passes of the builder (tools/pmc_traffic.sh)
        for rnd in ("r03", "r02", "r01"):
            tf = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, res["name"]))
            if os.path.exists(tf):
                traffic = json.load(open(tf))["launches"].get(dom, {}).get("hbm_bytes")
                src = "profiles/%s (builder's rocprofv3 --pmc run, not measured in this run)" % os.path.basename(tf)
                break
        roof.update({"traffic": traffic, "traffic_source": src, "kernel": dom, "avg_us": round(kern[dom][1] * 1e3, 2),
                     "share_of_step": round(kern[dom][1] * 1e3 / step_us, 4),
                     "algorithmic_bytes": by, "algorithmic_flop": fl})
    in_step = [n for n in work if n in kern or n == "prep"]
    step_bytes = sum(work[n][0] for n in in_step)
    step_flop = sum(work[n][1] for n in in_step)
    gemm = [n for n in names if n.startswith(GEMM_LAUNCH_PREFIXES) and work[n][1] > 0]
    gemm_fl = sum(work[n][1] for n in gemm)
    gemm_us = sum(kern[n][1] for n in gemm) * 1e3
    per_launch = {n: {"us": round(kern[n][1] * 1e3, 2), "mfma_frac": round(work[n][1] / (kern[n][1] * 1e-3) / 1e12 / peak_tf, 4)} for n in gemm}
    return {"roofline": roof, "n_params": P,
            "step_roofline": {"algorithmic_bytes": step_bytes, "algorithmic_flop": step_flop,
                              "hbm_frac": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
                              "mfma_frac": round(step_flop / (dt / steps) / 1e12 / peak_tf, 4)},
            "gemm_launches": {"flop": gemm_fl, "us": round(gemm_us, 2),
                              "mfma_frac": round(gemm_fl / (gemm_us * 1e-6) / 1e12 / peak_tf, 4) if gemm_us else None,
                              "per_launch": per_launch}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=640)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of --steps; the median is reported")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short C4 / c2conv / C5 / C1 runs that ride behind the headline")
    ap.add_argument("--kernel-steps", type=int, default=200, help="steps of the per-kernel hipEvent pass")
    ap.add_argument("--single-step", action="store_true", help="submit every step on its own (avae_train_step) instead of in runs")
    ap.add_argument("--force-comm", action="store_true",
                    help="one GPU: run the step through the library's data-parallel pipeline (one-rank RCCL communicator, two buckets, "
                         "comm stream) to see what the pipeline itself costs; not the headline")
    ap.add_argument("--comm", default="auto", choices=["auto", "ipc", "library", "torch"],
                    help="N > 1: who runs the gradient all-reduce.  auto = the library's one-shot all-reduce over hipIpc peers, checked "
                         "against torch.distributed on a few steps first; falls back to the library's RCCL communicator, then to "
                         "torch.distributed, if the check fails on any rank")
    ap.add_argument("--comm-buckets", type=int, default=0, choices=[0, 1, 2],
                    help="2: decoder bucket first, its all-reduce beside the encoder's backward pass; 1: one all-reduce of the whole buffer "
                         "(no split weight-gradient / Adam launches); 0 (default): N > 1 times both for a few steps and keeps the faster")
    ap.add_argument("--wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="gradient element type on the wire.  auto (default): fp32 for fp32 operands; for bf16 operands N > 1 also trains a few "
                         "steps with bf16 on the wire (sum in fp32, rounded once; the cost travels as fp32), checks their costs against the "
                         "torch.distributed reference at north_star's 1e-3, times both and keeps the faster -- both are reported")
    ap.add_argument("--host-input", action="store_true",
                    help="also time the step fed from pinned host batches (PCIe-inclusive rate; reported beside `value`, never as it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    g.build()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # Rehearsal of the N > 1 path on a ONE-GPU box (AVAE_BENCH_ONE_GPU=1): every rank on cuda:0, gloo for the bootstrap -- RCCL refuses
    # two ranks on one device, the hipIpc all-reduce does not.  The line says so ("rehearsal"); it is not a scaling measurement.
    one_gpu = world > 1 and os.environ.get("AVAE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        os.environ.setdefault("AVAE_IPC_BLOCKS", "64")       # the ranks' spinning exchange kernels share one GPU: all of them must be resident
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    comm, comm_log, bf16_wire_ok, bf16_wire_note = None, [], False, None
    if world > 1:
        comm, comm_log, bf16_wire_ok, bf16_wire_note = choose_comm(args, world, rank, local_rank)
    buckets, wire, trials = None, None, None
    if world > 1 and comm != "torch":
        # Two buckets overlap the decoder side's all-reduce with the encoder's backward pass but split the weight-gradient and Adam
        # launches in two; with a fast (mesh) collective the single all-reduce behind ONE weight-gradient launch can be the shorter
        # step.  bf16 on the wire halves the bytes.  Every (buckets, wire) combination the flags leave open is timed for a few steps
        # on this node (MAX over ranks, so every rank picks the same) and the fastest becomes the headline; all are reported.
        dt_cfg = args.dtype or CONFIGS[args.config][2]
        b_opts = [args.comm_buckets] if args.comm_buckets else [2, 1]
        w_opts = [args.wire] if args.wire != "auto" else (["fp32", "bf16"] if (dt_cfg == "bf16" and bf16_wire_ok) else ["fp32"])
        trials = {}
        if len(b_opts) * len(w_opts) > 1:
            for nb in b_opts:
                for wd in w_opts:
                    try:
                        r = measure(args.config, args, world, rank, local_rank, max(64, args.steps // 5), 32, 3, 0, args.dtype, comm=comm, buckets=nb, wire=wd)
                        trials["%d_buckets_%s_wire" % (nb, wd)] = round(r["dt"] / r["steps"] * 1e3, 5)
                    except Exception as e:
                        comm_log.append("buckets=%d wire=%s failed: %s" % (nb, wd, repr(e)[:120]))
            if trials:
                best = min(trials, key=trials.get)
                buckets, wire = int(best[0]), best.split("_")[2]
        else:
            buckets, wire = b_opts[0], w_opts[0]
    res = measure(args.config, args, world, rank, local_rank, args.steps, args.warmup, args.repeats, args.kernel_steps, args.dtype, comm=comm, buckets=buckets, wire=wire)
    coll = None
    if world > 1:
        coll = {"backend": res["comm"], "buckets": res["buckets"], "wire": res["wire"], "selection": comm_log}
        if trials:
            coll["trials_ms_per_step"] = trials
        coll["bf16_wire_check"] = bf16_wire_note
        try:    # the same pipeline on one rank (split weight-gradient / Adam launches, the collective's launch, nothing on the wire):
                # what is left of the N-rank step beyond it is the exposed, non-overlapped share of the collective
            lo = measure(args.config, args, world, rank, local_rank, max(64, args.steps // 4), 32, 3, 0, args.dtype,
                         comm="ipc" if res["comm"] == "ipc" else "library", local_only=True, buckets=res["buckets"], wire=res["wire"])
            coll["pipeline_one_rank_ms_per_step"] = round(lo["dt"] / lo["steps"] * 1e3, 5)
            coll["exposed_us_per_step"] = round((res["dt"] / res["steps"] - lo["dt"] / lo["steps"]) * 1e6, 2)
        except Exception as e:
            coll["pipeline_one_rank_error"] = repr(e)[:200]
        for k in ("allreduce_dec", "allreduce_enc", "allreduce"):
            if k in res["kern"]:
                coll[k + "_us"] = round(res["kern"][k][1] * 1e3, 2)       # eager pass; includes waiting for the slowest peer
    extras = {}
    if world == 1 and not args.no_extras and args.config == "c2":
        # the other configurations of BASELINE.json on the same path, witnessed by the same run (short: they are parity-test
        # cases, not the headline).  C4 = north_star's MFMA target: per-launch MFMA fractions of its GEMM launches.
        for nm, st, wu, ks in (("c4", 48, 16, 20), ("c2conv", 160, 32, 40), ("c5", 320, 32, 0), ("c1", 320, 32, 0)):
            try:
                r = measure(nm, args, world, rank, local_rank, st, wu, 3, ks)
            except Exception as e:                       # an extra must never cost the headline
                extras[nm] = {"error": repr(e)[:200]}
                continue
            e = {"workload": r["label"], "ms_per_step": round(r["dt"] / st * 1e3, 5), "value": round(r["B"] * st / r["dt"], 1),
                 "unit": "paired-samples/s", "steps": st, "repeats": 3, "dtype": r["dtype"], "last_cost": r["last_cost"]}
            if r["kern"]:
                pr = price(r)
                e.update({"step_mfma_frac": pr["step_roofline"]["mfma_frac"], "step_hbm_frac": pr["step_roofline"]["hbm_frac"],
                          "gemm_mfma_frac": pr["gemm_launches"]["mfma_frac"], "gemm_launches_us": pr["gemm_launches"]["per_launch"],
                          "kernels_us": {n: round(v[1] * 1e3, 2) for n, v in sorted(r["kern"].items())}})
            extras[nm] = e

    if rank == 0:
        B, dt, steps = res["B"], res["dt"], res["steps"]
        pr = price(res)
        out = {
            "metric": "paired-samples/sec (img+jnt assoc-VAE train step)",
            "value": round(B * world * steps / dt, 1),
            "unit": "paired-samples/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": res["dtype"], "data": "synthetic",
            "config": {"workload": res["label"], "global_batch": B * world, "per_gpu_batch": B, "n_params": pr["n_params"] or res["n_params"],
                       "parallelism": "dp%d" % world, "graph": not args.no_graph,
                       "collective": ("%s, %d bucket(s), %s on the wire" % ({"ipc": "library-owned one-shot all-reduce over hipIpc peers",
                                                                                   "library": "library-owned RCCL communicator",
                                                                                   "torch": "torch.distributed all_reduce, host-stepped"}[res["comm"]],
                                                                                  res["buckets"], res["wire"])
                                      if (world > 1 or args.force_comm) else "none (one replica)"),
                       "submission": "per step" if args.single_step else "runs of <=16 consecutive resident batches"},
            "timing": {"repeats": args.repeats, "statistic": "median of the repeats, each EXACTLY --steps steps between barrier + synchronise",
                       "ms_per_step_all": [round(d / steps * 1e3, 5) for d in res["dts"]]},
            "roofline": pr["roofline"],
            "step_roofline": pr["step_roofline"],
            "gemm_launches": {k: v for k, v in pr["gemm_launches"].items() if k != "per_launch"},
            "kernels_us": {n: round(v[1] * 1e3, 2) for n, v in sorted(res["kern"].items())},
            "launch_order": [n for n in res["kern"] if n not in ("prep", "_null_kernel")],      # one step's launches in issue order (tools/per_launch.py)
            "last_cost": res["last_cost"],
        }
        if coll:
            out["collective"] = coll
            if one_gpu:
                out["rehearsal"] = "%d ranks sharing ONE GPU (gloo bootstrap): exercises the N > 1 code path, not a scaling measurement" % world
        if extras:
            out.update(extras)
        if res["pcie"]:
            out["pcie_inclusive"] = res["pcie"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(res["archs"], B)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
