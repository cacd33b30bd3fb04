#!/usr/bin/env python3
"""Diagnostic: rate at which two back-to-back single steps end differently from the same steps with a host synchronisation in between,
per plan switch / configuration variant.  This is how the identity-activation race of k_small_latb was cornered (a counted wait whose
count the compiler had changed by dropping two dead loads): base 21 / 1200, bin_TF 51 / 1200 before the fix, 0 / 2500 after.

    python tools/determinism_probe.py [repetitions] [variant,variant,...]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_arch, synth_batch
from vae_assoc_amd import vae_assoc as V
import test_gpu_parity as T
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
base = dict(dtype="fp32", B=64, nz=32, act="identity", spec=[(784, [500, 500]), (147, [200, 200])], binary=[False, True], w=[1.0, 50.0], lam=0.0, n=2, env={})
variants = {"base": {}, "NO_TAIL": dict(env={"AVAE_NO_TAIL": "1"}), "NO_ADAM_FUSE": dict(env={"AVAE_NO_ADAM_FUSE": "1"}), "NO_LEAN": dict(env={"AVAE_NO_LEAN": "1"}),
            "NO_XCD_PIECES": dict(env={"AVAE_NO_XCD_PIECES": "1"}), "NO_GRAPH": dict(graph=False), "relu": dict(act="relu"), "nz20": dict(nz=20), "bin_TF": dict(binary=[True, False]),
            "bf16": dict(dtype="bf16"), "c2_like": dict(dtype="bf16", B=256, nz=20, act="relu", binary=[True, False], w=[50.0, 1.0], lam=8.0)}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(variants)
ENVS = ("AVAE_NO_TAIL", "AVAE_NO_ADAM_FUSE", "AVAE_NO_LEAN", "AVAE_NO_XCD_PIECES")
for name in only:
    cfg = dict(base); cfg.update(variants[name])
    for k in ENVS: os.environ.pop(k, None)
    os.environ.update(cfg["env"])
    dtype, B, nz, act, spec, binary, w, lam, n = (cfg[k] for k in ("dtype", "B", "nz", "act", "spec", "binary", "w", "lam", "n"))
    archs = [make_arch("m%d" % i, ni, 0, 0, nz, n_hidden=hs) for i, (ni, hs) in enumerate(spec)]
    widths = [a["n_input"] for a in archs]
    rng = np.random.default_rng(5)
    model, _ref = T.build_pair(V, archs, binary, w, lam, act, B, dtype, seed=3001)
    p0 = model.get_params()
    data = np.concatenate(synth_batch(rng, n * B, widths, binary), axis=1)
    dev = torch.as_tensor(data).cuda()
    cols = np.cumsum([0] + widths)
    Xd = [dev[:, cols[m]:cols[m + 1]] for m in range(len(archs))]
    eps = torch.as_tensor(rng.standard_normal((n * B, nz)).astype(np.float32)).cuda()
    kw = dict(use_graph=False) if cfg.get("graph") is False else {}
    def run(sync):
        mm = V.AssocVariationalAutoEncoder(archs, binary=binary, transfer_fct=act, weights=w, assoc_lambda=lam, batch_size=B, compute_dtype=dtype, seed=7, **kw)
        mm.set_params(p0)
        for i in range(n):
            mm.partial_fit([x[i * B:(i + 1) * B] for x in Xd], eps[i * B:(i + 1) * B], return_cost=False)
            if sync:
                torch.cuda.synchronize()
        return mm.get_params()
    truth = run(True)
    bad = sum(1 for r in range(REPS) if not np.array_equal(run(False), truth))
    print("%-14s %3d of %d differ" % (name, bad, REPS), flush=True)
