"""Round-3 fixtures: BASELINE C3 and C5 at their GLOBAL batch (2048), from the CPU oracle (PARITY UNPINNED against the reference
itself, as make_golden.py explains: it cannot run here and holds no vectors).

  c3_b2048.npz  img 784-500-500 + jnt 147-200-200, n_z 20, relu, weights [50, 1], lambda 8, lr 1e-3, ONE batch of 2048 rows per step
                (what 8 ranks x 256 rows add up to, vae_assoc.py:319-371: Bernoulli / KL terms carry 1/2048, Gaussian recon and the
                association term carry 1): step-0 cost, per tensor gradient maxima / L2 norms / 1024 sampled entries, the costs of
                three Adam steps and sampled weights after them -- from the fp64 oracle and from the oracle that rounds where the
                bf16 kernels round.
  c5_b2048.npz  + the 256-d aux modality (256-200-200, Gaussian), fp32 configuration: the same quantities from the fp64 oracle for
                every lambda of BASELINE's sweep {0, 1e-5, 1e-2, 1, 8, 50}.

Inputs / weights / eps are regenerated from seeded NumPy generators (`inputs(name)`), checksums are stored.
Run from the repo root:  python tests/golden/make_golden_dp.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import vae_assoc_oracle as O  # noqa: E402
from make_golden_big import arch, nonzero_biases, synth, tensor_slices  # noqa: E402

N_SAMPLES = 1024
STEPS = 3
LAMBDAS = [0.0, 1e-5, 1e-2, 1.0, 8.0, 50.0]
CFG = {
    "c3": dict(archs=[arch("image", 784, [500, 500], 20), arch("joint", 147, [200, 200], 20)], binary=[True, False],
               weights=[50.0, 1.0], assoc_lambda=8.0, lr=1e-3, B=2048, act="relu", seed=20260301),
    "c5": dict(archs=[arch("image", 784, [500, 500], 20), arch("joint", 147, [200, 200], 20), arch("aux", 256, [200, 200], 20)],
               binary=[True, False, False], weights=[50.0, 1.0, 1.0], assoc_lambda=8.0, lr=1e-3, B=2048, act="relu", seed=20260305),
}


def inputs(name):
    """(config, X, eps [STEPS, B, n_z], p0) -- used by this script and by tests/test_gpu_global_batch.py."""
    c = CFG[name]
    rng = np.random.default_rng(c["seed"])
    X = synth(rng, c["B"], [a["n_input"] for a in c["archs"]], c["binary"])
    eps = rng.standard_normal((STEPS, c["B"], 20)).astype(np.float32)
    p0 = O.flatten_params(c["archs"], O.init_params(c["archs"], np.random.default_rng(c["seed"] + 1))).astype(np.float32)
    p0 = nonzero_biases(c["archs"], p0.copy(), rng)
    return c, X, eps, p0


def run(c, X, eps, p0, lam, quant):
    model = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], lam, c["lr"], c["B"], dtype=np.float64,
                             params_flat=p0.astype(np.float64), quant=quant)
    cost0, g0, _ = model.cost_and_grads(X, eps[0])
    costs = [model.partial_fit(X, eps[s]) for s in range(STEPS)]
    assert costs[0] == cost0
    return cost0, g0, np.array(costs), model.get_params()


def make(name):
    c, X, eps, p0 = inputs(name)
    sl = tensor_slices(c["archs"])
    rng = np.random.default_rng(77)
    idx = [np.sort(rng.choice(n, size=min(N_SAMPLES, n), replace=False)) + off for _nm, off, n in sl]
    idx_all = np.concatenate(idx).astype(np.int64)
    out = dict(config=np.array(json.dumps(c)), names=np.array([nm for nm, _o, _n in sl]), sample_idx=idx_all,
               sample_ptr=np.cumsum([0] + [len(i) for i in idx]).astype(np.int64),
               checksum=np.array([float(x.astype(np.float64).sum()) for x in X] + [float(eps.astype(np.float64).sum()), float(p0.astype(np.float64).sum())]))
    runs = [("f64", c["assoc_lambda"], None), ("bf16", c["assoc_lambda"], "bf16")] if name == "c3" else \
           [("f64_lam%d" % i, lam, None) for i, lam in enumerate(LAMBDAS)]
    if name == "c5":
        out["lambdas"] = np.array(LAMBDAS)
    for tag, lam, quant in runs:
        cost0, g0, costs, p3 = run(c, X, eps, p0, lam, quant)
        out["cost_" + tag] = np.float64(cost0)
        out["gmax_" + tag] = np.array([np.abs(g0[o:o + n]).max() for _nm, o, n in sl])
        out["gl2_" + tag] = np.array([np.linalg.norm(g0[o:o + n]) for _nm, o, n in sl])
        out["gsample_" + tag] = g0[idx_all]
        out["costs_" + tag] = costs
        out["p3sample_" + tag] = p3[idx_all]
        print(name, tag, "lambda", lam, "costs", costs, flush=True)
    np.savez_compressed(os.path.join(HERE, "%s_b2048.npz" % name), **out)


if __name__ == "__main__":
    for k in sys.argv[1:] or ["c3", "c5"]:
        make(k)
