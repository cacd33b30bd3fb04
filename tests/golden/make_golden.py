"""Generates the golden fixtures of tests/golden/ from the fp64 CPU oracle.

PARITY UNPINNED (see oracle/vae_assoc_oracle.py): the reference cannot run here and holds no
vectors for this path, so these are outputs of the oracle's reading of vae_assoc.py, not of
the reference.  They pin the oracle against silent edits and give the GPU tests fixed inputs.

The two configurations follow SURVEY.md 8 "config discrepancy": the experiment script's live
hyper-parameters (n_z=4, batch 64, lambda 8, weights [50,1], vae_assoc_ujichar_img_jnt.py:38-47)
and BASELINE's C1 (n_z=20, batch 100, train() default lambda 1e-5, weights 1).  Input widths
(784 / 147) are the reference's; hidden widths are shrunk so the fixtures stay small.

Also writes dataset_order.npz: the batch order produced by the REFERENCE's own dataset.py
(numpy-only, importable under Python 3; imported from /root/reference in this container only)
for a seeded run, which pins vae_assoc_amd/dataset.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import vae_assoc_oracle as O  # noqa: E402


def arch(scope, n_in, h1, h2, n_z):
    return dict(scope=scope, hidden_conv=False, n_hidden_recog_1=h1, n_hidden_recog_2=h2,
                n_hidden_gener_1=h1, n_hidden_gener_2=h2, n_input=n_in, n_z=n_z)


ROWS = 16   # rows kept of the wide [B, n_input] outputs (keeps the fixtures small)

CONFIGS = {
    "script_nz4_b64": dict(archs=[arch("image", 784, 24, 16, 4), arch("joint", 147, 16, 12, 4)],
                           binary=[True, False], weights=[50.0, 1.0], assoc_lambda=8.0, lr=1e-3, B=64, act="relu"),
    "c1_nz20_b100": dict(archs=[arch("image", 784, 24, 20, 20), arch("joint", 147, 20, 16, 20)],
                         binary=[True, False], weights=[1.0, 1.0], assoc_lambda=1e-5, lr=1e-3, B=100, act="relu"),
}


def synth_inputs(rng, B):
    """dark-background stroke-like images in [0,1], z-scored joint features."""
    img = np.clip(rng.beta(0.25, 1.5, size=(B, 784)), 0, 1) * (rng.random((B, 784)) >= 0.7)
    jnt = rng.standard_normal((B, 147))
    return [img.astype(np.float32), jnt.astype(np.float32)]


def make(name, c):
    rng = np.random.default_rng(20260104)
    archs, B, nz = c["archs"], c["B"], c["archs"][0]["n_z"]
    p0 = O.flatten_params(archs, O.init_params(archs, np.random.default_rng(0))).astype(np.float32)
    # non-zero biases so that bias handling is actually exercised
    off = 0
    p0 = p0.copy()
    for na in archs:
        for nm, shp in O.layer_shapes(na):
            n = int(np.prod(shp))
            if len(shp) == 1:
                p0[off:off + n] = (0.05 * rng.standard_normal(n)).astype(np.float32)
            off += n
    X = synth_inputs(rng, B)
    eps = rng.standard_normal((3, B, nz)).astype(np.float32)
    model = O.OracleAssocVAE(archs, c["binary"], c["act"], c["weights"], c["assoc_lambda"], c["lr"], B,
                             dtype=np.float64, params_flat=p0.astype(np.float64))
    cost0, g0, fw = model.cost_and_grads(X, eps[0])
    terms = O.loss_terms(archs, fw, [x.astype(np.float64) for x in X], c["binary"], c["weights"], c["assoc_lambda"])
    assert abs(terms["cost"] - cost0) < 1e-9 * abs(cost0)
    out = dict(config=np.array(json.dumps(c)), params0=p0, eps=eps, cost0=np.float64(cost0),
               grads0=g0.astype(np.float32))
    for m in range(len(archs)):
        out["x%d" % m] = X[m]
        out["mu%d" % m] = fw[m]["mu"]
        out["lv%d" % m] = fw[m]["lv"]
        out["z%d" % m] = fw[m]["z"]
        out["xhat%d" % m] = fw[m]["xhat"][:ROWS].astype(np.float32)
        out["vae_cost%d" % m] = np.float64(terms["vae_costs"][m])
    out["assoc"] = np.array(terms["assoc"], dtype=np.float64)
    costs = []
    for s in range(3):
        costs.append(model.partial_fit(X, eps[s]))
        if s == 0:
            out["params1"] = model.get_params().astype(np.float32)
    out["costs"] = np.array(costs)
    out["params3"] = model.get_params().astype(np.float32)
    out["adam_m3_norm"] = np.float64(np.linalg.norm(model.m))
    out["adam_v3_norm"] = np.float64(np.linalg.norm(model.v))
    # inference surface on the trained weights
    mus = model.transform(X)
    gen = model.generate(eps[1])
    rec = model.reconstruct(X, eps=[eps[1], eps[2]])
    for m in range(len(archs)):
        out["t_mu%d" % m] = mus[m]
        out["gen%d" % m] = gen[m][:ROWS].astype(np.float32)
        out["rec%d" % m] = rec[m][:ROWS].astype(np.float32)
    out["eval_cost3"] = np.float64(model.evaluate_cost(X, eps[2]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "P =", p0.size, "cost0 =", cost0, "costs =", costs)


def make_dataset_order():
    ref = "/root/reference/dataset.py"
    if not os.path.exists(ref):
        print("reference dataset.py not present; keeping the committed dataset_order.npz")
        return
    spec = importlib.util.spec_from_file_location("ref_dataset", ref)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    data = np.arange(53 * 3, dtype=np.float64).reshape(53, 3)
    np.random.seed(1234)
    ds = mod.construct_datasets(data, validation_ratio=.1, test_ratio=.1)
    batches = [ds.train.next_batch(8)[0][:, 0].copy() for _ in range(17)]
    np.savez_compressed(os.path.join(HERE, "dataset_order.npz"), data=data, seed=np.int64(1234), batch=np.int64(8),
                        batches=np.array(batches),
                        n_train=np.int64(ds.train._data.shape[0]), validation=ds.validation._data[:, 0],
                        test=ds.test._data[:, 0])
    print("dataset_order: train", ds.train._data.shape, "val", ds.validation._data.shape, "test", ds.test._data.shape)


if __name__ == "__main__":
    for k, v in CONFIGS.items():
        make(k, v)
    make_dataset_order()
