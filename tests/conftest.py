import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_arch(scope, n_in, h1, h2, n_z, n_hidden=None):
    na = dict(scope=scope, hidden_conv=False, n_hidden_recog_1=h1, n_hidden_recog_2=h2,
              n_hidden_gener_1=h1, n_hidden_gener_2=h2, n_input=n_in, n_z=n_z)
    if n_hidden is not None:
        na["n_hidden"] = list(n_hidden)
    return na


def synth_batch(rng, B, widths, binary):
    """Synthetic inputs of SURVEY.md 8d: stroke-like images in [0,1] for Bernoulli modalities,
    standard-normal features for Gaussian ones."""
    X = []
    for w, b in zip(widths, binary):
        if b:
            X.append((np.clip(rng.beta(0.25, 1.5, size=(B, w)), 0, 1) * (rng.random((B, w)) >= 0.7)).astype(np.float32))
        else:
            X.append(rng.standard_normal((B, w)).astype(np.float32))
    return X


@pytest.fixture(scope="session")
def golden():
    import json
    out = {}
    for name in ("script_nz4_b64", "c1_nz20_b100"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        d = {k: z[k] for k in z.files}
        d["config"] = json.loads(str(d["config"]))
        out[name] = d
    return out


def hip_relu_masks(model, archs):
    """The relu decisions of the HIP path's LAST forward pass, in the shape oracle.backward(masks=...) takes: per modality
    {"enc": [bool [B, width] per hidden layer], "dec": [...]} from the stored activations (avae_debug_fetch "E<m>_<k>" / "D<m>_<k>").
    With them the oracle takes the same side of every relu kink as the kernels did, and a gradient comparison on the benchmark's
    transfer function is a comparison of arithmetic, at the plain tolerance."""
    import ctypes as C
    B = model.batch_size
    out = []
    for m, na in enumerate(archs):
        hs = na["n_hidden"] if na.get("n_hidden") else [na["n_hidden_recog_1"], na["n_hidden_recog_2"]]
        d = {"enc": [], "dec": []}
        for side, key in (("E", "enc"), ("D", "dec")):
            for k, w in enumerate(hs):
                buf = np.empty(B * int(w), dtype=np.float32)
                cnt = C.c_size_t(0)
                rc = model._L.avae_debug_fetch(model._h, ("%s%d_%d" % (side, m, k)).encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(cnt))
                assert rc == 0 and cnt.value == buf.size, model._L.avae_last_error(model._h)
                d[key].append(buf.reshape(B, int(w)) > 0)
        out.append(d)
    return out


def concat_masks(parts):
    """row-wise concatenation of hip_relu_masks of several shards"""
    return [{key: [np.concatenate([p[m][key][k] for p in parts]) for k in range(len(parts[0][m][key]))] for key in ("enc", "dec")}
            for m in range(len(parts[0]))]


def shadow_err(model):
    """(max |W - theta|, max |W^T - theta|, layers checked) over every dense layer and conv stage (avae_debug_fetch "shadow_err"): the
    compute-dtype shadows the optimiser rewrites must BE the parameters, rounded once.  A stale shadow entry would only show in the
    next step's arithmetic; this sees it at once."""
    import ctypes as C
    buf = np.zeros(4, np.float32)
    cnt = C.c_size_t(0)
    rc = model._L.avae_debug_fetch(model._h, b"shadow_err", buf.ctypes.data_as(C.c_void_p), 4, C.byref(cnt))
    assert rc == 0 and cnt.value == 4, model._L.avae_last_error(model._h)
    return float(buf[0]), float(buf[1]), int(buf[2])
