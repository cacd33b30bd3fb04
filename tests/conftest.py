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
