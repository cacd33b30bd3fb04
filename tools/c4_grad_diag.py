"""Diagnostic: C4 fixture, HIP gradients vs the fixture's sampled entries, per tensor (max / 99th percentile / count above 1e-4)."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.build()
from vae_assoc_amd import vae_assoc as V
spec = importlib.util.spec_from_file_location("big", os.path.join(ROOT, "tests", "golden", "make_golden_big.py"))
big = importlib.util.module_from_spec(spec); spec.loader.exec_module(big)
G = np.load(os.path.join(ROOT, "tests", "golden", "c4_b4096.npz"))
c = big.C4
X, eps, p0 = big.c4_inputs()
act = sys.argv[2] if len(sys.argv) > 2 else c["act"]
for dtype, tag in ((sys.argv[1] if len(sys.argv) > 1 else "fp32", "f64"),):
    m = V.AssocVariationalAutoEncoder(c["archs"], binary=c["binary"], transfer_fct=act, weights=c["weights"], assoc_lambda=c["assoc_lambda"],
                                      learning_rate=c["lr"], batch_size=c["B"], compute_dtype=dtype)
    m.set_params(p0)
    cost = m.partial_fit(X, eps)
    print(dtype, "cost", cost, "fixture", float(G["cost_" + tag]))
    gr = m.get_grads().astype(np.float64)
    ptr, idx = G["sample_ptr"], G["sample_idx"]
    for t, name in enumerate(G["names"]):
        sl = slice(int(ptr[t]), int(ptr[t + 1]))
        mx = float(G["gmax_" + tag][t])
        e = np.abs(gr[idx[sl]] - G["gsample_" + tag][sl]) / mx
        print("%-14s max %.2e  p99 %.2e  median %.2e  n>1e-4: %d / %d" % (name, e.max(), np.percentile(e, 99), np.median(e), int((e > 1e-4).sum()), e.size))
