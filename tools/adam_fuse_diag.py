#!/usr/bin/env python3
"""Diagnostic (not part of the product): the fused weight-gradient + Adam launch against its unfused twin (AVAE_NO_ADAM_FUSE=1) on a
golden fixture's configuration -- parameters, moments and gradients after every step, first differing tensor and position."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "script_nz4_b64"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    from vae_assoc_amd import vae_assoc as V
    from oracle import vae_assoc_oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    G = {k: z[k] for k in z.files}
    c = json.loads(str(G["config"]))
    M = len(c["archs"])
    X = [G["x%d" % m] for m in range(M)]

    def make(env):
        for k in ("AVAE_NO_ADAM_FUSE",):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = V.AssocVariationalAutoEncoder(c["archs"], binary=c["binary"], transfer_fct=c["act"], weights=c["weights"],
                                          assoc_lambda=c["assoc_lambda"], learning_rate=c["lr"], batch_size=c["B"], compute_dtype=dtype)
        m.set_params(G["params0"])
        return m
    a, b = make({}), make({"AVAE_NO_ADAM_FUSE": "1"})
    spans, off = [], 0
    for mi, na in enumerate(c["archs"]):
        for n, shp in O.layer_shapes(na):
            cnt = int(np.prod(shp))
            spans.append(("m%d.%s%s" % (mi, n, tuple(shp)), off, off + cnt, shp))
            off += cnt

    def where(i):
        for n, lo, hi, shp in spans:
            if lo <= i < hi:
                return "%s@%s" % (n, np.unravel_index(i - lo, shp) if len(shp) > 1 else (i - lo,))
        return str(i)
    for s in range(3):
        ca, cb = a.partial_fit(X, G["eps"][s]), b.partial_fit(X, G["eps"][s])
        pa, pb = a.get_params(), b.get_params()
        ga, gb = a.get_grads(), b.get_grads()
        ma, va, _ = a.get_opt_state()
        mb, vb, _ = b.get_opt_state()
        print("step %d cost %.4f %.4f" % (s, ca, cb))
        for nm, x, y in (("params", pa, pb), ("grads", ga, gb), ("m", ma, mb), ("v", va, vb)):
            d = np.flatnonzero(x != y)
            print("   %-7s differing entries: %d of %d%s" % (nm, d.size, x.size, ("  first %s  last %s  max|d| %.3e" % ([where(i) for i in d[:4]], where(d[-1]), np.abs(x - y).max())) if d.size else ""))


if __name__ == "__main__":
    main()
