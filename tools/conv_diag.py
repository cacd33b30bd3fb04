#!/usr/bin/env python3
"""Per-tensor gradient error of the conv/deconv branch against the oracle (diagnostic): python tools/conv_diag.py [fp32|bf16] [small|bench]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
g.build()
from conftest import make_arch, synth_batch
from oracle import vae_assoc_oracle as O
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
which = sys.argv[2] if len(sys.argv) > 2 else "small"
if which == "small":
    archs = [dict(make_arch("image", 784, 8, 24, 6), hidden_conv=True, n_hidden_gener_1=24, n_hidden_gener_2=8), make_arch("joint", 147, 40, 32, 6)]
    B = 12
else:
    archs = [dict(make_arch("image", 784, 16, 64, 20), hidden_conv=True, n_hidden_gener_1=64, n_hidden_gener_2=16), make_arch("joint", 147, 200, 200, 20)]
    B = 256
rng = np.random.default_rng(3)
X = synth_batch(rng, B, [784, 147], [True, False])
eps = rng.standard_normal((B, archs[0]["n_z"])).astype(np.float32)
m = AssocVariationalAutoEncoder(archs, binary=[True, False], transfer_fct="relu", weights=[5.0, 1.0], assoc_lambda=0.5, batch_size=B, compute_dtype=dtype, seed=2)
p0 = m.get_params()
ref = O.OracleAssocVAE(archs, [True, False], "relu", [5.0, 1.0], 0.5, 1e-3, B, params_flat=p0.astype(np.float64), quant=None if dtype == "fp32" else "bf16")
c_ref, g_ref, _ = ref.cost_and_grads(X, eps)
c = m.partial_fit(X, eps)
gh = m.get_grads().astype(np.float64)
print("cost", c, c_ref, abs(c - c_ref) / abs(c_ref))
off = 0
for mi, na in enumerate(archs):
    for nm, shp in O.layer_shapes(na):
        n = int(np.prod(shp))
        a, b = gh[off:off + n], g_ref[off:off + n]
        print("m%d.%-10s max|ref| %.3e  err/max %.2e" % (mi, nm, np.abs(b).max(), np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
        off += n
