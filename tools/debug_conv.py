import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_arch, synth_batch
from oracle import vae_assoc_oracle as O
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
img = dict(make_arch("image", 784, 8, 12, 6), hidden_conv=True, n_hidden_gener_1=12, n_hidden_gener_2=6)
archs=[img]; B=10
m = AssocVariationalAutoEncoder(archs, binary=True, transfer_fct="relu", weights=1.0, assoc_lambda=1.0, batch_size=B, compute_dtype="fp32", seed=5)
rng=np.random.default_rng(0)
p0=m.get_params(); p0 += (0.02*rng.standard_normal(p0.size)).astype(np.float32); m.set_params(p0)
ref=O.OracleAssocVAE(archs, True, "relu", 1.0, 1.0, 1e-3, B, params_flat=p0.astype(np.float64))
X=synth_batch(rng,B,[784],[True]); eps=rng.standard_normal((B,6)).astype(np.float32)
c,g_ref,_=ref.cost_and_grads(X,eps)
m._backward(X,eps); g=m.get_grads()
off=0
for name,shp in O.layer_shapes(img):
    n=int(np.prod(shp)); a=g[off:off+n]; b=g_ref[off:off+n]
    print("%-10s %-16s |got| %.4e |ref| %.4e  maxerr/maxref %.3e  corr %.4f" % (name, shp, np.abs(a).max(), np.abs(b).max(), np.abs(a-b).max()/np.abs(b).max(), float(np.dot(a,b)/(np.linalg.norm(a)*np.linalg.norm(b)+1e-30))))
    off+=n
