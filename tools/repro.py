import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import make_arch, synth_batch
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
archs=[make_arch("a", 60, 20, 16, 5), make_arch("b", 21, 12, 10, 5)]
B=9
m = AssocVariationalAutoEncoder(archs, binary=[True, False], transfer_fct="relu", weights=[50.0,1.0], assoc_lambda=8.0, batch_size=B, compute_dtype=dtype, seed=5)
rng=np.random.default_rng(0)
X=synth_batch(rng,B,[60,21],[True,False]); eps=rng.standard_normal((B,5)).astype(np.float32)
print("transform", flush=True); m.transform(X)
print("eval", flush=True); print(m.evaluate_cost(X,eps))
print("fit", flush=True); print(m.partial_fit(X,eps))
