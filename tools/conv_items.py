import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
archs, B, dtype, label = bench.CONFIGS["c2conv"]
m = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, use_graph=False, **bench.HYPER)
rng = np.random.default_rng(0)
img, jnt = bench.synth(rng, B)
m.partial_fit([img, jnt])
