import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import make_arch, synth_batch
from vae_assoc_amd import vae_assoc as V
def shadow(m):
    buf = np.zeros(4, np.float32); cnt = C.c_size_t(0)
    rc = m._L.avae_debug_fetch(m._h, b"shadow_err", buf.ctypes.data_as(C.c_void_p), 4, C.byref(cnt))
    assert rc == 0, m._L.avae_last_error(m._h)
    return buf.tolist()
rng = np.random.default_rng(0)
cases = {
 "c2": ([make_arch("image", 784, 500, 500, 20), make_arch("joint", 147, 200, 200, 20)], 256),
 "conv": ([dict(make_arch("image", 784, 16, 64, 20), hidden_conv=True, n_hidden_gener_1=64, n_hidden_gener_2=16), make_arch("joint", 147, 200, 200, 20)], 64),
 "conv_odd": ([dict(make_arch("image", 784, 7, 30, 5), hidden_conv=True, n_hidden_gener_1=20, n_hidden_gener_2=3), make_arch("joint", 147, 33, 21, 5)], 17),
 "c4": ([make_arch("image", 784, 0, 0, 64, n_hidden=[1024]*4), make_arch("joint", 147, 0, 0, 64, n_hidden=[1024]*4)], 4096),
}
for name, (archs, B) in cases.items():
    for dtype in ("fp32", "bf16"):
        m = V.AssocVariationalAutoEncoder(archs, binary=[True, False], transfer_fct="relu", weights=[50, 1], assoc_lambda=8.0, batch_size=B, compute_dtype=dtype, seed=1)
        X = synth_batch(rng, B, [784, 147], [True, False])
        s0 = shadow(m)
        for _ in range(2): m.partial_fit(X)
        print(name, dtype, "after init", s0, "after 2 steps", shadow(m), flush=True)
        del m
