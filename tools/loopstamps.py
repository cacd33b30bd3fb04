#!/usr/bin/env python3
"""Diagnostic (not part of the product): builds libavae with -DAVAE_STAMPS -DAVAE_LOOPSTAMPS into gpurun_out/loopstamps/
and prints, per launch of one eager step, the shader-clock cycles an average K-loop iteration of a workgroup's wave 0
spends in: vmcnt wait | barrier | slab-0 LDS reads | DMA issue | slab-1 LDS reads | MFMAs (drained).
The build skips the epilogue: results are garbage, only the K loop is being looked at."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
    out = os.path.join(ROOT, "gpurun_out", "loopstamps")
    pkg = os.path.join(out, "vae_assoc_amd")
    if os.path.exists(out):
        shutil.rmtree(out)
    shutil.copytree(os.path.join(ROOT, "vae_assoc_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    src = [os.path.join(ROOT, "vae_assoc_amd", "csrc", f) for f in ("avae_kernels.hip", "avae_host.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DAVAE_STAMPS", "-DAVAE_LOOPSTAMPS"]
                   + os.environ.get("EXTRA_DEFS", "").split() + src + ["-o", os.path.join(pkg, "libavae.so")], check=True)
    import torch
    import bench
    sys.path.insert(0, out)      # after bench (which puts ROOT first): the diagnostic build must win
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    archs, B, dtype, label = bench.CONFIGS[cfg]
    model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **bench.HYPER)
    rng = np.random.default_rng(0)
    img, jnt = bench.synth(rng, 2 * B)
    data = torch.as_tensor(np.concatenate([img, jnt], axis=1)).cuda()
    batches = [[data[i * B:(i + 1) * B, :784], data[i * B:(i + 1) * B, 784:]] for i in range(2)]
    for i in range(6):
        model.partial_fit(batches[i % 2], return_cost=False)
    torch.cuda.synchronize()
    nl, nb, nw = 32, 1024, 8
    buf = np.zeros(nl * nb * nw, dtype=np.uint64)
    cnt = C.c_size_t(0)
    rc = model._L.avae_debug_fetch(model._h, b"stamps", buf.ctypes.data_as(C.c_void_p), buf.size * 2, C.byref(cnt))
    assert rc == 0, model._L.avae_last_error(model._h)
    st = buf.reshape(nl, nb, nw).astype(np.int64)
    print("%-6s %6s %5s | %7s %7s %7s %7s %7s %7s | %7s   (cycles per K tile, mean over workgroups)" % (
        "launch", "blocks", "nk", "vmwait", "barrier", "lds0", "dma", "lds1", "mfma", "total"))
    for l in range(nl):
        s = st[l]
        live = (s[:, 0] > 0) & (s[:, 7] > 0)
        if not live.any():
            continue
        s = s[live]
        per = s[:, 1:7] / s[:, 7:8]
        m = per.mean(axis=0)
        print("L%-5d %6d %5d | %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f | %7.0f" % (l, live.sum(), int(np.median(s[:, 7])), *m, m.sum()))


if __name__ == "__main__":
    main()
