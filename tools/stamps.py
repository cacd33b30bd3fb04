#!/usr/bin/env python3
"""Diagnostic (not part of the product): builds libavae with -DAVAE_STAMPS into
gpurun_out/stamps/, runs the bench workload and prints, per launch of one graph-replayed step,
where a block's time goes (item lookup / first tile / K loop / epilogue), the block start spread
and the shader clock.  Read the SHARES, not the absolute length (stamps fence the schedule)."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    out = os.path.join(ROOT, "gpurun_out", "stamps")
    pkg = os.path.join(out, "vae_assoc_amd")
    if os.path.exists(out):
        shutil.rmtree(out)
    shutil.copytree(os.path.join(ROOT, "vae_assoc_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    src = [os.path.join(ROOT, "vae_assoc_amd", "csrc", f) for f in ("avae_kernels.hip", "avae_host.hip", "avae_comm.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DAVAE_STAMPS"] + os.environ.get("EXTRA_DEFS", "").split()
                   + src + ["-o", os.path.join(pkg, "libavae.so")], check=True)
    import torch
    import bench
    sys.path.insert(0, out)      # after bench (which puts ROOT first): the stamps build must win
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    archs, B, dtype, label = bench.CONFIGS[cfg]
    model = AssocVariationalAutoEncoder(archs, transfer_fct="relu", batch_size=B, compute_dtype=dtype, seed=0, **bench.HYPER)
    rng = np.random.default_rng(0)
    img, jnt = bench.synth(rng, 4 * B)
    data = torch.as_tensor(np.concatenate([img, jnt], axis=1)).cuda()
    batches = [[data[i * B:(i + 1) * B, :784], data[i * B:(i + 1) * B, 784:]] for i in range(4)]
    for i in range(300 if cfg != "c4" else 20):
        model.partial_fit(batches[i % 4], return_cost=False)
    torch.cuda.synchronize()
    nl, nb, nw = 32, 1024, 8
    buf = np.zeros(nl * nb * nw, dtype=np.uint64)
    cnt = C.c_size_t(0)
    rc = model._L.avae_debug_fetch(model._h, b"stamps", buf.ctypes.data_as(C.c_void_p), buf.size * 2, C.byref(cnt))
    assert rc == 0, model._L.avae_last_error(model._h)
    st = buf.reshape(nl, nb, nw).astype(np.int64)
    names = ['fwd_enc1', 'fwd_enc2', 'fwd_head', 'fwd_dec1', 'fwd_dec2', 'fwd_out_loss', 'bwd_out', 'bwd_dec2',
             'bwd_dec1_latent', 'bwd_head', 'bwd_enc2', 'wgrad']
    if cfg != "c4" and not os.environ.get("AVAE_NO_TAIL"):      # tail products: two launches fewer (avae_host.hip::fuse_tail)
        names = ['fwd_enc1', 'fwd_enc2', 'fwd_head+fwd_dec1', 'fwd_dec2', 'fwd_out_loss', 'bwd_out', 'bwd_dec2',
                 'bwd_dec1_latent+bwd_head', 'bwd_enc2', 'wgrad(+adam)']
    print("%-16s %6s %8s | %7s %7s %7s %7s | %8s %7s  (us; realtime ticks are 10 ns)" % (
        "launch", "blocks", "span", "lookup", "tile0", "kloop", "epilog", "startspr", "clkMHz"))
    prev_end = None
    if cfg == "c4":
        names = ['fwd_enc1', 'fwd_enc2', 'fwd_enc3', 'fwd_enc4', 'fwd_head', 'fwd_dec1', 'fwd_dec2', 'fwd_dec3', 'fwd_dec4',
                 'fwd_out_loss', 'bwd_out', 'bwd_dec4', 'bwd_dec3', 'bwd_dec2', 'bwd_dec1_latent', 'bwd_head',
                 'bwd_enc4', 'bwd_enc3', 'bwd_enc2', 'wgrad1', 'wgrad2']
    for l in range(min(nl, len(names))):
        s = st[l]
        live = s[:, 0] > 0
        if not live.any():
            continue
        s = s[live]
        gemm = s[:, 1] > 0
        t0, t4 = s[:, 0], s[:, 4]
        span = (t4.max() - t0.min()) / 100.0
        gap = (t0.min() - prev_end) / 100.0 if prev_end is not None else 0.0
        prev_end = t4.max()
        g = s[gemm]
        def seg(a, b):
            return ((g[:, b] - g[:, a]).mean() / 100.0) if len(g) else 0.0
        e1 = g[g[:, 6] > 0]
        ep = "acc->lds %.2f pass1 %.2f rest %.2f" % (((e1[:, 5] - e1[:, 3]).mean() / 100.0), ((e1[:, 6] - e1[:, 5]).mean() / 100.0),
                                                     ((e1[:, 4] - e1[:, 6]).mean() / 100.0)) if len(e1) else ""
        if len(e1) and (e1[:, 7] > 0).any():
            ep += " 6->7 %.2f" % ((e1[:, 7] - e1[:, 6]).mean() / 100.0)
        if "AVAE_STAMPS_PRO" in os.environ.get("EXTRA_DEFS", "") and len(g):
            ep = "entry->item %.2f  item->addresses %.2f  addresses->loop %.2f" % (
                (g[:, 6] - g[:, 0]).mean() / 100.0, (g[:, 7] - g[:, 6]).mean() / 100.0, (g[:, 1] - g[:, 7]).mean() / 100.0)
        if os.environ.get("STAMPS_DETAIL") and os.environ["STAMPS_DETAIL"] in (names[l] if l < len(names) else ""):
            base = t0.min()
            q = [0, 10, 25, 50, 75, 90, 100]
            for nm, col in (("entry", 0), ("loop start", 1), ("tile0 landed", 2), ("loop end", 3), ("end", 4)):
                print("      %-13s percentiles %s us after the first block's entry: %s" % (
                    nm, q, " ".join("%.2f" % ((np.percentile(g[:, col], x) - base) / 100.0) for x in q)))
        print("%-16s %6d %8.2f | %7.2f %7.2f %7.2f %7.2f | %8.2f  gap_before=%.2f  %s" % (
            names[l] if l < len(names) else "L%d" % l, live.sum(), span, seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4),
            (t0.max() - t0.min()) / 100.0, gap, ep))


if __name__ == "__main__":
    main()
