"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol
include/avae.h declares, sizes a workspace on the host, and refuses to run without a GPU
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, make_arch


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from vae_assoc_amd import _capi
    return _capi


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "avae.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(avae_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "libavae.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared, "python binding and include/avae.h disagree"


def _config(capi, n_z=20, B=100):
    cfg = capi.Config()
    cfg.abi_version = capi.AVAE_ABI_VERSION
    cfg.n_modalities = 2
    for m, (n_in, h) in enumerate(((784, 500), (147, 200))):
        cfg.mod[m].n_input = n_in
        cfg.mod[m].n_hidden_layers = 2
        cfg.mod[m].n_hidden[0] = h
        cfg.mod[m].n_hidden[1] = h
        cfg.mod[m].binary = 1 - m
        cfg.mod[m].weight = 1.0
    cfg.n_z, cfg.batch_size, cfg.activation, cfg.compute_dtype = n_z, B, 1, 1
    cfg.learning_rate, cfg.assoc_lambda = 1e-3, 1.0
    return cfg


def test_workspace_size_is_host_only_and_sane(capi):
    L = capi.lib()
    n = C.c_size_t(0)
    assert L.avae_workspace_bytes(C.byref(_config(capi)), C.byref(n)) == 0
    P = 1468611
    assert 4 * 4 * P < n.value < 64 * 4 * P          # theta, m, v, g + shadows + activations
    cfg = _config(capi)
    cfg.compute_dtype = 0
    n32 = C.c_size_t(0)
    assert L.avae_workspace_bytes(C.byref(cfg), C.byref(n32)) == 0 and n32.value > n.value


def test_bad_configs_are_rejected_with_a_message(capi):
    L = capi.lib()
    n = C.c_size_t(0)
    for mutate, needle in ((lambda c: setattr(c, "abi_version", 99), "abi_version"),
                           (lambda c: setattr(c, "n_z", 65), "n_z"),
                           (lambda c: setattr(c, "batch_size", 0), "batch_size"),
                           (lambda c: setattr(c.mod[0], "hidden_conv", 1), "hidden_conv"),
                           (lambda c: setattr(c, "n_modalities", 5), "n_modalities")):
        cfg = _config(capi)
        mutate(cfg)
        assert L.avae_workspace_bytes(C.byref(cfg), C.byref(n)) != 0
        assert needle in L.avae_last_error(None).decode()


def test_no_cpu_fallback(capi):
    """Without a HIP device avae_create must fail loudly, and so must the Python model."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu suite")
    L = capi.lib()
    h = C.c_void_p()
    assert L.avae_create(C.byref(_config(capi)), C.byref(h)) != 0
    assert not h.value
    assert L.avae_last_error(None)
    from vae_assoc_amd.vae_assoc import AssocVariationalAutoEncoder
    with pytest.raises(RuntimeError):
        AssocVariationalAutoEncoder([make_arch("image", 784, 500, 500, 20)], batch_size=8)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vae_assoc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_bench_accounting_matches_survey_figures():
    """bench.py's algorithmic work per step (what roofline.achieved is priced with) against the figures of SURVEY.md 8d:
    train FLOPs per paired sample 7 946 000 (C2) / 87 390 208 (C4), weights only -- bench also counts the bias rows."""
    import bench
    for cfg, per_sample in (("c2", 7946000), ("c4", 87390208)):
        archs, B, dtype, _ = bench.CONFIGS[cfg]
        work, P = bench.launch_work(archs, B, 2)
        flops = sum(f for _, f in work.values())
        assert abs(flops / B - per_sample) / per_sample < 0.01, (cfg, flops / B)
        assert P == (1468611 if cfg == "c2" else 14900387)
        # every launch name bench prices is a launch the library reports (names mirror build_training_plan)
        assert {"prep", "adam", "fwd_head", "fwd_out_loss", "bwd_out", "bwd_dec1_latent", "bwd_head"} <= set(work)


def test_dp_plan_is_host_only_and_tiles_the_gradient_buffer(capi):
    """avae_dp_plan (no GPU needed): the data-parallel buckets of a configuration.  MLP models: the master layout holds the encoder
    side of EVERY modality first, then every decoder side, then the cost slot, so that bucket 1 (encoder) and bucket 0 (decoder +
    cost) are ONE contiguous range each (VERDICT r2 #2a: two collectives per step instead of 2 x modalities) that tile
    [0, P_int + 1) exactly; comm_buckets = 1 gives the single all-reduce of the whole buffer (north_star's literal design); P_int
    carries less than one 128-byte line of padding per matrix row over the reference's parameter count.  A conv modality: one bucket."""
    L = capi.lib()

    def plan(cfg):
        nb, nr = C.c_int32(0), (C.c_int32 * 2)()
        offs, cnts = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        assert L.avae_dp_plan(C.byref(cfg), C.byref(nb), nr, offs, cnts) == 0
        out, k = [], 0
        for b in range(2):
            out.append([(int(offs[k + i]), int(cnts[k + i])) for i in range(nr[b])])
            k += nr[b]
        return nb.value, out
    nb, (b0, b1) = plan(_config(capi, B=256))
    assert nb == 2 and len(b0) == 1 and len(b1) == 1
    (o1, c1), (o0, c0) = b1[0], b0[0]
    assert o1 == 0 and o0 == c1, "encoder sides first, decoder sides + cost right behind: no gap, no overlap"
    total = o0 + c0                                 # = P_int + 1 (cost slot)
    P = 1468611
    rows = (785 + 501 + 501 + 21 + 501 + 501) + (148 + 201 + 201 + 21 + 201 + 201)
    assert P + 1 <= total <= P + 1 + 31 * rows      # rows are whole 128-byte lines, not padded to the K unit
    assert (total - 1) * 4 < 1.05 * P * 4           # C2: 6.13 MB on the wire for 5.87 MB of parameters (6.6 MB with K padding)
    # encoder sides of both modalities: 500 -> 512, 200 -> 224, 40 -> 64 floats per row
    assert c1 == (785 * 512 + 501 * 512 + 501 * 64) + (148 * 224 + 201 * 224 + 201 * 64)
    assert c1 % 32 == 0 and (c0 - 1) % 32 == 0      # whole 128-byte lines: the exchange kernels move 8-float granules
    for world in range(1, 9):
        cfg = _config(capi, B=256)
        cfg.use_comm, cfg.world_size, cfg.rank = capi.COMM_IPC, world, world - 1
        assert plan(cfg) == (2, [b0, b1])           # the plan does not depend on the collective's backend or the rank
    cfg = _config(capi, B=256)
    cfg.comm_buckets = 1
    nb, (s0, s1) = plan(cfg)
    assert nb == 1 and s0 == [(0, total)] and s1 == []
    cfg = _config(capi, B=64)
    cfg.mod[0].hidden_conv = 1
    cfg.mod[0].n_hidden[0], cfg.mod[0].n_hidden[1] = 8, 16
    cfg.mod[0].conv_gener[0], cfg.mod[0].conv_gener[1] = 16, 8
    nb, (b0, b1) = plan(cfg)
    assert nb == 1 and len(b0) == 1 and b1 == [] and b0[0][0] == 0


def test_config_struct_matches_the_header(capi):
    """The ctypes mirror of avae_config has the size and field offsets the C compiler gives include/avae.h."""
    import subprocess
    import tempfile
    src = """#include <stdio.h>\n#include <stddef.h>\n#include "avae.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %d\\n",sizeof(avae_config),sizeof(avae_modality),
offsetof(avae_config,seed),offsetof(avae_config,use_comm),offsetof(avae_config,comm_buckets),offsetof(avae_config,nccl_id),offsetof(avae_config,wire_dtype),AVAE_ABI_VERSION);return 0;}"""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    Cf = capi.Config
    want = [C.sizeof(Cf), C.sizeof(capi.Modality), Cf.seed.offset, Cf.use_comm.offset, Cf.comm_buckets.offset, Cf.nccl_id.offset,
            Cf.wire_dtype.offset, capi.AVAE_ABI_VERSION]
    assert [int(x) for x in out] == want, (out, want)
