#!/usr/bin/env python3
"""Diagnostic (not part of the product): seeded random models against the oracle for a time budget, THREE steps each (the compute-
dtype shadows and moments a step leaves behind only show in the next one), plus the default plan against its unfused twin
(AVAE_NO_ADAM_FUSE=1, AVAE_NO_LEAN=1 ...) bitwise.  tests/test_gpu_parity.py::test_random_shapes is the fixed-seed subset that runs
in the suite; this tool is for spending GPU minutes on shapes nobody thought of.

    python tools/fuzz_parity.py [seconds] [seed] [conv]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    import test_gpu_parity as T
    from conftest import make_arch, synth_batch
    from vae_assoc_amd import vae_assoc as V
    rng = np.random.default_rng(seed)
    acts = ["relu", "softplus", "tanh", "sigmoid", "identity"]
    t0, case, bad = time.time(), 0, []
    conv_mode = len(sys.argv) > 3 and sys.argv[3] == "conv"
    while conv_mode and time.time() - t0 < budget:
        # conv / deconv image branches (random depths, 1-3 modalities of which at least one is conv), THREE steps, fp32 and bf16, the
        # default implicit-GEMM policy and every stage implicit / explicit -- each against the oracle (the routes sum in different
        # orders: no bitwise twin here)
        M = int(rng.integers(1, 4))
        nz = int(rng.choice([int(x) for x in os.environ["FUZZ_NZ"].split(",")] if os.environ.get("FUZZ_NZ") else [2, 5, 8, 20, 33, 64]))
        B = int(rng.choice([3, 8, 17, 32, 64]))
        dtype = str(rng.choice(["fp32", "fp32", "bf16"]))
        conv = [True] + [bool(rng.integers(0, 2)) for _ in range(M - 1)]
        rng.shuffle(conv)
        archs, binary, w = [], [], []
        for m in range(M):
            if conv[m]:
                r1, r2 = int(rng.integers(1, 25)), int(rng.integers(1, 81))
                g1, g2 = int(rng.choice([2, 6, 8, 16, 20, 64, 130, 160])), int(rng.integers(1, 21))
                archs.append(dict(make_arch("c%d" % m, 784, r1, r2, nz), hidden_conv=True, n_hidden_gener_1=g1, n_hidden_gener_2=g2))
                binary.append(True)
            else:
                archs.append(make_arch("m%d" % m, int(rng.integers(1, 200)), int(rng.integers(1, 90)), int(rng.integers(1, 90)), nz))
                binary.append(bool(rng.integers(0, 2)))
            w.append(float(rng.choice([0.5, 1.0, 3.0])))
        lam = float(rng.choice([0.0, 0.3, 8.0]))
        policy = str(rng.choice(["", "E:fwb,H:fwb,D1:fwb,DT:fwb", "none"]))
        desc = "conv case %d seed %d: %s nz=%d B=%d policy=%r conv=%s archs=%s" % (case, seed, dtype, nz, B, policy, conv, [
            (a.get("n_hidden_recog_1"), a.get("n_hidden_recog_2"), a.get("n_hidden_gener_1"), a.get("n_hidden_gener_2")) for a in archs])
        try:
            for k in ("AVAE_IMPL_POLICY", "AVAE_NO_IMPLICIT"):
                os.environ.pop(k, None)
            if policy == "none":
                os.environ["AVAE_NO_IMPLICIT"] = "1"
            elif policy:
                os.environ["AVAE_IMPL_POLICY"] = policy
            T.check_step_parity(V, archs, binary, w, lam, "relu", B, dtype, steps=3, seed=2000 + case, drift_tol=5e-3 if dtype == "fp32" else 2e-2)
        except Exception as e:
            msg = repr(e)
            import re
            nums = [float(x) for x in re.findall(r"(?<![\w.])(?:\d+\.\d+(?:e[-+]?\d+)?)", msg)] if "gradient mismatch" in msg else []
            structural = not isinstance(e, AssertionError) or (nums and max(nums) > 0.2)
            bad.append((desc, msg[:400], structural))
            print("STRUCTURAL" if structural else "tolerance", desc, "\n     ", msg[:400], flush=True)
        case += 1
        if case % 5 == 0:
            print("%d cases, %d beyond a tolerance or failed, %.0f s" % (case, len(bad), time.time() - t0), flush=True)
    while not conv_mode and time.time() - t0 < budget:
        M = int(rng.integers(1, 4))
        nz = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 20, 31, 32, 33, 48, 64]))
        B = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 129, 200, 256, 300]))
        dtype = str(rng.choice(["fp32", "bf16"]))
        archs, binary, w = [], [], []
        for m in range(M):
            top = int(rng.choice([40, 150, 600]))
            hs = [int(rng.integers(1, top + 1)) for _ in range(int(rng.integers(1, 4)))]
            archs.append(make_arch("m%d" % m, int(rng.integers(1, 801)), 0, 0, nz, n_hidden=hs))
            binary.append(bool(rng.integers(0, 2)))
            w.append(float(rng.choice([0.5, 1.0, 3.0, 50.0])))
        lam = float(rng.choice([0.0, 1e-5, 0.3, 8.0]))
        act = acts[case % len(acts)]
        desc = "case %d seed %d: %s M=%d nz=%d B=%d act=%s lam=%g archs=%s binary=%s" % (
            case, seed, dtype, M, nz, B, act, lam, [(a["n_input"], a["n_hidden"]) for a in archs], binary)
        try:
            for k in ("AVAE_NO_ADAM_FUSE", "AVAE_NO_LEAN", "AVAE_NO_TAIL", "AVAE_NO_XCD_PIECES"):
                os.environ.pop(k, None)
            model, _emu, X, eps = T.check_step_parity(V, archs, binary, w, lam, act, B, dtype, steps=3, seed=1000 + case)
            ref_state = (model.get_params(), model.get_grads()) + tuple(model.get_opt_state()[:2])
            del model
            for env in ({"AVAE_NO_ADAM_FUSE": "1"}, {"AVAE_NO_LEAN": "1", "AVAE_NO_TAIL": "1"}):
                os.environ.update(env)
                twin, _e, _x, _eps = T.check_step_parity(V, archs, binary, w, lam, act, B, dtype, steps=3, seed=1000 + case)
                st = (twin.get_params(), twin.get_grads()) + tuple(twin.get_opt_state()[:2])
                for k in env:
                    os.environ.pop(k)
                for nm, a, b in zip(("params", "grads", "m", "v"), ref_state, st):
                    if not np.array_equal(a, b):
                        raise AssertionError("default plan vs %s: %s differ in %d entries (max %.3e)" % (env, nm, int((a != b).sum()), float(np.abs(a - b).max())))
                del twin
        except Exception as e:                       # keep going: collect every failing shape
            # "tolerance": an oracle bound exceeded by relu-kink or bf16-rounding flips (the suite's bounds are set for its own
            # shapes; wider layers flip more often) -- small by construction; "STRUCTURAL": routes that disagree with each other,
            # errors of order one, or anything that is not an assertion
            import re
            msg = repr(e)
            nums = [float(x) for x in re.findall(r"(?<![\w.])(?:\d+\.\d+(?:e[-+]?\d+)?)", msg)] if "gradient mismatch" in msg else []
            structural = not isinstance(e, AssertionError) or "default plan vs" in msg or (nums and max(nums) > 0.2)
            bad.append((desc, msg[:400], structural))
            print("STRUCTURAL" if structural else "tolerance", desc, "\n     ", msg[:400], flush=True)
        case += 1
        if case % 10 == 0:
            print("%d cases, %d beyond a tolerance or failed, %.0f s" % (case, len(bad), time.time() - t0), flush=True)
    hard = [b for b in bad if b[2]]
    print("done: %d cases (seed %d): %d beyond an oracle tolerance, %d STRUCTURAL" % (case, seed, len(bad) - len(hard), len(hard)))
    for d, e, _s in hard:
        print(d, "\n    ", e)
    sys.exit(1 if hard else 0)


if __name__ == "__main__":
    main()
