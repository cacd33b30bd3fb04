"""CPU tests of the oracle itself (no GPU): it is checked against (1) an independent torch
autograd transcription of the reference's loss, (2) central finite differences, (3) closed-form
spot checks from SURVEY.md 8c, and (4) the committed golden fixtures."""
import itertools
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_arch, synth_batch
from oracle import vae_assoc_oracle as O


def torch_cost(archs, flat, X, eps, binary, weights, lam, act):
    """Independent transcription of reference vae_assoc.py:163-222,243-304,306-371 in torch fp64
    (autograd supplies the gradients the way TF's autodiff does)."""
    f = {"relu": torch.relu, "softplus": torch.nn.functional.softplus, "tanh": torch.tanh,
         "sigmoid": torch.sigmoid, "identity": lambda a: a}[act]
    off = 0
    mus, lvs, costs = [], [], []
    n_z = archs[0]["n_z"]
    for na, x, b, w in zip(archs, X, binary, weights):
        p = {}
        for name, shp in O.layer_shapes(na):
            n = int(np.prod(shp))
            p[name] = flat[off:off + n].reshape(shp)
            off += n
        L = len(O.hidden_sizes(na))
        h = x
        for i in range(L):
            h = f(h @ p["enc_W%d" % (i + 1)] + p["enc_b%d" % (i + 1)])
        mu = h @ p["enc_Wmu"] + p["enc_bmu"]
        lv = h @ p["enc_Wsig"] + p["enc_bsig"]
        z = mu + torch.sqrt(torch.exp(lv)) * eps
        g = z
        for i in range(L):
            g = f(g @ p["dec_W%d" % (i + 1)] + p["dec_b%d" % (i + 1)])
        a = g @ p["dec_Wout"] + p["dec_bout"]
        if b:
            xr = torch.sigmoid(a)
            r = -torch.sum(x * torch.log(1e-3 + xr) + (1 - x) * torch.log(1e-3 + 1 - xr), 1)
        else:
            xr = a
            r = torch.sum((x - xr) ** 2) / 2
        k = -0.5 * torch.sum(1 + lv - mu ** 2 - torch.exp(lv), 1)
        costs.append(torch.mean(r + k) * w)
        mus.append(mu)
        lvs.append(lv)
    cost = sum(costs)
    for i, j in itertools.combinations(range(len(archs)), 2):
        a1 = torch.sum(0.5 * (lvs[j].sum(1) - lvs[i].sum(1) - n_z + torch.exp(lvs[i] - lvs[j]).sum(1)
                              + ((mus[j] - mus[i]) ** 2 * torch.exp(-lvs[j])).sum(1)))
        a2 = torch.sum(0.5 * (lvs[i].sum(1) - lvs[j].sum(1) - n_z + torch.exp(lvs[j] - lvs[i]).sum(1)
                              + ((mus[i] - mus[j]) ** 2 * torch.exp(-lvs[i])).sum(1)))
        cost = cost + lam * (a1 + a2)
    return cost


CASES = [
    dict(archs=[make_arch("image", 60, 20, 16, 5), make_arch("joint", 21, 12, 10, 5)], binary=[True, False],
         weights=[50.0, 1.0], lam=8.0, act="relu", B=9),
    dict(archs=[make_arch("image", 60, 20, 16, 3), make_arch("joint", 21, 12, 10, 3), make_arch("aux", 17, 8, 8, 3)],
         binary=[True, False, False], weights=[2.0, 1.0, 0.5], lam=0.7, act="softplus", B=6),
    dict(archs=[make_arch("a", 30, 0, 0, 4, n_hidden=[12, 10, 8]), make_arch("b", 11, 0, 0, 4, n_hidden=[7])],
         binary=[False, True], weights=[1.0, 3.0], lam=1e-2, act="tanh", B=5),
    dict(archs=[make_arch("solo", 25, 9, 7, 2)], binary=[True], weights=[1.0], lam=3.0, act="sigmoid", B=4),
]


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_torch_autograd(case):
    rng = np.random.default_rng(5)
    archs, B = case["archs"], case["B"]
    nz = archs[0]["n_z"]
    X = synth_batch(rng, B, [a["n_input"] for a in archs], case["binary"])
    eps = rng.standard_normal((B, nz))
    m = O.OracleAssocVAE(archs, case["binary"], case["act"], case["weights"], case["lam"], 1e-3, B, seed=11)
    th = m.get_params() + 0.05 * rng.standard_normal(O.param_count(archs))     # non-zero biases
    m.set_params(th)
    cost, g, _ = m.cost_and_grads(X, eps)
    t = torch.tensor(th, dtype=torch.float64, requires_grad=True)
    tc = torch_cost(archs, t, [torch.tensor(x, dtype=torch.float64) for x in X], torch.tensor(eps),
                    case["binary"], case["weights"], case["lam"], case["act"])
    tc.backward()
    assert abs(cost - tc.item()) <= 1e-11 * abs(tc.item())
    tg = t.grad.numpy()
    assert np.abs(g - tg).max() <= 1e-9 * max(1.0, np.abs(tg).max())


def test_conv_primitives_match_torch():
    """conv2d / conv2d_transpose restatements (incl. TF SAME alignment: pad 1 before / 2 after for k=5, s=2;
    the transposed conv is the full one cropped 1 before) against torch, forward and both gradients."""
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    for (H, ci, co, k, s, pad) in [(28, 1, 4, 5, 2, "SAME"), (14, 4, 8, 5, 2, "SAME"), (7, 8, 6, 5, 1, "VALID")]:
        x, W = rng.standard_normal((3, H, H, ci)), rng.standard_normal((k, k, ci, co))
        y = O.conv2d(x, W, s, pad)
        xt, Wt = torch.tensor(x, requires_grad=True), torch.tensor(W, requires_grad=True)
        xp = xt.permute(0, 3, 1, 2)
        xp = F.pad(xp, (1, 2, 1, 2)) if pad == "SAME" else xp
        yt = F.conv2d(xp, Wt.permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1)
        assert np.abs(y - yt.detach().numpy()).max() < 1e-12
        dy = rng.standard_normal(y.shape)
        (yt * torch.tensor(dy)).sum().backward()
        dx, dW = O.conv2d_bwd(x, W, s, pad, dy)
        assert np.abs(dx - xt.grad.numpy()).max() < 1e-12 and np.abs(dW - Wt.grad.numpy()).max() < 1e-11
    for (H, ci, co, k, s, pad) in [(1, 5, 6, 3, 1, "VALID"), (3, 6, 3, 5, 1, "VALID"), (7, 3, 4, 5, 2, "SAME"), (14, 4, 1, 5, 2, "SAME")]:
        x, W = rng.standard_normal((2, H, H, ci)), rng.standard_normal((k, k, co, ci))
        y = O.deconv2d(x, W, s, pad)
        assert y.shape[1] == O.deconv_out_size(H, k, s, pad)
        xt, Wt = torch.tensor(x, requires_grad=True), torch.tensor(W, requires_grad=True)
        pb, OH = (1 if pad == "SAME" else 0), y.shape[1]
        yt = F.conv_transpose2d(xt.permute(0, 3, 1, 2), Wt.permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1)[:, pb:pb + OH, pb:pb + OH]
        assert np.abs(y - yt.detach().numpy()).max() < 1e-12
        u = rng.standard_normal(y.shape)          # adjoint of the conv that maps the output image back to x
        assert abs((y * u).sum() - (x * O.conv2d(u, W, s, pad)).sum()) < 1e-10
        dy = rng.standard_normal(y.shape)
        (yt * torch.tensor(dy)).sum().backward()
        dx, dW = O.deconv2d_bwd(x, W, s, pad, dy)
        assert np.abs(dx - xt.grad.numpy()).max() < 1e-12 and np.abs(dW - Wt.grad.numpy()).max() < 1e-11


def test_conv_branch_gradients_by_finite_differences():
    img = dict(make_arch("image", 784, 4, 6, 5), hidden_conv=True, n_hidden_gener_1=8, n_hidden_gener_2=4)
    jnt = make_arch("joint", 21, 12, 10, 5)
    archs = [img, jnt]
    assert [n for n, _ in O.layer_shapes(img)][:3] == ["enc_C1", "enc_C2", "enc_C3"]
    rng = np.random.default_rng(3)
    X = [rng.random((3, 784)), rng.standard_normal((3, 21))]
    eps = rng.standard_normal((3, 5))
    m = O.OracleAssocVAE(archs, [True, False], "relu", [2.0, 1.0], 0.5, 1e-3, 3, seed=1)
    th = m.get_params() + 0.02 * rng.standard_normal(O.param_count(archs))
    m.set_params(th)
    _, g, _ = m.cost_and_grads(X, eps)
    off, picks = 0, []
    for name, shp in O.layer_shapes(img):          # a few entries of every tensor of the conv modality
        n = int(np.prod(shp))
        picks += list(off + rng.integers(0, n, 3))
        off += n
    for i in picks:
        h = 1e-6
        t2 = th.copy(); t2[i] += h; m.set_params(t2); cp = m.evaluate_cost(X, eps)
        t2[i] -= 2 * h; m.set_params(t2); cm = m.evaluate_cost(X, eps)
        fd = (cp - cm) / (2 * h)
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_oracle_finite_differences():
    case = CASES[1]
    rng = np.random.default_rng(2)
    archs, B = case["archs"], case["B"]
    X = synth_batch(rng, B, [a["n_input"] for a in archs], case["binary"])
    eps = rng.standard_normal((B, archs[0]["n_z"]))
    m = O.OracleAssocVAE(archs, case["binary"], case["act"], case["weights"], case["lam"], 1e-3, B, seed=1)
    th = m.get_params()
    _, g, _ = m.cost_and_grads(X, eps)
    for i in rng.integers(0, th.size, 25):
        h = 1e-6
        t2 = th.copy(); t2[i] += h; m.set_params(t2); cp = m.evaluate_cost(X, eps)
        t2[i] -= 2 * h; m.set_params(t2); cm = m.evaluate_cost(X, eps)
        fd = (cp - cm) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(g[i]))


def test_closed_form_spot_checks():
    """mu=0,lv=0 => KL=0; identical posteriors => assoc=0 with zero assoc-grad;
    p=x in {0,1} => BCE = -log(1.001) per dim; Gaussian recon is a batch SUM (not a mean)."""
    archs = [make_arch("a", 6, 4, 4, 3), make_arch("b", 5, 4, 4, 3)]
    B = 4
    z = np.zeros((B, 3))
    fw = [{"mu": z, "lv": z, "xhat": np.array([[1., 0, 1, 0, 1, 0]] * B)},
          {"mu": z, "lv": z, "xhat": np.ones((B, 5))}]
    X = [np.array([[1., 0, 1, 0, 1, 0]] * B), np.zeros((B, 5))]
    t = O.loss_terms(archs, fw, X, [True, False], [1.0, 1.0], 2.0)
    assert all(np.allclose(k, 0) for k in t["latent"])
    assert np.allclose(t["assoc"], 0)
    assert np.allclose(t["recon"][0], -6 * np.log(1.001))
    assert np.isclose(t["recon"][1], B * 5 / 2.0)                      # sum over the batch
    assert np.isclose(t["vae_costs"][1], B * 5 / 2.0)                  # ... and NOT divided by B
    assert np.isclose(t["cost"], -6 * np.log(1.001) + B * 5 / 2.0)
    # assoc gradient vanishes for identical posteriors
    rng = np.random.default_rng(0)
    mu, lv = rng.standard_normal((B, 3)), rng.standard_normal((B, 3))
    fw2 = [{"mu": mu, "lv": lv, "xhat": fw[0]["xhat"]}, {"mu": mu.copy(), "lv": lv.copy(), "xhat": fw[1]["xhat"]}]
    t2 = O.loss_terms(archs, fw2, X, [True, False], [0.0, 0.0], 5.0)
    assert abs(t2["assoc"][0]) < 1e-12


def test_adam_first_step_moves_by_lr_sign():
    """TF-1 Adam, t=1: lr_t*m/(sqrt(v)+eps) = lr*sqrt(.001)*.1*g/((sqrt(.001)*|g|+1e-8)*.1) ~ lr*sign(g)."""
    g = np.array([3.0, -2.0, 1e-3, -50.0])
    th, m, v = O.adam_step(np.zeros(4), np.zeros(4), np.zeros(4), g, 1, 1e-3)
    assert np.allclose(th, -1e-3 * np.sign(g), rtol=1e-3)      # |g| >> 1e-8
    assert np.allclose(m, 0.1 * g) and np.allclose(v, 0.001 * g * g)
    # epsilon sits outside the bias correction: for tiny g the step shrinks exactly this way
    gs = np.array([1e-9])
    th2, _, _ = O.adam_step(np.zeros(1), np.zeros(1), np.zeros(1), gs, 1, 1e-3)
    expect = -1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * gs) / (np.sqrt(0.001 * gs * gs) + 1e-8)
    assert np.allclose(th2, expect)


def test_shard_sum_equals_full_batch():
    case = CASES[0]
    rng = np.random.default_rng(9)
    archs, B = case["archs"], 12
    X = synth_batch(rng, B, [a["n_input"] for a in archs], case["binary"])
    eps = rng.standard_normal((B, archs[0]["n_z"]))
    m = O.OracleAssocVAE(archs, case["binary"], case["act"], case["weights"], case["lam"], 1e-3, B, seed=4)
    c, g, _ = m.cost_and_grads(X, eps)
    cs, gs = 0.0, 0.0
    for r in range(3):
        sl = slice(4 * r, 4 * r + 4)
        c_, g_, _ = m.cost_and_grads([x[sl] for x in X], eps[sl], batch_global=B)
        cs, gs = cs + c_, gs + g_
    assert abs(c - cs) < 1e-10 * abs(c) and np.abs(g - gs).max() < 1e-10 * np.abs(g).max()


def test_flat_layout_is_reference_variable_order():
    na = make_arch("image", 784, 500, 500, 20)
    names = [n for n, _ in O.layer_shapes(na)]
    assert names == ["enc_W1", "enc_b1", "enc_W2", "enc_b2", "enc_Wmu", "enc_bmu", "enc_Wsig", "enc_bsig",
                     "dec_W1", "dec_b1", "dec_W2", "dec_b2", "dec_Wout", "dec_bout"]
    jn = make_arch("joint", 147, 200, 200, 20)
    assert O.param_count([na]) == 1316824 and O.param_count([jn]) == 151787
    # decoder is sized from n_hidden_recog_* even when n_hidden_gener_* differ (reference quirk)
    q = dict(na, n_hidden_gener_1=7, n_hidden_gener_2=9)
    assert O.layer_shapes(q) == O.layer_shapes(na)


@pytest.mark.parametrize("name", ["script_nz4_b64", "c1_nz20_b100"])
def test_oracle_reproduces_golden(golden, name):
    G = golden[name]
    c = G["config"]
    M = len(c["archs"])
    X = [G["x%d" % m] for m in range(M)]
    model = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], c["assoc_lambda"], c["lr"], c["B"],
                             params_flat=G["params0"].astype(np.float64))
    cost, g, fw = model.cost_and_grads(X, G["eps"][0])
    assert abs(cost - G["cost0"]) <= 1e-12 * abs(G["cost0"])
    assert np.abs(g - G["grads0"]).max() <= 1e-6 * np.abs(G["grads0"]).max()
    for m in range(M):
        assert np.allclose(fw[m]["mu"], G["mu%d" % m], rtol=0, atol=1e-12)
        assert np.allclose(fw[m]["lv"], G["lv%d" % m], rtol=0, atol=1e-12)
        assert np.allclose(fw[m]["xhat"][:16], G["xhat%d" % m], rtol=0, atol=1e-6)
    costs = [model.partial_fit(X, G["eps"][s]) for s in range(3)]
    assert np.allclose(costs, G["costs"], rtol=1e-12)
    assert np.abs(model.get_params() - G["params3"]).max() < 1e-6
    assert abs(np.linalg.norm(model.m) - G["adam_m3_norm"]) < 1e-9 * G["adam_m3_norm"]
    # the fp32 run of the oracle stays within the stated fp32 tolerance of the fp64 one
    m32 = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], c["assoc_lambda"], c["lr"], c["B"],
                           dtype=np.float32, params_flat=G["params0"])
    c32, _, _ = m32.cost_and_grads(X, G["eps"][0])
    assert abs(c32 - G["cost0"]) <= 1e-5 * abs(G["cost0"])


# ----------------------------------------------------------------------------- round-2 fixtures (sampled gradients)
def _load_big():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_big", os.path.join(GOLDEN, "make_golden_big.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name", ["conv_small", "c5_small"])
def test_oracle_reproduces_round2_small_fixtures(name):
    """conv/deconv branch and the 3-modality net: the committed fixtures pin the oracle on those branches against silent edits
    (cost, per-tensor gradient maxima / norms / sampled entries, weights after two Adam steps), fp64 and bf16-emulating runs."""
    big = _load_big()
    G = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    c, X, eps, p0 = big.small_inputs(name)
    chk = [float(x.astype(np.float64).sum()) for x in X] + [float(eps.astype(np.float64).sum()), float(p0.astype(np.float64).sum())]
    assert np.allclose(chk, G["checksum"], rtol=1e-12), "the seeded inputs are not the ones the fixture was made from"
    for tag, quant in (("f64", None), ("bf16", "bf16")):
        m = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], c["assoc_lambda"], c["lr"], c["B"],
                             params_flat=p0.astype(np.float64), quant=quant)
        cost, g, fw = m.cost_and_grads(X, eps[0])
        assert abs(cost - float(G["cost0_" + tag])) <= 1e-10 * abs(cost)
        assert np.allclose(g[G["sample_idx"]], G["gsample_" + tag], rtol=1e-9, atol=1e-12)
        costs = [m.partial_fit(X, eps[s]) for s in range(2)]
        assert np.allclose(costs, G["costs_" + tag], rtol=1e-10)
        assert np.allclose(m.get_params()[G["sample_idx"]], G["p2sample_" + tag], rtol=1e-9, atol=1e-12)


def test_oracle_reproduces_c4_fixture():
    """BASELINE C4 at its full size (4 x 1024, n_z = 64, batch 4096): ~6 s of fp64 BLAS here."""
    big = _load_big()
    G = np.load(os.path.join(GOLDEN, "c4_b4096.npz"), allow_pickle=False)
    c = big.C4
    X, eps, p0 = big.c4_inputs()
    chk = [float(X[0].astype(np.float64).sum()), float(X[1].astype(np.float64).sum()), float(eps.astype(np.float64).sum()), float(p0.astype(np.float64).sum())]
    assert np.allclose(chk, G["checksum"], rtol=1e-12)
    m = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], c["assoc_lambda"], c["lr"], c["B"], params_flat=p0.astype(np.float64))
    cost, g, _fw = m.cost_and_grads(X, eps)
    assert abs(cost - float(G["cost_f64"])) <= 1e-10 * abs(cost)
    assert np.allclose(g[G["sample_idx"]], G["gsample_f64"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("name,tag,li", [("c3", "f64", None), ("c3", "bf16", None), ("c5", "f64_lam0", 0), ("c5", "f64_lam5", 5)])
def test_oracle_reproduces_global_batch_fixtures(name, tag, li):
    """Round 3: BASELINE C3 / C5 at their global batch of 2048 rows (tests/golden/make_golden_dp.py).  The fixture pins the oracle
    against silent edits, and -- the data-parallel contract of SURVEY.md 8e -- the SUM over 8 shards of 256 rows, each computed with
    batch_global = 2048, is the one-batch gradient and cost (vae_assoc.py:319-371 decides which terms carry 1/B)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_dp", os.path.join(GOLDEN, "make_golden_dp.py"))
    dp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dp)
    G = np.load(os.path.join(GOLDEN, name + "_b2048.npz"), allow_pickle=False)
    c, X, eps, p0 = dp.inputs(name)
    chk = [float(x.astype(np.float64).sum()) for x in X] + [float(eps.astype(np.float64).sum()), float(p0.astype(np.float64).sum())]
    assert np.allclose(chk, G["checksum"], rtol=1e-12)
    lam = c["assoc_lambda"] if li is None else dp.LAMBDAS[li]
    quant = "bf16" if tag == "bf16" else None
    m = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], lam, c["lr"], c["B"], params_flat=p0.astype(np.float64), quant=quant)
    cost, g, _ = m.cost_and_grads(X, eps[0])
    assert abs(cost - float(G["cost_" + tag])) <= 1e-10 * abs(cost)
    assert np.allclose(g[G["sample_idx"]], G["gsample_" + tag], rtol=1e-8, atol=1e-12)
    if quant is None:               # shard sum (exact arithmetic: fp64 sums differ by rounding only)
        gs, cs = np.zeros_like(g), 0.0
        for r in range(8):
            sh = O.OracleAssocVAE(c["archs"], c["binary"], c["act"], c["weights"], lam, c["lr"], 256, params_flat=p0.astype(np.float64))
            cr, gr, _ = sh.cost_and_grads([x[256 * r:256 * (r + 1)] for x in X], eps[0][256 * r:256 * (r + 1)], batch_global=2048)
            gs += gr
            cs += cr
        assert abs(cs - cost) <= 1e-10 * abs(cost)
        assert np.abs(gs - g).max() <= 1e-10 * np.abs(g).max()
    costs = [m.partial_fit(X, eps[s]) for s in range(dp.STEPS)]
    assert np.allclose(costs, G["costs_" + tag], rtol=1e-10)
    assert np.allclose(m.get_params()[G["sample_idx"]], G["p3sample_" + tag], rtol=1e-9, atol=1e-12)
