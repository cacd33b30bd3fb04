"""CPU oracle: NumPy restatement of the reference's assoc-VAE hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
(``vae_assoc_amd``) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (navigator8972/vae_assoc, Python 2 + TensorFlow 1.x +
prettytensor) cannot run here and ships no tests, golden vectors or checkpoints for this
path.  This file is a line-by-line reading of the reference source;
it is validated (tests/test_oracle.py) against an independent torch-autograd derivation,
central finite differences and closed-form spot checks, not against reference outputs.

What follows which reference lines (all paths relative to /root/reference):
  xavier_init              vae_assoc.py:11-18
  layer sizes / wiring     vae_assoc.py:78-119   (one eps [B,n_z] shared by all modalities, :90)
  encoder (MLP branch)     vae_assoc.py:185-188, 201-204, 212-221
  decoder (MLP branch)     vae_assoc.py:257-260, 280-283, 293-303  (sized from n_hidden_recog_*)
  losses                   vae_assoc.py:306-371
  Adam                     vae_assoc.py:373-374  (tf.train.AdamOptimizer defaults, TF-1 update rule)
  method semantics         vae_assoc.py:378-425
  train loop               vae_assoc.py:498-583
TensorFlow semantics that are not visible in the reference source (marked [TF]) are from
knowledge of TF 1.x: tf.nn.l2_loss = sum(t**2)/2; Adam adds epsilon to sqrt(v), with
lr_t = lr*sqrt(1-b2^t)/(1-b1^t); softplus = log(1+exp(x)).

The flat parameter layout (the order every get/set_params API in this repo uses) is the
reference's variable-creation order, per modality:
  enc W1[n_in,H1] b1[H1] W2[H1,H2] b2[H2] ... Wmu[HL,n_z] bmu[n_z] Wsig[HL,n_z] bsig[n_z]
  dec V1[n_z,H1]  c1[H1] V2[H1,H2] c2[H2] ... Vout[HL,n_in] cout[n_in]
each matrix row-major [fan_in, fan_out] (vae_assoc.py:185-215, 257-300).
"""
from itertools import combinations

import numpy as np

ADAM_BETA1 = 0.9      # [TF] tf.train.AdamOptimizer defaults
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8


# ----------------------------------------------------------------------------- architecture
def hidden_sizes(na):
    """Hidden widths of one modality's encoder.  The reference is hard-wired to two hidden
    layers n_hidden_recog_1/2 (vae_assoc.py:185-204); the optional key ``n_hidden`` (a list)
    generalises it for the 4x1024 stress config.  The MLP decoder reuses the *recognition*
    sizes, not n_hidden_gener_* (vae_assoc.py:257,280,293,299) -- reproduced here."""
    if "n_hidden" in na and na["n_hidden"] is not None:
        return [int(h) for h in na["n_hidden"]]
    return [int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])]


def layer_shapes(na):
    """[(name, (fan_in, fan_out))...] for one modality in flat-layout order (W then b)."""
    hs = hidden_sizes(na)
    n_in, n_z = int(na["n_input"]), int(na["n_z"])
    shapes = []
    prev = n_in
    for i, h in enumerate(hs):
        shapes.append(("enc_W%d" % (i + 1), (prev, h)))
        shapes.append(("enc_b%d" % (i + 1), (h,)))
        prev = h
    shapes.append(("enc_Wmu", (prev, n_z)))
    shapes.append(("enc_bmu", (n_z,)))
    shapes.append(("enc_Wsig", (prev, n_z)))
    shapes.append(("enc_bsig", (n_z,)))
    prev = n_z
    for i, h in enumerate(hs):
        shapes.append(("dec_W%d" % (i + 1), (prev, h)))
        shapes.append(("dec_b%d" % (i + 1), (h,)))
        prev = h
    shapes.append(("dec_Wout", (prev, n_in)))
    shapes.append(("dec_bout", (n_in,)))
    return shapes


def param_count(archs):
    return int(sum(int(np.prod(s)) for na in archs for _, s in layer_shapes(na)))


def xavier_init(fan_in, fan_out, rng, constant=1.0):
    """vae_assoc.py:11-18.  U(-c*sqrt(6/(fi+fo)), +c*sqrt(6/(fi+fo))), shape (fan_in, fan_out).
    The TF RNG stream cannot be reproduced; ``rng`` is a numpy Generator."""
    high = constant * np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-high, high, size=(fan_in, fan_out))


def init_params(archs, rng, dtype=np.float64):
    """Weights xavier-uniform, biases zero (vae_assoc.py:185-215,257-300)."""
    params = []
    for na in archs:
        p = {}
        for name, shp in layer_shapes(na):
            if len(shp) == 2:
                p[name] = xavier_init(shp[0], shp[1], rng).astype(dtype)
            else:
                p[name] = np.zeros(shp, dtype=dtype)
        params.append(p)
    return params


def flatten_params(archs, params):
    return np.concatenate([np.asarray(p[name]).reshape(-1)
                           for na, p in zip(archs, params) for name, _ in layer_shapes(na)])


def unflatten_params(archs, flat, dtype=None):
    flat = np.asarray(flat)
    out, off = [], 0
    for na in archs:
        p = {}
        for name, shp in layer_shapes(na):
            n = int(np.prod(shp))
            a = flat[off:off + n].reshape(shp)
            p[name] = a.astype(dtype) if dtype is not None else a.copy()
            off += n
        out.append(p)
    assert off == flat.size, (off, flat.size)
    return out


# ----------------------------------------------------------------------------- activations
def _softplus(a):
    return np.logaddexp(a, 0.0)          # [TF] tf.nn.softplus = log(1 + exp(a))


def _sigmoid(a):
    out = np.empty_like(a)
    pos = a >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-a[pos]))
    e = np.exp(a[~pos])
    out[~pos] = e / (1.0 + e)
    return out


ACT = {
    "relu": (lambda a: np.maximum(a, 0.0), lambda a, y: (a > 0).astype(a.dtype)),
    "softplus": (_softplus, lambda a, y: _sigmoid(a)),
    "sigmoid": (_sigmoid, lambda a, y: y * (1.0 - y)),
    "tanh": (np.tanh, lambda a, y: 1.0 - y * y),
    "identity": (lambda a: a, lambda a, y: np.ones_like(a)),
}


def act_name(transfer_fct):
    """Accept 'relu'/'softplus'/... or a TF-like callable (tf.nn.relu.__name__ == 'relu')."""
    if transfer_fct is None:
        return "identity"
    if isinstance(transfer_fct, str):
        name = transfer_fct
    else:
        name = getattr(transfer_fct, "__name__", str(transfer_fct))
    name = name.lower()
    if name not in ACT:
        raise ValueError("unsupported transfer_fct %r" % (transfer_fct,))
    return name


# ----------------------------------------------------------------------------- bf16 emulation
def bf16_round(a):
    """Round-to-nearest-even to bfloat16, returned as float64 (what v_cvt_pk_bf16_f32 does)."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32)
    u2 = ((u + ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return u2.view(np.float32).astype(np.float64).reshape(np.shape(a))


# The HIP path with bf16 operands rounds exactly at these points (DESIGN.md "precision"):
# GEMM inputs (x, weights AND biases -- the bias rides in the GEMM as an extra weight row),
# every stored hidden activation, z, and every stored activation-gradient; products are
# accumulated in fp32 and mu / lv / logits / losses / weight gradients / Adam stay fp32.
# ``quant='bf16'`` reproduces that placement on the CPU so the bf16 kernels can be checked
# tightly (same roundings -> same relu decisions); ``quant=None`` is the plain restatement.
# Transfer-function derivatives are then taken from the STORED (rounded) output, as the kernels do.
DACT_FROM_OUTPUT = {
    "relu": lambda y: (y > 0).astype(y.dtype),
    "softplus": lambda y: 1.0 - np.exp(-y),
    "sigmoid": lambda y: y * (1.0 - y),
    "tanh": lambda y: 1.0 - y * y,
    "identity": lambda y: np.ones_like(y),
}


def _q(quant):
    if quant is None:
        return lambda a: a
    if quant == "bf16":
        return bf16_round
    raise ValueError("quant must be None or 'bf16'")


# ----------------------------------------------------------------------------- forward
def encode(na, p, x, act, quant=None):
    """_recognition_network, MLP branch (vae_assoc.py:185-188,201-204,212-221)."""
    f, _ = ACT[act]
    q = _q(quant)
    hs = hidden_sizes(na)
    h = q(x)
    acts, pre = [h], []
    for i in range(len(hs)):
        a = h @ q(p["enc_W%d" % (i + 1)]) + q(p["enc_b%d" % (i + 1)])
        h = q(f(a))
        pre.append(a)
        acts.append(h)
    mu = h @ q(p["enc_Wmu"]) + q(p["enc_bmu"])
    lv = h @ q(p["enc_Wsig"]) + q(p["enc_bsig"])
    return mu, lv, {"acts": acts, "pre": pre}


def decode(na, p, z, act, binary, quant=None):
    """_generator_network, MLP branch (vae_assoc.py:257-260,280-283,293-303)."""
    f, _ = ACT[act]
    q = _q(quant)
    hs = hidden_sizes(na)
    g = q(z)
    acts, pre = [g], []
    for i in range(len(hs)):
        a = g @ q(p["dec_W%d" % (i + 1)]) + q(p["dec_b%d" % (i + 1)])
        g = q(f(a))
        pre.append(a)
        acts.append(g)
    logits = g @ q(p["dec_Wout"]) + q(p["dec_bout"])
    xhat = _sigmoid(logits) if binary else logits
    return xhat, {"acts": acts, "pre": pre, "logits": logits}


def forward(archs, params, X, eps, binary, act, quant=None):
    """_create_network (vae_assoc.py:78-119): ONE eps [B,n_z] shared by every modality (:90),
    z = mu + sqrt(exp(lv))*eps (:102-103)."""
    out = []
    for na, p, x, b in zip(archs, params, X, binary):
        mu, lv, ec = encode(na, p, x, act, quant)
        z = mu + np.sqrt(np.exp(lv)) * eps
        xhat, dc = decode(na, p, z, act, b, quant)
        out.append({"mu": mu, "lv": lv, "z": z, "xhat": xhat, "enc": ec, "dec": dc})
    return out


# ----------------------------------------------------------------------------- loss
def loss_terms(archs, fw, X, binary, weights, assoc_lambda):
    """_create_loss_optimizer (vae_assoc.py:306-371), written the way the reference writes it."""
    n_z = int(archs[0]["n_z"])                       # vae_assoc.py:89
    recon, latent, vae_costs = [], [], []
    for f, x, b, w in zip(fw, X, binary, weights):
        xr, mu, lv = f["xhat"], f["mu"], f["lv"]
        if b:                                        # :321-324  -> [B]
            r = -np.sum(x * np.log(1e-3 + xr) + (1 - x) * np.log(1e-3 + 1 - xr), axis=1)
        else:                                        # :327-328  [TF] l2_loss = sum(t^2)/2 -> SCALAR over batch
            r = np.sum((x - xr) ** 2) / 2.0
        k = -0.5 * np.sum(1 + lv - mu ** 2 - np.exp(lv), axis=1)      # :335-337 -> [B]
        recon.append(r)
        latent.append(k)
        # :340  reduce_mean(reconstr_loss + latent_loss) * weight.  In the Gaussian case the
        # scalar r broadcasts over [B], so the recon term is NOT divided by B.
        vae_costs.append(np.mean(r + k) * w)
    assoc = []
    for i, j in combinations(range(len(archs)), 2):  # :346-366
        mi, mj, li, lj = fw[i]["mu"], fw[j]["mu"], fw[i]["lv"], fw[j]["lv"]
        a = np.sum(0.5 * (np.sum(lj, 1) - np.sum(li, 1) - n_z
                          + np.sum(np.exp(li - lj), 1)
                          + np.sum((mj - mi) ** 2 * np.exp(-lj), 1)))
        b_ = np.sum(0.5 * (np.sum(li, 1) - np.sum(lj, 1) - n_z
                           + np.sum(np.exp(lj - li), 1)
                           + np.sum((mi - mj) ** 2 * np.exp(-li), 1)))
        assoc.append(a + b_)
    cost = sum(vae_costs)
    if assoc:                                        # :368-371
        cost = cost + assoc_lambda * sum(assoc)
    return {"cost": cost, "recon": recon, "latent": latent, "vae_costs": vae_costs, "assoc": assoc}


# Data-parallel restatement (the build adds this; the reference is single-process).
def shard_cost(archs, fw, X, binary, weights, assoc_lambda, batch_global):
    """Cost contribution of one row-shard such that sum over shards == full-batch cost:
    mean terms scaled by 1/batch_global, sum terms (Gaussian recon, assoc) by 1."""
    n_z = int(archs[0]["n_z"])
    c = 0.0
    for f, x, b, w in zip(fw, X, binary, weights):
        xr, mu, lv = f["xhat"], f["mu"], f["lv"]
        k = -0.5 * np.sum(1 + lv - mu ** 2 - np.exp(lv))
        if b:
            r = -np.sum(x * np.log(1e-3 + xr) + (1 - x) * np.log(1e-3 + 1 - xr))
            c += w * (r + k) / batch_global
        else:
            c += w * (np.sum((x - xr) ** 2) / 2.0 + k / batch_global)
    for i, j in combinations(range(len(archs)), 2):
        mi, mj, li, lj = fw[i]["mu"], fw[j]["mu"], fw[i]["lv"], fw[j]["lv"]
        c += assoc_lambda * np.sum(0.5 * (np.exp(li - lj) + np.exp(lj - li) - 2.0
                                         + (mi - mj) ** 2 * (np.exp(-li) + np.exp(-lj))))
    return c


# ----------------------------------------------------------------------------- backward
def backward(archs, params, fw, X, eps, binary, weights, assoc_lambda, act, batch_global=None, quant=None):
    """Analytic gradient of ``cost`` w.r.t. every parameter (what TF autodiff of
    vae_assoc.py:373-374 produces).  Formulas: SURVEY.md 8(a) row A5.  Mean terms carry
    1/batch_global, sum terms (Gaussian recon, assoc) carry 1."""
    _, dact_pre = ACT[act]
    q = _q(quant)
    if quant is None:
        dact = dact_pre
    else:
        dact = lambda a, y: DACT_FROM_OUTPUT[act](y)      # noqa: E731  (from the stored, rounded output)
    M = len(archs)
    B = X[0].shape[0]
    Bg = B if batch_global is None else batch_global
    # gradients landing directly on (mu, lv): KL + association
    dmu = [w / Bg * f["mu"] for f, w in zip(fw, weights)]
    dlv = [w / (2.0 * Bg) * (np.exp(f["lv"]) - 1.0) for f, w in zip(fw, weights)]
    for i, j in combinations(range(M), 2):
        mi, mj, li, lj = fw[i]["mu"], fw[j]["mu"], fw[i]["lv"], fw[j]["lv"]
        d = mi - mj
        a = li - lj
        gm = assoc_lambda * d * (np.exp(-li) + np.exp(-lj))
        dmu[i] = dmu[i] + gm
        dmu[j] = dmu[j] - gm
        dlv[i] = dlv[i] + 0.5 * assoc_lambda * (np.exp(a) - np.exp(-a) - d ** 2 * np.exp(-li))
        dlv[j] = dlv[j] + 0.5 * assoc_lambda * (np.exp(-a) - np.exp(a) - d ** 2 * np.exp(-lj))
    grads = []
    for m, (na, p, f, x, b, w) in enumerate(zip(archs, params, fw, X, binary, weights)):
        hs = hidden_sizes(na)
        L = len(hs)
        g = {}
        xr = f["xhat"]
        if b:   # d/dlogit of -(x log(1e-3+p) + (1-x) log(1e-3+1-p)), p = sigmoid(logit)
            dl = (w / Bg) * xr * (1 - xr) * (-x / (1e-3 + xr) + (1 - x) / (1e-3 + 1 - xr))
        else:   # d/dxhat of w * sum (x-xhat)^2 / 2
            dl = w * (xr - x)
        dl = q(dl)
        dacts, dpre = f["dec"]["acts"], f["dec"]["pre"]
        g["dec_Wout"] = dacts[L].T @ dl
        g["dec_bout"] = dl.sum(0)
        dg = dl @ q(p["dec_Wout"]).T
        for i in range(L - 1, -1, -1):
            da = q(dg * dact(dpre[i], dacts[i + 1]))
            g["dec_W%d" % (i + 1)] = dacts[i].T @ da
            g["dec_b%d" % (i + 1)] = da.sum(0)
            dg = da @ q(p["dec_W%d" % (i + 1)]).T
        dz = dg
        # reparameterisation: z = mu + exp(lv/2)*eps
        gmu = q(dmu[m] + dz)
        glv = q(dlv[m] + dz * 0.5 * np.sqrt(np.exp(f["lv"])) * eps)
        eacts, epre = f["enc"]["acts"], f["enc"]["pre"]
        g["enc_Wmu"] = eacts[L].T @ gmu
        g["enc_bmu"] = gmu.sum(0)
        g["enc_Wsig"] = eacts[L].T @ glv
        g["enc_bsig"] = glv.sum(0)
        dh = gmu @ q(p["enc_Wmu"]).T + glv @ q(p["enc_Wsig"]).T
        for i in range(L - 1, -1, -1):
            da = q(dh * dact(epre[i], eacts[i + 1]))
            g["enc_W%d" % (i + 1)] = eacts[i].T @ da
            g["enc_b%d" % (i + 1)] = da.sum(0)
            if i > 0:
                dh = da @ q(p["enc_W%d" % (i + 1)]).T
        grads.append(g)
    return grads, {"dmu_direct": dmu, "dlv_direct": dlv}


# ----------------------------------------------------------------------------- Adam
def adam_step(theta, m, v, g, t, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """[TF] tf.train.AdamOptimizer (TF 1.x) dense update, t = 1 for the first step:
        lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
        theta -= lr_t * m / (sqrt(v) + eps)      (epsilon OUTSIDE the bias correction)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# ----------------------------------------------------------------------------- model
class OracleAssocVAE(object):
    """Mirror of AssocVariationalAutoEncoder (vae_assoc.py:20-463) on NumPy.

    Differences forced by the environment: weights and eps are explicit (TF's RNG stream is
    not reproducible); ``transfer_fct`` is a name; there is no session/checkpoint."""

    def __init__(self, network_architectures, binary=True, transfer_fct="softplus", weights=1.0,
                 assoc_lambda=1.0, learning_rate=0.001, batch_size=100, dtype=np.float64,
                 seed=0, params_flat=None, quant=None):
        self.network_architectures = network_architectures
        self.assoc_lambda = assoc_lambda
        M = len(network_architectures)
        if type(binary) is list:                      # vae_assoc.py:31-35
            assert len(binary) == M
            self.binary = binary
        else:
            self.binary = [binary] * M
        if type(weights) is list:                     # :37-41
            assert len(weights) == M
            self.weights = weights
        else:
            self.weights = [weights] * M
        self.act = act_name(transfer_fct)
        self.learning_rate = learning_rate
        self.batch_size = batch_size
        self.dtype = dtype
        self.quant = quant                                # None, or 'bf16' = emulate the bf16-operand kernels
        self.n_z = int(network_architectures[0]["n_z"])   # :89
        self.rng = np.random.default_rng(seed)
        if params_flat is None:
            self.params = init_params(network_architectures, self.rng, dtype)
        else:
            self.params = unflatten_params(network_architectures, params_flat, dtype)
        P = param_count(network_architectures)
        self.m = np.zeros(P, dtype=dtype)
        self.v = np.zeros(P, dtype=dtype)
        self.t = 0

    # -- helpers
    def get_params(self):
        return flatten_params(self.network_architectures, self.params)

    def set_params(self, flat):
        self.params = unflatten_params(self.network_architectures, flat, self.dtype)

    def _eps(self, eps, rows):
        if eps is None:
            eps = self.rng.standard_normal((rows, self.n_z))
        return np.asarray(eps, dtype=self.dtype)

    def _cast(self, X):
        return [np.asarray(x, dtype=self.dtype) for x in X]

    def cost_and_grads(self, X, eps, batch_global=None):
        X = self._cast(X)
        eps = self._eps(eps, X[0].shape[0])
        fw = forward(self.network_architectures, self.params, X, eps, self.binary, self.act, self.quant)
        if batch_global is None:
            cost = loss_terms(self.network_architectures, fw, X, self.binary, self.weights,
                              self.assoc_lambda)["cost"]
        else:
            cost = shard_cost(self.network_architectures, fw, X, self.binary, self.weights,
                              self.assoc_lambda, batch_global)
        grads, _ = backward(self.network_architectures, self.params, fw, X, eps, self.binary,
                            self.weights, self.assoc_lambda, self.act, batch_global, self.quant)
        return cost, flatten_params(self.network_architectures, grads), fw

    def apply_gradients(self, gflat):
        self.t += 1
        th, self.m, self.v = adam_step(self.get_params(), self.m, self.v,
                                       np.asarray(gflat, dtype=self.dtype), self.t,
                                       self.learning_rate)
        self.set_params(th)

    # -- reference surface
    def partial_fit(self, X, eps=None):
        """vae_assoc.py:378-386: one Adam step; returns the cost of the SAME forward pass
        that produced the gradients (pre-update weights)."""
        cost, g, _ = self.cost_and_grads(X, eps)
        self.apply_gradients(g)
        return float(cost)

    def evaluate_cost(self, X, eps=None):
        """vae_assoc.py:388-391."""
        X = self._cast(X)
        eps = self._eps(eps, X[0].shape[0])
        fw = forward(self.network_architectures, self.params, X, eps, self.binary, self.act, self.quant)
        return float(loss_terms(self.network_architectures, fw, X, self.binary, self.weights,
                                self.assoc_lambda)["cost"])

    def transform(self, X, sens_idx=None):
        """vae_assoc.py:393-403: posterior means only."""
        if sens_idx is None:
            return [encode(na, p, np.asarray(x, dtype=self.dtype), self.act, self.quant)[0]
                    for na, p, x in zip(self.network_architectures, self.params, X)]
        assert sens_idx < len(self.network_architectures)
        return encode(self.network_architectures[sens_idx], self.params[sens_idx],
                      np.asarray(X, dtype=self.dtype), self.act, self.quant)[0]

    def generate(self, z_mu=None):
        """vae_assoc.py:405-419: decoder only, z fed directly."""
        if z_mu is None:
            z_mu = np.random.normal(size=(self.batch_size, self.n_z))
        z_mu = np.asarray(z_mu, dtype=self.dtype)
        return [decode(na, p, z_mu, self.act, b, self.quant)[0]
                for na, p, b in zip(self.network_architectures, self.params, self.binary)]

    def reconstruct(self, X, eps=None):
        """vae_assoc.py:421-425: encode -> sample -> decode, one sess.run per modality, i.e.
        a FRESH eps per modality; ``eps`` may be a list (one per modality) for determinism."""
        out = []
        for m, (na, p, x, b) in enumerate(zip(self.network_architectures, self.params, X, self.binary)):
            x = np.asarray(x, dtype=self.dtype)
            mu, lv, _ = encode(na, p, x, self.act, self.quant)
            e = self._eps(None if eps is None else eps[m], x.shape[0])
            z = mu + np.sqrt(np.exp(lv)) * e
            out.append(decode(na, p, z, self.act, b, self.quant)[0])
        return out


def train(data_sets, network_architectures, binary=True, weights=1.0, assoc_lambda=1e-5,
          learning_rate=0.001, batch_size=100, training_epochs=10, display_step=5,
          early_stop=False, dtype=np.float64, seed=0, params_flat=None, eps_fn=None, verbose=False):
    """vae_assoc.py:498-583 on the oracle model (transfer_fct fixed to relu, :502)."""
    model = OracleAssocVAE(network_architectures, binary, "relu", weights, assoc_lambda,
                           learning_rate, batch_size, dtype=dtype, seed=seed, params_flat=params_flat)
    n_samples = data_sets.train._data.shape[0]
    sens = np.concatenate([[0], np.cumsum([na["n_input"] for na in network_architectures])])
    M = len(network_architectures)
    hist = []
    valid_cost = None
    step = 0
    for epoch in range(training_epochs):
        avg_cost = 0.
        total_batch = int(n_samples / batch_size)
        if early_stop and epoch % early_stop == 0:
            cur = 0
            nvb = int(data_sets.validation._data.shape[0] / batch_size)
            for _ in range(nvb):
                bx, _l = data_sets.validation.next_batch(batch_size)
                seg = [bx[:, sens[k]:sens[k + 1]] for k in range(M)]
                cur += model.evaluate_cost(seg, None if eps_fn is None else eps_fn(-1)) / nvb
            if valid_cost is not None and cur > valid_cost:
                break
            valid_cost = cur
        for _ in range(total_batch):
            bx, _l = data_sets.train.next_batch(batch_size)
            seg = [bx[:, sens[k]:sens[k + 1]] for k in range(M)]
            cost = model.partial_fit(seg, None if eps_fn is None else eps_fn(step))
            step += 1
            avg_cost += cost / n_samples * batch_size      # :576
            hist.append(avg_cost)                           # :577 (running sum, per batch)
        if verbose and epoch % display_step == 0:
            print("Epoch:", '%04d' % (epoch + 1), "cost=", "{:.9f}".format(avg_cost))
    return model, hist
