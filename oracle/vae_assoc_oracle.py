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
  conv encoder branch      vae_assoc.py:169-183, 190-199, 206-210, 480-489 (bias-less convs, NO activation)
  deconv decoder branch    vae_assoc.py:249-255, 262-278, 286-291, 491-496; deconv.py:76-128, 133-161
                           (conv2d_transpose + bias + sigmoid on every layer, then a dense n_input x n_input + sigmoid)
TensorFlow semantics that are not visible in the reference source (marked [TF]) are from
knowledge of TF 1.x: tf.nn.l2_loss = sum(t**2)/2; Adam adds epsilon to sqrt(v), with
lr_t = lr*sqrt(1-b2^t)/(1-b1^t); softplus = log(1+exp(x)).

The flat parameter layout (the order every get/set_params API in this repo uses) is the
reference's variable-creation order, per modality:
  enc W1[n_in,H1] b1[H1] W2[H1,H2] b2[H2] ... Wmu[HL,n_z] bmu[n_z] Wsig[HL,n_z] bsig[n_z]
  dec V1[n_z,H1]  c1[H1] V2[H1,H2] c2[H2] ... Vout[HL,n_in] cout[n_in]
each matrix row-major [fan_in, fan_out] (vae_assoc.py:185-215, 257-300).
For a modality with hidden_conv=True (binary modalities only; the non-binary conv decoder of the
reference is shape-broken, vae_assoc.py:299) the order is the creation order of that branch:
  enc C1[5,5,1,R1] C2[5,5,R1,2R1] C3[5,5,2R1,R2] Wmu[9R2,n_z] bmu Wsig[9R2,n_z] bsig
  dec T1.W[3,3,G1,n_z] T1.b[G1] T2.W[5,5,G1/2,G1] T2.b T3.W[5,5,G2,G1/2] T3.b T4.W[5,5,1,G2] T4.b
      Wout[n_in,n_in] bout[n_in]
(conv filters [k,k,in,out] as tf.nn.conv2d takes them, vae_assoc.py:482; transposed-conv filters
[k,k,out_depth,in_depth], deconv.py:78).
"""
from itertools import combinations

import numpy as np

ADAM_BETA1 = 0.9      # [TF] tf.train.AdamOptimizer defaults
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8


# ----------------------------------------------------------------------------- conv primitives (NHWC)
def _same_pad_before(in_size, k, s):
    """[TF] SAME padding: out = ceil(in/s), total pad = max((out-1)*s + k - in, 0), the SMALLER half
    goes before.  For k=5, s=2 on 28 or 14 pixels: 1 before / 2 after, i.e. out[o] = sum_k in[2o+k-1] W[k]."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


def conv2d(x, W, s, padding):
    """tf.nn.conv2d(x, W, strides=(1,s,s,1), padding) (vae_assoc.py:484-486); x [B,H,W,Ci], W [k,k,Ci,Co]."""
    B, H, Wd, Ci = x.shape
    k = W.shape[0]
    if padding == "SAME":
        OH, pb = _same_pad_before(H, k, s)
        OW, _ = _same_pad_before(Wd, k, s)
    else:
        OH, OW, pb = (H - k) // s + 1, (Wd - k) // s + 1, 0
    xp = np.zeros((B, (OH - 1) * s + k, (OW - 1) * s + k, Ci), dtype=x.dtype)
    hh, ww = min(H, xp.shape[1] - pb), min(Wd, xp.shape[2] - pb)
    xp[:, pb:pb + hh, pb:pb + ww] = x[:, :hh, :ww]
    y = np.zeros((B, OH, OW, W.shape[3]), dtype=np.result_type(x, W))
    for kh in range(k):
        for kw in range(k):
            y += xp[:, kh:kh + (OH - 1) * s + 1:s, kw:kw + (OW - 1) * s + 1:s] @ W[kh, kw]
    return y


def conv2d_bwd(x, W, s, padding, dy):
    """Gradients of conv2d w.r.t. its input and filter."""
    B, H, Wd, Ci = x.shape
    k = W.shape[0]
    OH, OW = dy.shape[1], dy.shape[2]
    pb = _same_pad_before(H, k, s)[1] if padding == "SAME" else 0
    PH, PW = (OH - 1) * s + k, (OW - 1) * s + k
    xp = np.zeros((B, PH, PW, Ci), dtype=x.dtype)
    hh, ww = min(H, PH - pb), min(Wd, PW - pb)
    xp[:, pb:pb + hh, pb:pb + ww] = x[:, :hh, :ww]
    dxp = np.zeros((B, PH, PW, Ci), dtype=dy.dtype)
    dW = np.zeros(W.shape, dtype=dy.dtype)
    for kh in range(k):
        for kw in range(k):
            sl = (slice(None), slice(kh, kh + (OH - 1) * s + 1, s), slice(kw, kw + (OW - 1) * s + 1, s))
            dxp[sl] += dy @ W[kh, kw].T
            dW[kh, kw] = np.tensordot(xp[sl], dy, axes=([0, 1, 2], [0, 1, 2]))
    dx = np.zeros(x.shape, dtype=dy.dtype)
    dx[:, :hh, :ww] = dxp[:, pb:pb + hh, pb:pb + ww]
    return dx, dW


def deconv_out_size(in_size, k, s, padding):
    """deconv.py:133-161 get2d_deconv_output_size: VALID (in-1)*s+k, SAME in*s."""
    return (in_size - 1) * s + k if padding == "VALID" else in_size * s


def deconv2d(x, W, s, padding):
    """tf.nn.conv2d_transpose(x, W, output_shape, (1,s,s,1), padding) (deconv.py:107); W [k,k,Co,Ci].
    [TF] It is the adjoint of the conv2d that maps the OUTPUT image to x: out[i*s + kh - pb] += x[i] W[kh]
    with pb that conv's pad-before (0 for VALID, 1 for k=5,s=2 SAME): the full transposed conv cropped pb
    before -- not what torch's conv_transpose2d(padding=2, output_padding=1) does."""
    B, H, Wd, Ci = x.shape
    k, Co = W.shape[0], W.shape[2]
    OH, OW = deconv_out_size(H, k, s, padding), deconv_out_size(Wd, k, s, padding)
    pb = _same_pad_before(OH, k, s)[1] if padding == "SAME" else 0
    full = np.zeros((B, (H - 1) * s + k, (Wd - 1) * s + k, Co), dtype=np.result_type(x, W))
    for kh in range(k):
        for kw in range(k):
            full[:, kh:kh + (H - 1) * s + 1:s, kw:kw + (Wd - 1) * s + 1:s] += x @ W[kh, kw].T
    y = np.zeros((B, OH, OW, Co), dtype=full.dtype)
    hh, ww = min(OH, full.shape[1] - pb), min(OW, full.shape[2] - pb)
    y[:, :hh, :ww] = full[:, pb:pb + hh, pb:pb + ww]
    return y


def deconv2d_bwd(x, W, s, padding, dy):
    """Gradients of deconv2d w.r.t. its input (a plain conv of dy) and its filter."""
    B, H, Wd, Ci = x.shape
    k = W.shape[0]
    OH, OW = dy.shape[1], dy.shape[2]
    pb = _same_pad_before(OH, k, s)[1] if padding == "SAME" else 0
    full = np.zeros((B, (H - 1) * s + k, (Wd - 1) * s + k, dy.shape[3]), dtype=dy.dtype)
    hh, ww = min(OH, full.shape[1] - pb), min(OW, full.shape[2] - pb)
    full[:, pb:pb + hh, pb:pb + ww] = dy[:, :hh, :ww]
    dx = np.zeros(x.shape, dtype=dy.dtype)
    dW = np.zeros(W.shape, dtype=dy.dtype)
    for kh in range(k):
        for kw in range(k):
            g = full[:, kh:kh + (H - 1) * s + 1:s, kw:kw + (Wd - 1) * s + 1:s]      # [B,H,W,Co]
            dx += g @ W[kh, kw]
            dW[kh, kw] = np.tensordot(g, x, axes=([0, 1, 2], [0, 1, 2]))
    return dx, dW


def conv_geometry(na):
    """Layer list of the conv/deconv branch for one modality (vae_assoc.py:169-199, 249-278)."""
    S = int(round(np.sqrt(na["n_input"])))
    assert S * S == int(na["n_input"]), "hidden_conv needs a square image"
    R1, R2 = int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])
    G1, G2 = int(na["n_hidden_gener_1"]), int(na["n_hidden_gener_2"])
    enc = [dict(k=5, s=2, pad="SAME", ci=1, co=R1), dict(k=5, s=2, pad="SAME", ci=R1, co=2 * R1),
           dict(k=5, s=1, pad="VALID", ci=2 * R1, co=R2)]
    size = S
    for L in enc:
        size = _same_pad_before(size, L["k"], L["s"])[0] if L["pad"] == "SAME" else (size - L["k"]) // L["s"] + 1
    flat = size * size * R2                        # vae_assoc.py:192,198-199 (py2 integer division: 28 -> 7 -> 3)
    dec = [dict(k=3, s=1, pad="VALID", ci=int(na["n_z"]), co=G1), dict(k=5, s=1, pad="VALID", ci=G1, co=G1 // 2),
           dict(k=5, s=2, pad="SAME", ci=G1 // 2, co=G2), dict(k=5, s=2, pad="SAME", ci=G2, co=1)]
    return S, enc, flat, dec


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
    if na.get("hidden_conv"):
        S, enc, flat, dec = conv_geometry(na)
        n_in, n_z = int(na["n_input"]), int(na["n_z"])
        shapes = [("enc_C%d" % (i + 1), (L["k"], L["k"], L["ci"], L["co"])) for i, L in enumerate(enc)]
        shapes += [("enc_Wmu", (flat, n_z)), ("enc_bmu", (n_z,)), ("enc_Wsig", (flat, n_z)), ("enc_bsig", (n_z,))]
        for i, L in enumerate(dec):
            shapes += [("dec_T%d_W" % (i + 1), (L["k"], L["k"], L["co"], L["ci"])), ("dec_T%d_b" % (i + 1), (L["co"],))]
        shapes += [("dec_Wout", (n_in, n_in)), ("dec_bout", (n_in,))]
        return shapes
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
            elif len(shp) == 4 and name.startswith("enc_C"):
                # weight_variable (vae_assoc.py:471-473): truncated_normal(stddev=0.1) [TF: redraw beyond 2 sigma]
                w = rng.standard_normal(shp)
                while np.any(np.abs(w) > 2):
                    bad = np.abs(w) > 2
                    w[bad] = rng.standard_normal(int(bad.sum()))
                p[name] = (0.1 * w).astype(dtype)
            elif len(shp) == 4:
                # deconv.py:83-84 xavier_init(out_depth*k*k, in_depth*k*k) [prettytensor: uniform +-sqrt(6/(a+b))]
                kk = shp[0] * shp[1]
                lim = np.sqrt(6.0 / (shp[2] * kk + shp[3] * kk))
                p[name] = rng.uniform(-lim, lim, size=shp).astype(dtype)
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
    if na.get("hidden_conv"):
        # vae_assoc.py:169-183,190-199: three bias-less convs with NO activation, flatten (h,w,c), two dense heads
        S, enc, flat, _dec = conv_geometry(na)
        h = q(x).reshape(-1, S, S, 1)
        acts = [h]
        for i, L in enumerate(enc):
            h = q(conv2d(h, q(p["enc_C%d" % (i + 1)]), L["s"], L["pad"]))
            acts.append(h)
        hf = h.reshape(h.shape[0], flat)
        mu = hf @ q(p["enc_Wmu"]) + q(p["enc_bmu"])
        lv = hf @ q(p["enc_Wsig"]) + q(p["enc_bsig"])
        return mu, lv, {"acts": acts, "conv": True}
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
    if na.get("hidden_conv"):
        # vae_assoc.py:249-278,286-291: four conv2d_transpose layers, each + bias + SIGMOID (deconv_2d's default
        # transfer_fct, :491, is never overridden), flatten, dense n_input x n_input + sigmoid
        assert binary, "the reference's non-binary conv decoder is shape-broken (vae_assoc.py:299)"
        S, _enc, _flat, dec = conv_geometry(na)
        g = q(z).reshape(-1, 1, 1, int(na["n_z"]))
        acts = [g]
        for i, L in enumerate(dec):
            g = q(_sigmoid(deconv2d(g, q(p["dec_T%d_W" % (i + 1)]), L["s"], L["pad"]) + q(p["dec_T%d_b" % (i + 1)])))
            acts.append(g)
        gf = g.reshape(g.shape[0], -1)
        logits = gf @ q(p["dec_Wout"]) + q(p["dec_bout"])
        return _sigmoid(logits), {"acts": acts, "logits": logits, "conv": True}
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
def backward(archs, params, fw, X, eps, binary, weights, assoc_lambda, act, batch_global=None, quant=None, masks=None):
    """Analytic gradient of ``cost`` w.r.t. every parameter (what TF autodiff of
    vae_assoc.py:373-374 produces).  Formulas: SURVEY.md 8(a) row A5.  Mean terms carry
    1/batch_global, sum terms (Gaussian recon, assoc) carry 1.

    ``masks`` (tests only, relu): ``masks[m]["enc"|"dec"][i]`` = boolean [B, width] "this hidden unit was active" decisions of
    ANOTHER run of the same forward pass (the HIP path's stored activations > 0).  A relu pre-activation within rounding of 0
    gets opposite decisions from two arithmetic types; with the decisions handed over, what is compared is the arithmetic."""
    _, dact_pre = ACT[act]
    q = _q(quant)
    if quant is None:
        dact = dact_pre
    else:
        dact = lambda a, y: DACT_FROM_OUTPUT[act](y)      # noqa: E731  (from the stored, rounded output)
    if masks is not None and act != "relu":
        raise ValueError("masks are relu decisions")
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
        if na.get("hidden_conv"):
            grads.append(_backward_conv(na, p, f, x, eps, w, Bg, dmu[m], dlv[m], q))
            continue
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
            d_i = dact(dpre[i], dacts[i + 1]) if masks is None else np.asarray(masks[m]["dec"][i], dtype=dg.dtype)
            da = q(dg * d_i)
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
            d_i = dact(epre[i], eacts[i + 1]) if masks is None else np.asarray(masks[m]["enc"][i], dtype=dh.dtype)
            da = q(dh * d_i)
            g["enc_W%d" % (i + 1)] = eacts[i].T @ da
            g["enc_b%d" % (i + 1)] = da.sum(0)
            if i > 0:
                dh = da @ q(p["enc_W%d" % (i + 1)]).T
        grads.append(g)
    return grads, {"dmu_direct": dmu, "dlv_direct": dlv}


def _backward_conv(na, p, f, x, eps, w, Bg, dmu_direct, dlv_direct, q):
    """Backward pass of the conv/deconv branch (binary modality)."""
    S, enc, flat, dec = conv_geometry(na)
    g = {}
    xr = f["xhat"]
    dl = q((w / Bg) * xr * (1 - xr) * (-x / (1e-3 + xr) + (1 - x) / (1e-3 + 1 - xr)))
    dacts = f["dec"]["acts"]
    gf = dacts[-1].reshape(dacts[-1].shape[0], -1)
    g["dec_Wout"] = gf.T @ dl
    g["dec_bout"] = dl.sum(0)
    dy = (dl @ q(p["dec_Wout"]).T).reshape(dacts[-1].shape)
    for i in range(len(dec) - 1, -1, -1):
        y = dacts[i + 1]
        da = q(dy * y * (1.0 - y))                                   # sigmoid' from the stored output
        dy, dW = deconv2d_bwd(dacts[i], q(p["dec_T%d_W" % (i + 1)]), dec[i]["s"], dec[i]["pad"], da)
        g["dec_T%d_W" % (i + 1)] = dW
        g["dec_T%d_b" % (i + 1)] = da.sum((0, 1, 2))
    dz = dy.reshape(dy.shape[0], -1)
    gmu = q(dmu_direct + dz)
    glv = q(dlv_direct + dz * 0.5 * np.sqrt(np.exp(f["lv"])) * eps)
    eacts = f["enc"]["acts"]
    hf = eacts[-1].reshape(eacts[-1].shape[0], flat)
    g["enc_Wmu"] = hf.T @ gmu
    g["enc_bmu"] = gmu.sum(0)
    g["enc_Wsig"] = hf.T @ glv
    g["enc_bsig"] = glv.sum(0)
    dh = q((gmu @ q(p["enc_Wmu"]).T + glv @ q(p["enc_Wsig"]).T).reshape(eacts[-1].shape))   # convs carry no activation
    for i in range(len(enc) - 1, -1, -1):
        dx, dW = conv2d_bwd(eacts[i], q(p["enc_C%d" % (i + 1)]), enc[i]["s"], enc[i]["pad"], dh)
        g["enc_C%d" % (i + 1)] = dW
        dh = q(dx)
    return g


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

    def cost_and_grads(self, X, eps, batch_global=None, masks=None):
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
                            self.weights, self.assoc_lambda, self.act, batch_global, self.quant, masks)
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
