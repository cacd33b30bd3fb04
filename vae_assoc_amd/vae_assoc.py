"""Drop-in host side of the associative VAE: the reference's Python surface over libavae.

Mirrors /root/reference/vae_assoc.py: class ``AssocVariationalAutoEncoder`` (:20-463) with
``partial_fit / evaluate_cost / transform / generate / reconstruct / save_model / restore_model``
and the module function ``train`` (:498-583), same argument names, defaults and error behaviour.
Everything numerical happens in hand-written gfx950 kernels behind the C ABI of
include/avae.h; PyTorch-ROCm tensors only hold device memory.  There is no CPU path.

Deliberate, documented differences from the reference:
  * ``transfer_fct`` is a name ('relu', 'softplus', ...) or any callable whose ``__name__`` is
    one (``tf.nn.relu`` would qualify); the reference passes TF callables (:26,:502).
  * weights are drawn with NumPy (TF's RNG stream is not reproducible); same distribution as
    ``xavier_init`` (:11-18), biases zero.
  * ``partial_fit`` / ``evaluate_cost`` / ``reconstruct`` accept an optional explicit ``eps``
    (the reference draws it inside the graph, :90); ``None`` uses the in-kernel Philox stream.
  * network dicts may carry an extra key ``n_hidden`` (list) for more than two hidden layers.
  * ``hidden_conv=True`` (conv encoder :169-210 / deconv decoder :249-291, deconv.py) is built for what
    the reference's branch can express: a binary 28x28 modality (n_input = 784).
"""
import ctypes as C
import datetime
import os
import time

import numpy as np
import torch

from . import _capi
from .parallel import GradSync, dp_train_step_bucketed

_ARCH_KEYS = ("scope", "hidden_conv", "n_hidden_recog_1", "n_hidden_recog_2",
              "n_hidden_gener_1", "n_hidden_gener_2", "n_input", "n_z")


def xavier_init(fan_in, fan_out, constant=1, rng=None):
    """Xavier initialisation of network weights (reference vae_assoc.py:11-18), as float32 NumPy."""
    low = -constant * np.sqrt(6.0 / (fan_in + fan_out))
    high = constant * np.sqrt(6.0 / (fan_in + fan_out))
    rng = np.random if rng is None else rng
    return rng.uniform(low, high, size=(fan_in, fan_out)).astype(np.float32)


def _act_name(transfer_fct):
    if transfer_fct is None:
        return "identity"
    name = transfer_fct if isinstance(transfer_fct, str) else getattr(transfer_fct, "__name__", str(transfer_fct))
    name = name.lower()
    if name not in _capi.ACT_IDS:
        raise ValueError("unsupported transfer_fct %r (supported: %s)" % (transfer_fct, sorted(_capi.ACT_IDS)))
    return name


def hidden_sizes(na):
    """Encoder widths of one modality.  The MLP decoder reuses them: the reference sizes its
    generator from n_hidden_recog_* and ignores n_hidden_gener_* (vae_assoc.py:257,280,293,299)."""
    if na.get("n_hidden") is not None:
        return [int(h) for h in na["n_hidden"]]
    return [int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])]


def layer_shapes(na):
    """Flat-parameter layout of one modality in the reference's variable-creation order
    (vae_assoc.py:185-215,257-300; conv branch :174-210,:251-300): [(name, shape), ...]."""
    if na.get("hidden_conv"):
        r1, r2 = int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])
        g1, g2 = int(na["n_hidden_gener_1"]), int(na["n_hidden_gener_2"])
        n_in, n_z = int(na["n_input"]), int(na["n_z"])
        shapes = [("enc_C1", (5, 5, 1, r1)), ("enc_C2", (5, 5, r1, 2 * r1)), ("enc_C3", (5, 5, 2 * r1, r2)),
                  ("enc_Wmu", (9 * r2, n_z)), ("enc_bmu", (n_z,)), ("enc_Wsig", (9 * r2, n_z)), ("enc_bsig", (n_z,))]
        for i, (k, co, ci) in enumerate(((3, g1, n_z), (5, g1 // 2, g1), (5, g2, g1 // 2), (5, 1, g2))):
            shapes += [("dec_T%d_W" % (i + 1), (k, k, co, ci)), ("dec_T%d_b" % (i + 1), (co,))]
        return shapes + [("dec_Wout", (n_in, n_in)), ("dec_bout", (n_in,))]
    hs = hidden_sizes(na)
    n_in, n_z = int(na["n_input"]), int(na["n_z"])
    shapes, prev = [], n_in
    for i, h in enumerate(hs):
        shapes += [("enc_W%d" % (i + 1), (prev, h)), ("enc_b%d" % (i + 1), (h,))]
        prev = h
    shapes += [("enc_Wmu", (prev, n_z)), ("enc_bmu", (n_z,)), ("enc_Wsig", (prev, n_z)), ("enc_bsig", (n_z,))]
    prev = n_z
    for i, h in enumerate(hs):
        shapes += [("dec_W%d" % (i + 1), (prev, h)), ("dec_b%d" % (i + 1), (h,))]
        prev = h
    shapes += [("dec_Wout", (prev, n_in)), ("dec_bout", (n_in,))]
    return shapes


class AssocVariationalAutoEncoder(object):
    """Associative VAE over M sensory modalities, trained on one MI355X (or one per rank).

    Same constructor signature as the reference (vae_assoc.py:26-27); keyword-only extras:
      compute_dtype  'bf16' (default: bf16 MFMA operands, fp32 accumulate/loss/Adam) or 'fp32'
      device         torch device / ordinal (default: current CUDA(HIP) device)
      seed           seeds the NumPy weight draw and the in-kernel eps generator
      use_graph      replay the step as a captured hipGraph
      data_parallel  True -> one replica per torch.distributed rank, sample-sharded batch, the gradient SUM-all-reduced per
                     step in two buckets (decoder side first, overlapping the encoder's backward pass), Adam per bucket
      comm           who runs that collective:
                       'ipc'     libavae's own one-shot all-reduce over hipIpc peers (push reduce-scatter + push all-gather, every
                                 xGMI link at once; torch.distributed only hands the exchange-block handles round);
                       'library' libavae's RCCL communicator (ncclAllReduce on the library's comm stream; torch.distributed only
                                 hands the ncclUniqueId round);
                       'torch'   torch.distributed.all_reduce over the same buckets, the host stepping the pipeline.
                     None = 'torch' for world > 1 (the path every multi-rank parity test runs; pass 'ipc' / 'library' to opt in --
                     bench.py does), 'torch' for one rank.  comm='library' / 'ipc' without data_parallel builds a one-rank
                     communicator (tests)
      comm_buckets   2 (default): decoder-side bucket first, its all-reduce beside the encoder's backward pass; 1: ONE all-reduce of
                     the whole gradient buffer after the backward pass (north_star's literal design)
      wire_dtype     'fp32' (default) or 'bf16': element type of the gradient on the wire (the cost always travels as fp32)
    """

    def __init__(self, network_architectures, binary=True, transfer_fct="softplus", weights=1.0,
                 assoc_lambda=1.0, learning_rate=0.001, batch_size=100, *, compute_dtype="bf16",
                 device=None, seed=0, use_graph=True, data_parallel=False, process_group=None, comm=None,
                 comm_buckets=2, wire_dtype="fp32"):
        self.network_architectures = network_architectures
        self.assoc_lambda = assoc_lambda
        n_mod = len(network_architectures)
        # check if binary data (vae_assoc.py:31-35)
        if type(binary) is list:
            assert len(binary) == n_mod
            self.binary = binary
        else:
            self.binary = [binary] * n_mod
        if type(weights) is list:              # :37-41
            assert len(weights) == n_mod
            self.weights = weights
        else:
            self.weights = [weights] * n_mod
        self.transfer_fct = transfer_fct
        self._act = _act_name(transfer_fct)
        self.learning_rate = learning_rate
        self.batch_size = int(batch_size)
        self.n_z = int(network_architectures[0]["n_z"])       # :89
        if n_mod > _capi.AVAE_MAX_MODALITIES:
            raise ValueError("at most %d modalities" % _capi.AVAE_MAX_MODALITIES)
        for na in network_architectures:
            if int(na["n_z"]) != self.n_z:
                raise ValueError("all modalities must share n_z (the reference builds one eps of modality 0's n_z, :89-91)")
            if na.get("hidden_conv") and int(na["n_input"]) != 784:
                raise ValueError("hidden_conv=True needs n_input = 784: the reference's branch is hard-wired to 28x28 images")
        if compute_dtype not in _capi.DTYPE_IDS:
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self.compute_dtype = compute_dtype

        if not torch.cuda.is_available():
            raise RuntimeError("vae_assoc_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self._sync = GradSync(process_group) if data_parallel else None
        world = self._sync.world_size if self._sync else 1
        rank = self._sync.rank if self._sync else 0
        if comm not in (None, "library", "torch", "ipc"):
            raise ValueError("comm must be None, 'library', 'ipc' or 'torch'")
        if comm is None:
            comm = "torch"
        if comm_buckets not in (1, 2):
            raise ValueError("comm_buckets must be 1 or 2")
        if wire_dtype not in _capi.DTYPE_IDS:
            raise ValueError("wire_dtype must be 'fp32' or 'bf16'")
        self._comm = comm
        self._comm_lib = comm in ("library", "ipc")
        self._wire_bf16 = _capi.DTYPE_IDS[wire_dtype] == 1

        cfg = _capi.Config()
        cfg.abi_version = _capi.AVAE_ABI_VERSION
        cfg.n_modalities = n_mod
        for m, na in enumerate(network_architectures):
            hs = hidden_sizes(na)
            if len(hs) > _capi.AVAE_MAX_HIDDEN:
                raise ValueError("at most %d hidden layers" % _capi.AVAE_MAX_HIDDEN)
            cfg.mod[m].n_input = int(na["n_input"])
            cfg.mod[m].n_hidden_layers = len(hs)
            for k, hsz in enumerate(hs):
                cfg.mod[m].n_hidden[k] = hsz
            cfg.mod[m].binary = 1 if self.binary[m] else 0
            cfg.mod[m].weight = float(self.weights[m])
            cfg.mod[m].hidden_conv = 1 if na.get("hidden_conv") else 0
            if na.get("hidden_conv"):
                if not self.binary[m]:
                    raise ValueError("hidden_conv=True needs a binary modality (the reference's non-binary conv decoder "
                                     "is shape-broken, vae_assoc.py:299)")
                cfg.mod[m].n_hidden_layers = 2
                cfg.mod[m].n_hidden[0], cfg.mod[m].n_hidden[1] = int(na["n_hidden_recog_1"]), int(na["n_hidden_recog_2"])
                cfg.mod[m].conv_gener[0], cfg.mod[m].conv_gener[1] = int(na["n_hidden_gener_1"]), int(na["n_hidden_gener_2"])
        cfg.n_z = self.n_z
        cfg.batch_size = self.batch_size
        cfg.batch_global = self.batch_size * world
        cfg.row_offset = rank * self.batch_size
        cfg.activation = _capi.ACT_IDS[self._act]
        cfg.compute_dtype = _capi.DTYPE_IDS[compute_dtype]
        cfg.device = self.device.index
        cfg.use_graph = 1 if use_graph else 0
        cfg.assoc_lambda = float(assoc_lambda)
        cfg.learning_rate = float(learning_rate)
        cfg.beta1 = cfg.beta2 = cfg.adam_eps = 0.0          # -> TF-1 AdamOptimizer defaults
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

        cfg.comm_buckets = comm_buckets
        cfg.wire_dtype = _capi.DTYPE_IDS[wire_dtype]
        L = _capi.lib()
        agree_dev = self.device if (self._sync is not None and self._sync.backend == "nccl") else "cpu"
        self._agree_dev = agree_dev
        if self._comm == "library" and world > 1:
            # ncclCommInitRank below is collective: a rank that cannot even load RCCL would leave the others waiting inside it.  Every
            # rank therefore probes the loader first (drawing an id is the cheapest call that needs it) and the ranks agree: all, or
            # the torch.distributed collective on the same buckets for everybody.
            probe = (C.c_uint8 * 128)()
            ok = 1.0 if L.avae_comm_unique_id(probe) == 0 else 0.0
            if self._sync.sum_scalar(ok, agree_dev) < world:
                if rank == 0:
                    print("[vae_assoc_amd] RCCL cannot be loaded on every rank: gradient all-reduce through torch.distributed")
                self._comm, self._comm_lib = "torch", False
        if self._comm == "library":
            # bootstrap only: rank 0 draws the ncclUniqueId, torch.distributed hands it round; the communicator itself is the library's
            idb = (C.c_uint8 * 128)()
            if rank == 0:
                _capi.check(None, L.avae_comm_unique_id(idb), "avae_comm_unique_id")
            raw = self._sync.broadcast_bytes(bytes(idb), 128, device=agree_dev) if self._sync is not None else bytes(idb)
            cfg.use_comm, cfg.world_size, cfg.rank = _capi.COMM_RCCL, world, rank
            for i in range(128):
                cfg.nccl_id[i] = raw[i]
        elif self._comm == "ipc":
            cfg.use_comm, cfg.world_size, cfg.rank = _capi.COMM_IPC, world, rank
        # data-parallel buckets (host-only query): [[(offset, count), ...] per bucket]
        nb, nr = C.c_int32(0), (C.c_int32 * 2)()
        offs, cnts = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        _capi.check(None, L.avae_dp_plan(C.byref(cfg), C.byref(nb), nr, offs, cnts), "avae_dp_plan")
        self._buckets = [[(int(offs[b]), int(cnts[b]))] for b in range(nb.value)]      # ONE contiguous range per bucket
        nbytes = C.c_size_t(0)
        _capi.check(None, L.avae_workspace_bytes(C.byref(cfg), C.byref(nbytes)), "avae_workspace_bytes")
        # PyTorch is the device allocator: one uint8 tensor holds the whole replica state
        self._ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
        base = self._ws.data_ptr()
        self._ws_off = (-base) % 256
        cfg.workspace = base + self._ws_off
        cfg.workspace_bytes = nbytes.value
        self._cfg = cfg
        h = C.c_void_p()
        torch.cuda.synchronize(self.device)
        rc = L.avae_create(C.byref(cfg), C.byref(h))
        if self._comm_lib and world > 1:
            # Bring-up is agreed between the ranks at every collective step: a communicator / exchange that came up on some ranks
            # only is of no use to any.  RCCL: ncclCommInitRank ran inside avae_create.  IPC: every rank that created its replica
            # exports its exchange block, the handles go round (all_gather), every rank maps its peers' blocks.
            ok = 1.0 if rc == 0 else 0.0
            all_ok = self._sync.sum_scalar(ok, agree_dev) >= world
            if all_ok and self._comm == "ipc":
                mine = (C.c_uint8 * _capi.AVAE_IPC_HANDLE_BYTES)()
                ok = 1.0 if L.avae_comm_ipc_handle(h, mine) == 0 else 0.0
                blob = self._sync.all_gather_bytes(bytes(mine), _capi.AVAE_IPC_HANDLE_BYTES, device=agree_dev)
                if ok:
                    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
                    ok = 1.0 if L.avae_comm_ipc_attach(h, buf) == 0 else 0.0
                    if not ok:
                        print("[vae_assoc_amd] rank %d: %s" % (rank, L.avae_last_error(h).decode("utf-8", "replace")))
                all_ok = self._sync.sum_scalar(ok, agree_dev) >= world
            if not all_ok:
                if rc == 0:
                    L.avae_destroy(h)
                if rank == 0:
                    print("[vae_assoc_amd] the library's %s collective did not come up on every rank: gradient all-reduce "
                          "through torch.distributed" % ("RCCL" if self._comm == "library" else "hipIpc"))
                self._comm, self._comm_lib = "torch", False
                cfg.use_comm = _capi.COMM_NONE
                h = C.c_void_p()
                rc = L.avae_create(C.byref(cfg), C.byref(h))
        _capi.check(None, rc, "avae_create")
        self._h = h
        self._L = L
        n = C.c_size_t(0)
        _capi.check(h, L.avae_param_count(h, C.byref(n)), "avae_param_count")
        self.n_params = n.value
        gp, gn = C.c_void_p(), C.c_size_t(0)
        _capi.check(h, L.avae_grad_buffer(h, C.byref(gp), C.byref(gn)), "avae_grad_buffer")
        goff = gp.value - base
        self._grad_view = self._ws[goff:goff + 4 * gn.value].view(torch.float32)

        # initial weights: xavier-uniform, zero biases (vae_assoc.py:185-215,257-300)
        rng = np.random.RandomState(int(seed) & 0x7FFFFFFF)
        flat = []
        for na in network_architectures:
            for name, shp in layer_shapes(na):
                if len(shp) == 2:
                    flat.append(xavier_init(shp[0], shp[1], rng=rng).reshape(-1))
                elif len(shp) == 4 and name.startswith("enc_C"):
                    # weight_variable (vae_assoc.py:471-473): truncated_normal(stddev=0.1)
                    w = rng.standard_normal(shp)
                    while np.any(np.abs(w) > 2):
                        bad = np.abs(w) > 2
                        w[bad] = rng.standard_normal(int(bad.sum()))
                    flat.append((0.1 * w).astype(np.float32).reshape(-1))
                elif len(shp) == 4:
                    # deconv.py:83-84: xavier over (out_depth*k*k, in_depth*k*k)
                    kk = shp[0] * shp[1]
                    lim = np.sqrt(6.0 / (shp[2] * kk + shp[3] * kk))
                    flat.append(rng.uniform(-lim, lim, size=shp).astype(np.float32).reshape(-1))
                else:
                    flat.append(np.zeros(shp, dtype=np.float32))
        self.set_params(np.concatenate(flat))

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.avae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, cols):
        """-> (float32 device tensor [rows, cols] with unit column stride, was_numpy)."""
        was_np = not torch.is_tensor(a)
        t = torch.as_tensor(np.asarray(a, dtype=np.float32) if was_np else a)
        if t.dim() != 2 or t.shape[1] != cols:
            raise ValueError("expected a [rows, %d] array, got %s" % (cols, tuple(t.shape)))
        t = t.to(device=self.device, dtype=torch.float32)
        if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < cols):
            t = t.contiguous()
        return t, was_np

    def _batch_args(self, X, eps, n_steps=1):
        assert len(X) == len(self.network_architectures)
        ts = []
        rows = self.batch_size * n_steps
        for x, na in zip(X, self.network_architectures):
            t, _ = self._dev(x, int(na["n_input"]))
            if t.shape[0] != rows:
                # the reference's eps has static shape (batch_size, n_z): every path through z
                # needs exactly batch_size rows (vae_assoc.py:90)
                raise ValueError("expected %d rows (batch_size%s), got %d" % (rows, " x n_steps" if n_steps > 1 else "", t.shape[0]))
            ts.append(t)
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        lds = (C.c_int32 * len(ts))(*[t.stride(0) if t.shape[0] > 1 else t.shape[1] for t in ts])
        e = None
        if eps is not None:
            e, _ = self._dev(eps, self.n_z)
            if e.shape[0] != rows:
                raise ValueError("eps must be [batch_size%s, n_z]" % (" x n_steps" if n_steps > 1 else ""))
            e = e.contiguous()
        return ts, ptrs, lds, e

    def get_params(self):
        out = np.empty(self.n_params, dtype=np.float32)
        _capi.check(self._h, self._L.avae_get_params(self._h, out.ctypes.data_as(C.c_void_p)), "avae_get_params")
        return out

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32).reshape(-1)
        if flat.size != self.n_params:
            raise ValueError("expected %d parameters, got %d" % (self.n_params, flat.size))
        _capi.check(self._h, self._L.avae_set_params(self._h, flat.ctypes.data_as(C.c_void_p)), "avae_set_params")

    def get_grads(self):
        out = np.empty(self.n_params, dtype=np.float32)
        _capi.check(self._h, self._L.avae_get_grads(self._h, out.ctypes.data_as(C.c_void_p)), "avae_get_grads")
        return out

    def get_opt_state(self):
        m = np.empty(self.n_params, dtype=np.float32)
        v = np.empty(self.n_params, dtype=np.float32)
        step = C.c_int64(0)
        _capi.check(self._h, self._L.avae_get_opt_state(self._h, m.ctypes.data_as(C.c_void_p),
                                                        v.ctypes.data_as(C.c_void_p), C.byref(step)), "avae_get_opt_state")
        return m, v, step.value

    def cost_history(self, n):
        out = np.empty(n, dtype=np.float32)
        last = C.c_int64(0)
        _capi.check(self._h, self._L.avae_cost_history(self._h, n, out.ctypes.data_as(C.c_void_p), C.byref(last)),
                    "avae_cost_history")
        return out

    def synchronize(self):
        _capi.check(self._h, self._L.avae_synchronize(self._h), "avae_synchronize")

    # ------------------------------------------------------------------ data-parallel seam (parallel.py protocol)
    def _backward(self, X, eps=None):
        """local forward + backward + every weight gradient: the gradient buffer (and the local cost in its last float) is complete"""
        self._stage(X, eps)
        for b in range(len(self._buckets)):
            self._backward_bucket(b)

    def _apply(self, want_cost=True):
        cost = None
        for b in range(len(self._buckets)):
            cost = self._apply_bucket(b, want_cost and b == len(self._buckets) - 1)
        return cost

    def _grad_tensor(self):
        return self._grad_view

    # bucketed seam (parallel.dp_train_step_bucketed): stage -> per bucket backward / all-reduce -> per bucket Adam
    def _stage(self, X, eps=None, n_steps=1):
        ts, ptrs, lds, e = self._batch_args(X, eps, n_steps)
        _capi.check(self._h, self._L.avae_stage_batches(self._h, n_steps, ptrs, lds, e.data_ptr() if e is not None else None,
                                                        self._stream()), "avae_stage_batches")
        self._staged_j = 0

    def _backward_bucket(self, b):
        _capi.check(self._h, self._L.avae_dp_backward(self._h, self._staged_j, b, self._stream()), "avae_dp_backward")

    def _apply_bucket(self, b, want_cost=True):
        cost = C.c_float(0.0)
        _capi.check(self._h, self._L.avae_dp_apply(self._h, b, C.byref(cost) if want_cost else None, self._stream()), "avae_dp_apply")
        return cost.value if want_cost else None

    # ------------------------------------------------------------------ reference surface
    def partial_fit(self, X, eps=None, return_cost=True):
        """Train model based on mini-batch of input data.  Return cost of mini-batch.
        (reference vae_assoc.py:378-386).  ``return_cost=False`` skips the host synchronise;
        the cost stays retrievable through ``cost_history``."""
        if self._sync is not None and self._sync.world_size > 1 and not self._comm_lib:
            # host-owned collective (torch.distributed) over the library's buckets
            cost = dp_train_step_bucketed(self, self._sync, self._buckets, X, eps)
            return cost if return_cost else None
        ts, ptrs, lds, e = self._batch_args(X, eps)      # (library-owned collective: avae_train_step runs the bucketed pipeline itself)
        cost = C.c_float(0.0)
        _capi.check(self._h, self._L.avae_train_step(self._h, ptrs, lds, e.data_ptr() if e is not None else None,
                                                     C.byref(cost) if return_cost else None, self._stream()),
                    "avae_train_step")
        return cost.value if return_cost else None

    def partial_fit_steps(self, X, n_steps, eps=None, return_cost=True):
        """``n_steps`` successive ``partial_fit`` calls in one submission: step i trains on rows
        [i*batch_size, (i+1)*batch_size) of every X[m] (and of ``eps``) -- what the reference's inner
        loop does with ``DataSet.next_batch``'s consecutive slices (vae_assoc.py:541-550).  Returns the
        last step's cost; every step's cost is in ``cost_history``."""
        n_steps = int(n_steps)
        if self._sync is not None and self._sync.world_size > 1 and not self._comm_lib:
            # host-owned collective: the batches are staged 16 at a time, every step runs the bucketed schedule
            ts, ptrs, lds, e = self._batch_args(X, eps, n_steps)
            B, cost, st = self.batch_size, None, self._stream()
            for i0 in range(0, n_steps, 16):
                n = min(16, n_steps - i0)
                p_i = (C.c_void_p * len(ts))(*[t.data_ptr() + i0 * B * t.stride(0) * 4 for t in ts])
                e_i = (e.data_ptr() + i0 * B * self.n_z * 4) if e is not None else None
                _capi.check(self._h, self._L.avae_stage_batches(self._h, n, p_i, lds, e_i, st), "avae_stage_batches")
                for j in range(n):
                    self._staged_j = j
                    pending = []
                    for b, ranges in enumerate(self._buckets):
                        self._backward_bucket(b)
                        pending.append(self._sync.all_reduce_ranges_(self._grad_view, ranges, async_op=True))
                    for b in range(len(self._buckets)):
                        for w in pending[b]:
                            w.wait()
                        cost = self._apply_bucket(b, return_cost and i0 + j == n_steps - 1 and b == len(self._buckets) - 1)
            return cost
        ts, ptrs, lds, e = self._batch_args(X, eps, n_steps)
        cost = C.c_float(0.0)
        _capi.check(self._h, self._L.avae_train_steps(self._h, n_steps, ptrs, lds, e.data_ptr() if e is not None else None,
                                                      C.byref(cost) if return_cost else None, self._stream()),
                    "avae_train_steps")
        return cost.value if return_cost else None

    def evaluate_cost(self, X, eps=None):
        """reference vae_assoc.py:388-391 (forward + loss with a fresh eps, no update)."""
        ts, ptrs, lds, e = self._batch_args(X, eps)
        cost = C.c_float(0.0)
        _capi.check(self._h, self._L.avae_eval_cost(self._h, ptrs, lds, e.data_ptr() if e is not None else None,
                                                    C.byref(cost), self._stream()), "avae_eval_cost")
        c = cost.value
        if self._sync is not None and self._sync.world_size > 1:
            c = self._sync.sum_scalar(c, self.device)
        return c

    def _encode(self, m, x, want_logvar=False):
        t, was_np = self._dev(x, int(self.network_architectures[m]["n_input"]))
        rows = t.shape[0]
        mu = torch.empty((rows, self.n_z), dtype=torch.float32, device=self.device)
        lv = torch.empty_like(mu) if want_logvar else None
        if rows:
            _capi.check(self._h, self._L.avae_encode(self._h, m, t.data_ptr(), t.stride(0) if rows > 1 else t.shape[1], rows,
                                                     mu.data_ptr(), lv.data_ptr() if want_logvar else None,
                                                     self._stream()), "avae_encode")
        conv = (lambda a: a.cpu().numpy()) if was_np else (lambda a: a)
        return (conv(mu), conv(lv)) if want_logvar else conv(mu)

    def transform(self, X, sens_idx=None):
        """Transform data by mapping it into the latent space (posterior means only).
        ``sens_idx`` is None (X = list over modalities) or an integer (X = one array)
        (reference vae_assoc.py:393-403)."""
        if sens_idx is None:
            return [self._encode(m, x) for m, x in enumerate(X)]
        assert sens_idx < len(self.network_architectures)
        return self._encode(sens_idx, X)

    def generate(self, z_mu=None):
        """Generate data by sampling from latent space: decoder only, z fed directly; returns the
        list of per-modality decoder means.  ``None`` draws z from the prior with NumPy's global
        RNG, batch_size rows (reference vae_assoc.py:405-419)."""
        if z_mu is None:
            z_mu = np.random.normal(size=(self.batch_size, self.n_z))
        z, was_np = self._dev(z_mu, self.n_z)
        z = z.contiguous()
        rows = z.shape[0]
        outs = [torch.empty((rows, int(na["n_input"])), dtype=torch.float32, device=self.device) for na in self.network_architectures]
        if rows:       # every modality's decoder in one submission (avae_generate: one graph replay for 1-64 rows)
            ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            _capi.check(self._h, self._L.avae_generate(self._h, z.data_ptr(), rows, ptrs, self._stream()), "avae_generate")
        return [o.cpu().numpy() for o in outs] if was_np else outs

    def reconstruct(self, X, eps=None):
        """Use VAE to reconstruct given data: encode -> sample z -> decode, per modality with its
        own eps draw as each sess.run of the reference makes one (vae_assoc.py:421-425).
        ``eps`` may be a list with one [rows, n_z] array per modality."""
        outs = []
        for m, (x, na) in enumerate(zip(X, self.network_architectures)):
            t, was_np = self._dev(x, int(na["n_input"]))
            rows = t.shape[0]
            e = None
            if eps is not None:
                e, _ = self._dev(eps[m], self.n_z)
                e = e.contiguous()
                assert e.shape[0] == rows
            o = torch.empty((rows, int(na["n_input"])), dtype=torch.float32, device=self.device)
            if rows:
                _capi.check(self._h, self._L.avae_reconstruct(self._h, m, t.data_ptr(), t.stride(0) if rows > 1 else t.shape[1],
                                                              e.data_ptr() if e is not None else None, rows, o.data_ptr(),
                                                              self._stream()), "avae_reconstruct")
            outs.append(o.cpu().numpy() if was_np else o)
        return outs

    def save_model(self, fname=None):
        """reference vae_assoc.py:427-435 (default name: timestamp + batch size)."""
        if fname is None:
            ts = time.time()
            ckpt_fname = 'vae_assoc_' + datetime.datetime.fromtimestamp(ts).strftime('%Y_%m_%d_%H_%M_%S') \
                + '_batchsize_{}.ckpt'.format(self.batch_size)
        else:
            ckpt_fname = fname
        print('Saving model to {}...'.format(ckpt_fname))
        _capi.check(self._h, self._L.avae_save(self._h, os.fsencode(ckpt_fname)), "avae_save")
        return

    def restore_model(self, folder=None, fname=None):
        """reference vae_assoc.py:437-463: newest-listed *.ckpt of ``folder`` ('output' by default)
        unless ``fname`` is given; every failure prints and returns, nothing raises."""
        model_folder = 'output' if folder is None else folder
        if os.path.isdir(model_folder) and os.path.exists(model_folder):
            if fname is None:
                files = [f for f in os.listdir(model_folder) if f.endswith('.ckpt')]
                if not files:
                    print('No valid model file.')
                    return
                model_file = files[-1]
            else:
                model_file = fname
            path = os.path.join(model_folder, model_file)
            if os.path.exists(path):
                print('Loading {}...'.format(path))
                rc = self._L.avae_load(self._h, os.fsencode(path))
                if rc != 0:
                    print('Invalid or non-exist model file. ({})'.format(self._L.avae_last_error(self._h).decode()))
            else:
                print('Invalid or non-exist model file.')
        else:
            print('Invalid or non-exist model folder.')
        return


def train(data_sets, network_architectures, binary=True, weights=1.0, assoc_lambda=1e-5, learning_rate=0.001,
          batch_size=100, training_epochs=10, display_step=5, early_stop=False, **model_kwargs):
    """Epoch/minibatch loop of the reference (vae_assoc.py:498-583): relu transfer (:502), column
    split of the [N, sum n_input] matrix (:510,:543), optional validation early stop (:520-537),
    ``avg_cost_hist`` = running within-epoch sum appended per batch (:576-577).

    Runs of consecutive ``next_batch`` slices (everything between two reshuffles) are handed over as one
    matrix and trained in one submission (``partial_fit_steps``); the matrix is split into modalities on
    the device by pointer offset + row stride (no per-modality copies); per-step costs are read back once
    per epoch from the device-side history, so the hot loop never synchronises.

    ``data_parallel=True`` (one process per GPU): ``batch_size`` is the per-rank batch; every rank walks the SAME
    data set in the SAME order in global batches of ``world * batch_size`` rows and trains on its own rows of each
    (see ``train_loop``), so the run equals the single-process run with ``batch_size * world``."""
    vae_assoc = AssocVariationalAutoEncoder(network_architectures, binary, transfer_fct="relu", weights=weights,
                                            assoc_lambda=assoc_lambda, learning_rate=learning_rate,
                                            batch_size=batch_size, **model_kwargs)
    return train_loop(vae_assoc, data_sets, network_architectures, batch_size, training_epochs, display_step, early_stop)


def train_loop(vae_assoc, data_sets, network_architectures, batch_size, training_epochs=10, display_step=5,
               early_stop=False, sync=None, device=None):
    """The loop of ``train`` on an already built model (any object with ``partial_fit`` / ``evaluate_cost``, and optionally
    ``partial_fit_steps`` / ``cost_history``; the CPU tests drive it with an oracle-backed replica).

    Data parallelism (``sync`` = the model's ``GradSync``, world > 1): the reference's loop (vae_assoc.py:516-577) is kept, with
    the global batch ``B_g = world * batch_size`` in the place of ``batch_size``: ``next_batch(B_g)`` on every rank -- rank 0's
    NumPy seed is broadcast first, and a checksum of the training matrix is compared, so that all ranks shuffle alike --
    rank r trains on rows [r*batch_size, (r+1)*batch_size) of it, ``total_batch = n_samples // B_g`` and the (already
    all-reduced, global-batch) cost enters ``avg_cost`` with weight ``B_g / n_samples``."""
    if sync is None:
        sync = getattr(vae_assoc, "_sync", None)
    world = sync.world_size if sync is not None else 1
    rank = sync.rank if sync is not None else 0
    n_samples = data_sets.train._data.shape[0]
    sens_indices = np.concatenate([[0], np.cumsum([na["n_input"] for na in network_architectures])])
    n_mod = len(network_architectures)
    avg_cost_hist = []
    valid_cost = None
    dev = device if device is not None else getattr(vae_assoc, "device", None)
    hist_cap = 4096
    batch_global = batch_size * world
    lo, hi = rank * batch_size, (rank + 1) * batch_size
    if world > 1:
        boot_dev = getattr(vae_assoc, "_agree_dev", "cpu" if dev is None else dev)
        np.random.seed(sync.broadcast_int(int(np.random.randint(0, 2 ** 31 - 1)), device=boot_dev))
        # every rank must hold the SAME training matrix in the SAME order: rank 0's digest of ~256 sampled rows (content and
        # position both enter) goes round, every rank compares its own with it, and the verdicts are all-reduced so that either
        # every rank raises or none does (a rank that raised alone would leave the others waiting in the next collective)
        d = data_sets.train._data
        rows = d[:: max(1, n_samples // 256)]
        rows = rows.double().cpu().numpy() if torch.is_tensor(rows) else np.asarray(rows, dtype=np.float64)
        wts = np.cos(np.arange(rows.size, dtype=np.float64).reshape(rows.shape) * 0.7548776662466927)
        digest = np.array([n_samples, rows.shape[1], float((rows * wts).sum()), float(np.abs(rows).sum())], dtype=np.float64)
        ref = np.frombuffer(sync.broadcast_bytes(digest.tobytes(), digest.nbytes, device=boot_dev), dtype=np.float64)
        same = bool(np.all(np.abs(ref - digest) <= 1e-9 * np.maximum(1.0, np.abs(ref))))
        if not sync.all_agree(same, boot_dev):
            raise ValueError("data_parallel train(): the ranks hold different training matrices (or different orders of one); "
                             "build the data sets from the same array with the same NumPy seed on every rank")

    def seg(batch_xs):
        # host batches (reference DataSet) are uploaded once per step; a dataset.DeviceDataSet hands device rows
        if dev is None:
            t = np.asarray(batch_xs)
        else:
            t = batch_xs if torch.is_tensor(batch_xs) else torch.as_tensor(np.ascontiguousarray(batch_xs, dtype=np.float32))
            t = t.to(dev)
        return [t[:, sens_indices[k]:sens_indices[k + 1]] for k in range(n_mod)]

    def shard_run(batch_xs, n):
        """rows of this rank inside each of the n consecutive global batches, as one matrix of n*batch_size rows"""
        if world == 1:
            return batch_xs
        t = batch_xs.reshape(n, batch_global, batch_xs.shape[1])[:, lo:hi]
        return t.reshape(n * batch_size, batch_xs.shape[1])

    multi = hasattr(vae_assoc, "partial_fit_steps") and hasattr(vae_assoc, "cost_history")
    for epoch in range(training_epochs):
        avg_cost = 0.
        total_batch = int(n_samples / batch_global)
        if early_stop:
            if epoch % early_stop == 0:
                curr_valid_cost = 0
                n_valid_batches = int(data_sets.validation._data.shape[0] / batch_global)
                for i in range(n_valid_batches):
                    batch_xs, _ = data_sets.validation.next_batch(batch_global)
                    curr_valid_cost += vae_assoc.evaluate_cost(seg(shard_run(batch_xs, 1))) / n_valid_batches
                print("Validation cost=", "{:.9f}".format(curr_valid_cost))
                if valid_cost is not None:
                    if curr_valid_cost > valid_cost:
                        print('Validation error increases. Early stop at epoch {} to prevent overfitting...'.format(epoch + 1))
                        break
                valid_cost = curr_valid_cost
        done = 0
        while done < total_batch:
            chunk = min(hist_cap, total_batch - done)
            got = 0
            costs = []
            while got < chunk:
                if multi and hasattr(data_sets.train, "next_batches"):     # a run of consecutive slices = one submission
                    batch_xs, _, n = data_sets.train.next_batches(batch_global, chunk - got)
                    vae_assoc.partial_fit_steps(seg(shard_run(batch_xs, n)), n, return_cost=False)
                else:                                             # a reference-style DataSet object
                    batch_xs, _ = data_sets.train.next_batch(batch_global)
                    c = vae_assoc.partial_fit(seg(shard_run(batch_xs, 1)), **({"return_cost": False} if multi else {}))
                    costs.append(c)
                    n = 1
                got += n
            for cost in (vae_assoc.cost_history(chunk) if multi else costs):
                avg_cost += float(cost) / n_samples * batch_global
                avg_cost_hist.append(avg_cost)
            done += chunk
        if epoch % display_step == 0:
            print("Epoch:", '%04d' % (epoch + 1), "cost=", "{:.9f}".format(avg_cost))
    return vae_assoc, avg_cost_hist
