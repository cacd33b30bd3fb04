"""vae_assoc_amd: MI355X-native associative-VAE training path (drop-in for the hot path of
navigator8972/vae_assoc).  Importing the package does not load the HIP library; constructing a
model does, and fails loudly if it is missing."""
import importlib

from . import dataset  # noqa: F401

_LAZY = {"AssocVariationalAutoEncoder": "vae_assoc", "train": "vae_assoc", "xavier_init": "vae_assoc",
         "GradSync": "parallel", "dp_train_step": "parallel"}


def __getattr__(name):
    # lazy: `import vae_assoc_amd` must work on a CPU-only box (tests of host logic), while the
    # model itself needs torch + libavae.so + a gfx950 device
    if name in ("vae_assoc", "parallel", "_capi"):
        return importlib.import_module("." + name, __name__)
    if name in _LAZY:
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
