"""vae_assoc_amd: MI355X-native associative-VAE training path (drop-in for the hot path of
navigator8972/vae_assoc).  Importing the package does not load the HIP library; constructing a
model does, and fails loudly if it is missing."""
from . import dataset  # noqa: F401


def __getattr__(name):
    # lazy: `import vae_assoc_amd` must work on a CPU-only box (tests of host logic), while the
    # model itself needs torch + libavae.so + a gfx950 device
    if name in ("AssocVariationalAutoEncoder", "train", "xavier_init", "vae_assoc"):
        from . import vae_assoc as _v
        return _v if name == "vae_assoc" else getattr(_v, name)
    if name in ("parallel", "GradSync", "dp_train_step"):
        from . import parallel as _p
        return _p if name == "parallel" else getattr(_p, name)
    raise AttributeError(name)
