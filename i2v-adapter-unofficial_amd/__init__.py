"""MI355X-native I2V-Adapter denoising path.

Host-side mirror of the reference's module / pipeline API over hand-written HIP kernels for
gfx950 reached through a C ABI (include/i2v_hip.h, csrc/).  Import name: `i2v_adapter_unofficial_amd`
(the directory carries the repository's hyphenated name; `i2v_adapter_unofficial_amd.py` at the repo root
registers it under the importable name).
"""
from . import _lib, kernels, streams  # noqa: F401
from ._lib import HipLibraryError  # noqa: F401


def __getattr__(name):
    # module classes are imported lazily so that `import i2v_adapter_unofficial_amd.kernels` stays light
    import importlib
    if name in ("blocks", "handle", "checkpoint", "sharding", "vae", "training", "profiling", "i2v_adapter", "unet_motion_cross_frame_attn",
                "pipeline_i2v_adapter", "image_processor"):
        return importlib.import_module(f"{__name__}.{name}")
    for mod in ("i2v_adapter", "unet_motion_cross_frame_attn", "pipeline_i2v_adapter", "blocks", "sharding", "vae",
                "image_processor", "checkpoint", "handle"):
        m = importlib.import_module(f"{__name__}.{mod}")
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
