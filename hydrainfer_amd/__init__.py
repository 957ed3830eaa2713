"""hydrainfer_amd — MI355X (gfx950) implementation of HydraInfer's attention + paged-KV
hot path behind the reference's own `hydrainfer._C.*` operator surface.

`install_as_hydrainfer()` registers the op modules under the reference's import paths
(`hydrainfer._C.kernel.flash_attn`, ...) so the unmodified reference engine picks them up
at its `try: from hydrainfer._C... import ...` call sites."""
import importlib
import sys
import types

__version__ = "0.1.0"

_OP_MODULES = [
    "_C.kernel.flash_attn",
    "_C.kernel.kv_cache_kernels",
    "_C.kernel.cache_kernels",
    "_C.kernel.norm",
    "_C.kernel.position_embedding",
    "_C.kernel.activation",
    "_C.kernel.moe",
    "_C.data_transfer.block_migration",
]


def install_as_hydrainfer() -> None:
    """Alias hydrainfer_amd._C.* as hydrainfer._C.* in sys.modules."""
    from hydrainfer_amd import _lib
    _lib.lib()  # fail now, loudly, if the shared library is missing
    for pkg in ("hydrainfer._C", "hydrainfer._C.kernel", "hydrainfer._C.data_transfer"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    for rel in _OP_MODULES:
        mod = importlib.import_module(f"hydrainfer_amd.{rel}")
        name = f"hydrainfer.{rel}"
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
