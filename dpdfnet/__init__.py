"""`import dpdfnet` -- the reference package's import name, served by the MI355X engine.

A user of the reference switches by putting this repository ahead of the reference package on `sys.path`: the public names
(reference package/src/dpdfnet/__init__.py:3-9: `enhance`, `enhance_file`, `available_models`, `download`, `StreamEnhancer`)
and the submodule names its callers and tests reach into (`dpdfnet.api`, `dpdfnet.audio`, `dpdfnet.models`, `dpdfnet.stream`,
`dpdfnet.onnx_backend` -- package/tests/test_package_behaviors.py:95-107 monkeypatches them) resolve to the SAME module
objects as `dpdfnet_amd.*` (a stub per submodule replaces itself in `sys.modules`), so patching one name patches both.
Nothing is implemented here: the engine is `dpdfnet_amd` + `dpdfnet_amd/libdpdfnet_hip.so`.  The reference's CLI, banner
and model download are out of scope and have no alias.
"""
import dpdfnet_amd as _impl

__all__ = list(_impl.__all__)


def __getattr__(name: str):
    try:
        return getattr(_impl, name)
    except AttributeError:
        raise AttributeError(f"module 'dpdfnet' has no attribute '{name}'") from None


def __dir__():
    return sorted(set(globals()) | set(__all__))
