"""Alias: `dpdfnet.onnx_backend` IS `dpdfnet_amd.ort_shim` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.ort_shim as _m

_sys.modules[__name__] = _m
