"""Alias: `dpdfnet.multi_gpu` IS `dpdfnet_amd.multi_gpu` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.multi_gpu as _m

_sys.modules[__name__] = _m
