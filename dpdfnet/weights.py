"""Alias: `dpdfnet.weights` IS `dpdfnet_amd.weights` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.weights as _m

_sys.modules[__name__] = _m
