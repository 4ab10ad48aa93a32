"""Alias: `dpdfnet.stream` IS `dpdfnet_amd.stream` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.stream as _m

_sys.modules[__name__] = _m
