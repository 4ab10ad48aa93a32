"""Alias: `dpdfnet.audio` IS `dpdfnet_amd.audio` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.audio as _m

_sys.modules[__name__] = _m
