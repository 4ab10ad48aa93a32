"""Alias: `dpdfnet.evalkit` IS `dpdfnet_amd.evalkit` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.evalkit as _m

_sys.modules[__name__] = _m
