"""Alias: `dpdfnet.runtime` IS `dpdfnet_amd.runtime` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.runtime as _m

_sys.modules[__name__] = _m
