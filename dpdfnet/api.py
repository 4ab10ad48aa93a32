"""Alias: `dpdfnet.api` IS `dpdfnet_amd.api` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.api as _m

_sys.modules[__name__] = _m
