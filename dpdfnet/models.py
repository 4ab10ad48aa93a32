"""Alias: `dpdfnet.models` IS `dpdfnet_amd.models` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.models as _m

_sys.modules[__name__] = _m
