"""Alias: `dpdfnet.backend` IS `dpdfnet_amd.backend` (same module object; see dpdfnet/__init__.py)."""
import sys as _sys

import dpdfnet_amd.backend as _m

_sys.modules[__name__] = _m
