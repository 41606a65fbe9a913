"""metakssd_amd -- MI355X-native engine for MetaKSSD's k-mer sketching hot path.

The product is the C-ABI shared library (metakssd_amd/lib/libmetakssd_hip.so, sources in
metakssd_amd/csrc) and the C command line (metakssd_amd/bin/metakssd).  This package is the thin
ctypes binding used by tests and bench.py; importing it requires the built library.
"""
from . import capi  # noqa: F401
from .capi import Engine, Shuf, MkError, CrowdedError, MK_MODE_KOC, MK_MODE_SET, MK_MODE_UNIQ_SET, MK_MODE_OCC_SET  # noqa: F401
