"""slacken_amd -- MI355X-native engine for Slacken's classify hot path (scan -> lookup -> per-read LCA).

The product is libslacken_amd.so (hand-written HIP for gfx950 behind the C ABI in include/slacken_amd.h).
This package is the thin Python plumbing used by tests and bench.py: a ctypes binding of that ABI.
There is no CPU fallback: importing works anywhere, computing needs the built library and a gfx950 device.
"""
from .capi import (Index, Stream, SlackenError, lib, lib_path, ClassifyParams, DEFAULT_TOGGLE_MASK,  # noqa: F401
                   TAXON_NONE, TAXON_ROOT, TAXON_AMBIGUOUS, TAXON_MATE_PAIR_BORDER,
                   FLAG_SEQUENCE, FLAG_AMBIGUOUS, FLAG_MATE_PAIR_BORDER)

__version__ = "0.1.0"
