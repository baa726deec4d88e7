"""MI355X-native Smith-Waterman batch aligner: drop-in for the src/sw hot path of
elizabethfong/SparkSmithWaterman (matrix fill + tied-maximum search + traceback).

The compute path is libswmi.so (hand-written HIP for gfx950, C ABI in include/swmi.h).
Importing this package does not load the library; the first Context does, and fails loudly
if it has not been built.
"""
from ._capi import SwmiError, TIE_SERIAL, TIE_STRICT, PAIR_DEGENERATE, LIB_PATH
from .aligner import Context, Batch, Stream, make_params, DEFAULT_SCORES, DEFAULT_TYPES
from .sw import SmithWaterman, DistributedSW, Distribution, default_context

__all__ = ["SwmiError", "TIE_SERIAL", "TIE_STRICT", "PAIR_DEGENERATE", "LIB_PATH", "Context", "Batch", "Stream",
           "make_params", "DEFAULT_SCORES", "DEFAULT_TYPES", "SmithWaterman", "DistributedSW",
           "Distribution", "default_context"]
