"""libnyquist_amd -- MI355X-native batched CELT inverse-MDCT path for libnyquist.

The product is the C-ABI shared library libnyq_imdct.so (include/nyq_imdct.h), built
from hand-written gfx950 HIP kernels under csrc/.  This Python package is the harness
used by tests/ and bench.py: a ctypes binding plus the in-tree build helper.
"""
from ._build import LIB as LIB_PATH, build  # noqa: F401
from .binding import EXPORTS, Context, NyqError, load, n2_of, pinned_empty  # noqa: F401
