"""libnyquist_amd -- MI355X-native batched CELT inverse-MDCT path for libnyquist.

The product is the C-ABI shared library libnyq_imdct.so (include/nyq_imdct.h), built
from hand-written gfx950 HIP kernels under csrc/.  This Python package is the harness
used by tests/ and bench.py: a ctypes binding plus the in-tree build helper.
"""
from ._build import LIB as LIB_PATH, LIB_AB as LIB_AB_PATH, build, build_ab  # noqa: F401
from . import binding  # noqa: F401
from .binding import EXPORTS, Context, NyqError, load, load_ab, n2_of, pinned_empty  # noqa: F401
