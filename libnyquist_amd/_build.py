"""Compile libnyq_imdct.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libnyq_imdct.so")
LIB_AB = os.path.join(ROOT, "tools", "libnyq_imdct_ab.so")   # the same + the A/B kernel forms (-DNYQ_AB_FORMS): tools and tests only
SOURCES = [os.path.join(CSRC, "nyq_imdct.hip")]
AB_DIR = os.path.join(ROOT, "tools", "ab")                     # sources of the measured-and-rejected kernel forms (A/B build only)
DEPS = sorted(set(SOURCES + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hip"))
                  + [os.path.join(ROOT, "include", "nyq_imdct.h")]))
DEPS_AB = sorted(set(DEPS + glob.glob(os.path.join(AB_DIR, "*.hpp"))))


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libnyq_imdct.so cannot be built (no CPU fallback exists)")


def stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in (DEPS_AB if lib == LIB_AB else DEPS))


def build(force=False, verbose=False):
    """Build libnyquist_amd/libnyq_imdct.so if missing or older than its sources."""
    if not force and not stale():
        return LIB
    # -fno-slp-vectorize: on gfx950 packed f32 VALU issues at half rate, so hipcc's SLP packing of
    # the butterflies buys nothing and costs ~480 v_mov plus 70 VGPRs (218 -> 148: 2 -> 3 waves/SIMD).
    cmd = [hipcc(), "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-Wl,-Bsymbolic-functions", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_ab(force=False, verbose=False):
    """Build tools/libnyq_imdct_ab.so: the product's sources with -DNYQ_AB_FORMS (round-1 post-filter kernels and the
    fused chain kernel compiled in, selectable through nyq_ctx_set_option).  Not part of the product."""
    if not force and not stale(LIB_AB):
        return LIB_AB
    cmd = [hipcc(), "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DNYQ_AB_FORMS", "-I" + AB_DIR, "-I" + CSRC,
           "-Wl,-Bsymbolic-functions", "-o", LIB_AB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_AB
