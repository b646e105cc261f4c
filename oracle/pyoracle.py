"""ctypes front-ends for the TEST-ONLY checkers under oracle/.

  Oracle  -> oracle/liboracle.so        (this repo's C restatement, nyq_oracle.c)
  Ref     -> oracle/_ref/libnyq_ref.so  (the reference's own sources, built by oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing here is on the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libnyq_ref.so")

HALF_OV = 60
OVERLAP = 120
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libnyq_ref.so."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def n2_of(shift):
    return 960 >> shift


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """The C restatement.  `tables` = (trig, window, tw) to pin the reference's static tables."""

    def __init__(self, tables=None):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.nyq_oracle_init_tables.argtypes = [_f32p, _f32p, _f32p]
        L.nyq_oracle_get_tables.argtypes = [_f32p, _f32p, _f32p]
        L.nyq_oracle_get_plan.argtypes = [C.c_int, _i16p, _i32p, _i32p]
        L.nyq_oracle_ifft_shared.argtypes = [C.c_int, _f32p, _f32p]
        L.nyq_oracle_ifft_own.argtypes = [C.c_int, _f32p, _f32p]
        L.nyq_oracle_ifft_batch.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_long, C.c_int]
        L.nyq_oracle_imdct.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.nyq_oracle_imdct_batch.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, C.c_int]
        L.nyq_oracle_imdct_chain.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long]
        L.nyq_oracle_celt_synth.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, C.c_long,
                                            C.c_int, C.c_int]
        L.nyq_oracle_vorbis_imdct.argtypes = [C.c_int, _f32p, _f32p, C.c_long]
        L.nyq_oracle_celt_post.argtypes = [C.c_int, _f32p, C.c_long, C.c_long, _i32p, _f32p, _i32p, _f32p, _f32p,
                                           _f32p, C.c_long, C.c_long, C.c_int]
        if tables is None:
            L.nyq_oracle_init_default()
        else:
            t, w, tw = (np.ascontiguousarray(x, dtype=np.float32) for x in tables)
            assert t.size == 481 and w.size == 120 and tw.size == 960
            L.nyq_oracle_init_tables(t, w, tw.reshape(-1))

    def max_threads(self):
        return int(self.lib.nyq_oracle_max_threads())

    def tables(self):
        t = np.empty(481, np.float32)
        w = np.empty(120, np.float32)
        tw = np.empty(960, np.float32)
        self.lib.nyq_oracle_get_tables(t, w, tw)
        return t, w, tw.reshape(480, 2)

    def plan(self, shift):
        perm = np.zeros(480, np.int16)
        radix = np.zeros(8, np.int32)
        rest = np.zeros(8, np.int32)
        n = self.lib.nyq_oracle_get_plan(shift, perm, radix, rest)
        return perm[: 480 >> shift].copy(), radix[:n].copy(), rest[:n].copy()

    def ifft_shared(self, shift, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        assert x.size == 2 * (480 >> shift)
        y = np.empty_like(x)
        assert self.lib.nyq_oracle_ifft_shared(shift, x, y) == 0
        return y

    def ifft_own(self, nfft, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        assert x.size == 2 * nfft
        y = np.empty_like(x)
        assert self.lib.nyq_oracle_ifft_own(nfft, x, y) == 0
        return y

    def ifft_batch(self, nfft, x, shared=False, nthreads=1):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 2 * nfft)
        y = np.empty_like(x)
        rc = self.lib.nyq_oracle_ifft_batch(nfft, int(shared), x.reshape(-1), y.reshape(-1), x.shape[0], nthreads)
        assert rc == 0
        return y

    def imdct(self, x, out, shift, stride=1):
        """Single call, reference layout: `out` (N2+60 floats) is read-modify-write."""
        x = np.ascontiguousarray(x, np.float32)
        assert out.dtype == np.float32 and out.size == n2_of(shift) + HALF_OV
        assert x.size >= (n2_of(shift) - 1) * stride + 1
        assert self.lib.nyq_oracle_imdct(x.ctypes.data_as(C.c_void_p), out, shift, stride) == 0
        return out

    def imdct_batch(self, shift, x, carry=None, nthreads=1, want_tail=True):
        n2 = n2_of(shift)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n2)
        b = x.shape[0]
        if carry is not None:
            carry = np.ascontiguousarray(carry, np.float32).reshape(b, HALF_OV)
        fin = np.empty((b, n2), np.float32)
        tail = np.empty((b, HALF_OV), np.float32) if want_tail else None
        rc = self.lib.nyq_oracle_imdct_batch(shift, x.reshape(-1), _opt(carry), fin.reshape(-1), _opt(tail), b, nthreads)
        assert rc == 0
        return fin, tail

    def imdct_chain(self, shift, x, carry0=None):
        n2 = n2_of(shift)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n2)
        b = x.shape[0]
        if carry0 is not None:
            carry0 = np.ascontiguousarray(carry0, np.float32).reshape(HALF_OV)
        pcm = np.empty((b, n2), np.float32)
        tail = np.empty(HALF_OV, np.float32)
        rc = self.lib.nyq_oracle_imdct_chain(shift, x.reshape(-1), _opt(carry0), pcm.reshape(-1), _opt(tail), b)
        assert rc == 0
        return pcm, tail


    def celt_synth(self, lm, freq, transient=None, state=None, nthreads=1):
        """freq [ns][nf][ch][120<<lm] -> (pcm [ns][ch][nf*N], state_out or None); see nyq_oracle_celt_synth."""
        n = 120 << lm
        freq = np.ascontiguousarray(freq, np.float32)
        ns, nf, ch, nn = freq.shape
        assert nn == n
        tr = None if transient is None else np.ascontiguousarray(transient, np.uint8).reshape(ns, nf)
        st = None if state is None else np.ascontiguousarray(state, np.float32).reshape(ns * ch, HALF_OV).copy()
        pcm = np.empty((ns, ch, nf * n), np.float32)
        rc = self.lib.nyq_oracle_celt_synth(lm, freq.reshape(-1), _opt(tr), pcm.reshape(-1), _opt(st), ns, nf, ch, nthreads)
        assert rc == 0
        return pcm, st


    def celt_post(self, lm, pcm, pre, pf_pitch, pf_gain, pf_tapset, pf_state=None, deemph=None):
        """Post-filter + de-emphasis over frame sequences (nyq_oracle_celt_post).
        pcm [ns][ch][pre + nf*N] (history then IMDCT output); returns
        (out [ns][nf*N][ch], pcm_filtered, pf_state_out [ns][6], deemph_out [ns*ch])."""
        n = 120 << lm
        pcm = np.ascontiguousarray(pcm, np.float32).copy()
        ns, ch, pitch = pcm.shape
        nf = (pitch - pre) // n
        pp = np.ascontiguousarray(pf_pitch, np.int32).reshape(ns, nf)
        pg = np.ascontiguousarray(pf_gain, np.float32).reshape(ns, nf)
        pt = np.ascontiguousarray(pf_tapset, np.int32).reshape(ns, nf)
        st = np.zeros((ns, 6), np.float32) if pf_state is None else np.ascontiguousarray(pf_state, np.float32).reshape(ns, 6).copy()
        dm = np.zeros(ns * ch, np.float32) if deemph is None else np.ascontiguousarray(deemph, np.float32).reshape(ns * ch).copy()
        out = np.empty((ns, nf * n, ch), np.float32)
        rc = self.lib.nyq_oracle_celt_post(lm, pcm.reshape(-1), pitch, pre, pp.reshape(-1), pg.reshape(-1), pt.reshape(-1),
                                           st.reshape(-1), dm, out.reshape(-1), ns, nf, ch)
        assert rc == 0
        return out, pcm, st, dm


    def vorbis_imdct(self, n, x):
        """libvorbis mdct_backward from its closed form (double precision): [rows][n/2] -> [rows][n]."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n // 2)
        y = np.empty((x.shape[0], n), np.float32)
        assert self.lib.nyq_oracle_vorbis_imdct(n, x.reshape(-1), y.reshape(-1), x.shape[0]) == 0
        return y


def ref_available():
    return os.path.exists(REF_SO)


class Ref:
    """The reference's own compiled hot path (only where oracle/_ref was built)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (run `make -C oracle` where /root/reference exists)")
        L = self.lib = C.CDLL(REF_SO)
        L.ref_get_tables.argtypes = [_f32p, _f32p, _f32p, _i16p, _i16p]
        L.ref_imdct.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int]
        L.ref_imdct_c2.argtypes = [C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int, C.c_int]
        L.ref_ifft_shared.argtypes = [C.c_int, _f32p, _f32p]
        L.test_opus_ifft.argtypes = [C.c_int, _f32p, _f32p]
        L.ref_imdct_bench.argtypes = [_f32p, _f32p, C.c_int, C.c_long, C.c_int]
        L.ref_imdct_bench.restype = C.c_double
        _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.ref_get_alloc_tables.argtypes = [_i16p, _i16p, _u8p, _u8p]

    def tables(self):
        t = np.empty(481, np.float32)
        w = np.empty(120, np.float32)
        tw = np.empty(960, np.float32)
        br = np.empty(900, np.int16)
        fa = np.empty(64, np.int16)
        n = self.lib.ref_get_tables(t, w, tw, br, fa)
        assert n == 1920
        logn = np.zeros(21, np.int16)
        cidx = np.zeros(105, np.int16)
        cbits = np.zeros(1024, np.uint8)
        ccaps = np.zeros(168, np.uint8)
        nb = self.lib.ref_get_alloc_tables(logn, cidx, cbits, ccaps)
        return dict(trig=t, window=w, tw=tw.reshape(480, 2), bitrev=br, factors=fa.reshape(4, 16),
                    logN=logn, cache_index=cidx, cache_bits=cbits[:nb].copy(), cache_caps=ccaps)

    def imdct(self, x, out, shift, stride=1):
        x = np.ascontiguousarray(x, np.float32)
        assert out.dtype == np.float32 and out.size == n2_of(shift) + HALF_OV
        assert x.size >= (n2_of(shift) - 1) * stride + 1
        self.lib.ref_imdct(x.ctypes.data_as(C.c_void_p), out, shift, stride)
        return out

    def imdct_c2(self, x0, x1, out0, out1, shift, stride=1):
        self.lib.ref_imdct_c2(x0.ctypes.data_as(C.c_void_p), x1.ctypes.data_as(C.c_void_p), out0, out1, shift, stride)

    def ifft_shared(self, shift, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        y = np.empty_like(x)
        self.lib.ref_ifft_shared(shift, x, y)
        return y

    def ifft_own(self, nfft, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        y = np.zeros_like(x)
        self.lib.test_opus_ifft(nfft, x, y)
        return y

    def bench(self, x, shift, reps):
        n2 = n2_of(shift)
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n2)
        scratch = np.zeros((x.shape[0], n2 + HALF_OV), np.float32)
        return float(self.lib.ref_imdct_bench(x.reshape(-1), scratch.reshape(-1), shift, x.shape[0], reps))


class VorbisRef:
    """libvorbis' own mdct.c compiled standalone (oracle/_ref/libvorbis_ref.so)."""

    class _Lookup(C.Structure):
        _fields_ = [("n", C.c_int), ("log2n", C.c_int), ("trig", C.c_void_p), ("bitrev", C.c_void_p), ("scale", C.c_float)]

    def __init__(self):
        so = os.path.join(HERE, "_ref", "libvorbis_ref.so")
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        L = self.lib = C.CDLL(so)
        L.mdct_init.argtypes = [C.POINTER(self._Lookup), C.c_int]
        L.mdct_backward.argtypes = [C.POINTER(self._Lookup), _f32p, _f32p]
        self._lu = {}

    def backward(self, n, x):
        if n not in self._lu:
            lu = self._Lookup()
            self.lib.mdct_init(C.byref(lu), n)
            self._lu[n] = lu
        x = np.ascontiguousarray(x, np.float32).reshape(-1, n // 2)
        y = np.zeros((x.shape[0], n), np.float32)
        for r in range(x.shape[0]):
            self.lib.mdct_backward(C.byref(self._lu[n]), x[r].copy(), y[r])
        return y
