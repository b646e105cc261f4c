"""ctypes binding of include/nyq_imdct.h -- the test/bench harness side of the C ABI.

The product is libnyq_imdct.so; this module only marshals numpy arrays and torch
device pointers into it.  It never computes anything itself and has no fallback:
a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

# nyq_ctx_set_option (include/nyq_imdct.h)
OPT_BLOCKS_PER_CU, OPT_POST_FORM, OPT_CHAIN_FUSED, OPT_CHAIN_WINDOW, OPT_CHAIN_OVERLAP, OPT_HOST_WINDOW = 1, 2, 3, 4, 5, 6
POST_FORM_PIPELINE, POST_FORM_WAVE_PER_CHANNEL, POST_FORM_WAVE_PER_PAIR = 0, 1, 2

HALF_OV = 60
OVERLAP = 120
NYQ_OK = 0
ERRORS = {-1: "NYQ_ERR_INVALID", -2: "NYQ_ERR_NO_DEVICE", -3: "NYQ_ERR_HIP", -4: "NYQ_ERR_ALLOC"}

# every symbol include/nyq_imdct.h declares (tests check the .so exports all of them)
EXPORTS = [
    "nyq_device_count", "nyq_celt_post_round_chains", "nyq_ctx_set_option", "nyq_ctx_get_option", "nyq_ab_forms_built",
    "nyq_ctx_create", "nyq_ctx_destroy", "nyq_last_error", "nyq_ctx_set_stream", "nyq_ctx_reset_stream", "nyq_ctx_get_stream",
    "nyq_ctx_synchronize", "nyq_ctx_set_tables", "nyq_ctx_get_tables", "nyq_ctx_device_info",
    "nyq_ifft_batch_dev", "nyq_imdct_batch_dev", "nyq_imdct_chain_dev",
    "nyq_celt_synth_work_floats", "nyq_celt_synth_dev", "nyq_celt_synth", "nyq_celt_post_dev",
    "nyq_celt_chain_fused_supported", "nyq_celt_chain_dev", "nyq_device_copy_forms", "nyq_device_copy_form_name", "nyq_device_copy_dev",
    "nyq_celt_chain_mapped_dev", "nyq_device_alloc", "nyq_device_free", "nyq_device_zero", "nyq_device_download", "nyq_device_dup_channel",
    "nyq_celt_frames_to_pcm_mapped", "nyq_celt_symbol_bytes", "nyq_celt_shape_dev", "nyq_celt_symbols_to_pcm_mapped",
    "nyq_celt_symbols_packed_to_pcm_mapped", "nyq_celt_symbol_bytes_lm", "nyq_celt_shape_lm_dev",
    "nyq_celt_entropy_tables_bytes", "nyq_celt_entropy_dev", "nyq_celt_entropy_split_dev", "nyq_celt_entropy_slot_bytes", "nyq_celt_shape_slots_dev",
    "nyq_ctx_set_entropy_tables", "nyq_celt_byte_slot", "nyq_celt_bytes_to_pcm_mapped",
    "nyq_celt_frames_to_pcm", "nyq_celt_frames_to_pcm_window", "nyq_celt_state_floats", "nyq_vorbis_imdct_batch_dev", "nyq_vorbis_imdct_batch",
    "nyq_ifft_batch", "nyq_imdct_batch", "nyq_imdct_chain", "nyq_host_alloc", "nyq_host_free",
    "processMDCTCuda", "processMDCTCudaB1C2", "processMDCTCudaB8C2", "cleanupCudaBuffers", "printCudaVersion", "nyq_shim_set_error_handler",
]


def pinned_empty(shape, dtype="float32"):
    """numpy array over page-locked host memory from nyq_host_alloc (freed when the array is collected)."""
    L = load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = L.nyq_host_alloc(max(n, 1))
    if not p:
        raise MemoryError(f"nyq_host_alloc({n}) failed")
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                L.nyq_host_free(self.ptr)
            except Exception:
                pass

    buf._nyq_owner = _Owner(p)   # every numpy view keeps `buf` (its base) alive, and with it the allocation
    return arr


class NyqError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{ERRORS.get(code, code)}: {text}")
        self.code = code


_lib = None
_lib_ab = None


def load_ab():
    """The tools' build of the library (-DNYQ_AB_FORMS, tools/libnyq_imdct_ab.so): the product's kernels plus the
    measured-and-rejected forms (round-1 post-filter kernels, fused chain) for A/B runs and their parity tests."""
    global _lib_ab
    if _lib_ab is None:
        try:
            _build.hipcc()
        except RuntimeError:                     # no hipcc here: use the library that travelled with the tree, if any
            path = _build.LIB_AB
            if not os.path.exists(path):
                raise
        else:
            path = _build.build_ab()             # (a compile error of the A/B sources propagates: no stale binary is loaded)
        _lib_ab = load(path)
    return _lib_ab


def load(path=None):
    """dlopen libnyq_imdct.so (building it first when hipcc is available and it is stale)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # One process must use ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; if it is going
    # to be used beside this library (device buffers, streams), it has to be loaded first so that our
    # DT_NEEDED libamdhip64.so resolves to the same, already loaded runtime.  Harmless when absent.
    if os.environ.get("NYQ_NO_TORCH") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    p = path or _build.LIB
    if path is None:
        try:
            _build.build()
        except Exception:
            if not os.path.exists(p):
                raise
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p}: HIP extension missing -- run __graft_entry__.build(); there is no CPU fallback")
    L = C.CDLL(p)
    vp, i, sz, fp = C.c_void_p, C.c_int, C.c_size_t, C.c_void_p
    L.nyq_ctx_create.argtypes = [C.POINTER(vp), i]
    L.nyq_ctx_set_option.argtypes = [vp, i, C.c_long]
    L.nyq_ctx_get_option.argtypes = [vp, i, C.POINTER(C.c_long)]
    L.nyq_ctx_destroy.argtypes = [vp]
    L.nyq_ctx_destroy.restype = None
    L.nyq_last_error.argtypes = [vp]
    L.nyq_last_error.restype = C.c_char_p
    L.nyq_ctx_set_stream.argtypes = [vp, vp]
    L.nyq_ctx_reset_stream.argtypes = [vp]
    L.nyq_ctx_get_stream.argtypes = [vp]
    L.nyq_ctx_get_stream.restype = vp
    L.nyq_ctx_synchronize.argtypes = [vp]
    L.nyq_ctx_set_tables.argtypes = [vp, fp, fp]
    L.nyq_ctx_get_tables.argtypes = [vp, fp, fp]
    L.nyq_ctx_device_info.argtypes = [vp, C.POINTER(i), C.c_char_p, sz]
    L.nyq_ifft_batch_dev.argtypes = [vp, i, fp, fp, sz]
    L.nyq_imdct_batch_dev.argtypes = [vp, i, fp, fp, fp, fp, sz]
    L.nyq_imdct_chain_dev.argtypes = [vp, i, fp, fp, fp, fp, fp, sz, sz]
    L.nyq_celt_synth_work_floats.argtypes = [sz, sz, i]
    L.nyq_celt_synth_work_floats.restype = sz
    L.nyq_celt_synth_dev.argtypes = [vp, i, fp, fp, fp, fp, fp, sz, sz, i]
    L.nyq_celt_synth.argtypes = [vp, i, fp, fp, fp, fp, sz, sz, i]
    L.nyq_celt_post_dev.argtypes = [vp, i, fp, fp, fp, fp, fp, fp, fp, fp, fp, sz, sz, i]
    L.nyq_celt_post_round_chains.argtypes = [vp]
    L.nyq_celt_post_round_chains.restype = sz
    L.nyq_celt_chain_fused_supported.argtypes = [i, i]
    L.nyq_celt_chain_mapped_dev.argtypes = [vp, i] + [vp] * 14 + [sz, sz, i]
    L.nyq_device_alloc.argtypes = [vp, sz]
    L.nyq_device_alloc.restype = vp
    L.nyq_device_free.argtypes = [vp, vp]
    L.nyq_device_free.restype = None
    L.nyq_device_zero.argtypes = [vp, vp, sz]
    L.nyq_device_download.argtypes = [vp, vp, vp, sz]
    L.nyq_device_dup_channel.argtypes = [vp, vp, i, i, i, sz]
    L.nyq_celt_frames_to_pcm_mapped.argtypes = [vp, i] + [vp] * 8 + [sz, sz, i, sz]
    L.nyq_celt_symbol_bytes.argtypes = [i]
    L.nyq_celt_symbol_bytes.restype = sz
    L.nyq_celt_shape_dev.argtypes = [vp, vp, vp, sz, sz, i, sz]
    L.nyq_celt_symbols_to_pcm_mapped.argtypes = [vp, i] + [vp] * 8 + [sz, sz, i, sz]
    L.nyq_celt_symbols_packed_to_pcm_mapped.argtypes = [vp, i, vp, vp, sz] + [vp] * 7 + [sz, sz, i, sz]
    L.nyq_celt_symbol_bytes_lm.argtypes = [i, i]
    L.nyq_celt_symbol_bytes_lm.restype = sz
    L.nyq_celt_shape_lm_dev.argtypes = [vp, i, vp, vp, sz, sz, i, sz]
    L.nyq_celt_entropy_tables_bytes.restype = sz
    L.nyq_celt_entropy_dev.argtypes = [vp, i, vp, vp, sz, vp, sz, sz, i, vp, sz, vp, vp, vp, i]
    L.nyq_celt_entropy_slot_bytes.argtypes = [i, i]
    L.nyq_celt_entropy_slot_bytes.restype = sz
    L.nyq_celt_shape_slots_dev.argtypes = [vp, i, vp, sz, vp, sz, sz, i]
    L.nyq_celt_entropy_split_dev.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.nyq_ctx_set_entropy_tables.argtypes = [vp, vp, sz]
    L.nyq_celt_byte_slot.restype = sz
    L.nyq_celt_bytes_to_pcm_mapped.argtypes = [vp, i, vp, vp, vp, vp, vp, sz, sz, i, sz]
    L.nyq_device_copy_forms.restype = i
    L.nyq_device_copy_form_name.argtypes = [i]
    L.nyq_device_copy_form_name.restype = C.c_char_p
    L.nyq_device_copy_dev.argtypes = [vp, vp, vp, sz, i]
    L.nyq_celt_chain_dev.argtypes = [vp, i] + [fp] * 13 + [sz, sz, i]
    L.nyq_celt_frames_to_pcm.argtypes = [vp, i, fp, fp, fp, fp, fp, fp, fp, sz, sz, i]
    L.nyq_celt_frames_to_pcm_window.argtypes = [vp, i, fp, fp, fp, fp, fp, fp, fp, sz, sz, i, sz]
    L.nyq_celt_state_floats.argtypes = [sz, i]
    L.nyq_celt_state_floats.restype = sz
    L.nyq_vorbis_imdct_batch_dev.argtypes = [vp, i, fp, fp, sz]
    L.nyq_vorbis_imdct_batch.argtypes = [vp, i, fp, fp, sz]
    L.nyq_ifft_batch.argtypes = [vp, i, fp, fp, sz]
    L.nyq_imdct_batch.argtypes = [vp, i, fp, fp, fp, fp, sz]
    L.nyq_imdct_chain.argtypes = [vp, i, fp, fp, fp, fp, sz, sz]
    L.nyq_host_alloc.argtypes = [sz]
    L.nyq_host_alloc.restype = vp
    L.nyq_host_free.argtypes = [vp]
    L.nyq_host_free.restype = None
    L.processMDCTCuda.argtypes = [fp, fp, fp, i, i, i, C.c_float, i, fp]
    L.processMDCTCuda.restype = None
    L.processMDCTCudaB1C2.argtypes = [C.POINTER(fp), C.POINTER(fp), fp, i, i, i, C.c_float, i, fp]
    L.processMDCTCudaB1C2.restype = None
    L.processMDCTCudaB8C2.argtypes = [C.POINTER(fp), C.POINTER(fp), fp, i, i, i, C.c_float, i, fp]
    L.processMDCTCudaB8C2.restype = None
    L.cleanupCudaBuffers.restype = None
    L.printCudaVersion.restype = None
    if path is None:
        _lib = L
    return L


def _np(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def n2_of(shift):
    return 960 >> shift


class Context:
    """nyq_ctx wrapper.  Host-buffer methods take/return numpy arrays; *_dev methods take
    raw device pointers (ints, e.g. torch.Tensor.data_ptr()) and are asynchronous."""

    def __init__(self, device=0, ab=False):
        """ab=True: a context of the tools' A/B build (load_ab) instead of the product library."""
        self.lib = load_ab() if ab else load()
        h = C.c_void_p()
        rc = self.lib.nyq_ctx_create(C.byref(h), int(device))
        if rc != NYQ_OK:
            raise NyqError(rc, (self.lib.nyq_last_error(None) or b"").decode())
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.nyq_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != NYQ_OK:
            raise NyqError(rc, (self.lib.nyq_last_error(self.h) or b"").decode())

    # -- context plumbing
    def set_option(self, option, value):
        self._ck(self.lib.nyq_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_long(0)
        self._ck(self.lib.nyq_ctx_get_option(self.h, int(option), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        """Run on the given hipStream_t (int handle); 0 is HIP's default stream."""
        self._ck(self.lib.nyq_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        self._ck(self.lib.nyq_ctx_reset_stream(self.h))

    def synchronize(self):
        self._ck(self.lib.nyq_ctx_synchronize(self.h))

    def set_tables(self, trig, window):
        t, w = _f32(trig), _f32(window)
        assert t.size == 481 and w.size == 120
        self._ck(self.lib.nyq_ctx_set_tables(self.h, _np(t), _np(w)))

    def get_tables(self):
        t, w = np.empty(481, np.float32), np.empty(120, np.float32)
        self._ck(self.lib.nyq_ctx_get_tables(self.h, _np(t), _np(w)))
        return t, w

    def device_info(self):
        cus = C.c_int(0)
        name = C.create_string_buffer(256)
        self._ck(self.lib.nyq_ctx_device_info(self.h, C.byref(cus), name, 256))
        return cus.value, name.value.decode()

    # -- host-buffer operators
    def ifft_batch(self, nfft, x):
        x = _f32(x, (-1, 2 * nfft))
        y = np.empty_like(x)
        self._ck(self.lib.nyq_ifft_batch(self.h, nfft, _np(x), _np(y), x.shape[0]))
        return y

    def imdct_batch(self, shift, x, carry=None, want_tail=True, pinned=False):
        """pinned=True returns page-locked outputs (pass inputs made by pinned_empty for DMA both ways)."""
        n2 = n2_of(shift)
        x = _f32(x, (-1, n2))
        b = x.shape[0]
        carry = None if carry is None else _f32(carry, (b, HALF_OV))
        empty = pinned_empty if pinned else np.empty
        fin = empty((b, n2), np.float32)
        tail = empty((b, HALF_OV), np.float32) if want_tail else None
        self._ck(self.lib.nyq_imdct_batch(self.h, shift, _np(x), _np(carry), _np(fin), _np(tail), b))
        return fin, tail

    def imdct_chain(self, shift, x, carry0=None, nchains=1, pinned=False):
        n2 = n2_of(shift)
        x = _f32(x, (-1, n2))
        rows = x.shape[0]
        assert rows % nchains == 0
        carry0 = None if carry0 is None else _f32(carry0, (nchains, HALF_OV))
        pcm = (pinned_empty if pinned else np.empty)((rows, n2), np.float32)
        tail = np.empty((nchains, HALF_OV), np.float32)
        self._ck(self.lib.nyq_imdct_chain(self.h, shift, _np(x), _np(carry0), _np(pcm), _np(tail), nchains,
                                          rows // nchains))
        return pcm, tail

    def celt_synth(self, lm, freq, transient=None, state=None, channels=2):
        """freq [nstreams][nframes][channels][120<<lm]; transient [nstreams][nframes] uint8 or None;
        state [nstreams*channels][60] or None.  Returns (pcm [nstreams][channels][nframes*N], state_out)."""
        n = 120 << lm
        freq = _f32(freq)
        assert freq.ndim == 4 and freq.shape[2] == channels and freq.shape[3] == n
        ns, nf = freq.shape[0], freq.shape[1]
        tr = None if transient is None else np.ascontiguousarray(transient, dtype=np.uint8).reshape(ns, nf)
        st = None if state is None else _f32(state, (ns * channels, HALF_OV)).copy()
        pcm = np.empty((ns, channels, nf * n), np.float32)
        self._ck(self.lib.nyq_celt_synth(self.h, lm, _np(freq), _np(tr), _np(pcm), _np(st), ns, nf, channels))
        return pcm, st

    def celt_frames_to_pcm(self, lm, freq, transient, pf_pitch, pf_gain, pf_tapset, channels, state=None):
        """freq [ns][nf][ch][N] (+ per-frame flags/parameters [ns][nf]) -> interleaved PCM [ns][nf*N][ch].
        `state`: float32 array of nyq_celt_state_floats(ns, ch) floats, updated in place (None = fresh)."""
        n = 120 << lm
        freq = _f32(freq)
        ns, nf = freq.shape[0], freq.shape[1]
        tr = None if transient is None else np.ascontiguousarray(transient, dtype=np.uint8).reshape(ns, nf)
        pp = np.ascontiguousarray(pf_pitch, dtype=np.int32).reshape(ns, nf)
        pg = np.ascontiguousarray(pf_gain, dtype=np.float32).reshape(ns, nf)
        pt = np.ascontiguousarray(pf_tapset, dtype=np.int32).reshape(ns, nf)
        out = np.empty((ns, nf * n, channels), np.float32)
        self._ck(self.lib.nyq_celt_frames_to_pcm(self.h, lm, _np(freq), _np(tr), _np(pp), _np(pg), _np(pt), _np(out),
                                                 _np(state), ns, nf, channels))
        return out

    def vorbis_imdct_batch(self, n, x):
        """libvorbis mdct_backward on rows: x [batch][n/2] -> [batch][n]."""
        x = _f32(x, (-1, n // 2))
        y = np.empty((x.shape[0], n), np.float32)
        self._ck(self.lib.nyq_vorbis_imdct_batch(self.h, n, _np(x), _np(y), x.shape[0]))
        return y

    def vorbis_imdct_batch_dev(self, n, d_in, d_out, batch):
        self._ck(self.lib.nyq_vorbis_imdct_batch_dev(self.h, n, C.c_void_p(d_in), C.c_void_p(d_out), batch))

    def celt_synth_work_floats(self, nstreams, nframes, channels):
        return int(self.lib.nyq_celt_synth_work_floats(nstreams, nframes, channels))

    def celt_synth_dev(self, lm, d_freq, d_transient, d_pcm, d_state, d_work, nstreams, nframes, channels):
        self._ck(self.lib.nyq_celt_synth_dev(self.h, lm, C.c_void_p(d_freq), C.c_void_p(d_transient or 0),
                                             C.c_void_p(d_pcm), C.c_void_p(d_state or 0), C.c_void_p(d_work),
                                             nstreams, nframes, channels))

    def celt_post_dev(self, lm, d_pcm, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out, d_hist,
                      d_deemph, d_out, nstreams, nframes, channels):
        V = lambda p: C.c_void_p(p or 0)
        self._ck(self.lib.nyq_celt_post_dev(self.h, lm, V(d_pcm), V(d_pf_pitch), V(d_pf_gain), V(d_pf_tapset),
                                            V(d_pf_state_in), V(d_pf_state_out), V(d_hist), V(d_deemph), V(d_out),
                                            nstreams, nframes, channels))

    def celt_chain_dev(self, lm, d_freq, d_transient, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out,
                       d_overlap, d_hist, d_deemph, d_out, d_pcm, d_work, nstreams, nframes, channels):
        """freq[] -> interleaved PCM (one fused launch for LM 3 stereo, the two kernels otherwise)."""
        V = lambda p: C.c_void_p(p or 0)
        self._ck(self.lib.nyq_celt_chain_dev(self.h, lm, V(d_freq), V(d_transient), V(d_pf_pitch), V(d_pf_gain), V(d_pf_tapset),
                                             V(d_pf_state_in), V(d_pf_state_out), V(d_overlap), V(d_hist), V(d_deemph),
                                             V(d_out), V(d_pcm), V(d_work), nstreams, nframes, channels))

    def celt_shape_dev(self, d_sym, d_freq, nstreams, nframes, channels, sstride=0, lm=3):
        """symbol records -> freq[] (the band shapes of frames of 120 << lm samples built on the device)."""
        self._ck(self.lib.nyq_celt_shape_lm_dev(self.h, lm, C.c_void_p(d_sym), C.c_void_p(d_freq), nstreams, nframes, channels, sstride))

    def celt_entropy_dev(self, lm, d_tables, d_payload, payload_bytes, d_desc, nstreams, nframes, channels, d_sym, d_info, d_energy, d_state, fresh=True,
                         slot_bytes=0):
        """frames' bytes -> spread symbol records, a frame per lane, then the energy pass, a wave per stream (nyq_celt_entropy_dev)"""
        V = lambda p: C.c_void_p(p or 0)
        self._ck(self.lib.nyq_celt_entropy_dev(self.h, lm, V(d_tables), V(d_payload), payload_bytes, V(d_desc), nstreams, nframes, channels, V(d_sym),
                                               slot_bytes, V(d_info), V(d_energy), V(d_state), 1 if fresh else 0))

    def celt_shape_slots_dev(self, lm, d_sym, slot_bytes, d_freq, nstreams, nframes, channels):
        """records in slots of slot_bytes (nyq_celt_entropy_slot_bytes: any frame fits) -> freq[]"""
        self._ck(self.lib.nyq_celt_shape_slots_dev(self.h, lm, C.c_void_p(d_sym), slot_bytes, C.c_void_p(d_freq), nstreams, nframes, channels))

    def celt_entropy_split_dev(self, d_info, n, d_transient, d_pf_pitch, d_pf_gain, d_pf_tapset):
        V = lambda p: C.c_void_p(p or 0)
        self._ck(self.lib.nyq_celt_entropy_split_dev(self.h, V(d_info), n, V(d_transient), V(d_pf_pitch), V(d_pf_gain), V(d_pf_tapset)))

    def set_entropy_tables(self, tables):
        self._ck(self.lib.nyq_ctx_set_entropy_tables(self.h, _np(tables), tables.nbytes))

    def celt_bytes_to_pcm(self, lm, frame_bytes, frame_words, nstreams, nframes, channels, state=None):
        """frames' bytes [nstreams][nframes][nyq_celt_byte_slot] + words [nstreams][nframes] -> interleaved PCM (entropy stage on the device)"""
        out = np.empty((nstreams, nframes * (120 << lm), channels), np.float32)
        self._ck(self.lib.nyq_celt_bytes_to_pcm_mapped(self.h, lm, _np(frame_bytes), _np(frame_words), _np(out), None, _np(state), nstreams, nframes,
                                                      channels, nframes))
        return out

    def celt_symbols_to_pcm(self, sym, transient, pf_pitch, pf_gain, pf_tapset, nstreams, nframes, channels, state=None, lm=3):
        """host symbol records [nstreams][nframes][nyq_celt_symbol_bytes_lm] -> interleaved PCM [nstreams][nframes * (120 << lm)][channels]"""
        out = np.empty((nstreams, nframes * (120 << lm), channels), np.float32)
        self._ck(self.lib.nyq_celt_symbols_to_pcm_mapped(self.h, lm, _np(sym), _np(transient), _np(pf_pitch), _np(pf_gain), _np(pf_tapset),
                                                        _np(out), None, _np(state), nstreams, nframes, channels, nframes))
        return out

    def celt_symbols_packed_to_pcm(self, sym, offsets, stream_bytes, transient, pf_pitch, pf_gain, pf_tapset, nstreams, nframes, channels,
                                   state=None):
        """the same with records packed back to back: offsets uint32 [nstreams][nframes + 1], 16-byte units from each stream's base"""
        out = np.empty((nstreams, nframes * 960, channels), np.float32)
        self._ck(self.lib.nyq_celt_symbols_packed_to_pcm_mapped(self.h, 3, _np(sym), _np(offsets), stream_bytes, _np(transient), _np(pf_pitch),
                                                               _np(pf_gain), _np(pf_tapset), _np(out), None, _np(state), nstreams, nframes,
                                                               channels, nframes))
        return out

    # -- device-resident operators (raw pointers)
    def ifft_batch_dev(self, nfft, d_in, d_out, batch):
        self._ck(self.lib.nyq_ifft_batch_dev(self.h, nfft, C.c_void_p(d_in), C.c_void_p(d_out), batch))

    def imdct_batch_dev(self, shift, d_in, d_carry, d_fin, d_tail, batch):
        self._ck(self.lib.nyq_imdct_batch_dev(self.h, shift, C.c_void_p(d_in), C.c_void_p(d_carry or 0),
                                              C.c_void_p(d_fin), C.c_void_p(d_tail or 0), batch))

    def imdct_chain_dev(self, shift, d_in, d_carry0, d_pcm, d_tail_out, d_work, nchains, length):
        self._ck(self.lib.nyq_imdct_chain_dev(self.h, shift, C.c_void_p(d_in), C.c_void_p(d_carry0 or 0),
                                              C.c_void_p(d_pcm), C.c_void_p(d_tail_out or 0), C.c_void_p(d_work),
                                              nchains, length))
