#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/libnyq_ref.so).

Run in the build container only (needs /root/reference):

    make -C oracle && python oracle/gen_golden.py

Writes DATA only (inputs + expected outputs):
  ifft_{input,output}_N{60,480}.bin   byte copies of the reference's own bundled test
                                      vectors (test_data/), the IFFT-stage known answers
  ref_tables.npz                      static 48 kHz mode tables read out of the reference
                                      (trig[481], window[120], tw[480][2], bitrev, factors)
  ref_imdct_s{0,1,2,3}.npz            clt_mdct_backward in/carry/out rows, stride 1
  ref_imdct_strided.npz               stride-8 (transient layout) and B1_C2 stereo calls
  ref_ifft_shared.npz                 opus_ifft through the mode's shared plans, all 4 sizes
  ref_chain.npz                       consecutive blocks of one channel, long/short mixed,
                                      emulating the decode_mem shift of celt_decoder_clean.c:625,641
  ref_synth.npz                       compute_inv_mdcts over 2 stereo streams x 11 frames with
                                      transient frames, through the reference's B1_C2 fast paths
  real_opus_frames.npz                REAL data: freq[] / isTransient / out_syn of frames 64..127 of
                                      test_data/short.opus as the reference decoder itself computed
                                      them (NyquistIO::Load with the recording tap), the transient
                                      maps of short.opus and sb-reverie.opus, the end-to-end sample
                                      counts / checksums (examples/src/Main.cpp:137-154); and for the
                                      same frames the post-filter parameters and state, the filtered
                                      history, the de-emphasis memory and the FINAL decoded PCM
                                      (AudioData::samples) -- the whole freq[] -> PCM chain
  ref_vorbis.npz                      libvorbis mdct_backward (mdct.c compiled standalone), block sizes 64..8192
  short.opus, short_opus_digest.npz   the bundled test file itself plus per-frame digests of the
                                      reference decoder's freq[] for all 220 frames, its post-filter
                                      parameters and its complete decoded PCM (end-to-end check of
                                      the host-side decoder + GPU chain)
Everything is seeded; re-running reproduces the files bit for bit.
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import HALF_OV, Ref, n2_of  # noqa: E402

REFDATA = "/root/reference/test_data"
OUT = os.path.join(ROOT, "tests", "golden")


def decoder_like(rng, rows, n2):
    """freq[]-shaped rows: rms ~30, peaks of a few thousand, top 1/6 of the band zero
    (measured statistics of real decoder input)."""
    x = rng.standard_normal((rows, n2)).astype(np.float32) * 30.0
    spikes = rng.integers(0, n2 * 5 // 6, size=(rows, 4))
    for r in range(rows):
        x[r, spikes[r]] *= 60.0
    x[:, n2 * 5 // 6:] = 0.0
    return x


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Ref()

    for n in (60, 480):
        for kind in ("input", "output"):
            shutil.copyfile(f"{REFDATA}/ifft_{kind}_N{n}.bin", f"{OUT}/ifft_{kind}_N{n}.bin")

    T = ref.tables()
    np.savez(f"{OUT}/ref_tables.npz", **T)

    rng = np.random.default_rng(20241223)
    for s in range(4):
        n2 = n2_of(s)
        rows = 16
        x = np.concatenate([
            rng.uniform(-1, 1, (4, n2)).astype(np.float32),        # unit scale
            decoder_like(rng, 8, n2),                              # decoder scale
            np.zeros((1, n2), np.float32),                         # silence
            np.eye(1, n2, 0, dtype=np.float32) * 1000.0,           # DC-bin impulse
            np.eye(1, n2, n2 - 1, dtype=np.float32) * 1000.0,      # last-bin impulse
            rng.uniform(-4000, 4000, (1, n2)).astype(np.float32),  # full-scale noise
        ])
        assert x.shape[0] == rows
        carry = (rng.standard_normal((rows, HALF_OV)) * 30).astype(np.float32)
        carry[::2] = 0.0                                           # every other row: zero carry
        out = np.zeros((rows, n2 + HALF_OV), np.float32)
        for r in range(rows):
            out[r, :HALF_OV] = carry[r]
            ref.imdct(x[r], out[r], s, 1)
        np.savez(f"{OUT}/ref_imdct_s{s}.npz", x=x, carry=carry, out=out)

    # transient layout: one channel-frame = 960 floats, coefficient k of short block b at X[b + 8k]
    # (celt_decoder_clean.c:292-300); 8 sequential calls, block b writes out_mem + 120*b, so
    # block b+1 sees block b's raw tail as carry.
    frames = 3
    X = decoder_like(rng, frames * 2, 960).reshape(frames, 2, 960)
    carry0 = (rng.standard_normal((frames, 2, HALF_OV)) * 30).astype(np.float32)
    syn = np.zeros((frames, 2, 960 + HALF_OV), np.float32)
    for f in range(frames):
        syn[f, :, :HALF_OV] = carry0[f]
        for b in range(8):
            o0 = syn[f, 0, 120 * b: 120 * b + 180]
            o1 = syn[f, 1, 120 * b: 120 * b + 180]
            ref.imdct_c2(X[f, 0, b:], X[f, 1, b:], o0, o1, 3, 8)
    # long stereo frame through B1_C2, stride 1
    XL = decoder_like(rng, 4, 960).reshape(2, 2, 960)
    cl = (rng.standard_normal((2, 2, HALF_OV)) * 30).astype(np.float32)
    synl = np.zeros((2, 2, 960 + HALF_OV), np.float32)
    for f in range(2):
        synl[f, :, :HALF_OV] = cl[f]
        ref.imdct_c2(XL[f, 0], XL[f, 1], synl[f, 0], synl[f, 1], 0, 1)
    # generic strided single-channel calls for shifts 1 and 2 (B = 2, 4 interleaved blocks)
    gen = {}
    for s, B in ((1, 2), (2, 4)):
        n2 = n2_of(s)
        xs = decoder_like(rng, 1, 960)[0]
        c0 = (rng.standard_normal(HALF_OV) * 30).astype(np.float32)
        o = np.zeros(960 + HALF_OV, np.float32)
        o[:HALF_OV] = c0
        for b in range(B):
            ref.imdct(xs[b:], o[n2 * b: n2 * b + n2 + HALF_OV], s, B)
        gen[f"g{s}_x"] = xs
        gen[f"g{s}_carry"] = c0
        gen[f"g{s}_out"] = o
    np.savez(f"{OUT}/ref_imdct_strided.npz", X=X, carry0=carry0, syn=syn,
             XL=XL, carryL=cl, synL=synl, **gen)

    sh = {}
    for s in range(4):
        nfft = 480 >> s
        xi = rng.uniform(-1, 1, (8, 2 * nfft)).astype(np.float32)
        yo = np.stack([ref.ifft_shared(s, xi[r]) for r in range(8)])
        sh[f"x{s}"] = xi
        sh[f"y{s}"] = yo
    np.savez(f"{OUT}/ref_ifft_shared.npz", **sh)

    # one channel, 10 frames of 960 samples, kinds: L = one shift-0 block, S = 8 shift-3 blocks
    kinds = "LLSLSSLLSL"
    freq = decoder_like(rng, len(kinds), 960)
    mem = np.zeros(960 * len(kinds) + HALF_OV, np.float32)
    mem[:HALF_OV] = (rng.standard_normal(HALF_OV) * 30).astype(np.float32)
    carry_in = mem[:HALF_OV].copy()
    for f, k in enumerate(kinds):
        base = 960 * f
        if k == "L":
            ref.imdct(freq[f], mem[base: base + 960 + HALF_OV], 0, 1)
        else:
            for b in range(8):
                ref.imdct(freq[f, b:], mem[base + 120 * b: base + 120 * b + 180], 3, 8)
    np.savez(f"{OUT}/ref_chain.npz", kinds=np.array(list(kinds)), freq=freq,
             carry_in=carry_in, pcm=mem[: 960 * len(kinds)].copy(),
             tail=mem[960 * len(kinds):].copy())
    # compute_inv_mdcts over frame sequences (celt_decoder_clean.c:264-312): 2 stereo streams,
    # 11 frames, LM = 3, long frames through B1_C2 stride 1, transient frames through 8 x B1_C2
    # stride 8 -- exactly the reference's two fast paths -- into one decode_mem-like buffer.
    ns, nf, ch = 2, 11, 2
    freq = decoder_like(rng, ns * nf * ch, 960).reshape(ns, nf, ch, 960)
    transient = np.array([[0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0],
                          [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]], np.uint8)
    state = (rng.standard_normal((ns * ch, HALF_OV)) * 30).astype(np.float32)
    mem = np.zeros((ns, ch, nf * 960 + HALF_OV), np.float32)
    mem[:, :, :HALF_OV] = state.reshape(ns, ch, HALF_OV)
    for s_ in range(ns):
        for f in range(nf):
            X = np.ascontiguousarray(freq[s_, f])
            if transient[s_, f]:
                for b in range(8):
                    ref.imdct_c2(X[0, b:], X[1, b:], mem[s_, 0, f * 960 + 120 * b: f * 960 + 120 * b + 180],
                                 mem[s_, 1, f * 960 + 120 * b: f * 960 + 120 * b + 180], 3, 8)
            else:
                ref.imdct_c2(X[0], X[1], mem[s_, 0, f * 960: f * 960 + 1020], mem[s_, 1, f * 960: f * 960 + 1020], 0, 1)
    np.savez(f"{OUT}/ref_synth.npz", freq=freq, transient=transient, state_in=state,
             pcm=mem[:, :, : nf * 960].copy(), state_out=mem[:, :, nf * 960:].reshape(ns * ch, HALF_OV).copy())
    # REAL decoder data: the reference decoder itself (oracle/_ref/ref_capture = NyquistIO::Load built
    # from the reference's sources with the recording tap) run on the bundled test files.
    import struct
    import subprocess
    import tempfile

    def capture(name, max_frames):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "cap.bin")
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_capture"), f"{REFDATA}/{name}", out,
                            str(max_frames)], check=True, stdout=subprocess.DEVNULL)
            b = open(out, "rb").read()
            p = open(out + ".post", "rb").read()
        _, ch, frames, ncalls = struct.unpack("<4i", b[:16])
        (fsum,) = struct.unpack("<f", b[16:20])
        (nsamp,) = struct.unpack("<q", b[20:28])
        flags = np.frombuffer(b[28:28 + frames], np.uint8).copy()
        pay = np.frombuffer(b[28 + frames:], np.float32).reshape(frames, ch, 1980)
        # post-filter calls: per LM=3 frame and channel one N=120 call and one N=840 call
        ncomb, _ = struct.unpack("<2i", p[:8])
        off, comb = 8, []
        for _k in range(ncomb):
            iv = struct.unpack("<5i", p[off:off + 20])
            fv = struct.unpack("<2f", p[off + 20:off + 28])
            (hh,) = struct.unpack("<i", p[off + 28:off + 32])
            off += 32
            hist = None
            if hh:
                hist = np.frombuffer(p[off:off + 1088 * 4], np.float32).copy()
                off += 1088 * 4
            comb.append(dict(T0=iv[0], T1=iv[1], N=iv[2], ts0=iv[3], ts1=iv[4], g0=fv[0], g1=fv[1], hist=hist))
        (npcm,) = struct.unpack("<q", p[off:off + 8])
        final = np.frombuffer(p[off + 8:], np.float32).reshape(-1, ch).copy()
        raw = open(f"{REFDATA}/{name}", "rb").read()
        h = raw.find(b"OpusHead")
        _v, _c, preskip, _r, gain, _fam = struct.unpack("<BBHIhB", raw[h + 8:h + 19])
        assert gain == 0
        return dict(channels=ch, flags=flags, freq=pay[:, :, :960].copy(), out=pay[:, :, 960:].copy(),
                    sum=fsum, samples=nsamp, calls=ncalls, comb=comb, final=final, preskip=preskip)

    sh_ = capture("short.opus", 100000)
    lo, hi = 64, 128                                  # 64 frames incl. transient frames 73, 97, 123, 124
    rv = capture("sb-reverie.opus", 100000)
    # post-filter parameters of frame f = second call (N=840) of channel 0: (T1, g1, tapset1); the state
    # before frame lo = the arguments of its first call; filtered history and de-emphasis memory before
    # frame lo; final PCM = AudioData::samples (pre-skip removed by opusfile).
    cb, ch_ = sh_["comb"], sh_["channels"]
    first = lambda f, c: cb[(f * ch_ + c) * 2]
    second = lambda f, c: cb[(f * ch_ + c) * 2 + 1]
    pf_pitch = np.array([second(f, 0)["T1"] for f in range(lo, hi)], np.int32)
    pf_gain = np.array([second(f, 0)["g1"] for f in range(lo, hi)], np.float32)
    pf_tapset = np.array([second(f, 0)["ts1"] for f in range(lo, hi)], np.int32)
    a = first(lo, 0)
    z_ = first(hi, 0)
    pf_state_in = np.array([a["T0"], a["T1"], a["g0"], a["g1"], a["ts0"], a["ts1"]], np.float32)
    pf_state_out = np.array([z_["T0"], z_["T1"], z_["g0"], z_["g1"], z_["ts0"], z_["ts1"]], np.float32)
    hist_in = np.stack([first(lo, c)["hist"] for c in range(ch_)])
    ps = sh_["preskip"]
    final = sh_["final"][lo * 960 - ps: hi * 960 - ps].copy()           # [64*960][2]
    deemph_in = (np.float32(0.85000610) * (sh_["final"][lo * 960 - ps - 1] * np.float32(32768.0))).astype(np.float32)
    deemph_out = (np.float32(0.85000610) * (sh_["final"][hi * 960 - ps - 1] * np.float32(32768.0))).astype(np.float32)
    np.savez_compressed(
        f"{OUT}/real_opus_frames.npz",
        pf_pitch=pf_pitch[None], pf_gain=pf_gain[None], pf_tapset=pf_tapset[None], pf_state_in=pf_state_in[None],
        pf_state_out=pf_state_out[None], hist_in=hist_in[None], deemph_in=deemph_in, deemph_out=deemph_out,
        final=final[None], preskip=np.int64(ps),
        freq=sh_["freq"][lo:hi][None], transient=sh_["flags"][lo:hi][None],
        state_in=sh_["out"][lo - 1, :, 960:].copy(),              # raw tail of frame lo-1 = carry of frame lo
        pcm=sh_["out"][lo:hi, :, :960].transpose(1, 0, 2).reshape(1, 2, -1).copy(),
        state_out=sh_["out"][hi - 1, :, 960:].copy(),
        short_opus=np.array([sh_["samples"], sh_["calls"], len(sh_["flags"]), int(sh_["flags"].sum())], np.int64),
        short_opus_sum=np.float32(sh_["sum"]), short_opus_transient=sh_["flags"],
        sb_reverie=np.array([rv["samples"], rv["calls"], len(rv["flags"]), int(rv["flags"].sum())], np.int64),
        sb_reverie_sum=np.float32(rv["sum"]), sb_reverie_transient=rv["flags"],
        window=np.array([lo, hi]))
    # The bundled Opus test file itself (a data file the reference's tests hold) so that the host-side
    # decoder can be exercised where /root/reference does not exist, with per-frame digests of what the
    # reference decoder computed for EVERY frame of it: sum, sum of squares and peak of freq[] per
    # channel, the transient flag and the post-filter parameters.
    shutil.copyfile(f"{REFDATA}/short.opus", f"{OUT}/short.opus")
    fr = sh_["freq"].astype(np.float64)
    np.savez_compressed(
        f"{OUT}/short_opus_digest.npz",
        freq_sum=fr.sum(axis=2), freq_sq=(fr ** 2).sum(axis=2), freq_peak=np.abs(fr).max(axis=2),
        transient=sh_["flags"],
        pf_pitch=np.array([second(f, 0)["T1"] for f in range(len(sh_["flags"]))], np.int32),
        pf_gain=np.array([second(f, 0)["g1"] for f in range(len(sh_["flags"]))], np.float32),
        pf_tapset=np.array([second(f, 0)["ts1"] for f in range(len(sh_["flags"]))], np.int32),
        final=sh_["final"], preskip=np.int64(ps), samples=np.int64(sh_["samples"]))
    # A 4-channel MULTISTREAM file (channel mapping family 1: two coupled streams, mapping [2,0,3,1]) muxed
    # from the packets of short.opus by tools/oggopus.py (stream B = the same packets rotated by 57), decoded
    # by the reference decoder: ground truth for the multistream / channel-mapping path
    # (opus_multistream_decoder.c:184-331).  The file is rebuilt from short.opus by the tests; only digests
    # of the reference's output are stored.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oggopus
    pk, _gr, last = oggopus.read_packets(open(f"{REFDATA}/short.opus", "rb").read())
    aud = pk[2:]
    ms = oggopus.mux_family1([aud, aud[57:220] + aud[:57] + aud[220:]], 2, [2, 0, 3, 1], 312,
                             [(i + 1) * 960 for i in range(220)] + [last])
    with tempfile.TemporaryDirectory() as td:
        src, cap = os.path.join(td, "ms4.opus"), os.path.join(td, "cap.bin")
        open(src, "wb").write(ms)
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_capture"), src, cap, "0"], check=True, stdout=subprocess.DEVNULL)
        p = open(cap + ".post", "rb").read()
    ncomb, mch = struct.unpack("<2i", p[:8])
    off = 8
    for _k in range(ncomb):
        (hh,) = struct.unpack("<i", p[off + 28:off + 32])
        off += 32 + (1088 * 4 if hh else 0)
    mpcm = np.frombuffer(p[off + 8:], np.float32).reshape(-1, mch)
    assert mch == 4 and np.array_equal(mpcm[:, 1], sh_["final"][:, 0]) and np.array_equal(mpcm[:, 3], sh_["final"][:, 1])
    nb = mpcm.shape[0] // 960
    np.savez_compressed(f"{OUT}/multistream_digest.npz", samples=np.int64(mpcm.size), channels=np.int64(mch),
                        block_sum=mpcm[: nb * 960].astype(np.float64).reshape(nb, 960, mch).sum(axis=1),
                        block_sq=(mpcm[: nb * 960].astype(np.float64) ** 2).reshape(nb, 960, mch).sum(axis=1),
                        head=mpcm[:9600].copy(), tail=mpcm[-2000:].copy(), mapping=np.array([2, 0, 3, 1]), rotate=np.int64(57))
    # libvorbis mdct_backward (third_party/libvorbis/src/mdct.c compiled standalone): 3 rows per block size
    from oracle.pyoracle import VorbisRef
    vr = VorbisRef()
    vb = {}
    for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        xv = np.concatenate([rng.uniform(-1, 1, (1, n // 2)), rng.standard_normal((1, n // 2)) * 30,
                             np.eye(1, n // 2, 3) * 100.0]).astype(np.float32)
        vb[f"x{n}"] = xv
        vb[f"y{n}"] = vr.backward(n, xv)
    np.savez(f"{OUT}/ref_vorbis.npz", **vb)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} files, {total/1024:.0f} KiB -> {OUT}")


if __name__ == "__main__":
    main()
