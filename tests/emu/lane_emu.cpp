// tests/emu/lane_emu.cpp -- CPU replay of the HIP kernels' lane program.  TEST ONLY.
//
// Includes the same __host__ __device__ headers the gfx950 kernels are built from
// (libnyquist_amd/csrc/nyq_imdct_lanes.hpp) and executes them lane by lane, phase by
// phase, with plain arrays standing in for the wave's LDS slice and tail ring.  The phase
// order mirrors nyq_kernels.hpp (run_group, synth_long_kernel, synth_short_kernel,
// synth_fixup_kernel).  It exists so the index maps (prime-factor slots, stage tasks, TDAC
// placement, in-wave chaining) can be checked against the oracle in the CPU-only test tier;
// it is not reachable from the product API.
#include <cstring>
#include <vector>

#include "nyq_imdct_lanes.hpp"
#include "nyq_vorbis_lanes.hpp"

using namespace nyq;

template <int N2R>
struct WaveEmu {
    using Gm = Geo<N2R>;
    std::vector<cpx> lds;
    std::vector<float> ring;
    std::vector<LaneConst<N2R>> K;
    WaveEmu(const float *trig, const float *window)
        : lds(Gm::LDS_CPX, cpx{0, 0}), ring(Gm::RING_FLOATS + 4, 0.f), K(kWave) {
        for (int l = 0; l < kWave; l++) lane_init<N2R>(K[l], l, trig, window);
    }
    void fft() {
        for (int l = 0; l < kWave; l++) pass1<N2R>(l, lds.data());
        for (int it = 0; it < Gm::P2_ITERS; it++) {
            cpx v[kWave][15];
            int g[kWave], n2[kWave];
            bool ok[kWave];
            for (int l = 0; l < kWave; l++) ok[l] = pass2_load<N2R>(l, it, lds.data(), v[l], g[l], n2[l]);
            for (int l = 0; l < kWave; l++)
                if (ok[l]) pass2_store<N2R>(g[l], n2[l], lds.data(), v[l]);
        }
    }
    // nyq_kernels.hpp: stage_in_load + run_group
    template <class Rows>
    void group(const Rows &rows) {
        std::vector<StageRegs<N2R>> R(kWave);
        std::vector<HeadRegs<N2R>> H(kWave);
        for (int l = 0; l < kWave; l++) stage_in_load<N2R, 0>(R[l], l, rows);
        for (int l = 0; l < kWave; l++) stage_in_store<N2R>(R[l], K[l], l, lds.data());
        fft();
        for (int l = 0; l < kWave; l++) stage_out<N2R, 0>(K[l], l, lds.data(), ring.data(), rows, H[l]);
        if (Rows::CHAINS)
            for (int l = 0; l < kWave; l++) stage_out_heads<N2R, 0>(K[l], l, ring.data(), rows, H[l]);
    }
    void rotate() {
        for (int l = 0; l < kWave; l++) ring_rotate<N2R>(l, ring.data());
    }
};

template <int N2R>
static void emu_imdct(const float *in, const float *carry, float *fin, float *tail, long nrows,
                      const float *trig, const float *window) {
    WaveEmu<N2R> W(trig, window);
    for (long row0 = 0; row0 < nrows; row0 += Geo<N2R>::G) {
        IndepRows<N2R> rows{in, carry, fin, tail, row0, nrows};
        W.group(rows);
    }
}

template <int N2R>
static void emu_ifft(const float *in, float *out, long nrows) {
    using Gm = Geo<N2R>;
    std::vector<float> dummy_t(481, 1.f), dummy_w(120, 1.f);
    WaveEmu<N2R> W(dummy_t.data(), dummy_w.data());
    for (long row0 = 0; row0 < nrows; row0 += Gm::G) {
        for (int l = 0; l < kWave; l++) ifft_stage_in<N2R>(l, in, W.lds.data(), row0, nrows);
        W.fft();
        for (int l = 0; l < kWave; l++) ifft_stage_out<N2R>(l, W.lds.data(), out, row0, nrows);
    }
    (void)sizeof(Gm);
}

template <int N2R>
static void emu_synth_long(const SynthArgs &A, const float *trig, const float *window) {
    WaveEmu<N2R> W(trig, window);
    const long nchunks = A.nstreams * A.channels * FrameLongRows<N2R>::chunks_per_channel(A.nframes);
    for (long ci = 0; ci < nchunks; ci++)
        for (int qq = 0; qq < kChainGroups; qq++) {
            FrameLongRows<N2R> rows(A, ci, qq);
            if (!rows.any()) continue;
            W.group(rows);
            W.rotate();
        }
}

// nyq_kernels.hpp synth_short_kernel<LMc>: 64 units per scan step, the transient ones Rows::K at a time as one group
template <int LMc>
static void emu_synth_short(const SynthArgs &A, const float *trig, const float *window) {
    using Rows = FrameShortPacked<LMc>;
    WaveEmu<4> W(trig, window);
    const long units = A.nstreams * A.channels * A.nframes;
    for (long base = 0; base < units; base += kWave) {
        std::vector<ShortUnit> hits;
        for (long u = base; u < units && u < base + kWave; u++) {
            const long sc = u / A.nframes, f = u - sc * A.nframes, s = sc / A.channels, c = sc - s * A.channels;
            if (A.transient[s * A.fs() + f]) hits.push_back(short_unit<LMc>(A, s, c, f));
        }
        for (size_t h0 = 0; h0 < hits.size(); h0 += Rows::K) {
            ShortUnit tab[Rows::K];
            int n = 0;
            for (int k = 0; k < Rows::K; k++) {
                tab[k] = hits[h0];
                if (h0 + k < hits.size()) { tab[k] = hits[h0 + k]; n = k + 1; }
            }
            Rows rows(tab, n);
            W.group(rows);
        }
    }
}

static void emu_synth_fixup(const SynthArgs &A, int N, int chain_frames, const float *window) {
    const long units = A.nstreams * A.channels * A.nframes;
    for (long u = 0; u < units; u++) {
        const long sc = u / A.nframes, f = u - sc * A.nframes, s = sc / A.channels;
        if (head_done_in_wave(A.transient ? A.transient + s * A.nframes : nullptr, f, chain_frames)) continue;
        for (int i = 0; i < kHalfOv; i++) {
            const float cv = f == 0 ? (A.state_in ? A.state_in[sc * (long)kHalfOv + i] : 0.f)
                                    : A.tails[(sc * (A.nframes + 1) + f) * (long)kHalfOv + i];
            float *o = A.pcm + (sc * A.nframes + f) * (long)N;
            o[i] += window[kOverlap - 1 - i] * cv;
            o[kOverlap - 1 - i] += window[i] * cv;
            if (f == 0 && A.state_out) A.state_out[sc * (long)kHalfOv + i] = A.tails[(sc * (A.nframes + 1) + A.nframes) * (long)kHalfOv + i];
        }
    }
}

extern "C" int emu_imdct_batch(int shift, const float *in, const float *carry, float *fin, float *tail,
                               long nrows, const float *trig, const float *window) {
    switch (shift) {
    case 0: emu_imdct<32>(in, carry, fin, tail, nrows, trig, window); return 0;
    case 1: emu_imdct<16>(in, carry, fin, tail, nrows, trig, window); return 0;
    case 2: emu_imdct<8>(in, carry, fin, tail, nrows, trig, window); return 0;
    case 3: emu_imdct<4>(in, carry, fin, tail, nrows, trig, window); return 0;
    }
    return -1;
}

extern "C" int emu_ifft_batch(int nfft, const float *in, float *out, long nrows) {
    switch (nfft) {
    case 480: emu_ifft<32>(in, out, nrows); return 0;
    case 240: emu_ifft<16>(in, out, nrows); return 0;
    case 120: emu_ifft<8>(in, out, nrows); return 0;
    case 60: emu_ifft<4>(in, out, nrows); return 0;
    }
    return -1;
}

// mirrors nyq_celt_synth_dev (nyq_imdct.hip): long, short, fix-up (which takes the state in and hands the state out)
extern "C" int emu_celt_synth(int LM, const float *freq, const unsigned char *transient, float *pcm, float *state,
                              long nstreams, long nframes, int channels, const float *trig, const float *window) {
    if (LM < 0 || LM > 3) return -1;
    const long nsc = nstreams * channels;
    std::vector<float> tails((size_t)nsc * (nframes + 1) * kHalfOv, 0.f);
    SynthArgs A{freq, LM > 0 ? transient : nullptr, pcm, tails.data(), nstreams, nframes, channels, state, state};
    int chain_frames;
    switch (LM) {
    case 3: emu_synth_long<32>(A, trig, window); chain_frames = Geo<32>::CHAIN_FRAMES; break;
    case 2: emu_synth_long<16>(A, trig, window); chain_frames = Geo<16>::CHAIN_FRAMES; break;
    case 1: emu_synth_long<8>(A, trig, window); chain_frames = Geo<8>::CHAIN_FRAMES; break;
    default: emu_synth_long<4>(A, trig, window); chain_frames = Geo<4>::CHAIN_FRAMES; break;
    }
    if (A.transient) {
        if (LM == 3) emu_synth_short<3>(A, trig, window);
        else if (LM == 2) emu_synth_short<2>(A, trig, window);
        else emu_synth_short<1>(A, trig, window);
    }
    emu_synth_fixup(A, 120 << LM, chain_frames, window);
    return 0;
}

// mirrors vorbis_imdct_kernel (nyq_kernels.hpp)
template <int LOGN4>
static void emu_vorbis(const float *in, float *out, long nrows, const float *rot, const float *twid) {
    using V = VGeo<LOGN4>;
    std::vector<cpx> lds(V::LDS_CPX, cpx{0, 0});
    VTables T{rot, twid};
    std::vector<VTwid<LOGN4>> W(kWave);
    for (int l = 0; l < kWave; l++) v_twid_init<LOGN4>(W[l], l, T);
    std::vector<VStage<LOGN4>> R(kWave);
    for (long row0 = 0; row0 < nrows; row0 += V::G) {
        for (int l = 0; l < kWave; l++) v_stage_in_load<LOGN4>(R[l], l, in, row0, nrows);
        for (int l = 0; l < kWave; l++) v_stage_in_store<LOGN4>(R[l], l, lds.data(), T);
        for (int it = 0; it < V::P1_ITERS; it++) {
            std::vector<std::vector<cpx>> u(kWave, std::vector<cpx>(V::R2));
            int g[kWave], k1[kWave];
            bool ok[kWave];
            for (int l = 0; l < kWave; l++) {
                cpx tmp[V::R2];
                ok[l] = v_pass1_load<LOGN4>(l, it, lds.data(), tmp, g[l], k1[l]);
                for (int k = 0; k < V::R2; k++) u[l][k] = tmp[k];
            }
            for (int l = 0; l < kWave; l++)
                if (ok[l]) {
                    cpx tmp[V::R2];
                    for (int k = 0; k < V::R2; k++) tmp[k] = u[l][k];
                    v_pass1_store<LOGN4>(g[l], k1[l], lds.data(), tmp);
                }
        }
        for (int it = 0; it < V::P2_ITERS; it++) {
            std::vector<std::vector<cpx>> v(kWave, std::vector<cpx>(V::R1));
            int g[kWave], n2[kWave];
            bool ok[kWave];
            for (int l = 0; l < kWave; l++) {
                cpx tmp[V::R1];
                ok[l] = v_pass2_load<LOGN4>(l, it, lds.data(), W[l], tmp, g[l], n2[l]);
                for (int k = 0; k < V::R1; k++) v[l][k] = tmp[k];
            }
            for (int l = 0; l < kWave; l++)
                if (ok[l]) {
                    cpx tmp[V::R1];
                    for (int k = 0; k < V::R1; k++) tmp[k] = v[l][k];
                    v_pass2_store<LOGN4>(g[l], n2[l], lds.data(), tmp);
                }
        }
        for (int l = 0; l < kWave; l++) v_stage_out<LOGN4>(l, lds.data(), out, row0, nrows, T);
    }
}

// rot: (cos, sin)(2 pi (i + 1/8)/n), i < n/4;  twid: (cos, sin)(2 pi k/(n/4)), k < n/4 -- as nyq_imdct.hip builds them
extern "C" int emu_vorbis_imdct(int n, const float *in, float *out, long nrows, const float *rot, const float *twid) {
    switch (n) {
    case 64: emu_vorbis<4>(in, out, nrows, rot, twid); return 0;
    case 128: emu_vorbis<5>(in, out, nrows, rot, twid); return 0;
    case 256: emu_vorbis<6>(in, out, nrows, rot, twid); return 0;
    case 512: emu_vorbis<7>(in, out, nrows, rot, twid); return 0;
    case 1024: emu_vorbis<8>(in, out, nrows, rot, twid); return 0;
    case 2048: emu_vorbis<9>(in, out, nrows, rot, twid); return 0;
    case 4096: emu_vorbis<10>(in, out, nrows, rot, twid); return 0;
    case 8192: emu_vorbis<11>(in, out, nrows, rot, twid); return 0;
    }
    return -1;
}

// single in-register DFT, for unit-testing nyq_fft_core.hpp
extern "C" int emu_dft(int r, float *io) {
    switch (r) {
#define CASE(R) case R: { cpx v[R]; std::memcpy(v, io, sizeof v); Dft<R>::run(v); std::memcpy(io, v, sizeof v); return 0; }
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(8) CASE(15) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
    return -1;
}

// ---- the transform wave of the one-launch frames -> PCM kernel (nyq_fuse_lanes.hpp, celt_chain_kernel's XF role) ----
#include "nyq_fuse_lanes.hpp"

// One stereo stream: freq [nframes][2][960], transient [nframes] or null, state [2][60] in/out or null,
// pcm [2][nframes * 960].  Phase order as in nyq_chain_kernel.hpp.
extern "C" int emu_fuse_synth(const float *freq, const unsigned char *transient, float *pcm, float *state, long nframes,
                              const float *trig, const float *window) {
    using namespace nyq::fx;
    std::vector<float> r0(kN, 0.f), r1(kN, 0.f);
    float *const reg[2] = {r0.data(), r1.data()};
    std::vector<XfRot> K(kWave);
    std::vector<XfTw> W(kWave);
    std::vector<XfWin> Wn(kWave);
    std::vector<XfShortConst> S(kWave);
    std::vector<f4> tail[2] = {std::vector<f4>(kWave, f4{0, 0, 0, 0}), std::vector<f4>(kWave, f4{0, 0, 0, 0})};
    for (int l = 0; l < kWave; l++) {
        xf_init_rot(K[l], l, trig);
        xf_init_tw(W[l], l);
        xf_init_win(Wn[l], l, window);
        xf_short_init(S[l], l, trig);
        for (int c = 0; c < 2; c++)
            if (state && tail_lane(l)) tail[c][l] = *reinterpret_cast<const f4 *>(state + tail_offset(l, c));
    }
    std::vector<XfRegs> R(kWave);
    for (long f = 0; f < nframes; f++) {
        const float *frame = freq + f * 2 * kN;
        for (int l = 0; l < kWave; l++) xf_load<0>(R[l], l, frame);
        if (!(transient && transient[f])) {
            for (int r = 0; r < 2; r++)
                for (int l = 0; l < kWave; l++) xf_long_s0(R[l], K[l], W[l], l, r, reg[r]);
            {
                cpx u[kWave][16];
                for (int l = 0; l < kWave; l++) xf_long_s2_load(l, reg, u[l]);
                for (int l = 0; l < kWave; l++) xf_long_s2_store(l, reg, u[l]);
            }
            {
                cpx v[kWave][15];
                for (int l = 0; l < kWave; l++) xf_long_s3_load(l, reg, v[l]);
                for (int l = 0; l < kWave; l++) xf_long_s3_store(l, reg, v[l]);
            }
            for (int r = 0; r < 2; r++) {
                std::vector<XfOut> O(kWave);
                for (int l = 0; l < kWave; l++) xf_long_s4_load(K[l], l, reg[r], O[l]);
                for (int l = 0; l < kWave; l++) xf_long_s4_store(Wn[l], l, reg[r], O[l], tail[r][l]);
            }
        } else {
            for (int l = 0; l < kWave; l++) xf_short_t0(R[l], l, reg);
            for (int s = 0; s < 4; s++) {
                std::vector<XfShortIn> I(kWave);
                for (int l = 0; l < kWave; l++) xf_short_t1_load(l, s, reg, I[l]);
                for (int l = 0; l < kWave; l++) xf_short_t1_store(S[l], l, s, reg, I[l]);
            }
            for (int it = 0; it < 4; it++)
                for (int l = 0; l < kWave; l++) xf_short_t2(l, it, reg);
            {
                cpx v[kWave][15];
                for (int l = 0; l < kWave; l++) xf_short_t3_load(l, reg, v[l]);
                for (int l = 0; l < kWave; l++) xf_short_t3_store(l, reg, v[l]);
            }
            std::vector<f4> bk(4 * kWave);
            for (int s = 0; s < 4; s++)
                for (int l = 0; l < kWave; l++) bk[s * kWave + l] = xf_short_t4(S[l], l, s, reg);
            for (int c = 0; c < 2; c++)
                for (int h = 1; h >= 0; h--) {
                    std::vector<XfMirror> M(kWave);
                    for (int l = 0; l < kWave; l++) xf_short_t5_load(l, c, h, reg, tail[c][l], M[l]);
                    for (int l = 0; l < kWave; l++) xf_short_t5_store(Wn[l], l, c, h, reg, M[l]);
                }
            for (int c = 0; c < 2; c++)
                for (int l = 0; l < kWave; l++)
                    if (tail_lane(l)) tail[c][l] = bk[(2 * c + 1) * kWave + short_tail_src(l)];
        }
        for (int c = 0; c < 2; c++) std::memcpy(pcm + (c * nframes + f) * kN, reg[c], kN * sizeof(float));
    }
    if (state)
        for (int c = 0; c < 2; c++)
            for (int l = 0; l < kWave; l++)
                if (tail_lane(l)) *reinterpret_cast<f4 *>(state + tail_offset(l, c)) = tail[c][l];
    return 0;
}
