// tests/emu/lane_emu.cpp -- CPU replay of the HIP kernel's lane program.  TEST ONLY.
//
// Includes the same __host__ __device__ headers the gfx950 kernels are built from
// (libnyquist_amd/csrc/nyq_imdct_lanes.hpp) and executes them lane by lane, phase by
// phase, with a plain array standing in for the wave's LDS slice.  It exists so the
// index maps (prime-factor slots, stage tasks, TDAC placement) can be checked against
// the oracle in the CPU-only test tier; it is not reachable from the product API.
#include <cstring>
#include <vector>

#include "nyq_imdct_lanes.hpp"

using namespace nyq;

template <int N2R>
static void emu_imdct(const float *in, const float *carry, float *fin, float *tail, long nrows,
                      const float *trig, const float *window) {
    using Gm = Geo<N2R>;
    std::vector<cpx> lds(Gm::LDS_CPX, cpx{0, 0});
    std::vector<LaneConst<N2R>> K(kWave);
    for (int l = 0; l < kWave; l++) lane_init<N2R>(K[l], l, trig, window);
    std::vector<StageRegs<N2R>> R(kWave);
    for (long row0 = 0; row0 < nrows; row0 += kGroup) {
        for (int l = 0; l < kWave; l++) stage_in_load<N2R>(R[l], l, in, row0, nrows);
        for (int l = 0; l < kWave; l++) stage_in_store<N2R>(R[l], K[l], l, lds.data());
        for (int l = 0; l < kWave; l++) pass1<N2R>(l, lds.data());
        for (int it = 0; it < Gm::P2_ITERS; it++) {
            cpx v[kWave][15];
            int g[kWave], n2[kWave];
            bool ok[kWave];
            for (int l = 0; l < kWave; l++) ok[l] = pass2_load<N2R>(l, it, lds.data(), v[l], g[l], n2[l]);
            for (int l = 0; l < kWave; l++)
                if (ok[l]) pass2_store<N2R>(g[l], n2[l], lds.data(), v[l]);
        }
        for (int l = 0; l < kWave; l++) stage_out<N2R>(K[l], l, lds.data(), carry, fin, tail, row0, nrows);
    }
}

template <int N2R>
static void emu_ifft(const float *in, float *out, long nrows) {
    using Gm = Geo<N2R>;
    std::vector<cpx> lds(Gm::LDS_CPX, cpx{0, 0});
    for (long row0 = 0; row0 < nrows; row0 += kGroup) {
        for (int l = 0; l < kWave; l++) ifft_stage_in<N2R>(l, in, lds.data(), row0, nrows);
        for (int l = 0; l < kWave; l++) pass1<N2R>(l, lds.data());
        for (int it = 0; it < Gm::P2_ITERS; it++) {
            cpx v[kWave][15];
            int g[kWave], n2[kWave];
            bool ok[kWave];
            for (int l = 0; l < kWave; l++) ok[l] = pass2_load<N2R>(l, it, lds.data(), v[l], g[l], n2[l]);
            for (int l = 0; l < kWave; l++)
                if (ok[l]) pass2_store<N2R>(g[l], n2[l], lds.data(), v[l]);
        }
        for (int l = 0; l < kWave; l++) ifft_stage_out<N2R>(l, lds.data(), out, row0, nrows);
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

// single in-register DFT, for unit-testing nyq_fft_core.hpp
extern "C" int emu_dft(int r, float *io) {
    switch (r) {
#define CASE(R) case R: { cpx v[R]; std::memcpy(v, io, sizeof v); Dft<R>::run(v); std::memcpy(io, v, sizeof v); return 0; }
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(8) CASE(15) CASE(16) CASE(32)
#undef CASE
    }
    return -1;
}
