// nyq_kernels.hpp -- the gfx950 __global__ kernels of the batched CELT IMDCT.
// Included by nyq_imdct.hip (the product library) and by tools/kbench.hip (the tuning
// harness that times kernel configurations against each other in one process).
//
// Kernels (all wave-autonomous: 64 lanes own 4 rows, meet only in their LDS slice):
//   imdct_rows_kernel<N2R,Cfg>  clt_mdct_backward  (mdct.c:267-379)  on independent rows
//   ifft_rows_kernel<N2R,WPB>   opus_ifft          (kiss_fft.c:696-747)
//   synth_frames_kernel<N2R,LMc,Cfg>  compute_inv_mdcts (celt_decoder_clean.c:264-312): workgroups [0, nlong) the long
//                               frames (chunks of 4 G frames chained in-wave), the others the interleaved short blocks
//                               of the transient frames
//   synth_fixup_kernel          adds the carry terms of the TDAC mirror (mdct.c:371-372) to the
//                               few heads that could not be chained in-wave
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_imdct_lanes.hpp"

namespace nyq {

// Lanes of one wave exchange data through LDS without a workgroup barrier.  The LDS
// unit executes one wave's accesses in issue order, so only the COMPILER must be told
// not to move LDS accesses across a phase boundary.
#define NYQ_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// kernel configuration (compile time)
//   WPB       waves per workgroup (each wave is autonomous; WPB only sets the LDS granule)
//   PREFETCH  issue the next group's global loads right after pre-rotation of the current one
//   NT        bit 0: non-temporal loads, bit 1: non-temporal stores
template <int WPB_, bool PREFETCH_, int NT_, int MAP_ = 0>
struct KCfg {
    // which groups a wave takes: 0 grid-stride over all waves (neighbouring groups on neighbouring XCDs),
    // 1 one contiguous run per wave, 2 one contiguous eighth of the batch per XCD (workgroups are dealt
    // round-robin to the 8 XCDs), grid-stride inside it.  Measured with tools/kbench: see DESIGN.md 4.1.
    static constexpr int MAP = MAP_;
    static constexpr bool CHUNKED = MAP_ == 1;
    static constexpr int WPB = WPB_;
    static constexpr bool PREFETCH = PREFETCH_;
    static constexpr int NT_LD = NT_ & 1;
    static constexpr int NT_ST = (NT_ >> 1) & 1;
};

#ifndef NYQ_DEFAULT_WPB
#define NYQ_DEFAULT_WPB 1
#endif
// Waves kept resident per CU.  Measured on MI355X with tools/kbench (2^20 nfft-480 rows,
// interleaved A/B): 3 -> 2.02 ms, 4 -> 1.67, 5 -> 1.52, 6 -> 1.49, 7 -> 1.57, 8 -> 1.59,
// 10 (the LDS limit) -> 2.05.  Six autonomous waves already keep ~90 KB of loads in flight per
// CU; beyond that the memory system loses efficiency (plain copies on the same box show the
// same shape), so the persistent grid is capped here rather than at the occupancy limit.
#ifndef NYQ_WAVES_PER_CU
#define NYQ_WAVES_PER_CU 6
#endif
using DefaultCfg = KCfg<NYQ_DEFAULT_WPB, false, 0>;

template <int N2R>
__device__ __forceinline__ void fft_passes(int lane, cpx *lds) {
    using Gm = Geo<N2R>;
    pass1<N2R>(lane, lds);
    NYQ_WAVE_SYNC();
#pragma unroll
    for (int it = 0; it < Gm::P2_ITERS; it++) {
        cpx v[15];
        int g, n2;
        bool ok = pass2_load<N2R>(lane, it, lds, v, g, n2);
        NYQ_WAVE_SYNC();
        if (ok) pass2_store<N2R>(g, n2, lds, v);
    }
    NYQ_WAVE_SYNC();
}

// One group through phases A..D.  `R` holds the group's input registers (already loaded).
// the value, but not a loop invariant or a common subexpression to the optimiser
__device__ __forceinline__ int relane(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int N2R, typename Cfg, class Rows>
__device__ __forceinline__ void run_group(const LaneConst<N2R> &K, int lane, cpx *lds, float *ring,
                                          const Rows &rows, StageRegs<N2R> &R) {
    NYQ_WAVE_SYNC();
    stage_in_store<N2R>(R, K, lane, lds);
    NYQ_WAVE_SYNC();
    fft_passes<N2R>(lane, lds);
    HeadRegs<N2R> H;
    if constexpr (Rows::CHAINS && N2R == 8) {
        // Chained 240-sample frame rows: the two output phases derive their per-lane row indices, masks and 64-bit
        // addresses from a FRESH copy of the lane id.  Otherwise they are computed once per group, shared between the
        // phases and kept alive across the FFT passes: synth_long_kernel<8> needed 260 registers that way -- ONE wave per
        // SIMD, four resident waves per CU instead of six -- against 111 of imdct_rows_kernel<8>; now 156.  (The other
        // sizes stay below two waves' worth either way and lose 3-8 % to the recomputation: left as they are.)
        stage_out<N2R, Cfg::NT_ST>(K, relane(lane), lds, ring, rows, H);
        NYQ_WAVE_SYNC();
        stage_out_heads<N2R, Cfg::NT_ST>(K, relane(lane), ring, rows, H);
    } else if constexpr (Rows::CHAINS) {
        stage_out<N2R, Cfg::NT_ST>(K, lane, lds, ring, rows, H);
        NYQ_WAVE_SYNC();
        stage_out_heads<N2R, Cfg::NT_ST>(K, lane, ring, rows, H);
    } else {
        stage_out<N2R, Cfg::NT_ST>(K, lane, lds, ring, rows, H);
    }
}

// per-wave LDS carve: rows slice, then (chained kernels only) the (G + 1) x 60-float tail ring
template <int N2R, int WPB, bool RING>
struct WaveLds {
    static constexpr int ROW_FLOATS = 2 * Geo<N2R>::LDS_CPX;
    static constexpr int PER_WAVE = ROW_FLOATS + (RING ? Geo<N2R>::RING_FLOATS + 4 : 0);   // keep 16-B multiples
    static constexpr int TOTAL = WPB * PER_WAVE;
    static_assert(PER_WAVE % 4 == 0, "LDS slices must stay 16-byte aligned");
};

template <int N2R, typename Cfg>
__global__ __launch_bounds__(kWave *Cfg::WPB) void imdct_rows_kernel(
    const float *__restrict__ in, const float *__restrict__ carry, float *__restrict__ fin,
    float *__restrict__ tail, long nrows, const float *__restrict__ trig,
    const float *__restrict__ window) {
    using Gm = Geo<N2R>;
    using WL = WaveLds<N2R, Cfg::WPB, false>;
    __shared__ __attribute__((aligned(16))) float smem[WL::TOTAL];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = reinterpret_cast<cpx *>(smem + wv * WL::PER_WAVE);

    LaneConst<N2R> K;
    lane_init<N2R>(K, lane, trig, window);

    const long ngroups_all = (nrows + Gm::G - 1) / Gm::G;
    const long nwaves_all = (long)gridDim.x * Cfg::WPB;
    const long wid = (long)blockIdx.x * Cfg::WPB + wv;
    // grid-stride: wave w takes groups w, w+W, ...; chunked: wave w takes one contiguous run
    const long per = (ngroups_all + nwaves_all - 1) / nwaves_all;
    long nwaves = Cfg::CHUNKED ? 1 : nwaves_all;
    long gi = Cfg::CHUNKED ? wid * per : wid;
    long ngroups = Cfg::CHUNKED ? (gi + per < ngroups_all ? gi + per : ngroups_all) : ngroups_all;
    if constexpr (Cfg::MAP == 2) {
        const long xcd = blockIdx.x & 7, bx = blockIdx.x >> 3;
        const long blocks_here = ((long)gridDim.x + 7 - xcd) / 8;
        const long per_x = (ngroups_all + 7) / 8;
        gi = xcd * per_x + bx * Cfg::WPB + wv;
        nwaves = blocks_here * Cfg::WPB;
        ngroups = (xcd + 1) * per_x < ngroups_all ? (xcd + 1) * per_x : ngroups_all;
    }
    StageRegs<N2R> R;
    if constexpr (Cfg::PREFETCH) {
        // Software pipeline over this wave's groups: the float4 loads of group g+1 are issued
        // as soon as group g's registers have been pre-rotated into LDS.  (Measured: no gain.)
        if (gi < ngroups) {
            IndepRows<N2R> rows{in, carry, fin, tail, gi * Gm::G, nrows};
            stage_in_load<N2R, Cfg::NT_LD>(R, lane, rows);
        }
    }
    for (; gi < ngroups; gi += nwaves) {
        IndepRows<N2R> rows{in, carry, fin, tail, gi * Gm::G, nrows};
        if constexpr (!Cfg::PREFETCH) {
            stage_in_load<N2R, Cfg::NT_LD>(R, lane, rows);
            run_group<N2R, Cfg>(K, lane, lds, nullptr, rows, R);
        } else {
            NYQ_WAVE_SYNC();
            stage_in_store<N2R>(R, K, lane, lds);
            if (gi + nwaves < ngroups) {
                IndepRows<N2R> nxt{in, carry, fin, tail, (gi + nwaves) * Gm::G, nrows};
                stage_in_load<N2R, Cfg::NT_LD>(R, lane, nxt);
            }
            NYQ_WAVE_SYNC();
            fft_passes<N2R>(lane, lds);
            HeadRegs<N2R> H;
            stage_out<N2R, Cfg::NT_ST>(K, lane, lds, nullptr, rows, H);
        }
    }
}

// ---- frame sequences: the compute_inv_mdcts replacement (celt_decoder_clean.c:264-312) ----
// ONE launch, two roles by workgroup index (round 3): workgroups [0, nlong) run the long frames, the rest the transient
// frames.  (Rounds 1-2 launched the transient-frame kernel on a side stream between two events: whether it then ran BESIDE
// the long-frame kernel or behind it depended on which hardware queue the runtime had given that stream -- identical
// builds of the chain differed by 0.1 ms, 6 %, inside one process.  One dispatch has no such lottery, and two events and
// a launch less.)  LMc = log2 of the short blocks per transient frame; 0 = the call has no transient frames.

// Long frames: chunks of 4 G consecutive frames of one (stream, channel), chained in-wave.
template <int N2R, typename Cfg>
__device__ __forceinline__ void synth_long_role(const SynthArgs &A, const float *__restrict__ trig, const float *__restrict__ window,
                                                float *smem, long bid, long nblocks) {
    using WL = WaveLds<N2R, Cfg::WPB, true>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = reinterpret_cast<cpx *>(smem + wv * WL::PER_WAVE);
    float *ring = smem + wv * WL::PER_WAVE + WL::ROW_FLOATS;

    LaneConst<N2R> K;
    lane_init<N2R>(K, lane, trig, window);

    const long nchunks = A.nstreams * A.channels * FrameLongRows<N2R>::chunks_per_channel(A.nframes);
    const long nwaves = nblocks * Cfg::WPB;
    // The transient flags of a chunk in one go -- lane i holds frame (chunk start - 1 + i), a ballot makes the mask --
    // and one chunk AHEAD: the flag bytes of the next chunk are on their way while this one is transformed.
    auto flag_of = [&](long ci) {
        bool tr = false;
        if (A.transient && ci < nchunks) {
            const long cpc = FrameLongRows<N2R>::chunks_per_channel(A.nframes);
            const long sc = ci / cpc, k = ci - sc * cpc;
            const long f = k * FrameLongRows<N2R>::kChainFrames - 1 + lane;
            if (lane <= FrameLongRows<N2R>::kChainFrames + 1 && f >= 0 && f < A.nframes) tr = A.transient[(sc / A.channels) * A.fs() + f] != 0;
        }
        return tr;
    };
    bool tr_next = flag_of(bid * Cfg::WPB + wv);
    for (long ci = bid * Cfg::WPB + wv; ci < nchunks; ci += nwaves) {
        const unsigned long long tmask = __ballot(tr_next);
        tr_next = flag_of(ci + nwaves);
#pragma unroll 1
        for (int qq = 0; qq < kChainGroups; qq++) {
            FrameLongRows<N2R> rows(A, ci, qq, tmask);
            if (!rows.any()) continue;   // four transient (or out-of-range) frames: nothing to do here
            StageRegs<N2R> R;
            stage_in_load<N2R, Cfg::NT_LD>(R, lane, rows);
            run_group<N2R, Cfg>(K, lane, lds, ring, rows, R);
            NYQ_WAVE_SYNC();
            ring_rotate<N2R>(lane, ring);
        }
    }
}

// Transient frames: a wave scans 64 (stream * channel, frame) units at a time and runs the short blocks of up to
// 16 / B of the transient ones among them as ONE group of 16 rows (FrameShortPacked: unit k owns rows k B ..); blocks of a
// unit chain through the tail ring, its last block publishes the frame's tail.  LMc = log2 B.
template <int LMc, typename Cfg>
__device__ __forceinline__ void synth_short_role(const SynthArgs &A, const float *__restrict__ trig, const float *__restrict__ window,
                                                 float *smem, ShortUnit *utab_all, long bid, long nblocks) {
    using WL = WaveLds<4, Cfg::WPB, true>;
    using Rows = FrameShortPacked<LMc>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = reinterpret_cast<cpx *>(smem + wv * WL::PER_WAVE);
    float *ring = smem + wv * WL::PER_WAVE + WL::ROW_FLOATS;
    ShortUnit *utab = utab_all + wv * Rows::K;

    LaneConst<4> K;
    lane_init<4>(K, lane, trig, window);

    // Transient frames are rare (a few per cent), so the scan is vectorised: the wave looks at 64
    // consecutive (stream*channel, frame) units at once -- one flag byte per lane, one ballot --
    // and then takes the set bits Rows::K at a time: the lane that owns the r-th of them writes row r of the unit table
    // from the indices it computed for the scan.
    const long units = A.nstreams * A.channels * A.nframes;
    const long nwaves = nblocks * Cfg::WPB;
    for (long base = (bid * Cfg::WPB + wv) * kWave; base < units; base += nwaves * kWave) {
        const long mine = base + lane;
        bool hit = false;
        long s = 0, c = 0, f = 0;
        if (mine < units) {
            const long sc = mine / A.nframes;
            f = mine - sc * A.nframes;
            s = sc / A.channels;
            c = sc - s * A.channels;
            hit = A.transient[s * A.fs() + f] != 0;
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int rank = __builtin_popcountll(todo & ((1ull << lane) - 1ull));
            NYQ_WAVE_SYNC();                                   // (the previous group has finished reading the table)
            if (((todo >> lane) & 1ull) && rank < Rows::K) utab[rank] = short_unit<LMc>(A, s, c, f);
            const int avail = __builtin_popcountll(todo);
            const int n = avail < Rows::K ? avail : Rows::K;
#pragma unroll
            for (int k = 0; k < Rows::K; k++) todo &= todo - 1;   // (clearing a bit of zero leaves zero)
            NYQ_WAVE_SYNC();
            Rows rows(utab, n);
            StageRegs<4> R;
            stage_in_load<4, 0>(R, lane, rows);
            run_group<4, Cfg>(K, lane, lds, ring, rows, R);
        }
    }
}

template <int N2R, int LMc, typename Cfg>
__global__ __launch_bounds__(kWave *Cfg::WPB) void synth_frames_kernel(SynthArgs A, const float *__restrict__ trig,
                                                                        const float *__restrict__ window, int nlong) {
    using WLL = WaveLds<N2R, Cfg::WPB, true>;
    using WLS = WaveLds<4, Cfg::WPB, true>;
    constexpr int kFloats = (LMc > 0 && WLS::TOTAL > WLL::TOTAL) ? WLS::TOTAL : WLL::TOTAL;
    __shared__ __attribute__((aligned(16))) float smem[kFloats];
    if constexpr (LMc > 0) {
        __shared__ ShortUnit utab[Cfg::WPB * FrameShortPacked<LMc>::K];
        if ((int)blockIdx.x >= nlong) {
            synth_short_role<LMc, Cfg>(A, trig, window, smem, utab, (long)blockIdx.x - nlong, (long)gridDim.x - nlong);
            return;
        }
    }
    synth_long_role<N2R, Cfg>(A, trig, window, smem, (long)blockIdx.x, (long)nlong);
}

// Heads that were mirrored against zeros receive their carry: slot f of the tails buffer holds
// the raw tail that precedes frame f (slot 0: the state handed in).  mdct.c:371-372 is linear in
// the carry, so  out[i] += w[119-i] c[i];  out[119-i] += w[i] c[i]  completes the mirror exactly.
constexpr int kFixupWaves = 4;   // waves per block; each wave scans 64 (stream*channel, frame) units
__global__ __launch_bounds__(kWave *kFixupWaves) void synth_fixup_kernel(SynthArgs A, int N, int chain_frames,
                                                                          const float *__restrict__ window) {
    // Few heads need work (one in 4 G plus the neighbours of transient frames), so scan 64 units per wave with one flag
    // test per lane and a ballot, then patch the hits FOUR AT A TIME: a quarter of the wave (15 of its 16 lanes, one float4
    // of the 60-float carry each) per hit.  (Round 2 patched one hit per step with 60 scalar lanes: at 240-sample frames --
    // four times the heads per byte -- the pass took 80 us of the call's 1000.)
    // grid: x = (stream, channel), y = blocks of 256 frames -- no index arithmetic beyond one wave-uniform division
    const int lane = threadIdx.x & (kWave - 1);
    const int q = lane & 15, h = lane >> 4;
    const long sc = blockIdx.x;
    const long f_base = ((long)blockIdx.y * kFixupWaves + (threadIdx.x >> 6)) * kWave;
    if (f_base >= A.nframes) return;
    const int f_mine = (int)f_base + lane;
    const bool need = f_mine < A.nframes &&
                      !head_done_in_wave(A.transient ? A.transient + (sc / A.channels) * A.fs() : nullptr, f_mine, chain_frames);
    unsigned long long todo = __ballot(need);
    // window weights of this lane's four carry samples i = 4q .. 4q+3: out[i] += w[119-i] c[i], out[119-i] += w[i] c[i]
    f4 W0 = {0, 0, 0, 0}, W1 = {0, 0, 0, 0};
    if (q < 15) {
        W0 = *reinterpret_cast<const f4 *>(window + 4 * q);
        W1 = *reinterpret_cast<const f4 *>(window + kOverlap - 4 - 4 * q);
    }
    while (todo) {
        // the h-th of the next four hits belongs to quarter h
        int bit = -1;
        unsigned long long t = todo;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int b = t ? __builtin_ctzll(t) : -1;
            if (k == h) bit = b;
            t &= t - 1;
        }
        todo = t;
        const int f = (int)f_base + bit;
        if (bit >= 0 && q < 15) {
            // frame 0 takes the overlap state handed in; that chain's state is then replaced by the tail behind its last
            // frame (same lanes, read before write: state_in and state_out may be one buffer; frame 0 always lands here)
            f4 c = {0, 0, 0, 0};
            if (f == 0) {
                if (A.state_in) c = *reinterpret_cast<const f4 *>(A.state_in + sc * (long)kHalfOv + 4 * q);
            } else {
                c = *reinterpret_cast<const f4 *>(A.tails + (sc * (A.nframes + 1) + f) * (long)kHalfOv + 4 * q);
            }
            float *o = A.pcm + (sc * A.nframes + f) * (long)N;
            f4 lo = *reinterpret_cast<const f4 *>(o + 4 * q), hi = *reinterpret_cast<const f4 *>(o + kOverlap - 4 - 4 * q);
            lo.x += W1.w * c.x; lo.y += W1.z * c.y; lo.z += W1.y * c.z; lo.w += W1.x * c.w;
            hi.w += W0.x * c.x; hi.z += W0.y * c.y; hi.y += W0.z * c.z; hi.x += W0.w * c.w;
            *reinterpret_cast<f4 *>(o + 4 * q) = lo;
            *reinterpret_cast<f4 *>(o + kOverlap - 4 - 4 * q) = hi;
            if (f == 0 && A.state_out)
                *reinterpret_cast<f4 *>(A.state_out + sc * (long)kHalfOv + 4 * q) =
                    *reinterpret_cast<const f4 *>(A.tails + (sc * (A.nframes + 1) + A.nframes) * (long)kHalfOv + 4 * q);
        }
    }
}

template <int N2R, int WPB>
__global__ __launch_bounds__(kWave *WPB) void ifft_rows_kernel(const float *__restrict__ in,
                                                                float *__restrict__ out, long nrows) {
    using Gm = Geo<N2R>;
    __shared__ cpx lds_all[WPB * Gm::LDS_CPX];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = lds_all + wv * Gm::LDS_CPX;
    const long ngroups = (nrows + Gm::G - 1) / Gm::G;
    const long nwaves = (long)gridDim.x * WPB;
    for (long gi = (long)blockIdx.x * WPB + wv; gi < ngroups; gi += nwaves) {
        const long row0 = gi * Gm::G;
        NYQ_WAVE_SYNC();
        ifft_stage_in<N2R>(lane, in, lds, row0, nrows);
        NYQ_WAVE_SYNC();
        fft_passes<N2R>(lane, lds);
        ifft_stage_out<N2R>(lane, lds, out, row0, nrows);
    }
}

}  // namespace nyq

// ---- Vorbis inverse MDCT (row f4) ------------------------------------------------------------------
#include "nyq_vorbis_lanes.hpp"

namespace nyq {

template <int LOGN4, int WPB>
__global__ __launch_bounds__(kWave *WPB) void vorbis_imdct_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                   long nrows, VTables T) {
    using V = VGeo<LOGN4>;
    __shared__ __attribute__((aligned(16))) float smem[WPB * 2 * V::LDS_CPX];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = reinterpret_cast<cpx *>(smem) + wv * V::LDS_CPX;
    VTwid<LOGN4> W;
    v_twid_init<LOGN4>(W, lane, T);
    const long ngroups = (nrows + V::G - 1) / V::G;
    const long nwaves = (long)gridDim.x * WPB;
    for (long gi = (long)blockIdx.x * WPB + wv; gi < ngroups; gi += nwaves) {
        const long row0 = gi * V::G;
        VStage<LOGN4> R;
        v_stage_in_load<LOGN4>(R, lane, in, row0, nrows);
        NYQ_WAVE_SYNC();
        v_stage_in_store<LOGN4>(R, lane, lds, T);
        NYQ_WAVE_SYNC();
#pragma unroll
        for (int it = 0; it < V::P1_ITERS; it++) {
            cpx u[V::R2];
            int g, k1;
            const bool ok = v_pass1_load<LOGN4>(lane, it, lds, u, g, k1);
            NYQ_WAVE_SYNC();
            if (ok) v_pass1_store<LOGN4>(g, k1, lds, u);
            NYQ_WAVE_SYNC();
        }
#pragma unroll
        for (int it = 0; it < V::P2_ITERS; it++) {
            cpx v[V::R1];
            int g, n2;
            const bool ok = v_pass2_load<LOGN4>(lane, it, lds, W, v, g, n2);
            NYQ_WAVE_SYNC();
            if (ok) v_pass2_store<LOGN4>(g, n2, lds, v);
            NYQ_WAVE_SYNC();
        }
        v_stage_out<LOGN4>(lane, lds, out, row0, nrows, T);
    }
}

}  // namespace nyq
