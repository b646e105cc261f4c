// nyq_imdct_lanes.hpp -- the per-lane program of the fused batched IMDCT.
//
// One wavefront (64 lanes) owns a GROUP of 4 consecutive rows (one row = one
// clt_mdct_backward call of the reference, third_party/opus/celt/mdct.c:267-379)
// and takes them through five phases that meet only in that wave's private LDS
// slice -- no workgroup barrier anywhere:
//
//   A  stage-in + pre-rotation   (mdct.c:295-313)   global float4 x2 -> 4 complex points -> LDS
//   B  pass 1: radix-N2R DFT     (kiss_fft.c:696-747 replaced) 15 lanes per row, in place
//   C  pass 2: radix-15  DFT     N2R lanes per row, in place -> natural order
//   D  stage-out: post-rotation  (mdct.c:322-359), TDAC mirror (mdct.c:362-377),
//      float4 stores of the finished samples and the 60-float raw tail.
//
// nfft = N4 = 15 * N2R with N2R in {32,16,8,4} (shift 0..3 of the static 48 kHz
// mode).  15 and N2R are coprime, so the two passes are joined by the Good-Thomas
// prime-factor maps and need NO inter-pass twiddle multiplications:
//     k = (N2R*k1 + 15*k2) mod N4         k1 in [0,15), k2 in [0,N2R)
//     n = (N2R*T1*n1 + 15*T2*n2) mod N4   T1 = N2R^-1 mod 15, T2 = 15^-1 mod N2R
// LDS holds one row as N4 complex floats at float2 granularity:
//     phases A->B->C use slot 15*k2 + k1   (B is in place on the same slots,
//                                           C reads 15*n2 + k1: 30-dword lane stride,
//                                           conflict free for ds_read_b64)
//     phases C->D use natural order n      (C writes with a 225-slot lane stride:
//                                           2*225 mod 32 = 2, conflict free)
// Row stride S = N4 rounded up to 16 mod 32 so the two rows that share a 32-lane
// LDS group in pass 1 land on disjoint banks.
//
// Everything here is __host__ __device__: tests/emu replays the same lane program
// on the CPU (lane by lane, phase by phase) to check the index maps without a GPU.
#pragma once
#include "nyq_fft_core.hpp"

namespace nyq {

struct alignas(16) f4 {
    float x, y, z, w;
};

// 16-byte global accesses.  NT != 0 marks the access non-temporal (streamed once, do not keep
// in L2/MALL); on the host replay it is an ordinary access.
template <int NT>
NYQ_HD f4 ld_f4(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v4f __attribute__((ext_vector_type(4)));
    if constexpr (NT != 0) {
        v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return f4{v.x, v.y, v.z, v.w};
    }
#endif
    return *reinterpret_cast<const f4 *>(p);
}
template <int NT>
NYQ_HD void st_f4(float *p, f4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v4f __attribute__((ext_vector_type(4)));
    if constexpr (NT != 0) {
        v4f w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p));
        return;
    }
#endif
    *reinterpret_cast<f4 *>(p) = v;
}

// rows per wave-group of the two short transforms (measurement switches: tools/variant_ab.py; see Geo::G)
#ifndef NYQ_G60
#define NYQ_G60 16
#endif
#ifndef NYQ_G120
#define NYQ_G120 8
#endif
constexpr int kOverlap = 120;   // static_modes_float.h:579
constexpr int kHalfOv = 60;
constexpr int kWave = 64;

template <int N2R>
struct Geo {
    static_assert(N2R == 32 || N2R == 16 || N2R == 8 || N2R == 4, "nfft must be 15 * 2^a");
    static constexpr int N4 = 15 * N2R;               // complex points (nfft)
    static constexpr int NIN = 2 * N4;                // real coefficients per row (N2)
    static constexpr int SHIFT = N2R == 32 ? 0 : N2R == 16 ? 1 : N2R == 8 ? 2 : 3;
    static constexpr int NT = N4 / 4;                 // stage tasks per row (4 points each)
    static constexpr int SLOT = NT > 64 ? 128 : NT > 32 ? 64 : NT > 16 ? 32 : 16;
    // rows per wave-group: enough that the radix-15 pass (N2R lanes per row) fills the wavefront and that a
    // group keeps >= 7.5 KB of loads in flight: 4, 8, 8, 16 rows for nfft 480, 240, 120, 60 (16 rows at nfft 120 cost the
    // frame-synthesis kernel its second wave per SIMD: 256 VGPRs)
    static constexpr int G = N2R == 32 ? 4 : N2R == 4 ? NYQ_G60 : N2R == 8 ? NYQ_G120 : 8;
    static constexpr int SUBS = G * SLOT / kWave;        // stage sub-iterations per group
    static constexpr int JSETS = SLOT > kWave ? SLOT / kWave : 1;
    static constexpr int S = (N4 % 32 <= 16) ? (N4 - N4 % 32 + 16) : (N4 - N4 % 32 + 48);
    static constexpr int T1 = inv_mod(N2R, 15);
    static constexpr int T2 = inv_mod(15, N2R);
    static constexpr int P1_ITERS = G / 4;            // pass 1: four rows (16-lane slots) per iteration
    static constexpr int P2_ROWS = kWave / N2R;       // rows per pass-2 iteration
    static constexpr int P2_ITERS = P2_ROWS >= G ? 1 : G / P2_ROWS;
    static constexpr int LDS_CPX = G * S;             // per-wave LDS slice for the rows, in cpx
    static constexpr int RING_FLOATS = (G + 1) * 60;  // tail ring (chained rows only)
    static constexpr int CHAIN_FRAMES = 4 * G;        // frames of one channel a wave chains in-wave (kChainGroups groups)
    // sine = 2*PI*0.125/N with the reference's float PI (mdct.c:292, mathops.h:83)
    static constexpr float SINE = (float)2 * 3.141592653f * (.125f) / (float)(4 * N4);
};

// slot of point k in the A/B/C layout
template <int N2R>
NYQ_HD int slot_of(int k) {
    using Gm = Geo<N2R>;
    int k1 = (k * Gm::T1) % 15;
    int k2 = (k * Gm::T2) % N2R;
    return 15 * k2 + k1;
}

// lane-invariant values, fetched once per wave
template <int N2R>
struct LaneConst {
    // per j-set: t[2j], t[2j+1], t[2j+2], t[N4-2-2j], t[N4-1-2j], t[N4-2j] (index << SHIFT)
    float tr[Geo<N2R>::JSETS][6];
    int xs[Geo<N2R>::JSETS][4];   // LDS slots of points 2j, 2j+1, N4-2-2j, N4-1-2j
    float wlo[4], whi[4];          // window[56-4j..59-4j], window[60+4j..63+4j] (j < 15 only)
};

template <int N2R>
NYQ_HD void lane_init(LaneConst<N2R> &K, int lane, const float *trig, const float *window) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int s = 0; s < Gm::JSETS; s++) {
        int j = (s * kWave + lane) % Gm::SLOT;
        if (j >= Gm::NT) j = 0;   // idle lane: any valid index
        K.tr[s][0] = trig[(2 * j) << Gm::SHIFT];
        K.tr[s][1] = trig[(2 * j + 1) << Gm::SHIFT];
        K.tr[s][2] = trig[(2 * j + 2) << Gm::SHIFT];
        K.tr[s][3] = trig[(Gm::N4 - 2 - 2 * j) << Gm::SHIFT];
        K.tr[s][4] = trig[(Gm::N4 - 1 - 2 * j) << Gm::SHIFT];
        K.tr[s][5] = trig[(Gm::N4 - 2 * j) << Gm::SHIFT];
        K.xs[s][0] = slot_of<N2R>(2 * j);
        K.xs[s][1] = slot_of<N2R>(2 * j + 1);
        K.xs[s][2] = slot_of<N2R>(Gm::N4 - 2 - 2 * j);
        K.xs[s][3] = slot_of<N2R>(Gm::N4 - 1 - 2 * j);
    }
    {
        int j = lane % Gm::SLOT;
        if (j >= 15) j = 0;
        K.wlo[0] = window[56 - 4 * j];
        K.wlo[1] = window[57 - 4 * j];
        K.wlo[2] = window[58 - 4 * j];
        K.wlo[3] = window[59 - 4 * j];
        K.whi[0] = window[60 + 4 * j];
        K.whi[1] = window[61 + 4 * j];
        K.whi[2] = window[62 + 4 * j];
        K.whi[3] = window[63 + 4 * j];
    }
}

// (row-in-group, task, active) of stage sub-iteration `sub` for this lane
template <int N2R>
NYQ_HD void stage_slot(int sub, int lane, int &g, int &j, bool &on) {
    using Gm = Geo<N2R>;
    int sigma = sub * kWave + lane;
    g = sigma / Gm::SLOT;
    j = sigma % Gm::SLOT;
    on = j < Gm::NT;
}

// mdct.c:303-312: one pre-rotated point from x1 = in[2i], x2 = in[N2-1-2i]
NYQ_HD cpx prerot(float x1, float x2, float c, float s, float sine) {
    float yr = -x2 * c + x1 * s;
    float yi = -x2 * s - x1 * c;
    return {yr - yi * sine, yi + yr * sine};
}

// mdct.c:330-355: post-rotate FFT output k; .re -> raw[2k], .im -> raw[N2-1-2k]
NYQ_HD cpx postrot(cpx v, float c, float s, float sine) {
    float yr = v.re * c - v.im * s;
    float yi = v.im * c + v.re * s;
    return {-(yr - yi * sine), yi + yr * sine};
}

// ---- row sources ---------------------------------------------------------------
// A "Rows" object tells the lane program where the G rows of the current group live.  It is a
// handful of wave-uniform scalars with inline accessors, so addresses are recomputed from
// g instead of being kept per row:
//   static constexpr bool STRIDED   input elements are `stride()` floats apart (interleaved
//                                   short blocks of a transient frame, celt_decoder_clean.c:292-300)
//   static constexpr bool CHAINS    row g may take its carry from row g-1's raw tail
//   bool valid(g); const float *in(g); int stride(); float *fin(g);
//   float *head(g)  where out[0..60) of row g goes (fin(g) unless a caller keeps the row in a rotated layout)
//   float *tail(g)  (nullptr: do not store);  const float *carry(g)  (nullptr: zeros);
//   bool chain(g)   (CHAINS only: carry of row g = raw tail of row g-1 of this group,
//                    or for g == 0 the tail left in the ring by the previous group)
// Independent rows of one [nrows][N2] batch (nyq_imdct_batch_dev):
template <int N2R>
struct IndepRows {
    static constexpr bool STRIDED = false;
    static constexpr bool CHAINS = false;
    const float *in_;
    const float *carry_;
    float *fin_;
    float *tail_;
    long row0, nrows;
    NYQ_HD bool valid(int g) const { return row0 + g < nrows; }
    NYQ_HD const float *in(int g) const { return in_ + (row0 + g) * (long)Geo<N2R>::NIN; }
    NYQ_HD int stride() const { return 1; }
    NYQ_HD float *fin(int g) const { return fin_ + (row0 + g) * (long)Geo<N2R>::NIN; }
    NYQ_HD float *head(int g) const { return fin(g); }
    NYQ_HD float *tail(int g) const { return tail_ ? tail_ + (row0 + g) * kHalfOv : nullptr; }
    NYQ_HD const float *carry(int g) const { return carry_ ? carry_ + (row0 + g) * kHalfOv : nullptr; }
    NYQ_HD bool chain(int) const { return false; }
};

// ---- frame sequences (compute_inv_mdcts, celt_decoder_clean.c:264-312,622-656) ----------
// `nstreams` independent decoders, each `nframes` frames of `channels` channels, all of frame
// size N = 120 << LM.  Layouts:
//   freq      [stream][frame][channel][N]   as the decoder leaves freq[] (channel c at freq + c*N);
//                                           a transient frame holds B = 2^LM interleaved short
//                                           blocks, coefficient k of block b at [b + B*k]
//   transient [stream][frame]               non-zero = transient (shortBlocks) frame
//   pcm       [stream][channel][frame*N..]  time-contiguous per channel (the out_syn history)
//   tails     [stream*channels][nframes+1][60]   slot f+1 = raw tail after frame f (slot 0 is unused: the overlap
//                                           state before frame 0 is read from state_in by the fix-up pass)
//   state_in / state_out [stream*channels][60] or null; may be the same buffer (the fix-up pass reads a chain's state
//                                           before it writes it, in the same lanes)
struct SynthArgs {
    const float *freq;
    const unsigned char *transient;
    float *pcm;
    float *tails;
    long nstreams, nframes;
    int channels;
    const float *state_in;
    float *state_out;
    // freq and transient may be WINDOWS into longer per-stream arrays: consecutive streams are `fstride` frames apart
    // (0 = dense: nframes).  pcm, tails and the states are always dense.
    long fstride;
    NYQ_HD long fs() const { return fstride ? fstride : nframes; }
};

// Long frames.  A wave visits CHUNKS of kChainGroups groups = 4 G consecutive frames of one
// (stream, channel); inside a chunk every long frame that follows a long frame takes its carry
// in-wave (LDS tail ring, handed from group to group by ring_rotate).  Chunk-first frames and
// frames after a transient frame are mirrored against zeros here and receive their carry in
// synth_fixup; their predecessors publish a tail to the tails buffer.
constexpr int kChainGroups = 4;

template <int N2R>
struct FrameLongRows {
    static constexpr bool STRIDED = false;
    static constexpr bool CHAINS = true;
    static constexpr int N = Geo<N2R>::NIN;
    static constexpr int G = Geo<N2R>::G;
    static constexpr int kChainFrames = Geo<N2R>::CHAIN_FRAMES;
    const float *in0;   // frame f0, this channel
    float *fin0;        // pcm of frame f0
    float *tail0;       // tails slot f0+1
    long in_step;       // channels * N
    unsigned longmask;  // bit g+1 set <=> frame f0+g exists and is long (g = -1..G)
    int qq;             // group inside the chunk
    // chunk ci = sc * chunks_per_channel + k covers frames 4Gk .. 4Gk+4G-1 of channel sc
    NYQ_HD static long chunks_per_channel(long nframes) { return (nframes + kChainFrames - 1) / kChainFrames; }
    // Bit i + 1 of the chunk's transient mask <=> frame (first frame of the chunk) + i is transient, i = -1 .. 4 G
    // (frames outside the stream count as not transient; `valid` checks the range).  The kernel fetches the mask
    // ONCE per chunk (one byte per lane + a ballot) instead of G + 2 flag bytes in front of every group's loads.
    NYQ_HD static unsigned long long chunk_transient_mask(const SynthArgs &A, long ci) {
        const long cpc = chunks_per_channel(A.nframes);
        const long sc = ci / cpc, k = ci - sc * cpc;
        const long s = sc / A.channels;
        unsigned long long m = 0;
        if (A.transient) {
            const unsigned char *t = A.transient + s * A.fs();
            for (int i = -1; i <= kChainFrames; i++) {
                const long f = k * kChainFrames + i;
                if (i + 1 < 64 && f >= 0 && f < A.nframes && t[f]) m |= 1ull << (i + 1);   // (4 G + 2 <= 64 whenever flags exist: LM >= 1)
            }
        }
        return m;
    }
    NYQ_HD FrameLongRows(const SynthArgs &A, long ci, int qq_) : FrameLongRows(A, ci, qq_, chunk_transient_mask(A, ci)) {}
    NYQ_HD FrameLongRows(const SynthArgs &A, long ci, int qq_, unsigned long long tmask) {
        const long cpc = chunks_per_channel(A.nframes);
        const long sc = ci / cpc, k = ci - sc * cpc;
        const long s = sc / A.channels, c = sc - s * A.channels;
        const long f0 = k * kChainFrames + (long)qq_ * G;
        qq = qq_;
        in_step = (long)A.channels * N;
        in0 = A.freq + ((s * A.fs() + f0) * A.channels + c) * (long)N;
        fin0 = A.pcm + (sc * A.nframes + f0) * (long)N;
        tail0 = A.tails + (sc * (A.nframes + 1) + f0 + 1) * (long)kHalfOv;
        longmask = 0;
#pragma unroll
        for (int g = -1; g <= G; g++) {
            const long f = f0 + g;
            const int bit = qq_ * G + g + 1;
            const bool tr = bit < 64 && ((tmask >> bit) & 1ull);
            if (f >= 0 && f < A.nframes && !tr) longmask |= 1u << (g + 1);
        }
    }
    NYQ_HD bool any() const { return (longmask & (((1u << G) - 1u) << 1)) != 0; }
    NYQ_HD bool is_long(int g) const { return (longmask >> (g + 1)) & 1u; }
    NYQ_HD bool valid(int g) const { return is_long(g); }
    NYQ_HD const float *in(int g) const { return in0 + g * in_step; }
    NYQ_HD int stride() const { return 1; }
    NYQ_HD float *fin(int g) const { return fin0 + g * (long)N; }
    NYQ_HD float *head(int g) const { return fin(g); }
    // row 0 of a later group of the chunk chains to the ring slot the previous group left
    NYQ_HD bool chain(int g) const { return (g > 0 || qq > 0) && is_long(g - 1); }
    NYQ_HD const float *carry(int) const { return nullptr; }
    // publish the tail unless the next frame's head is completed in-wave
    NYQ_HD float *tail(int g) const {
        const bool next_in_wave = is_long(g + 1) && (g < G - 1 || qq < kChainGroups - 1);
        return next_in_wave ? nullptr : tail0 + g * (long)kHalfOv;
    }
};

// The short blocks of UP TO 16 / B transient frames -- any (stream, channel, frame) units -- as ONE group of 16 rows
// (round 3): unit k owns rows k B .. k B + B - 1, its blocks chain through the tail ring (row g takes
// the tail row g - 1 left, block 0 of every unit is mirrored against zeros and completed by the fix-up pass).  A
// transient frame of B = 2 blocks used to fill 2 of the 16 rows of its own group (LM 1 synthesis ran at 3800 GB/s where
// the other sizes reach 4500); eight of them now share one.  LMc = log2 B (1, 2, 3), K = 16 >> LMc units per group.
// where one transient frame of one (stream, channel) lives: a row of the wave's unit table (LDS; the lane that found the
// frame in the scan fills it from the indices it already holds)
struct ShortUnit {
    const float *in0;      // freq of the frame and channel
    float *fin0;           // pcm of the frame
    float *tslot;          // tails slot f + 1 of the channel
};
template <int LMc>
NYQ_HD ShortUnit short_unit(const SynthArgs &A, long s, long c, long f) {
    constexpr long N = 120L << LMc;
    const long sc = s * A.channels + c;
    return ShortUnit{A.freq + ((s * A.fs() + f) * A.channels + c) * N, A.pcm + (sc * A.nframes + f) * N,
                     A.tails + (sc * (A.nframes + 1) + f + 1) * (long)kHalfOv};
}

template <int LMc>
struct FrameShortPacked {
    static constexpr bool STRIDED = true;
    static constexpr bool CHAINS = true;
    static constexpr int B = 1 << LMc;
    static constexpr int K = Geo<4>::G / B;
    const ShortUnit *tab;  // K rows, the first nunits valid
    int nunits;
    NYQ_HD FrameShortPacked(const ShortUnit *tab_, int n) : tab(tab_), nunits(n) {}
    NYQ_HD bool valid(int g) const { return (g >> LMc) < nunits; }
    NYQ_HD const float *in(int g) const { return tab[g >> LMc].in0 + (g & (B - 1)); }
    NYQ_HD int stride() const { return B; }
    NYQ_HD float *fin(int g) const { return tab[g >> LMc].fin0 + 120L * (g & (B - 1)); }
    NYQ_HD float *head(int g) const { return fin(g); }
    NYQ_HD bool chain(int g) const { return (g & (B - 1)) > 0; }
    NYQ_HD const float *carry(int) const { return nullptr; }
    NYQ_HD float *tail(int g) const { return (g & (B - 1)) == B - 1 ? tab[g >> LMc].tslot : nullptr; }
};

// Is the head of frame f (its first 120 samples) already mirrored against the true carry?
// Only long frames that chained in-wave are; every other head gets its carry in synth_fixup.
NYQ_HD bool head_done_in_wave(const unsigned char *t, long f, int chain_frames) {
    return (f % chain_frames) != 0 && !(t && (t[f] || t[f - 1]));
}

// ---- phase A -----------------------------------------------------------
template <int N2R>
struct StageRegs {
    f4 a[Geo<N2R>::SUBS], b[Geo<N2R>::SUBS];
};

// issue every global load of the group
template <int N2R, int NT, class Rows>
NYQ_HD void stage_in_load(StageRegs<N2R> &R, int lane, const Rows &rows) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (on && rows.valid(g)) {
            const float *row = rows.in(g);
            if constexpr (Rows::STRIDED) {
                const long st = rows.stride();
                const float *pa = row + (long)(4 * j) * st;
                const float *pb = row + (long)(Gm::NIN - 4 - 4 * j) * st;
                R.a[sub] = f4{pa[0], pa[st], pa[2 * st], pa[3 * st]};
                R.b[sub] = f4{pb[0], pb[st], pb[2 * st], pb[3 * st]};
            } else {
                R.a[sub] = ld_f4<NT>(row + 4 * j);
                R.b[sub] = ld_f4<NT>(row + Gm::NIN - 4 - 4 * j);
            }
        } else {
            R.a[sub] = f4{0, 0, 0, 0};
            R.b[sub] = f4{0, 0, 0, 0};
        }
    }
}

template <int N2R>
NYQ_HD void stage_in_store(const StageRegs<N2R> &R, const LaneConst<N2R> &K, int lane, cpx *lds) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (on) {
            const int s = sub % Gm::JSETS;
            const f4 A = R.a[sub], B = R.b[sub];
            cpx *row = lds + g * Gm::S;
            row[K.xs[s][0]] = prerot(A.x, B.w, K.tr[s][0], K.tr[s][5], Gm::SINE);
            row[K.xs[s][1]] = prerot(A.z, B.y, K.tr[s][1], K.tr[s][4], Gm::SINE);
            row[K.xs[s][2]] = prerot(B.x, A.w, K.tr[s][3], K.tr[s][2], Gm::SINE);
            row[K.xs[s][3]] = prerot(B.z, A.y, K.tr[s][4], K.tr[s][1], Gm::SINE);
        }
    }
}

// ---- phase B: radix-N2R over k2, 15 lanes per row, 16-lane slots ---------
template <int N2R>
NYQ_HD void pass1(int lane, cpx *lds) {
    using Gm = Geo<N2R>;
    const int k1 = lane & 15;
    if (k1 < 15) {
#pragma unroll
        for (int it = 0; it < Gm::P1_ITERS; it++) {       // in place, every (row, k1) column on its own
            const int g = 4 * it + (lane >> 4);
            cpx *p = lds + g * Gm::S + k1;
            cpx u[N2R];
#pragma unroll
            for (int k2 = 0; k2 < N2R; k2++) u[k2] = p[15 * k2];
            Dft<N2R>::run(u);
#pragma unroll
            for (int n2 = 0; n2 < N2R; n2++) p[15 * n2] = u[n2];
        }
    }
}

// ---- phase C: radix-15 over k1, N2R lanes per row --------------------------
template <int N2R>
NYQ_HD bool pass2_load(int lane, int it, const cpx *lds, cpx (&v)[15], int &g, int &n2) {
    using Gm = Geo<N2R>;
    g = it * Gm::P2_ROWS + lane / N2R;
    n2 = lane % N2R;
    if (g >= Gm::G) return false;
    const cpx *p = lds + g * Gm::S + 15 * n2;
#pragma unroll
    for (int k1 = 0; k1 < 15; k1++) v[k1] = p[k1];
    return true;
}

template <int N2R>
NYQ_HD void pass2_store(int g, int n2, cpx *lds, cpx (&v)[15]) {
    using Gm = Geo<N2R>;
    Dft<15>::run(v);
    cpx *row = lds + g * Gm::S;
    const int base = (15 * Gm::T2 * n2) % Gm::N4;
#pragma unroll
    for (int n1 = 0; n1 < 15; n1++) {
        constexpr int dummy = 0;
        (void)dummy;
        int n = base + (N2R * Gm::T1 * n1) % Gm::N4;
        if (n >= Gm::N4) n -= Gm::N4;
        row[n] = v[n1];
    }
}

// ---- phase D -----------------------------------------------------------------
// Sub-iterations whose lanes can own a TDAC head (task j < 15): only the first 64-lane slice of
// each row slot.
template <int N2R>
NYQ_HD constexpr bool sub_has_heads(int sub) {
    return Geo<N2R>::SLOT <= kWave || (sub % Geo<N2R>::JSETS) == 0;
}

// The TDAC mirror of one head task (mdct.c:362-377): i = 59-4j-e, x1 = raw[59-i] = F[e],
// x2 = carry[i] = C[3-e];  out[i] = w[119-i] x2 - w[i] x1,  out[119-i] = w[i] x2 + w[119-i] x1.
template <int N2R>
NYQ_HD void tdac_mix(const LaneConst<N2R> &K, f4 F, f4 C, f4 &hi, f4 &lo) {
    hi.x = K.wlo[3] * C.w + K.whi[0] * F.x;   // out[60+4j+0]
    hi.y = K.wlo[2] * C.z + K.whi[1] * F.y;
    hi.z = K.wlo[1] * C.y + K.whi[2] * F.z;
    hi.w = K.wlo[0] * C.x + K.whi[3] * F.w;
    lo.w = K.whi[0] * C.w - K.wlo[3] * F.x;   // out[59-4j-0]
    lo.z = K.whi[1] * C.z - K.wlo[2] * F.y;
    lo.y = K.whi[2] * C.y - K.wlo[1] * F.z;
    lo.x = K.whi[3] * C.x - K.wlo[0] * F.w;
}

// Heads kept in registers between the two halves of phase D when rows chain.
template <int N2R>
struct HeadRegs {
    f4 F[Geo<N2R>::SUBS], Bk[Geo<N2R>::SUBS];
};

// Part 1: post-rotation, body stores; head lanes (j < 15) either finish at once (no chaining)
// or park F/Bk and publish their raw tail in the wave's LDS tail ring: ring[g+1] = tail of row g
// (ring[0] = tail left by the previous group).  ring: (G + 1) x 60 floats.
template <int N2R, int NT, class Rows>
NYQ_HD void stage_out(const LaneConst<N2R> &K, int lane, const cpx *lds, float *ring, const Rows &rows,
                      HeadRegs<N2R> &H) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (!(on && rows.valid(g))) continue;
        const int s = sub % Gm::JSETS;
        const cpx *row = lds + g * Gm::S;
        cpx q0 = postrot(row[2 * j], K.tr[s][0], K.tr[s][5], Gm::SINE);
        cpx q1 = postrot(row[2 * j + 1], K.tr[s][1], K.tr[s][4], Gm::SINE);
        cpx q2 = postrot(row[Gm::N4 - 2 - 2 * j], K.tr[s][3], K.tr[s][2], Gm::SINE);
        cpx q3 = postrot(row[Gm::N4 - 1 - 2 * j], K.tr[s][4], K.tr[s][1], Gm::SINE);
        // raw[4j..4j+3] and raw[NIN-4-4j..NIN-1-4j]
        f4 F = {q0.re, q3.im, q1.re, q2.im};
        f4 Bk = {q2.re, q1.im, q3.re, q0.im};
        float *orow = rows.fin(g);
        if (sub_has_heads<N2R>(sub) && j < 15) {
            if constexpr (Rows::CHAINS) {
                H.F[sub] = F;
                H.Bk[sub] = Bk;
                *reinterpret_cast<f4 *>(ring + (g + 1) * kHalfOv + 56 - 4 * j) = Bk;
            } else {
                f4 C = {0, 0, 0, 0};
                const float *cy = rows.carry(g);
                if (cy) C = *reinterpret_cast<const f4 *>(cy + 56 - 4 * j);
                f4 hi, lo;
                tdac_mix<N2R>(K, F, C, hi, lo);
                st_f4<NT>(orow + 60 + 4 * j, hi);
                st_f4<NT>(rows.head(g) + 56 - 4 * j, lo);
                float *tl = rows.tail(g);
                if (tl) st_f4<NT>(tl + 56 - 4 * j, Bk);
            }
        } else {
            st_f4<NT>(orow + 60 + 4 * j, F);
            st_f4<NT>(orow + Gm::NIN + 56 - 4 * j, Bk);
        }
    }
}

// Part 2 (CHAINS only; all lanes must have finished part 1): heads take their carry from the
// tail ring (chained rows) or from memory / zeros, mix, store, and emit tails where asked.
template <int N2R, int NT, class Rows>
NYQ_HD void stage_out_heads(const LaneConst<N2R> &K, int lane, const float *ring, const Rows &rows,
                            const HeadRegs<N2R> &H) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        if (!sub_has_heads<N2R>(sub)) continue;
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (!(on && rows.valid(g) && j < 15)) continue;
        f4 C = {0, 0, 0, 0};
        if (rows.chain(g)) {
            C = *reinterpret_cast<const f4 *>(ring + g * kHalfOv + 56 - 4 * j);
        } else {
            const float *cy = rows.carry(g);
            if (cy) C = *reinterpret_cast<const f4 *>(cy + 56 - 4 * j);
        }
        f4 hi, lo;
        tdac_mix<N2R>(K, H.F[sub], C, hi, lo);
        float *orow = rows.fin(g);
        st_f4<NT>(orow + 60 + 4 * j, hi);
        st_f4<NT>(rows.head(g) + 56 - 4 * j, lo);
        float *tl = rows.tail(g);
        if (tl) st_f4<NT>(tl + 56 - 4 * j, H.Bk[sub]);
    }
}

// Part 3 (CHAINS only, after part 2): hand the last row's tail to the next group of this wave.
template <int N2R>
NYQ_HD void ring_rotate(int lane, float *ring) {
    if (lane < 15) {
        f4 v = *reinterpret_cast<const f4 *>(ring + Geo<N2R>::G * kHalfOv + 4 * lane);
        *reinterpret_cast<f4 *>(ring + 4 * lane) = v;
    }
}

// ---- IFFT-only variant (opus_ifft, kiss_fft.c:696-747; golden-vector op) -----
// in/out: [nrows][N4] interleaved complex, natural order, unscaled inverse.
// The G x N4/2 float4 pieces (two complex points each) of a group are dealt to the lanes round robin, so
// every load / store instruction has all 64 lanes busy whatever the row length.
template <int N2R>
NYQ_HD void ifft_stage_in(int lane, const float *in, cpx *lds, long row0, long nrows) {
    using Gm = Geo<N2R>;
    constexpr int PAIRS = Gm::N4 / 2;   // float4 = two complex points
    constexpr int ITERS = (Gm::G * PAIRS + kWave - 1) / kWave;
    f4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int idx = it * kWave + lane, g = idx / PAIRS, p = idx - g * PAIRS;
        v[it] = f4{0, 0, 0, 0};
        if (g < Gm::G && row0 + g < nrows) v[it] = *reinterpret_cast<const f4 *>(in + (row0 + g) * (long)(2 * Gm::N4) + 4 * p);
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int idx = it * kWave + lane, g = idx / PAIRS, p = idx - g * PAIRS;
        if (g < Gm::G) {
            cpx *lrow = lds + g * Gm::S;
            lrow[slot_of<N2R>(2 * p)] = cpx{v[it].x, v[it].y};
            lrow[slot_of<N2R>(2 * p + 1)] = cpx{v[it].z, v[it].w};
        }
    }
}

template <int N2R>
NYQ_HD void ifft_stage_out(int lane, const cpx *lds, float *out, long row0, long nrows) {
    using Gm = Geo<N2R>;
    constexpr int PAIRS = Gm::N4 / 2;
    constexpr int ITERS = (Gm::G * PAIRS + kWave - 1) / kWave;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int idx = it * kWave + lane, g = idx / PAIRS, p = idx - g * PAIRS;
        if (g < Gm::G && row0 + g < nrows) {
            const cpx *lrow = lds + g * Gm::S;
            const cpx a = lrow[2 * p], b = lrow[2 * p + 1];
            *reinterpret_cast<f4 *>(out + (row0 + g) * (long)(2 * Gm::N4) + 4 * p) = f4{a.re, a.im, b.re, b.im};
        }
    }
}

}  // namespace nyq
