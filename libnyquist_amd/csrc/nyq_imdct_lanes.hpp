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

constexpr int kOverlap = 120;   // static_modes_float.h:579
constexpr int kHalfOv = 60;
constexpr int kGroup = 4;       // rows per wave-group
constexpr int kWave = 64;

template <int N2R>
struct Geo {
    static_assert(N2R == 32 || N2R == 16 || N2R == 8 || N2R == 4, "nfft must be 15 * 2^a");
    static constexpr int N4 = 15 * N2R;               // complex points (nfft)
    static constexpr int NIN = 2 * N4;                // real coefficients per row (N2)
    static constexpr int SHIFT = N2R == 32 ? 0 : N2R == 16 ? 1 : N2R == 8 ? 2 : 3;
    static constexpr int NT = N4 / 4;                 // stage tasks per row (4 points each)
    static constexpr int SLOT = NT > 64 ? 128 : NT > 32 ? 64 : NT > 16 ? 32 : 16;
    static constexpr int SUBS = kGroup * SLOT / kWave;   // stage sub-iterations per group
    static constexpr int JSETS = SLOT > kWave ? SLOT / kWave : 1;
    static constexpr int S = (N4 % 32 <= 16) ? (N4 - N4 % 32 + 16) : (N4 - N4 % 32 + 48);
    static constexpr int T1 = inv_mod(N2R, 15);
    static constexpr int T2 = inv_mod(15, N2R);
    static constexpr int P2_ROWS = kWave / N2R;       // rows per pass-2 iteration (may exceed 4)
    static constexpr int P2_ITERS = P2_ROWS >= kGroup ? 1 : kGroup / P2_ROWS;
    static constexpr int LDS_CPX = kGroup * S;        // per-wave LDS slice, in cpx
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

// ---- phase A -----------------------------------------------------------
template <int N2R>
struct StageRegs {
    f4 a[Geo<N2R>::SUBS], b[Geo<N2R>::SUBS];
};

// issue every global load of the group (rows row0 .. row0+3, clipped to nrows)
template <int N2R, int NT = 0>
NYQ_HD void stage_in_load(StageRegs<N2R> &R, int lane, const float *in, long row0, long nrows) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (on && row0 + g < nrows) {
            const float *row = in + (row0 + g) * (long)Gm::NIN;
            R.a[sub] = ld_f4<NT>(row + 4 * j);
            R.b[sub] = ld_f4<NT>(row + Gm::NIN - 4 - 4 * j);
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
    const int g = lane >> 4, k1 = lane & 15;
    if (k1 < 15) {
        cpx *p = lds + g * Gm::S + k1;
        cpx u[N2R];
#pragma unroll
        for (int k2 = 0; k2 < N2R; k2++) u[k2] = p[15 * k2];
        Dft<N2R>::run(u);
#pragma unroll
        for (int n2 = 0; n2 < N2R; n2++) p[15 * n2] = u[n2];
    }
}

// ---- phase C: radix-15 over k1, N2R lanes per row --------------------------
template <int N2R>
NYQ_HD bool pass2_load(int lane, int it, const cpx *lds, cpx (&v)[15], int &g, int &n2) {
    using Gm = Geo<N2R>;
    g = it * Gm::P2_ROWS + lane / N2R;
    n2 = lane % N2R;
    if (g >= kGroup) return false;
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
// carry: [nrows][60] or nullptr (zeros).  fin: [nrows][NIN].  tail: [nrows][60] or nullptr.
template <int N2R, int NT = 0>
NYQ_HD void stage_out(const LaneConst<N2R> &K, int lane, const cpx *lds, const float *carry,
                      float *fin, float *tail, long row0, long nrows) {
    using Gm = Geo<N2R>;
#pragma unroll
    for (int sub = 0; sub < Gm::SUBS; sub++) {
        int g, j;
        bool on;
        stage_slot<N2R>(sub, lane, g, j, on);
        if (!(on && row0 + g < nrows)) continue;
        const int s = sub % Gm::JSETS;
        const long r = row0 + g;
        const cpx *row = lds + g * Gm::S;
        cpx q0 = postrot(row[2 * j], K.tr[s][0], K.tr[s][5], Gm::SINE);
        cpx q1 = postrot(row[2 * j + 1], K.tr[s][1], K.tr[s][4], Gm::SINE);
        cpx q2 = postrot(row[Gm::N4 - 2 - 2 * j], K.tr[s][3], K.tr[s][2], Gm::SINE);
        cpx q3 = postrot(row[Gm::N4 - 1 - 2 * j], K.tr[s][4], K.tr[s][1], Gm::SINE);
        // raw[4j..4j+3] and raw[NIN-4-4j..NIN-1-4j]
        f4 F = {q0.re, q3.im, q1.re, q2.im};
        f4 Bk = {q2.re, q1.im, q3.re, q0.im};
        float *orow = fin + r * (long)Gm::NIN;
        if (j < 15) {
            // TDAC mirror, mdct.c:362-377: i = 59-4j-e, x1 = raw[59-i] = F[e], x2 = carry[i]
            f4 C = {0, 0, 0, 0};
            if (carry) C = *reinterpret_cast<const f4 *>(carry + r * kHalfOv + 56 - 4 * j);
            f4 hi, lo;
            hi.x = K.wlo[3] * C.w + K.whi[0] * F.x;   // out[60+4j+0]
            hi.y = K.wlo[2] * C.z + K.whi[1] * F.y;
            hi.z = K.wlo[1] * C.y + K.whi[2] * F.z;
            hi.w = K.wlo[0] * C.x + K.whi[3] * F.w;
            lo.w = K.whi[0] * C.w - K.wlo[3] * F.x;   // out[59-4j-0]
            lo.z = K.whi[1] * C.z - K.wlo[2] * F.y;
            lo.y = K.whi[2] * C.y - K.wlo[1] * F.z;
            lo.x = K.whi[3] * C.x - K.wlo[0] * F.w;
            st_f4<NT>(orow + 60 + 4 * j, hi);
            st_f4<NT>(orow + 56 - 4 * j, lo);
            if (tail) st_f4<NT>(tail + r * kHalfOv + 56 - 4 * j, Bk);
        } else {
            st_f4<NT>(orow + 60 + 4 * j, F);
            st_f4<NT>(orow + Gm::NIN + 56 - 4 * j, Bk);
        }
    }
}

// ---- IFFT-only variant (opus_ifft, kiss_fft.c:696-747; golden-vector op) -----
// in/out: [nrows][N4] interleaved complex, natural order, unscaled inverse.
template <int N2R>
NYQ_HD void ifft_stage_in(int lane, const float *in, cpx *lds, long row0, long nrows) {
    using Gm = Geo<N2R>;
    constexpr int PAIRS = Gm::N4 / 2;   // float4 = two complex points
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
        if (row0 + g >= nrows) continue;
        const float *row = in + (row0 + g) * (long)(2 * Gm::N4);
        cpx *lrow = lds + g * Gm::S;
        for (int p = lane; p < PAIRS; p += kWave) {
            f4 v = *reinterpret_cast<const f4 *>(row + 4 * p);
            lrow[slot_of<N2R>(2 * p)] = cpx{v.x, v.y};
            lrow[slot_of<N2R>(2 * p + 1)] = cpx{v.z, v.w};
        }
    }
}

template <int N2R>
NYQ_HD void ifft_stage_out(int lane, const cpx *lds, float *out, long row0, long nrows) {
    using Gm = Geo<N2R>;
    constexpr int PAIRS = Gm::N4 / 2;
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
        if (row0 + g >= nrows) continue;
        float *row = out + (row0 + g) * (long)(2 * Gm::N4);
        const cpx *lrow = lds + g * Gm::S;
        for (int p = lane; p < PAIRS; p += kWave) {
            cpx a = lrow[2 * p], b = lrow[2 * p + 1];
            *reinterpret_cast<f4 *>(row + 4 * p) = f4{a.re, a.im, b.re, b.im};
        }
    }
}

}  // namespace nyq
