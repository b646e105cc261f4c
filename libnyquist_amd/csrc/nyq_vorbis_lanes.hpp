// nyq_vorbis_lanes.hpp -- lane program of the batched Vorbis inverse MDCT.
//
// libvorbis' mdct_backward (third_party/libvorbis/src/mdct.c:397-491; tables mdct_init :52-91) maps
// n/2 coefficients to n samples, n a power of two (Vorbis block sizes 64..8192, all built):
//     out[i] = sum_k X[k] cos(2 pi / n (i + 1/2 + n/4)(k + 1/2)),   i = 0 .. n-1
// which is the same transform as CELT's with exact rotations, the middle half
// raw[j] = out[n/4 + j] being what the N/4-point complex FFT produces and the outer quarters its
// mirrors:  out[i] = -raw[n/4-1-i]  (i < n/4),   out[3n/4 + i] = raw[n/2-1-i]  (i < n/4).
// So the structure of nyq_imdct_lanes.hpp is reused -- one wavefront owns a group of rows; stage-in with
// pre-rotation (two float4 loads = four complex points), two register-DFT passes through the wave's LDS
// slice, stage-out with post-rotation and float4 stores -- with three differences:
//   * N/4 = 2^m has no coprime split, so the passes are joined Cooley-Tukey style (k = k1 + R1 k2,
//     n = R2 n1 + n2) with one twiddle multiplication W^(k1 n2) between them; each pass-2 lane keeps its
//     R1 twiddles in registers for the life of the wave;
//   * rotations use an exact (cos, sin) table of angle 2 pi (i + 1/8) / n;
//   * stage-out writes all n samples (four float4 stores per task), no TDAC.
// Host+device so tests/emu can replay it on the CPU.
#pragma once
#include "nyq_imdct_lanes.hpp"

namespace nyq {

template <int LOGN4>
struct VGeo {
    static_assert(LOGN4 >= 4 && LOGN4 <= 11, "Vorbis block sizes 64 .. 8192 (n/4 = 16 .. 2048)");
    static constexpr int N4 = 1 << LOGN4;              // complex points
    static constexpr int N2 = 2 * N4;                  // coefficients per row
    static constexpr int N = 4 * N4;                   // samples per row
    static constexpr int R1 = 1 << (LOGN4 / 2);        // pass-2 radix
    static constexpr int R2 = N4 / R1;                 // pass-1 radix (R2 >= R1)
    // rows per group: enough for the radix-R1 pass to fill the wavefront, at least 4 -- except n = 8192, where
    // two rows already are 32 KB of LDS and 16 stage tasks (32 float4 registers) per lane
    static constexpr int G = LOGN4 == 11 ? 2 : (kWave / R1) > 4 ? (kWave / R1) : 4;
    static constexpr int NT = N4 / 4;                  // stage tasks per row
    static constexpr int TASKS = G * NT;               // per group
    static constexpr int TPL = (TASKS + kWave - 1) / kWave;         // tasks per lane
    static constexpr int ZROW = R2 * (R1 + 1);         // padded pass-1 -> pass-2 layout
    static constexpr int S0 = ZROW > N4 ? ZROW : N4;
    static constexpr int S = S0 + ((R1 % 32) - (S0 % 32) + 32) % 32;   // S = R1 (mod 32): rows of a 32-lane group on disjoint banks
    static constexpr int LDS_CPX = G * S;
    static constexpr int P1_ITERS = (G * R1 + kWave - 1) / kWave;
    static constexpr int P2_ITERS = (G * R2 + kWave - 1) / kWave;
};

// tables for one block size, built on the host in double precision:
//   rot[i] = (cos, sin)(2 pi (i + 1/8) / n),  i < n/4        twid[m] = (cos, sin)(2 pi m / (n/4)),  m < n/4
struct VTables {
    const float *rot;
    const float *twid;
};

template <int LOGN4>
struct VStage {
    f4 a[VGeo<LOGN4>::TPL], b[VGeo<LOGN4>::TPL];
};

template <int LOGN4>
NYQ_HD void v_task(int it, int lane, int &g, int &j, bool &on) {
    using V = VGeo<LOGN4>;
    const int t = it * kWave + lane;
    g = t / V::NT;
    j = t % V::NT;
    on = t < V::TASKS;
}

template <int LOGN4>
NYQ_HD void v_stage_in_load(VStage<LOGN4> &R, int lane, const float *in, long row0, long nrows) {
    using V = VGeo<LOGN4>;
#pragma unroll
    for (int it = 0; it < V::TPL; it++) {
        int g, j;
        bool on;
        v_task<LOGN4>(it, lane, g, j, on);
        if (on && row0 + g < nrows) {
            const float *row = in + (row0 + g) * (long)V::N2;
            R.a[it] = ld_f4<0>(row + 4 * j);
            R.b[it] = ld_f4<0>(row + V::N2 - 4 - 4 * j);
        } else {
            R.a[it] = f4{0, 0, 0, 0};
            R.b[it] = f4{0, 0, 0, 0};
        }
    }
}

// exact pre-rotation of point i from x1 = in[2i], x2 = in[N2-1-2i] by angle 2 pi (i + 1/8) / n
NYQ_HD cpx v_prerot(float x1, float x2, float c, float s) { return {-x2 * c + x1 * s, -x2 * s - x1 * c}; }
// exact post-rotation of FFT output k: .re -> raw[2k], .im -> raw[N2-1-2k]
NYQ_HD cpx v_postrot(cpx v, float c, float s) { return {-(v.re * c - v.im * s), v.im * c + v.re * s}; }

template <int LOGN4>
NYQ_HD void v_stage_in_store(const VStage<LOGN4> &R, int lane, cpx *lds, const VTables &T) {
    using V = VGeo<LOGN4>;
#pragma unroll
    for (int it = 0; it < V::TPL; it++) {
        int g, j;
        bool on;
        v_task<LOGN4>(it, lane, g, j, on);
        if (!on) continue;
        const f4 lo = ld_f4<0>(T.rot + 4 * j);                         // (c,s) of points 2j, 2j+1
        const f4 hi = ld_f4<0>(T.rot + 2 * (V::N4 - 2 - 2 * j));       // (c,s) of points N4-2-2j, N4-1-2j
        const f4 A = R.a[it], B = R.b[it];
        cpx *row = lds + g * V::S;
        row[2 * j] = v_prerot(A.x, B.w, lo.x, lo.y);
        row[2 * j + 1] = v_prerot(A.z, B.y, lo.z, lo.w);
        row[V::N4 - 2 - 2 * j] = v_prerot(B.x, A.w, hi.x, hi.y);
        row[V::N4 - 1 - 2 * j] = v_prerot(B.z, A.y, hi.z, hi.w);
    }
}

// pass 1: R1 lanes per row, radix-R2 over k2 (input k = k1 + R1 k2); result Z[k1][n2] -> slot n2 (R1+1) + k1
template <int LOGN4>
NYQ_HD bool v_pass1_load(int lane, int it, const cpx *lds, cpx (&u)[VGeo<LOGN4>::R2], int &g, int &k1) {
    using V = VGeo<LOGN4>;
    const int idx = it * kWave + lane;
    g = idx / V::R1;
    k1 = idx % V::R1;
    if (g >= V::G) return false;
    const cpx *p = lds + g * V::S + k1;
#pragma unroll
    for (int k2 = 0; k2 < V::R2; k2++) u[k2] = p[V::R1 * k2];
    return true;
}

template <int LOGN4>
NYQ_HD void v_pass1_store(int g, int k1, cpx *lds, cpx (&u)[VGeo<LOGN4>::R2]) {
    using V = VGeo<LOGN4>;
    Dft<V::R2>::run(u);
    cpx *p = lds + g * V::S + k1;
#pragma unroll
    for (int n2 = 0; n2 < V::R2; n2++) p[n2 * (V::R1 + 1)] = u[n2];
}

// the R1 twiddles W^(k1 n2) of pass-2 lane n2 = lane % R2 (lane invariant because R2 divides 64)
template <int LOGN4>
struct VTwid {
    cpx w[VGeo<LOGN4>::R1];
};

template <int LOGN4>
NYQ_HD void v_twid_init(VTwid<LOGN4> &W, int lane, const VTables &T) {
    using V = VGeo<LOGN4>;
    const int n2 = lane % V::R2;
#pragma unroll
    for (int k1 = 0; k1 < V::R1; k1++) {
        const int m = (k1 * n2) % V::N4;
        W.w[k1] = cpx{T.twid[2 * m], T.twid[2 * m + 1]};
    }
}

// pass 2: R2 lanes per row, radix-R1 over k1 with twiddles; Y[R2 n1 + n2] in natural order
template <int LOGN4>
NYQ_HD bool v_pass2_load(int lane, int it, const cpx *lds, const VTwid<LOGN4> &W, cpx (&v)[VGeo<LOGN4>::R1], int &g, int &n2) {
    using V = VGeo<LOGN4>;
    const int idx = it * kWave + lane;
    g = idx / V::R2;
    n2 = idx % V::R2;
    if (g >= V::G) return false;
    const cpx *p = lds + g * V::S + n2 * (V::R1 + 1);
#pragma unroll
    for (int k1 = 0; k1 < V::R1; k1++) {
        const cpx z = p[k1], w = W.w[k1];
        v[k1] = cpx{z.re * w.re - z.im * w.im, z.re * w.im + z.im * w.re};
    }
    return true;
}

template <int LOGN4>
NYQ_HD void v_pass2_store(int g, int n2, cpx *lds, cpx (&v)[VGeo<LOGN4>::R1]) {
    using V = VGeo<LOGN4>;
    Dft<V::R1>::run(v);
    cpx *row = lds + g * V::S;
#pragma unroll
    for (int n1 = 0; n1 < V::R1; n1++) row[V::R2 * n1 + n2] = v[n1];
}

// stage-out: post-rotation and the four float4 stores of a task (body + both mirrors, mdct.c:455-489)
template <int LOGN4>
NYQ_HD void v_stage_out(int lane, const cpx *lds, float *out, long row0, long nrows, const VTables &T) {
    using V = VGeo<LOGN4>;
#pragma unroll
    for (int it = 0; it < V::TPL; it++) {
        int g, j;
        bool on;
        v_task<LOGN4>(it, lane, g, j, on);
        if (!(on && row0 + g < nrows)) continue;
        const f4 lo = ld_f4<0>(T.rot + 4 * j);
        const f4 hi = ld_f4<0>(T.rot + 2 * (V::N4 - 2 - 2 * j));
        const cpx *row = lds + g * V::S;
        const cpx q0 = v_postrot(row[2 * j], lo.x, lo.y);
        const cpx q1 = v_postrot(row[2 * j + 1], lo.z, lo.w);
        const cpx q2 = v_postrot(row[V::N4 - 2 - 2 * j], hi.x, hi.y);
        const cpx q3 = v_postrot(row[V::N4 - 1 - 2 * j], hi.z, hi.w);
        const f4 F = {q0.re, q3.im, q1.re, q2.im};       // raw[4j .. 4j+3]            (< n/4)
        const f4 Bk = {q2.re, q1.im, q3.re, q0.im};      // raw[N2-4-4j .. N2-1-4j]    (>= n/4)
        float *o = out + (row0 + g) * (long)V::N;
        st_f4<0>(o + V::N4 + 4 * j, F);                                  // out[n/4 + p] = raw[p]
        st_f4<0>(o + V::N4 - 4 - 4 * j, f4{-F.w, -F.z, -F.y, -F.x});     // out[n/4-1-p] = -raw[p]
        st_f4<0>(o + V::N4 + V::N2 - 4 - 4 * j, Bk);
        st_f4<0>(o + 3 * V::N4 + 4 * j, f4{Bk.w, Bk.z, Bk.y, Bk.x});     // out[5n/4-1-p] = raw[p]
    }
}

}  // namespace nyq
