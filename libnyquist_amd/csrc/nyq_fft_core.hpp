// nyq_fft_core.hpp -- in-register inverse DFT building blocks (gfx950 device code,
// also compilable on the host so tests/emu can replay the exact lane program).
//
// Replaces, for the MI355X path, the mixed-radix butterflies of the reference
// (third_party/opus/celt/kiss_fft.c:82-110,158-200,258-306,385-455: ki_bfly2/4/3/5
// driven by opus_ifft :696-747).  The reference walks memory stage by stage with a
// run-time twiddle table; here every transform a lane owns is a compile-time
// expression tree over registers:
//   * radix 2/3/4/5 kernels written directly,
//   * power-of-two radices 8/16/32 composed by Cooley-Tukey with constant twiddles
//     folded into immediates,
//   * radix 15 composed by the Good-Thomas prime-factor map (3 x 5, no twiddles).
// Sign convention: unscaled inverse DFT, Y[n] = sum_k x[k] e^{+2 pi i k n / R}
// (kiss_fft.c:696 opus_ifft: conjugate twiddles via C_MULC, no 1/N).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NYQ_HD __host__ __device__ __forceinline__
#else
#define NYQ_HD inline __attribute__((always_inline))
#endif

namespace nyq {

struct cpx {
    float re, im;
};

NYQ_HD cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
NYQ_HD cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
// multiply by +i (quarter turn of the inverse transform)
NYQ_HD cpx cmul_i(cpx a) { return {-a.im, a.re}; }
NYQ_HD cpx cmul_ni(cpx a) { return {a.im, -a.re}; }

// cos(2 pi m / 64), m = 0..16 (first quarter turn), double literals rounded once to float.
// All power-of-two twiddles are read out of this table by symmetry.
NYQ_HD constexpr double cos64_q(int m) {
    constexpr double c[17] = {1.0,
                              0.99518472667219688624483695310948,
                              0.98078528040323044912618223613424,
                              0.95694033573220886493579788698027,
                              0.92387953251128675612818318939679,
                              0.88192126434835502971275686366039,
                              0.83146961230254523707878837761791,
                              0.77301045336273696081090660975847,
                              0.70710678118654752440084436210485,
                              0.63439328416364549821517161322549,
                              0.55557023301960222474283081394853,
                              0.47139673682599764855638762590525,
                              0.38268343236508977172845998403040,
                              0.29028467725446236763619237581740,
                              0.19509032201612826784828486847702,
                              0.09801714032956060199419556388864,
                              0.0};
    return c[m];
}
// cos(2 pi m / 64) for any m
NYQ_HD constexpr double cos64(int m) {
    m %= 64;
    if (m < 0) m += 64;
    if (m > 32) m = 64 - m;      // cos is even about pi
    return m <= 16 ? cos64_q(m) : -cos64_q(32 - m);
}
NYQ_HD constexpr double sin64(int m) { return cos64(m - 16); }

// a * e^{+2 pi i m / N} for N in {4,8,16,32,64}, m a compile-time constant.
// Trivial rotations cost nothing, odd multiples of 1/8 turn cost 2 adds + 2 muls.
template <int N, int M>
NYQ_HD cpx rot(cpx a) {
    static_assert(64 % N == 0, "power-of-two twiddles up to the 64-gon");
    constexpr int m64 = ((M % N) + N) % N * (64 / N);   // position on the 64-gon
    if constexpr (m64 == 0) return a;
    else if constexpr (m64 == 16) return cmul_i(a);
    else if constexpr (m64 == 32) return {-a.re, -a.im};
    else if constexpr (m64 == 48) return cmul_ni(a);
    else if constexpr (m64 % 16 == 8) {
        constexpr float h = (float)cos64_q(8);
        if constexpr (m64 == 8) return {(a.re - a.im) * h, (a.re + a.im) * h};
        else if constexpr (m64 == 24) return {-(a.re + a.im) * h, (a.re - a.im) * h};
        else if constexpr (m64 == 40) return {(a.im - a.re) * h, -(a.re + a.im) * h};
        else return {(a.re + a.im) * h, (a.im - a.re) * h};
    } else {
        constexpr float c = (float)cos64(m64), s = (float)sin64(m64);
        return {a.re * c - a.im * s, a.re * s + a.im * c};
    }
}

template <int R>
struct Dft;   // static void run(cpx (&v)[R]) : natural order in, natural order out

template <>
struct Dft<1> {
    static NYQ_HD void run(cpx (&)[1]) {}
};

template <>
struct Dft<2> {
    static NYQ_HD void run(cpx (&v)[2]) {
        cpx a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <>
struct Dft<4> {
    static NYQ_HD void run(cpx (&v)[4]) {
        cpx s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
        cpx s13 = cadd(v[1], v[3]), d13 = cmul_i(csub(v[1], v[3]));
        v[0] = cadd(s02, s13);
        v[2] = csub(s02, s13);
        v[1] = cadd(d02, d13);
        v[3] = csub(d02, d13);
    }
};

template <>
struct Dft<3> {
    static NYQ_HD void run(cpx (&v)[3]) {
        constexpr float s60 = 0.86602540378443864676372317075294f;   // sin(2 pi / 3)
        cpx s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
        cpx m = {v[0].re - 0.5f * s.re, v[0].im - 0.5f * s.im};
        cpx q = {-s60 * d.im, s60 * d.re};                           // i * s60 * d
        v[0] = cadd(v[0], s);
        v[1] = cadd(m, q);
        v[2] = csub(m, q);
    }
};

template <>
struct Dft<5> {
    static NYQ_HD void run(cpx (&v)[5]) {
        constexpr float c1 = 0.30901699437494742410229341718282f;    // cos(2 pi / 5)
        constexpr float c2 = -0.80901699437494742410229341718282f;   // cos(4 pi / 5)
        constexpr float s1 = 0.95105651629515357211643933337938f;    // sin(2 pi / 5)
        constexpr float s2 = 0.58778525229247312916870595463907f;    // sin(4 pi / 5)
        cpx a = v[0];
        cpx p1 = cadd(v[1], v[4]), m1 = csub(v[1], v[4]);
        cpx p2 = cadd(v[2], v[3]), m2 = csub(v[2], v[3]);
        cpx e1 = {a.re + c1 * p1.re + c2 * p2.re, a.im + c1 * p1.im + c2 * p2.im};
        cpx e2 = {a.re + c2 * p1.re + c1 * p2.re, a.im + c2 * p1.im + c1 * p2.im};
        // i * (s1 m1 + s2 m2) and i * (s2 m1 - s1 m2)
        cpx o1 = {-(s1 * m1.im + s2 * m2.im), s1 * m1.re + s2 * m2.re};
        cpx o2 = {-(s2 * m1.im - s1 * m2.im), s2 * m1.re - s1 * m2.re};
        v[0] = {a.re + p1.re + p2.re, a.im + p1.im + p2.im};
        v[1] = cadd(e1, o1);
        v[4] = csub(e1, o1);
        v[2] = cadd(e2, o2);
        v[3] = csub(e2, o2);
    }
};

// Cooley-Tukey A x B with constant twiddles: k = a + A b, n = B na + nb.
template <int A, int B>
struct CtDft {
    static constexpr int N = A * B;
    template <int a, int nb>
    static NYQ_HD void tw_col(cpx (&t)[N], const cpx (&u)[B]) {
        t[a * B + nb] = rot<N, a * nb>(u[nb]);
        if constexpr (nb + 1 < B) tw_col<a, nb + 1>(t, u);
    }
    template <int a>
    static NYQ_HD void rows(cpx (&t)[N], const cpx (&v)[N]) {
        cpx u[B];
#pragma unroll
        for (int b = 0; b < B; b++) u[b] = v[a + A * b];
        Dft<B>::run(u);
        tw_col<a, 0>(t, u);
        if constexpr (a + 1 < A) rows<a + 1>(t, v);
    }
    static NYQ_HD void run(cpx (&v)[N]) {
        cpx t[N];
        rows<0>(t, v);
#pragma unroll
        for (int nb = 0; nb < B; nb++) {
            cpx u[A];
#pragma unroll
            for (int a = 0; a < A; a++) u[a] = t[a * B + nb];
            Dft<A>::run(u);
#pragma unroll
            for (int na = 0; na < A; na++) v[B * na + nb] = u[na];
        }
    }
};

template <>
struct Dft<8> {
    static NYQ_HD void run(cpx (&v)[8]) { CtDft<2, 4>::run(v); }
};
template <>
struct Dft<16> {
    static NYQ_HD void run(cpx (&v)[16]) { CtDft<4, 4>::run(v); }
};
template <>
struct Dft<32> {
    static NYQ_HD void run(cpx (&v)[32]) { CtDft<4, 8>::run(v); }
};
template <>
struct Dft<64> {   // Vorbis n = 8192 only (n/4 = 2048 = 32 x 64)
    static NYQ_HD void run(cpx (&v)[64]) { CtDft<8, 8>::run(v); }
};

constexpr int inv_mod(int a, int m) {
    a %= m;
    for (int x = 1; x < m; x++)
        if ((a * x) % m == 1) return x;
    return 0;
}

// Good-Thomas A x B (coprime): k = (B a + A b) mod N in, n = (B tA na + A tB nb) mod N out.
template <int A, int B>
struct PfaDft {
    static constexpr int N = A * B;
    static NYQ_HD void run(cpx (&v)[N]) {
        constexpr int tA = inv_mod(B, A), tB = inv_mod(A, B);
        cpx t[N];
#pragma unroll
        for (int a = 0; a < A; a++) {
            cpx u[B];
#pragma unroll
            for (int b = 0; b < B; b++) u[b] = v[(B * a + A * b) % N];
            Dft<B>::run(u);
#pragma unroll
            for (int nb = 0; nb < B; nb++) t[a * B + nb] = u[nb];
        }
#pragma unroll
        for (int nb = 0; nb < B; nb++) {
            cpx u[A];
#pragma unroll
            for (int a = 0; a < A; a++) u[a] = t[a * B + nb];
            Dft<A>::run(u);
#pragma unroll
            for (int na = 0; na < A; na++) v[(B * tA * na + A * tB * nb) % N] = u[na];
        }
    }
};

template <>
struct Dft<15> {
    static NYQ_HD void run(cpx (&v)[15]) { PfaDft<3, 5>::run(v); }
};

}  // namespace nyq
