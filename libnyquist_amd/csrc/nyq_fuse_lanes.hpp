// nyq_fuse_lanes.hpp -- lane program of the TRANSFORM WAVE of the one-launch frames -> PCM kernel
// (nyq_chain_kernel.hpp): compute_inv_mdcts (celt_decoder_clean.c:264-312) for ONE stereo 20 ms frame per
// iteration -- two clt_mdct_backward rows (mdct.c:267-379) or, in a transient frame, 2 x 8 interleaved short
// blocks (:292-300) -- computed by ONE wavefront IN PLACE in the two 3840-byte frame regions of the post-filter's
// LDS buffers, where the time-domain frame has to land anyway.  No LDS of its own: that is what lets the transform
// sit in the post-filter's workgroup (the two earlier fusions failed on exactly this).
//
// Long frame.  nfft 480 = 15 x 32 by the Good-Thomas map (no inter-factor twiddles), the 32 as a DIF 2 x 16:
//   S0  (once per row) lane v < 60 owns stage tasks j in {v, 119-v}: 4 float4 of coefficients (prefetched a frame
//       ahead) = the 8 complex points {2v, 2v+1, 238-2v, 239-2v} + 240 m, m = 0, 1.  Pre-rotation (mdct.c:295-313), the
//       FIRST butterflies (radix 2 over m) and their W32 twiddles run in registers: no LDS round trip between stage-in
//       and the first pass.  8 ds_write_b64 per row.
//   S2  lane (row r = lane >> 5, v = lane & 31 < 30): radix-16 over slots v + 30 k, in place.  Both rows at once.
//   S3  lane (row, n2): radix-15 over slots 15 n2 + k1, natural order out.  All 64 lanes.
//   S4  (once per row) same tasks as S0: post-rotation (mdct.c:322-359) -> raw[], TDAC mirror (mdct.c:362-377) of the head
//       against the tail the SAME lane kept from the previous frame (registers: lane j < 15 holds tail[56-4j..59-4j]),
//       finished samples out[0..960) back into the region.  S4 (and the transient program's T4, T5) of a row run on the
//       chain's COMB wave, one iteration later, right before it filters the frame: the transform wave was the slowest role.
// LDS layout of a row while it is being transformed (float2 slots): point (k1, k2lo, na) at 15 (na + 2 k2lo) + k1 after
// S0, (k1, n2 = na + 2 nb) at 15 n2 + k1 after S2, natural order after S3.  Bank conflicts: S2, S3 none, S0 a few
// 2-way ones that a ds_write_b64's issue time covers (searched: tools/scripts/fuse_layout_search.py).
// (A 4 x 8 split with both rows in one S0 pass -- 30 lanes per row, four tasks per lane -- is the same instruction count
// but needs 60 lane-resident constants instead of 32: 48 VGPRs spilled under the kernel's 128-register budget.)
//
// Transient frame (the 8 short blocks of each channel, coefficient k of block b at freq[b + 8 k]): the same prefetched
// registers are de-interleaved into the regions as 16 rows of 120 floats (row (c, b) at float 120 b of region c), then
// the nfft-60 program (4 x 15, as nyq_imdct_lanes.hpp's <4>) runs with a row stride of exactly 60 slots, post-rotates
// in place and mirrors every block against its predecessor's second half (block 0: the register tail).
//
// Everything here is __host__ __device__: tests/emu replays it lane by lane on the CPU against the oracle.
#pragma once
#include "nyq_imdct_lanes.hpp"

namespace nyq {
namespace fx {

constexpr int kN = 960;            // samples per channel-frame (LM 3)
constexpr int kStageLanes = 60;    // active lanes of S0 / S4 (one row per pass)

NYQ_HD bool stage_lane(int lane) { return lane < kStageLanes; }
// the two stage tasks of lane v (task j: float4 j and float4 239 - j of the row)
NYQ_HD int task_of(int v, int t) { return t == 0 ? v : 119 - v; }

// region c of the two, as a select (a dynamically indexed local array of pointers would live in scratch memory and its
// pointers would lose their LDS address space)
NYQ_HD float *region_of(float *const (&reg)[2], int c) { return c ? reg[1] : reg[0]; }

NYQ_HD cpx cmul(cpx a, cpx w) { return {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re}; }

// prime-factor coordinates of point k of a 480-point row: k = (32 k1 + 15 k2) mod 480
NYQ_HD int pfa_k1(int k) { return (8 * k) % 15; }
NYQ_HD int pfa_k2(int k) { return (15 * k) & 31; }

// lane-invariant values.  The transform is split over two waves (nyq_chain_kernel.hpp): S0 .. S3 run on the transform
// wave (XfRot + XfTw), S4 / T4 / T5 on the chain's own comb wave right before it filters the frame (XfRot + XfWin).
struct XfRot {
    float tr[2][6];   // task t: trig[2j], [2j+1], [2j+2], [478-2j], [479-2j], [480-2j]
};
struct XfTw {
    cpx tw[4];        // point set R: e^{+2 pi i k2b / 32} (k2b = k2 of the set's m = 0 point)
    int sb01, sb23;   // point set R: slot of its na = 0 output, 30 k2lo + k1 (na = 1: + 15); two 16-bit fields per register
};
struct XfWin {
    float wlo[4], whi[4];   // window[56-4j..59-4j], window[60+4j..63+4j], j = lane & 15 (< 15): TDAC lanes of both programs
};

// residue (mod 240) of point set R of lane v: 2v, 2v+1, 238-2v, 239-2v
NYQ_HD int set_residue(int v, int R) { return R == 0 ? 2 * v : R == 1 ? 2 * v + 1 : R == 2 ? 238 - 2 * v : 239 - 2 * v; }

// e^{+2 pi i m / 32} (evaluated once per wave, in xf_init)
NYQ_HD cpx w32(int m) {
    m &= 31;
#if defined(__HIP_DEVICE_COMPILE__)
    return {cospif((float)m * 0.0625f), sinpif((float)m * 0.0625f)};
#else
    return {(float)__builtin_cos(3.14159265358979323846 * m / 16.0), (float)__builtin_sin(3.14159265358979323846 * m / 16.0)};
#endif
}

NYQ_HD void xf_init_rot(XfRot &K, int lane, const float *trig) {
    const int v = lane < kStageLanes ? lane : 0;   // idle lane: any valid index
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = task_of(v, t);
        K.tr[t][0] = trig[2 * j];
        K.tr[t][1] = trig[2 * j + 1];
        K.tr[t][2] = trig[2 * j + 2];
        K.tr[t][3] = trig[478 - 2 * j];
        K.tr[t][4] = trig[479 - 2 * j];
        K.tr[t][5] = trig[480 - 2 * j];
    }
}
NYQ_HD void xf_init_tw(XfTw &K, int lane) {
    const int v = lane < kStageLanes ? lane : 0;
#pragma unroll
    for (int R = 0; R < 4; R++) {
        const int k0 = set_residue(v, R);
        const int k1 = pfa_k1(k0), k2b = pfa_k2(k0);
        const int sb = 30 * (k2b & 15) + k1;
        if (R == 0) K.sb01 = sb;
        if (R == 1) K.sb01 |= sb << 16;
        if (R == 2) K.sb23 = sb;
        if (R == 3) K.sb23 |= sb << 16;
        K.tw[R] = w32(k2b);
    }
}
NYQ_HD void xf_init_win(XfWin &K, int lane, const float *window) {
    int j = lane & 15;
    if (j >= 15) j = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        K.wlo[e] = window[56 - 4 * j + e];
        K.whi[e] = window[60 + 4 * j + e];
    }
}

// the prefetched coefficients of one stereo frame: row r, task t: a = in[4j..4j+3], b = in[956-4j..959-4j]
struct XfRegs {
    f4 a[2][2], b[2][2];
};

// issue the loads: frame = freq of (stream, frame), channel r at + r * 960
template <int NT>
NYQ_HD void xf_load(XfRegs &R, int lane, const float *frame) {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (lane < kStageLanes) {
                const int j = task_of(lane, t);
                R.a[r][t] = ld_f4<NT>(frame + r * kN + 4 * j);
                R.b[r][t] = ld_f4<NT>(frame + r * kN + 956 - 4 * j);
            } else {
                R.a[r][t] = f4{0, 0, 0, 0};
                R.b[r][t] = f4{0, 0, 0, 0};
            }
        }
}

constexpr float kSineLong = Geo<32>::SINE;
constexpr float kSineShort = Geo<4>::SINE;

// ---- long frame ------------------------------------------------------------------------------------------
// S0 of row r: pre-rotation, radix-2 over m, twiddle, scatter.  region = the row's 960 floats = 480 slots.
NYQ_HD void xf_long_s0(const XfRegs &R, const XfRot &K, const XfTw &W, int lane, int r, float *region) {
    if (!stage_lane(lane)) return;
    cpx *row = reinterpret_cast<cpx *>(region);
    cpx p[2][4];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const f4 A = R.a[r][t], B = R.b[r][t];
        p[t][0] = prerot(A.x, B.w, K.tr[t][0], K.tr[t][5], kSineLong);   // point 2j
        p[t][1] = prerot(A.z, B.y, K.tr[t][1], K.tr[t][4], kSineLong);   // 2j+1
        p[t][2] = prerot(B.x, A.w, K.tr[t][3], K.tr[t][2], kSineLong);   // 478-2j
        p[t][3] = prerot(B.z, A.y, K.tr[t][4], K.tr[t][1], kSineLong);   // 479-2j
    }
    // point set R: residue (m = 0), residue + 240 (m = 1)
    const cpx s0[4] = {p[0][0], p[0][1], p[1][0], p[1][1]};     // 2v, 2v+1, 238-2v, 239-2v
    const cpx s1[4] = {p[1][2], p[1][3], p[0][2], p[0][3]};     // 240+2v, 241+2v, 478-2v, 479-2v
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int pk = q < 2 ? W.sb01 : W.sb23;
        cpx *o = row + ((q & 1) ? (pk >> 16) : (pk & 0xffff));
        o[0] = cadd(s0[q], s1[q]);
        o[15] = cmul(csub(s0[q], s1[q]), W.tw[q]);
    }
}

// S2: radix-16 over k2lo, lane (row = lane >> 5, v = lane & 31 < 30), both rows at once
NYQ_HD void xf_long_s2_load(int lane, float *const (&reg)[2], cpx (&u)[16]) {
    const int v = (lane & 31) < 30 ? (lane & 31) : 0;
    const cpx *row = reinterpret_cast<const cpx *>(region_of(reg, lane >> 5)) + v;
#pragma unroll
    for (int k = 0; k < 16; k++) u[k] = row[30 * k];
}
NYQ_HD void xf_long_s2_store(int lane, float *const (&reg)[2], cpx (&u)[16]) {
    if ((lane & 31) >= 30) return;
    cpx *row = reinterpret_cast<cpx *>(region_of(reg, lane >> 5)) + (lane & 31);
    Dft<16>::run(u);
#pragma unroll
    for (int n = 0; n < 16; n++) row[30 * n] = u[n];
}

// S3: radix-15 over k1, lane (row, n2); natural order out: n = (225 n2 + 256 n1) mod 480
NYQ_HD void xf_long_s3_load(int lane, float *const (&reg)[2], cpx (&v)[15]) {
    const cpx *row = reinterpret_cast<const cpx *>(region_of(reg, lane >> 5)) + 15 * (lane & 31);
#pragma unroll
    for (int k1 = 0; k1 < 15; k1++) v[k1] = row[k1];
}
NYQ_HD void xf_long_s3_store(int lane, float *const (&reg)[2], cpx (&v)[15]) {
    cpx *row = reinterpret_cast<cpx *>(region_of(reg, lane >> 5));
    Dft<15>::run(v);
    const int base = (225 * (lane & 31)) % 480;
#pragma unroll
    for (int n1 = 0; n1 < 15; n1++) {
        int n = base + (256 * n1) % 480;
        if (n >= 480) n -= 480;
        row[n] = v[n1];
    }
}

// the TDAC mirror of one head task (nyq_imdct_lanes.hpp tdac_mix, on this program's window registers)
NYQ_HD void xf_tdac(const XfWin &K, f4 F, f4 C, f4 &hi, f4 &lo) {
    hi.x = K.wlo[3] * C.w + K.whi[0] * F.x;
    hi.y = K.wlo[2] * C.z + K.whi[1] * F.y;
    hi.z = K.wlo[1] * C.y + K.whi[2] * F.z;
    hi.w = K.wlo[0] * C.x + K.whi[3] * F.w;
    lo.w = K.whi[0] * C.w - K.wlo[3] * F.x;
    lo.z = K.whi[1] * C.z - K.wlo[2] * F.y;
    lo.y = K.whi[2] * C.y - K.wlo[1] * F.z;
    lo.x = K.whi[3] * C.x - K.wlo[0] * F.w;
}

// post-rotation of one stage task: FFT outputs 2j, 2j+1 (P01) and 478-2j, 479-2j (P23) -> raw[4j..4j+3], raw[956-4j..959-4j]
NYQ_HD void xf_postrot4(f4 P01, f4 P23, const float (&tr)[6], float sine, f4 &F, f4 &Bk) {
    const cpx q0 = postrot(cpx{P01.x, P01.y}, tr[0], tr[5], sine);
    const cpx q1 = postrot(cpx{P01.z, P01.w}, tr[1], tr[4], sine);
    const cpx q2 = postrot(cpx{P23.x, P23.y}, tr[3], tr[2], sine);
    const cpx q3 = postrot(cpx{P23.z, P23.w}, tr[4], tr[1], sine);
    F = f4{q0.re, q3.im, q1.re, q2.im};
    Bk = f4{q2.re, q1.im, q3.re, q0.im};
}

// S4 of one row in two halves (every lane's reads come before any lane's writes: the samples move by 60 floats)
struct XfOut {
    f4 F[2], Bk[2];
};
NYQ_HD void xf_long_s4_load(const XfRot &K, int lane, const float *region, XfOut &O) {
    const int v = stage_lane(lane) ? lane : 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = task_of(v, t);
        const f4 P01 = *reinterpret_cast<const f4 *>(region + 4 * j);
        const f4 P23 = *reinterpret_cast<const f4 *>(region + 956 - 4 * j);
        xf_postrot4(P01, P23, K.tr[t], kSineLong, O.F[t], O.Bk[t]);
    }
}
// out[60 + 4j ..] = raw[4j ..]; out[1016 - 4j ..] = raw[956 - 4j ..] (j >= 15); heads j < 15: mirror against `tail`,
// which then becomes this frame's raw[956-4j..959-4j]
NYQ_HD void xf_long_s4_store(const XfWin &K, int lane, float *region, const XfOut &O, f4 &tail) {
    if (!stage_lane(lane)) return;
    const int v = lane;
    {
        const int j = task_of(v, 1);
        *reinterpret_cast<f4 *>(region + 60 + 4 * j) = O.F[1];
        *reinterpret_cast<f4 *>(region + 1016 - 4 * j) = O.Bk[1];
    }
    if (v < 15) {
        f4 hi, lo;
        xf_tdac(K, O.F[0], tail, hi, lo);
        *reinterpret_cast<f4 *>(region + 60 + 4 * v) = hi;
        *reinterpret_cast<f4 *>(region + 56 - 4 * v) = lo;
        tail.x = O.Bk[0].x; tail.y = O.Bk[0].y; tail.z = O.Bk[0].z; tail.w = O.Bk[0].w;
    } else {
        *reinterpret_cast<f4 *>(region + 60 + 4 * v) = O.F[0];
        *reinterpret_cast<f4 *>(region + 1016 - 4 * v) = O.Bk[0];
    }
}

// ---- transient frame: 2 x 8 short blocks, nfft 60 = 4 x 15 -------------------------------------------------
// per-lane rotation values of short-block task j = lane & 15: trig[(2j) << 3] ...
struct XfShortConst {
    float ts[6];
};
NYQ_HD void xf_short_init(XfShortConst &S, int lane, const float *trig) {
    int j = lane & 15;
    if (j >= 15) j = 0;
    S.ts[0] = trig[(2 * j) << 3];
    S.ts[1] = trig[(2 * j + 1) << 3];
    S.ts[2] = trig[(2 * j + 2) << 3];
    S.ts[3] = trig[(58 - 2 * j) << 3];
    S.ts[4] = trig[(59 - 2 * j) << 3];
    S.ts[5] = trig[(60 - 2 * j) << 3];
}

// T0: the interleaved coefficients (float m of the frame = coefficient m >> 3 of block m & 7) out of the prefetch
// registers into rows: block b at floats [120 b, 120 b + 120) of the channel's region
NYQ_HD void xf_short_t0(const XfRegs &R, int lane, float *const (&reg)[2]) {
    if (!stage_lane(lane)) return;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = task_of(lane, t);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int u = h == 0 ? j : 239 - j;          // float4 index inside the row
                const f4 q = h == 0 ? R.a[r][t] : R.b[r][t];
                float *d = reg[r] + 480 * (u & 1) + (u >> 1);   // block 4 (u & 1) + i, coefficient u >> 1
                d[0] = q.x;
                d[120] = q.y;
                d[240] = q.z;
                d[360] = q.w;
            }
        }
}

// row g = 8 c + b of the 16: region c, float offset 120 b
NYQ_HD float *short_row(float *const (&reg)[2], int g) { return region_of(reg, g >> 3) + 120 * (g & 7); }
NYQ_HD int slot60(int k) { return 15 * ((3 * k) & 3) + (4 * k) % 15; }

// T1: stage-in of sub-iteration s (rows 4s .. 4s+3, 16 lanes each): pre-rotation into prime-factor slots.  Row-local,
// but a task's slots are other tasks' inputs: two halves (every lane's reads come before any lane's writes)
struct XfShortIn {
    f4 A, B;
};
NYQ_HD void xf_short_t1_load(int lane, int s, float *const (&reg)[2], XfShortIn &I) {
    const int j = (lane & 15) < 15 ? (lane & 15) : 0;
    const float *rowf = short_row(reg, 4 * s + (lane >> 4));
    I.A = *reinterpret_cast<const f4 *>(rowf + 4 * j);
    I.B = *reinterpret_cast<const f4 *>(rowf + 116 - 4 * j);
}
NYQ_HD void xf_short_t1_store(const XfShortConst &S, int lane, int s, float *const (&reg)[2], const XfShortIn &I) {
    const int j = lane & 15;
    if (j >= 15) return;
    cpx *row = reinterpret_cast<cpx *>(short_row(reg, 4 * s + (lane >> 4)));
    const f4 A = I.A, B = I.B;
    row[slot60(2 * j)] = prerot(A.x, B.w, S.ts[0], S.ts[5], kSineShort);
    row[slot60(2 * j + 1)] = prerot(A.z, B.y, S.ts[1], S.ts[4], kSineShort);
    row[slot60(58 - 2 * j)] = prerot(B.x, A.w, S.ts[3], S.ts[2], kSineShort);
    row[slot60(59 - 2 * j)] = prerot(B.z, A.y, S.ts[4], S.ts[1], kSineShort);
}

// T2: radix-4 over k2 (15 lanes per row), iteration `it` = rows 4 it .. 4 it + 3
NYQ_HD void xf_short_t2(int lane, int it, float *const (&reg)[2]) {
    const int k1 = lane & 15;
    if (k1 >= 15) return;
    cpx *p = reinterpret_cast<cpx *>(short_row(reg, 4 * it + (lane >> 4))) + k1;
    cpx u[4] = {p[0], p[15], p[30], p[45]};
    Dft<4>::run(u);
    p[0] = u[0];
    p[15] = u[1];
    p[30] = u[2];
    p[45] = u[3];
}

// T3: radix-15 over k1, lane (row g = lane >> 2, n2 = lane & 3); natural order out: n = (45 n2 + 16 n1) mod 60
NYQ_HD void xf_short_t3_load(int lane, float *const (&reg)[2], cpx (&v)[15]) {
    const cpx *p = reinterpret_cast<const cpx *>(short_row(reg, lane >> 2)) + 15 * (lane & 3);
#pragma unroll
    for (int k1 = 0; k1 < 15; k1++) v[k1] = p[k1];
}
NYQ_HD void xf_short_t3_store(int lane, float *const (&reg)[2], cpx (&v)[15]) {
    cpx *row = reinterpret_cast<cpx *>(short_row(reg, lane >> 2));
    Dft<15>::run(v);
    const int base = (45 * (lane & 3)) % 60;
#pragma unroll
    for (int n1 = 0; n1 < 15; n1++) {
        int n = base + (16 * n1) % 60;
        if (n >= 60) n -= 60;
        row[n] = v[n1];
    }
}

// T4: post-rotation of sub-iteration s, in place (task j reads and writes floats 4j.. and 116-4j.. of its row);
// returns the row's raw[116-4j..119-4j] (the lanes of block 7 keep it: the frame's tail)
NYQ_HD f4 xf_short_t4(const XfShortConst &S, int lane, int s, float *const (&reg)[2]) {
    const int j = lane & 15;
    if (j >= 15) return f4{0, 0, 0, 0};
    float *rowf = short_row(reg, 4 * s + (lane >> 4));
    const f4 P01 = *reinterpret_cast<const f4 *>(rowf + 4 * j);
    const f4 P23 = *reinterpret_cast<const f4 *>(rowf + 116 - 4 * j);
    f4 F, Bk;
    xf_postrot4(P01, P23, S.ts, kSineShort, F, Bk);
    *reinterpret_cast<f4 *>(rowf + 4 * j) = F;
    *reinterpret_cast<f4 *>(rowf + 116 - 4 * j) = Bk;
    return Bk;
}
// which lane holds a channel's tail after T4 (block 7 sits in the last 16-lane slot of sub-iterations 1 and 3), for the
// lanes j < 15 that keep it
NYQ_HD int short_tail_src(int lane) { return 48 + (lane & 15); }

// T5: every block's 120 samples = TDAC mirror of its raw first half against the previous block's raw second half
// (block 0: the register tail).  One call = channel c, blocks 4h .. 4h+3; call h = 1 BEFORE h = 0 (h = 1 reads block
// 3's second half, which h = 0 overwrites).  Reads and writes are two halves (all lanes read before any writes).
struct XfMirror {
    f4 F, C;
};
NYQ_HD void xf_short_t5_load(int lane, int c, int h, float *const (&reg)[2], const f4 &tail, XfMirror &M) {
    const int j = (lane & 15) < 15 ? (lane & 15) : 0;
    const int blk = 4 * h + (lane >> 4);
    const float *r = reg[c];
    M.F = *reinterpret_cast<const f4 *>(r + 120 * blk + 4 * j);
    M.C = blk > 0 ? *reinterpret_cast<const f4 *>(r + 120 * (blk - 1) + 116 - 4 * j) : tail;   // (blk 0 <=> lane j: `tail` = the channel's)
}
NYQ_HD void xf_short_t5_store(const XfWin &K, int lane, int c, int h, float *const (&reg)[2], const XfMirror &M) {
    const int j = lane & 15;
    if (j >= 15) return;
    const int blk = 4 * h + (lane >> 4);
    float *r = reg[c];
    f4 hi, lo;
    xf_tdac(K, M.F, M.C, hi, lo);
    *reinterpret_cast<f4 *>(r + 120 * blk + 60 + 4 * j) = hi;
    *reinterpret_cast<f4 *>(r + 120 * blk + 56 - 4 * j) = lo;
}

// ---- overlap state <-> tail registers: lane j < 15 holds state[c][56-4j .. 59-4j] of both channels --------------
NYQ_HD bool tail_lane(int lane) { return lane < 15; }
NYQ_HD int tail_offset(int lane, int c) { return c * kHalfOv + 56 - 4 * lane; }

}  // namespace fx
}  // namespace nyq
