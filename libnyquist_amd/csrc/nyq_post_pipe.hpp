// nyq_post_pipe.hpp -- the post-filter stage as a WORKGROUP PIPELINE (round 2).
//
// Same operator as nyq_post_kernels.hpp (comb_filter celt.c:114-172 as applied by
// celt_decoder_clean.c:658-683, then deemphasis() :192-256 with scaling and channel interleave), other
// execution shape.  The comb filter is a recursion along time: ONE chain per (stream, channel), a few
// thousand chains per job, so a chain's own instruction stream is what bounds the stage (measured round 1:
// 874 wave-instructions per channel-frame on the chain's wave, two waves per SIMD, 13 k cycles per frame).
// Here a chain's wave issues nothing but the recursion; everything else moves to a helper wave of the
// same workgroup that runs beside it:
//
//   workgroup = 2 chains (units 2p, 2p+1 = the two channels of a stereo stream, or two mono streams, ...)
//     wave 0, 1   comb wave of chain 0 / 1: per frame, the comb steps of that frame and nothing else
//     wave 2      I/O wave for both chains: global prefetch of frame f+2, raw frame f+1 into LDS,
//                 de-emphasis + interleave + global store of frame f-1, post-filter parameters of
//                 frame f+1 into an LDS slot, carry-over of the history that survives
//   one s_barrier per frame.
//
// LDS per chain: two buffers A, B of [1040 filtered history | frame] floats (kPipeHist).  The comb wave filters
// frame f in place in `cur` and writes every output a second time into `nxt`'s history region
// (nxt[idx - N]); meanwhile the I/O wave fills the rest of `nxt` (the older part of the history from
// `cur`, the raw frame f+1) -- so the next frame starts right after the barrier, nothing is moved on the
// chain's critical path, and every tap address is the output's address minus a wave-uniform constant.
// The de-emphasis of frame f-1 reads cur's history region (final since the previous barrier) and stages
// its interleaved output in nxt's frame region before the raw frame f+1 lands there.
// 32 544 B of LDS and at most 128 VGPRs per workgroup: four workgroups (8 chains, 12 waves) per CU, with a fifth
// workgroup's LDS and a fourth wave's registers per SIMD to spare -- the workgroups live for the whole launch and must
// all be placed at once (kPipeHist, and the kernel's launch bounds, say why).
#pragma once
#include "nyq_post_common.hpp"

namespace nyq {

constexpr int kPipeWaves = 3;          // waves per workgroup: 2 comb + 1 I/O

// Diagnostic build only (tools/chainbench.hip, -DNYQ_PIPE_STAMPS): per-role cycle accounting with s_memtime, summed
// into a buffer of its own that nothing else reads.  The product build compiles none of it.
#ifdef NYQ_PIPE_STAMPS
__device__ unsigned long long g_pipe_stamps[16];
#define NYQ_STAMP_DECL() unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0}
#define NYQ_STAMP(slot)                                              \
    do {                                                             \
        const unsigned long long st_n = __builtin_amdgcn_s_memtime(); \
        st_acc[slot] += st_n - st_t;                                 \
        st_t = st_n;                                                 \
    } while (0)
#define NYQ_STAMP_FLUSH(base)                                                        \
    do {                                                                             \
        if (lane == 0)                                                               \
            for (int st_i = 0; st_i < 6; st_i++) atomicAdd(&g_pipe_stamps[(base) + st_i], st_acc[st_i]); \
    } while (0)
__device__ unsigned long long g_stamp_mid_t;   // (unused placeholder: keeps the macro below self-contained)
// placement trace: per workgroup {HW_ID, XCC_ID, first cycle, last cycle} (which CU it ran on, when it started / ended)
__device__ unsigned long long g_pipe_wg[4096][4];
#define NYQ_WG_TRACE(slot)                                                                                \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                      \
            if ((slot) == 2) {                                                                            \
                g_pipe_wg[blockIdx.x][0] = __builtin_amdgcn_s_getreg(63492);                              \
                g_pipe_wg[blockIdx.x][1] = __builtin_amdgcn_s_getreg(63508);                              \
            }                                                                                             \
            g_pipe_wg[blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime();                               \
        }                                                                                                 \
    } while (0)
#define NYQ_STAMP_MID()                                                  \
    do {                                                                 \
        if (stm) {                                                       \
            const unsigned long long st_n = __builtin_amdgcn_s_memtime(); \
            stm[1] += st_n - stm[0];                                     \
            stm[0] = st_n;                                               \
        }                                                                \
    } while (0)
#else
#define NYQ_STAMP_MID() do { } while (0)
#define NYQ_WG_TRACE(slot) do { } while (0)
#define NYQ_STAMP_DECL() do { } while (0)
#define NYQ_STAMP(slot) do { } while (0)
#define NYQ_STAMP_FLUSH(base) do { } while (0)
#endif
constexpr int kPipeUnits = 2;          // chains per workgroup
// Streaming hints of the I/O wave's global accesses: bit 0 loads, bit 1 stores.  Every byte is touched once.  Measured in one
// process against the plain forms (tools/variant_ab.py, profiles/r03_c_*): both hints together 0.6-0.9 % faster on the
// stage, 0.7-1.7 % on the chain, in all six cases; either one alone 0.3-1.5 % SLOWER.  Kept at both.
#ifndef NYQ_PIPE_NT
#define NYQ_PIPE_NT 3
#endif
__device__ __forceinline__ vf4 pipe_ld(const vf4 *p) {
    if (NYQ_PIPE_NT & 1) return __builtin_nontemporal_load(p);
    return *p;
}
__device__ __forceinline__ void pipe_st(vf4 *p, const vf4 &v) {
    if (NYQ_PIPE_NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// History in front of the frame inside an LDS buffer: >= COMBFILTER_MAXPERIOD + 2 + 3 (the 16-byte tap reads start up to three
// floats early) = 1029, and small enough that a workgroup's LDS (4 buffers of 1040 + 960 floats + 544 B = 32544 B) stays under
// 32 KB = a FIFTH of the CU's 160 KB although only four workgroups per CU are wanted: with the 1088 floats of the state
// hand-over (33312 B) four fit only when the CU's LDS is handed out from offset 0, and after other kernels it is not -- some
// CUs then take three workgroups for the whole launch (measurements: see the launch bounds of the kernel below).
constexpr int kPipeHist = 1040;

// copy cur[i0, i0+n) -> mir[i0, i0+n) (n, i0 multiples of 4), whole wave
template <bool MIR>
__device__ __forceinline__ void pipe_copy(const float *cur, float *mir, int lane, int i0, int n) {
    if (!MIR) return;                                          // single-buffer callers filter in place: nothing to copy
#pragma unroll 2
    for (int v = lane; v < n / 4; v += kWave) sts4(mir, i0 + 4 * v, *reinterpret_cast<const vf4 *>(cur + i0 + 4 * v));
}

// The constant part of a comb_filter() call (comb_filter_const, celt.c:87-110) for one tap alignment AL: outputs
// [i0, i0+n) of `ring`, w1 per step (four adjacent ones per lane), every output also stored at mir[idx].
template <bool MIR, int AL>
__device__ __forceinline__ void pipe_const_wide(float *ring, float *mir, int lane, int i0, int n, int T1, float g10,
                                                float g11, float g12, int w1) {
    const int o = 4 * lane;
    int idx = i0 + o;
    int rb = idx - T1 - 2 - AL;
    auto step = [&]() {
        const f4 q0 = lds4(ring, rb), q1 = lds4(ring, rb + 4), q2 = lds4(ring, rb + 8), cen = lds4(ring, idx);
        float x[8];
        pick8<AL>(q0, q1, q2, x);
        float y[4] = {cen.x, cen.y, cen.z, cen.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            y[u] += g10 * x[u + 2];
            y[u] += g11 * (x[u + 3] + x[u + 1]);
            y[u] += g12 * (x[u + 4] + x[u]);
        }
        sts4(ring, idx, f4{y[0], y[1], y[2], y[3]});
        if (MIR) sts4(mir, idx, f4{y[0], y[1], y[2], y[3]});
    };
    const int nfull = n / w1, rem = n - nfull * w1;
    if (o < w1) {
        for (int sidx = 0; sidx < nfull; sidx++) {
            step();
            idx += w1;
            rb += w1;
            NYQ_POST_SYNC();
        }
    }
    if (o < rem) step();
    NYQ_POST_SYNC();
}

// ---- lane shifts without LDS (gfx9 DPP) ---------------------------------------------------------------
// dpp_shr1(old, v): lane l >= 1 receives v[l-1], lane 0 keeps old[0]  (v_mov_b32_dpp wave_shr:1, bound_ctrl:0)
__device__ __forceinline__ float dpp_shr1(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_zero(float v) {   // lanes without a source (or rows masked off) receive 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, ROWMASK == 0xf));
}

// A run of n outputs ring[i0, i0+n) of ONE period T with w = T-2 <= 60 outputs per step, the recursion kept in
// REGISTERS.  What bounds a step is the latency of its dependent chain (three DPP moves, a DPP add, an FMA, the carry move:
// about 110 cycles with no LDS instruction in the loop, measured) plus the waits on its two LDS loads; the layout keeps
// both the chain and the instruction count (11 VALU per constant-filter step) as short as the one-lane DPP shift allows:
//   * lane L >= 4 owns output L-4 of a step; lanes 0..3 of the register Y that holds the previous step's outputs carry the
//     four outputs in FRONT of them.  The five taps y[n-T+2 .. n-T-2] of every owning lane are then Y and four plain
//     one-lane shifts of it (v_mov_b32_dpp wave_shr:1, lane 0 <- 0): no lane has to be fed from elsewhere.
//   * the four carry lanes of the NEXT step are the last four outputs of the previous one, final since its write: one
//     ds_read_b32 issued right behind that write (lanes 0..3 read ring[base-4 ..]; the same address expression as the
//     raw-sample read of that step, which the other lanes issued a step earlier) and ONE DPP move (row 0, bank 0) that
//     drops them into lanes 0..3 at the end of the step.  Lanes 0..3 run the arithmetic with zero gains and store what they
//     hold: they rewrite four outputs with their own values (also where the I/O wave copies or reads the same words
//     meanwhile: same values), so no lane mask changes inside the loop.
//   * raw samples and window weights are read two steps ahead into alternating register sets.
// 11 VALU instructions per constant-filter step (the first register form: 22); outputs still go to ring[] and mir[].
//   XF = 0: comb_filter_const (celt.c:87-110), gains ga[0..2]
//   XF = 1: the cross-fade loop of comb_filter (celt.c:139-160) when BOTH tap sets have this period: window
//           weights from win2[i], set 0 gains ga, set 1 gains gb (same order of operations as the LDS form)
//   XF = 2: the cross-fade loop when set 0 is switched off (gain 0: the filter fades IN on set 1's period T)
//   XF = 3: the cross-fade loop when set 1 is switched off (the filter fades OUT on set 0's period T)
constexpr int kDppMaxW = kWave - 4;

__device__ __forceinline__ float dpp_shr1z(float v) {            // lane l >= 1 receives v[l-1], lane 0 receives 0
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_low4(float keep, float v) {  // lanes 0..3 receive v's, every other lane keeps `keep`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), 0xE4, 0x1, 0x1, false));
}

template <bool MIR, int XF>
__device__ __forceinline__ void pipe_run_dpp(float *ring, float *mir, int lane, int i0, int n, int T, const float (&ga)[3],
                                             const float (&gb)[3], const float *win2) {
    const int w = T - 2;
    const int nfull = n / w, rem = n - nfull * w;
    float *rc = ring + (i0 + lane - 4);
    float *mc = mir + (i0 + lane - 4);
    const float *wp = win2 + (lane - 4);
    // per-lane gains: zero in the carry lanes
    const float own = lane >= 4 ? 1.f : 0.f;
    const float a0 = ga[0] * own, a1 = ga[1] * own, a2 = ga[2] * own, b0 = gb[0] * own, b1 = gb[1] * own, b2 = gb[2] * own;
    float Y = 0.f, xa = 0.f, fa = 0.f, xb = 0.f, fb = 0.f;
    if (lane < w + 4) {
        Y = rc[-w];
        xa = rc[0];                                            // (lanes 0..3: the four outputs in front of the run)
        xb = rc[w];
        if (XF) { fa = wp[0]; fb = wp[w]; }
    }
    float Ra = xa, Rb = 0.f;
    auto taps = [&](float &y, float f, float S1, float S2, float S3, float S4) {
        if (XF == 1 || XF == 3) {
            const float nf = 1.0f - f;
            y += (nf * a0) * S2;
            y += (nf * a1) * (S1 + S3);
            y += (nf * a2) * (Y + S4);
        }
        if (XF == 1 || XF == 2) {
            y += (f * b0) * S2;
            y += (f * b1) * (S1 + S3);
            y += (f * b2) * (Y + S4);
        }
        if (XF == 0) {
            y += a0 * S2;
            y += a1 * (S1 + S3);
            y += a2 * (Y + S4);
        }
    };
    // one step at offset `o` floats from rc (o = 0 or w): x = its raw samples, f = its window weights, R = (lanes 0..3) the
    // last four outputs of the step before it; returns with Y = the next step's input register
    auto step = [&](int o, float x, float f, float R, float &Rnext) {
        const float S1 = dpp_shr1z(Y), S2 = dpp_shr1z(S1), S3 = dpp_shr1z(S2), S4 = dpp_shr1z(S3);
        float y = x;
        taps(y, f, S1, S2, S3, S4);
        y = dpp_low4(y, R);
        rc[o] = y;
        Rnext = rc[o + w];                                     // lanes 0..3: the last four outputs just written (first in
        if (MIR) mc[o] = y;                                    // the LDS queue behind that write: it is the next step's
        Y = y;                                                 // only LDS dependency)
    };
    int done = 0;
    if (lane < w + 4) {
        for (; done + 2 <= nfull; done += 2) {
            step(0, xa, fa, Ra, Rb);
            xa = rc[2 * w];                                    // (reads up to two steps past the run: inside the LDS, unused)
            if (XF) fa = wp[2 * w];
            NYQ_POST_SYNC();
            step(w, xb, fb, Rb, Ra);
            xb = rc[3 * w];
            if (XF) fb = wp[3 * w];
            NYQ_POST_SYNC();
            rc += 2 * w; mc += 2 * w; wp += 2 * w;
        }
        // (tried on the first register form: peeling the first pair so that the s_waitcnt at the loop top is exact, plus
        // scheduling barriers between the two steps -- precise waits, 2 % slower; and here: the carry lanes merged at the
        // START of the next step with the stores under a lane mask instead of the rewrite -- a branch per step, 7 % slower)
        if (done < nfull) {                                    // odd count: one more whole step, then b is the next set
            step(0, xa, fa, Ra, Rb);
            NYQ_POST_SYNC();
            rc += w; mc += w; wp += w;
            xa = xb; fa = fb;
            Ra = Rb;
        }
    }
    // remainder: fewer than w outputs (the shifts only read lanes below the last owning one: all enabled)
    if (rem > 0 && lane < rem + 4) step(0, xa, fa, Ra, Rb);
    NYQ_POST_SYNC();
}

// One comb_filter() call (celt.c:114-172) on ring[r0, r0+n): cross-fade (T0,g0,ts0) -> (T1,g1,ts1) over the first 120
// samples, constant after; every output y[idx] is stored at ring[idx] AND mir[idx].
template <bool MIR>
__device__ __forceinline__ void pipe_comb_call(float *ring, float *mir, int lane, int r0, int n, int T0, int T1,
                                               float g0, float g1, int ts0, int ts1, const float *win2,
                                               unsigned long long *stm = nullptr) {
    if (g0 == 0.f && g1 == 0.f) {                             // celt.c:126-132: y = x
        pipe_copy<MIR>(ring, mir, lane, r0, n);
        return;
    }
    float g00, g01, g02, g10, g11, g12;
    comb_gains(g0, ts0, g00, g01, g02);
    comb_gains(g1, ts1, g10, g11, g12);
    // Outputs i .. i+w-1 are independent when w <= T-2 for every ACTIVE tap set (a switched-off side may carry any
    // period, even 0: the reference multiplies those taps by zero, here they are skipped and do not bound w).
    const int o = 4 * lane;
    const float ga[3] = {g00, g01, g02}, gb[3] = {g10, g11, g12};
    int tmin = 4 * kWave + 2;
    if (g0 != 0.f && T0 < tmin) tmin = T0;
    if (g1 != 0.f && T1 < tmin) tmin = T1;
    if (T0 == T1 && g0 == g1 && ts0 == ts1 && T1 - 2 <= kDppMaxW) {
        // the same tap set on both sides -- a frame's first 120 samples from the second frame on (celt_decoder_clean.c:678-683
        // makes old = current).  The reference still evaluates the cross-fade expression (celt.c:139-160), whose weights
        // (1-f) g + f g add up to the constant filter's g: run as comb_filter_const, equal to it up to rounding (the
        // parity tests hold the chain to 1e-5 against the oracle, which cross-fades) at half the arithmetic per step
        pipe_run_dpp<MIR, 0>(ring, mir, lane, r0, kOverlap, T1, gb, gb, win2);
    } else if (T0 == T1 && g0 == g1 && ts0 == ts1) {
        // the same, long period: four adjacent outputs per lane (one step covers the 120 samples from period 122 on)
        const int w1 = (T1 - 2 < 4 * kWave ? T1 - 2 : 4 * kWave) & ~3;
        switch ((r0 - T1 - 2) & 3) {
            case 0: pipe_const_wide<MIR, 0>(ring, mir, lane, r0, kOverlap, T1, g10, g11, g12, w1); break;
            case 1: pipe_const_wide<MIR, 1>(ring, mir, lane, r0, kOverlap, T1, g10, g11, g12, w1); break;
            case 2: pipe_const_wide<MIR, 2>(ring, mir, lane, r0, kOverlap, T1, g10, g11, g12, w1); break;
            default: pipe_const_wide<MIR, 3>(ring, mir, lane, r0, kOverlap, T1, g10, g11, g12, w1); break;
        }
    } else if (T0 == T1 && g0 != 0.f && g1 != 0.f && T1 - 2 <= kDppMaxW) {
        // both tap sets on one short period (always the case for the first 120 samples of a frame once the filter
        // runs: celt_decoder_clean.c:678-683 makes old = current): recursion in registers
        pipe_run_dpp<MIR, 1>(ring, mir, lane, r0, kOverlap, T1, ga, gb, win2);
    } else if (g0 == 0.f && T1 - 2 <= kDppMaxW) {
        pipe_run_dpp<MIR, 2>(ring, mir, lane, r0, kOverlap, T1, ga, gb, win2);   // fading in: only set 1 has taps
    } else if (g1 == 0.f && T0 - 2 <= kDppMaxW) {
        pipe_run_dpp<MIR, 3>(ring, mir, lane, r0, kOverlap, T0, ga, gb, win2);   // fading out: only set 0 has taps
    } else if (tmin - 2 <= kWave) {
        // short periods: one output per lane
        const int w = tmin - 2;
        const int nfull = kOverlap / w, rem = kOverlap - nfull * w;
        float *rc = ring + (r0 + lane);
        float *mc = mir + (r0 + lane);
        const float *wp = win2 + lane;
        auto step = [&]() {
            const float f = *wp;
            const float nf = 1.0f - f;
            float y = rc[0];
            if (g0 != 0.f) {
                const float *t = rc - T0 - 2;
                const float x0 = t[0], x1 = t[1], x2 = t[2], x3 = t[3], x4 = t[4];
                y += (nf * g00) * x2;
                y += (nf * g01) * (x3 + x1);
                y += (nf * g02) * (x4 + x0);
            }
            if (g1 != 0.f) {
                const float *t = rc - T1 - 2;
                const float x0 = t[0], x1 = t[1], x2 = t[2], x3 = t[3], x4 = t[4];
                y += (f * g10) * x2;
                y += (f * g11) * (x3 + x1);
                y += (f * g12) * (x4 + x0);
            }
            rc[0] = y;
            if (MIR) mc[0] = y;
        };
        if (lane < w) {
            for (int sidx = 0; sidx < nfull; sidx++) {
                step();
                rc += w;
                mc += w;
                wp += w;
                NYQ_POST_SYNC();
            }
        }
        if (lane < rem) step();
        NYQ_POST_SYNC();
    } else {
        const int w = (tmin - 2) & ~3;
        for (int base = 0; base < kOverlap; base += w) {
            if (o < w && base + o < kOverlap) {
                const int idx = r0 + base + o;
                const f4 fw = lds4(win2, base + o);
                const float f[4] = {fw.x, fw.y, fw.z, fw.w};
                const f4 cen = lds4(ring, idx);
                float y[4] = {cen.x, cen.y, cen.z, cen.w};
                if (g0 != 0.f) {
                    float x[1][8];
                    taps8<1>(ring, idx - T0 - 2, x);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float nf = 1.0f - f[u];
                        y[u] += (nf * g00) * x[0][u + 2];
                        y[u] += (nf * g01) * (x[0][u + 3] + x[0][u + 1]);
                        y[u] += (nf * g02) * (x[0][u + 4] + x[0][u]);
                    }
                }
                if (g1 != 0.f) {
                    float x[1][8];
                    taps8<1>(ring, idx - T1 - 2, x);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        y[u] += (f[u] * g10) * x[0][u + 2];
                        y[u] += (f[u] * g11) * (x[0][u + 3] + x[0][u + 1]);
                        y[u] += (f[u] * g12) * (x[0][u + 4] + x[0][u]);
                    }
                }
                sts4(ring, idx, f4{y[0], y[1], y[2], y[3]});
                if (MIR) sts4(mir, idx, f4{y[0], y[1], y[2], y[3]});
            }
            NYQ_POST_SYNC();
        }
    }
    const int i0 = r0 + kOverlap, nc = n - kOverlap;
    if (nc <= 0) return;
    NYQ_STAMP_MID();
    if (g1 == 0.f) {                                          // celt.c:163-169
        pipe_copy<MIR>(ring, mir, lane, i0, nc);
        return;
    }
    if (T1 - 2 <= kDppMaxW) {
        // short periods (the common case in real streams): one output per lane, recursion in registers
        pipe_run_dpp<MIR, 0>(ring, mir, lane, i0, nc, T1, gb, gb, win2);
        return;
    }
    const int w1 = (T1 - 2 < 4 * kWave ? T1 - 2 : 4 * kWave) & ~3;
    switch ((i0 - T1 - 2) & 3) {
        case 0: pipe_const_wide<MIR, 0>(ring, mir, lane, i0, nc, T1, g10, g11, g12, w1); break;
        case 1: pipe_const_wide<MIR, 1>(ring, mir, lane, i0, nc, T1, g10, g11, g12, w1); break;
        case 2: pipe_const_wide<MIR, 2>(ring, mir, lane, i0, nc, T1, g10, g11, g12, w1); break;
        default: pipe_const_wide<MIR, 3>(ring, mir, lane, i0, nc, T1, g10, g11, g12, w1); break;
    }
}

// ---- de-emphasis of one frame (celt_decoder_clean.c:243-248) -----------------------------------------
// t[j] = a[j] + c t[j-1] over N = NL * CH samples: lane l < NL runs the recurrence over its own CH consecutive
// samples with a zero carry-in, ONE weighted wavefront scan with ratio q = c^CH turns the lane-end values into the
// true ones, and each lane then adds c^k times the value entering its chunk.  The scan runs on DPP lane shifts
// (row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast 15 and 31 across them): six VALU steps, no LDS.
template <int N>
struct DeGeo {
    static constexpr int CH = (N + kWave - 1) / kWave;   // 15, 8, 4, 2
    static constexpr int NL = N / CH;                    // 64, 60, 60, 60
    static_assert(NL * CH == N, "frame splits evenly over the lanes");
};

struct DeConst {
    float cstep[4];   // q^(2^k)
    float wA;         // q^((lane & 15) + 1): weight of the previous row's last value
    float wB;         // q^(lane - 31)      : weight of lane 31's value in rows 2, 3
    float pw;         // q^lane
    float pwEnd;      // q^NL
};

// c^j, j = 0 .. 15 (c = the de-emphasis coefficient), as compile-time constants
struct DeemphPow {
    float p[16];
    constexpr DeemphPow() : p{} {
        float v = 1.f;
        for (int j = 0; j < 16; j++) {
            p[j] = v;
            v *= kPreemph;
        }
    }
};
constexpr DeemphPow kDeemphPow{};

template <int N>
__device__ __forceinline__ void deemph_init(DeConst &D, int lane) {
    constexpr int CH = DeGeo<N>::CH, NL = DeGeo<N>::NL;
    float q = 1.f;
#pragma unroll
    for (int k = 0; k < CH; k++) q *= kPreemph;
    float st[7];
    st[0] = q;
#pragma unroll
    for (int k = 1; k < 7; k++) st[k] = st[k - 1] * st[k - 1];
    auto powq = [&](int e) {                                   // q^e, 0 <= e < 128
        float r = 1.f;
#pragma unroll
        for (int k = 0; k < 7; k++)
            if (e & (1 << k)) r *= st[k];
        return r;
    };
#pragma unroll
    for (int k = 0; k < 4; k++) D.cstep[k] = st[k];
    D.wA = powq((lane & 15) + 1);
    D.wB = powq(lane >= 32 ? lane - 31 : 0);
    D.pw = powq(lane);
    D.pwEnd = powq(NL);
}

// K frames side by side (their instruction streams interleave): src[k] the filtered frame (LDS, N floats),
// stage[k] N floats of LDS that receive the scaled samples, mem[k] = coef0 * (last output), celt_decoder_clean.c:246
template <int N, int K>
__device__ __forceinline__ void deemph_frames(const float *const (&src)[K], float *const (&stage)[K], float (&mem)[K],
                                              const bool (&on)[K], int lane, const DeConst &D) {
    constexpr int CH = DeGeo<N>::CH, NL = DeGeo<N>::NL;
    float loc[K][CH];
    const int li = lane < NL ? lane : 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!on[k]) continue;
        if constexpr (CH % 4 == 0) {
#pragma unroll
            for (int j = 0; j < CH; j += 4) {
                const f4 q = lds4(src[k], li * CH + j);
                loc[k][j] = q.x; loc[k][j + 1] = q.y; loc[k][j + 2] = q.z; loc[k][j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CH; j++) loc[k][j] = src[k][li * CH + j];
        }
    }
    // lane-local recurrence with a zero carry-in, run as TWO independent half chains (samples [0, H) and [H, CH)) that are
    // joined afterwards (hi[j] += c^(j-H+1) lo[H-1]): the dependent chain of the I/O wave is half as long
    constexpr int H = CH / 2;
    float e[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        float lo = 0.f, hi = 0.f;
        if (on[k]) {
#pragma unroll
            for (int j = 0; j < CH - H; j++) {
                if (j < H) {
                    lo = (loc[k][j] + 1e-30f) + kPreemph * lo;    // + VERY_SMALL
                    loc[k][j] = lo;
                }
                hi = (loc[k][H + j] + 1e-30f) + kPreemph * hi;
                loc[k][H + j] = hi;
            }
            if (H > 0) {
#pragma unroll
                for (int j = H; j < CH; j++) loc[k][j] += kDeemphPow.p[j - H + 1] * lo;
            }
        }
        e[k] = lane < NL ? loc[k][CH - 1] : 0.f;               // e[l] -> sum_{i<=l} q^(l-i) acc[i]
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        e[k] += D.cstep[0] * dpp_zero<0x111, 0xf>(e[k]);      // row_shr:1
        e[k] += D.cstep[1] * dpp_zero<0x112, 0xf>(e[k]);      // row_shr:2
        e[k] += D.cstep[2] * dpp_zero<0x114, 0xf>(e[k]);      // row_shr:4
        e[k] += D.cstep[3] * dpp_zero<0x118, 0xf>(e[k]);      // row_shr:8
        e[k] += D.wA * dpp_zero<0x142, 0xa>(e[k]);            // row_bcast:15 into rows 1, 3
        e[k] += D.wB * dpp_zero<0x143, 0xc>(e[k]);            // row_bcast:31 into rows 2, 3
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!on[k]) continue;
        // value entering lane l's chunk: c t[l CH - 1] = c e[l-1] + q^l mem, with mem = c t[-1]
        const float prevEnd = dpp_shr1(0.f, e[k]);
        const float cp = kPreemph * prevEnd + D.pw * mem[k];   // (lane 0: prevEnd = 0, pw = 1)
#pragma unroll
        for (int j = 0; j < CH; j++) loc[k][j] = (loc[k][j] + cp * kDeemphPow.p[j]) * (1.f / 32768.f);   // (independent of each other)
        if (lane < NL) {
            if constexpr (CH % 4 == 0) {
#pragma unroll
                for (int j = 0; j < CH; j += 4) sts4(stage[k], lane * CH + j, f4{loc[k][j], loc[k][j + 1], loc[k][j + 2], loc[k][j + 3]});
            } else {
#pragma unroll
                for (int j = 0; j < CH; j++) stage[k][lane * CH + j] = loc[k][j];
            }
        }
        mem[k] = kPreemph * __shfl(e[k], NL - 1) + D.pwEnd * mem[k];   // c t[N-1]
    }
}

// post-filter parameters of one frame as the I/O wave hands them to the comb wave (LDS slot of 4 dwords)
struct PipeParams {
    int T;
    float g;
    int ts;
    int pad;
};

// Four waves per SIMD, i.e. at most 128 VGPRs -- NOT for occupancy (the LDS admits four workgroups = 12 waves per CU, three
// per SIMD) but for PLACEMENT, like kPipeHist above: a workgroup lives for the whole launch, so one that cannot be placed at
// once waits ~0.8 ms for a retiring one and holds up every workgroup queued behind it on its XCD.  Measured with
// tools/placement_trace.py (s_memrealtime + HW_ID per workgroup, behind synthesis kernels / fills / itself):
//   168 VGPRs (3 waves fill a SIMD's file exactly), 33312 B LDS (4 fill the CU's LDS but for 30 KB): 4-40 of 1024
//        workgroups 0.72-0.85 ms late in most launches that follow another kernel -> 1.6 instead of 0.95 ms
//   121 VGPRs, 33312 B: the same;   137 VGPRs, 32544 B: the same;   119 VGPRs, 32544 B (this build): none, any predecessor.
// Both allocators evidently hand out space from wherever the previous kernels left off, so "exactly four fit" only holds
// from offset zero; with one more wave's registers and one more workgroup's LDS to spare four always fit.
#ifndef NYQ_PIPE_MINWAVES
#define NYQ_PIPE_MINWAVES 4
#endif
#pragma clang diagnostic ignored "-Wpass-failed"   // ("occupancy target 4, final occupancy 3": the LDS bound, intended -- see above)
// PAIR: stereo streams (channels == 2, so both units of every workgroup exist and are the channels of one stream): the
// I/O wave's frame loop then carries none of the per-iteration tests for missing units and other channel counts
template <int LM, bool PAIR = false>
#ifdef NYQ_PIPE_NUM_VGPR
__attribute__((amdgpu_num_vgpr(NYQ_PIPE_NUM_VGPR)))
#endif
__global__ __launch_bounds__(kWave *kPipeWaves, NYQ_PIPE_MINWAVES) void celt_post_pipe_kernel(PostArgs A, const float *__restrict__ window) {
    constexpr int N = 120 << LM;
    constexpr int NV = N / 4;
    constexpr int NLD = (NV + kWave - 1) / kWave;   // float4 per lane and frame (4, 2, 1, 1)
    constexpr int R0 = kPipeHist;                   // frame start inside a buffer
    constexpr int KEEP = kPipeHist - N;             // history samples that survive a frame: cur[N, R0) -> nxt[0, KEEP)
    constexpr int OLD = kPostHist - kPipeHist;      // samples of the handed-over history (1088) that are older than the buffer's
    static_assert(N >= OLD, "the previous buffer still holds the history the state hand-over needs");
    __shared__ __attribute__((aligned(16))) float bufs[kPipeUnits][2][kPipeHist + N];
    __shared__ __attribute__((aligned(16))) float win2[kOverlap];
    __shared__ __attribute__((aligned(16))) PipeParams pslot[kPipeUnits][2];
    for (int i = threadIdx.x; i < kOverlap; i += kWave * kPipeWaves) win2[i] = window[i] * window[i];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long nunits = A.nstreams * A.channels;
    const long npairs = (nunits + kPipeUnits - 1) / kPipeUnits;
    const long nfr = A.nframes;
    const long pst = A.ps();                       // frames per stream in pf_* and out (a window of longer arrays, or nfr)
    NYQ_WG_TRACE(2);
    __syncthreads();

    // The two roles are two separate loops over the same workgroup units (the register allocator then sees two
    // disjoint live ranges); both execute the same number of barriers per unit: one after the I/O wave's prologue,
    // one per frame, one closing iteration -- the last also separates this unit's LDS use from the next prologue.
#ifndef NYQ_PIPE_DBG_NO_COMB
    if (wave < kPipeUnits) {
#ifndef NYQ_PIPE_NO_PRIO
        __builtin_amdgcn_s_setprio(3);                         // the chain's wave wins instruction arbitration on its SIMD
#endif
        for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
            const long u0 = pair * kPipeUnits;
            // ---------------- comb wave of unit u0 + wave ----------------
            const long u = u0 + wave;
            const bool live = u < nunits;
            const long s = live ? u / A.channels : 0;
            const int c = (int)(u - s * A.channels);
            float *bA = bufs[wave][0], *bB = bufs[wave][1];
            int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
            float g_old = 0.f, g_cur = 0.f;
            if (live && A.pf_state) {
                const float *ps = A.pf_state + 6 * s;
                T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
            }
            __syncthreads();                                   // prologue of the I/O wave done
            NYQ_STAMP_DECL();
            for (long f = 0; f <= nfr; f++) {
                NYQ_STAMP(1);                                  // slot 1: waiting at the barrier
                if (live && f < nfr) {
                    float *cur = (f & 1) ? bB : bA, *nxt = (f & 1) ? bA : bB;
                    float *mir = nxt - N;                      // mir[idx] = nxt[idx - N]
                    const PipeParams P = pslot[wave][f & 1];
                    const int T_new = __builtin_amdgcn_readfirstlane(P.T), ts_new = __builtin_amdgcn_readfirstlane(P.ts);
                    const float g_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P.g)));
                    if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;   // celt_decoder_clean.c:661-662
                    if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
                    if (T_cur > kCombMaxPeriod) T_cur = kCombMaxPeriod;   // (a decoder never produces more: keeps taps in the buffer)
                    if (T_old > kCombMaxPeriod) T_old = kCombMaxPeriod;
                    const int T_nw = T_new < kCombMinPeriod ? kCombMinPeriod : T_new > kCombMaxPeriod ? kCombMaxPeriod : T_new;
                    NYQ_STAMP(2);                              // slot 2: parameters, set-up
                    pipe_comb_call<true>(cur, mir, lane, R0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
                    NYQ_STAMP(3);                              // slot 3: first 120 samples (old -> current parameters)
#ifdef NYQ_PIPE_STAMPS
                    unsigned long long stm[2] = {st_t, 0};
                    if (LM != 0)
                        pipe_comb_call<true>(cur, mir, lane, R0 + kOverlap, N - kOverlap, T_cur, T_nw, g_cur, g_new, ts_cur, ts_new, win2, stm);
                    st_acc[4] += stm[1];                       // slot 4: cross-fade of the second call; slot 0: its constant part
                    st_t = stm[0];
#else
                    if (LM != 0)
                        pipe_comb_call<true>(cur, mir, lane, R0 + kOverlap, N - kOverlap, T_cur, T_nw, g_cur, g_new, ts_cur, ts_new, win2);
#endif
                    T_old = T_cur; g_old = g_cur; ts_old = ts_cur;       // :672-677
                    T_cur = T_new; g_cur = g_new; ts_cur = ts_new;
                    if (LM != 0) { T_old = T_cur; g_old = g_cur; ts_old = ts_cur; }   // :678-683
                }
                NYQ_STAMP(0);                                  // slot 0: the comb steps of the frame
                __syncthreads();
            }
            if (wave == 0) NYQ_STAMP_FLUSH(0);
            if (live && A.pf_state_out && c == 0 && lane == 0) {
                float *ps = A.pf_state_out + 6 * s;
                ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
            }
        }
    }
#endif
#ifndef NYQ_PIPE_DBG_NO_IO
    if (wave >= kPipeUnits) {
#ifdef NYQ_PIPE_IO_PRIO
        __builtin_amdgcn_s_setprio(NYQ_PIPE_IO_PRIO);
#endif
        DeConst D;
        deemph_init<N>(D, lane);
        for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
            const long u0 = pair * kPipeUnits;
            // ---------------- I/O wave for both units ----------------
            const bool pairOut = PAIR || A.channels == 2;      // both units are the channels of one stream: interleaved 16-byte stores
            bool live[kPipeUnits];
            long sU[kPipeUnits];
            int cU[kPipeUnits];
            const vf4 *src[kPipeUnits];
            float mem[kPipeUnits];
            // mapped output (row f3): the descriptor of each unit's stream (PAIR: one stream); null base = dense output
            const OutDesc D0 = load_desc(A.desc, (PAIR || u0 < nunits) ? u0 / A.channels : 0);
            const OutDesc D1 = PAIR ? D0 : load_desc(A.desc, u0 + 1 < nunits ? (u0 + 1) / A.channels : 0);
            // the prefetched frame (+ its parameters): frame f+1 lands from these registers in iteration f, right after
            // that frame f+2 is fetched into them -- the loads have a whole frame time to arrive, and the wait for them
            // sits in front of this iteration's stores (vmcnt counts loads and stores in issue order)
            vf4 nx[kPipeUnits][NLD];
            int pT[kPipeUnits], pS[kPipeUnits];
            float pG[kPipeUnits];
#pragma unroll
            for (int k = 0; k < kPipeUnits; k++) {
                const long u = u0 + k;
                live[k] = PAIR || u < nunits;
                sU[k] = live[k] ? u / A.channels : 0;
                cU[k] = (int)(u - sU[k] * A.channels);
                src[k] = reinterpret_cast<const vf4 *>(A.pcm + (live[k] ? u : 0) * nfr * N);
                // (through an SGPR: the value is wave-uniform, and the s_waitcnt for this load then sits HERE -- kept as
                // a loaded VGPR across the frame loop, every iteration's first use of it waited for vmcnt(0), i.e. for the
                // frame prefetch issued just before and the stores in flight)
                mem[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                             __builtin_bit_cast(int, (live[k] && A.deemph) ? A.deemph[u] : 0.f)));
                pT[k] = 0; pS[k] = 0; pG[k] = 0.f;
#pragma unroll
                for (int q = 0; q < NLD; q++) nx[k][q] = vf4{0, 0, 0, 0};
            }
            // frame fidx (float4 columns q0 .. q1-1 of the lane) and, with column 0, its parameters -> registers
            auto fetch = [&](long fidx, int q0, int q1) {
                const int ln = opaque(lane);
#pragma unroll
                for (int k = 0; k < kPipeUnits; k++) {
                    if (!live[k]) continue;
                    const vf4 *fr = src[k] + fidx * NV;
#pragma unroll
                    for (int q = 0; q < NLD; q++) {
                        const int v = ln + q * kWave;
                        if (q >= q0 && q < q1) nx[k][q] = pipe_ld(fr + (v < NV ? v : NV - 1)); // lanes past the frame re-read its last vector
                    }
                    if (q0 == 0) {
                        const long pi = sU[k] * pst + fidx;
                        pT[k] = A.pf_pitch[pi];
                        pG[k] = A.pf_gain[pi];
                        pS[k] = A.pf_tapset[pi];
                    }
                }
            };
            // registers -> frame region of buffer b (float4 columns q0 .. q1-1 of the lane), parameter slot fidx & 1
            auto land = [&](long fidx, int b, int q0, int q1) {
                const int ln = opaque(lane);
#pragma unroll
                for (int k = 0; k < kPipeUnits; k++) {
                    if (!live[k]) continue;
#pragma unroll
                    for (int q = 0; q < NLD; q++) {
                        const int v = ln + q * kWave;
                        if (q >= q0 && q < q1 && v < NV) sts4(bufs[k][b], R0 + 4 * v, nx[k][q]);
                    }
                    if (q0 == 0 && ln == 0) {
                        PipeParams *ps = &pslot[k][fidx & 1];
                        ps->T = pT[k];
                        ps->g = pG[k];
                        ps->ts = pS[k];
                    }
                }
            };
            // prologue: buffer A = [history before frame 0 | frame 0]
#pragma unroll
            for (int k = 0; k < kPipeUnits; k++) {
                if (!live[k]) continue;
#pragma unroll 1
                for (int j = lane; j < kPipeHist; j += kWave) bufs[k][0][j] = A.hist ? A.hist[(u0 + k) * kPostHist + OLD + j] : 0.f;
            }
            if (nfr > 0) {
                fetch(0, 0, NLD);
                land(0, 0, 0, NLD);
                if (nfr > 1) fetch(1, 0, NLD);
            }
            __syncthreads();
            NYQ_STAMP_DECL();
            for (long f = 0; f <= nfr; f++) {
                NYQ_STAMP(5);                                  // slot 5: waiting at the barrier
                const int cb = (int)(f & 1), nb = cb ^ 1;
                // the part of cur's history that is still history after frame f
                if (f < nfr && KEEP > 0) {
#pragma unroll
                    for (int k = 0; k < kPipeUnits; k++)
                        if (live[k]) pipe_copy<true>(bufs[k][cb] + N, bufs[k][nb], lane, 0, KEEP);
                }
                __builtin_amdgcn_sched_barrier(0);             // (or the copy's writes sink to the end of the iteration, registers held)
                NYQ_STAMP(0);                                  // slot 0: history carry-over
                // frame f-1 is final in cur[R0 - N, R0): de-emphasis, staged in nxt's frame region, picked up again
                // in the output's interleaved order (registers) -- before the raw frame f+1 lands in that region
                // The pick-up / land / store sequence runs in HALVES of the frame (long frames): the interleaved output of
                // one half is picked up into registers, the raw frame f+1 lands over exactly that half of the stage, the
                // half is stored -- 16 instead of 32 registers of output in flight (the kernel has to stay at 128 VGPRs:
                // see the launch bounds).
                constexpr int HALVES = NLD % 2 == 0 ? 2 : 1;
                constexpr int QH = NLD / HALVES;               // float4 columns of a unit's frame per half
                const float *const dstg[kPipeUnits] = {bufs[0][nb] + R0, bufs[1][nb] + R0};
                if (f >= 1) {
                    const float *const dsrc[kPipeUnits] = {bufs[0][cb] + R0 - N, bufs[1][cb] + R0 - N};
                    float *const dst[kPipeUnits] = {bufs[0][nb] + R0, bufs[1][nb] + R0};
#ifndef NYQ_PIPE_DEEMPH_PAIRED   // (one unit after the other: 15 fewer registers; the I/O wave has the time)
#pragma unroll
                    for (int k = 0; k < kPipeUnits; k++) {
                        const float *const s1[1] = {dsrc[k]};
                        float *const d1[1] = {dst[k]};
                        float m1[1] = {mem[k]};
                        const bool l1[1] = {live[k]};
                        deemph_frames<N, 1>(s1, d1, m1, l1, opaque(lane), D);
                        mem[k] = m1[0];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
                    deemph_frames<N, kPipeUnits>(dsrc, dst, mem, live, opaque(lane), D);
#endif
                    NYQ_POST_SYNC();
                    if (!pairOut && A.channels != 1) {
                        // any other channel count: strided 4-byte stores straight from the stage (rare shapes)
#pragma unroll
                        for (int k = 0; k < kPipeUnits; k++) {
                            if (!live[k]) continue;
                            float *o = A.out + (sU[k] * pst * N + (f - 1) * N) * A.channels + cU[k];
                            const int lg = opaque(lane);       // (no per-lane 64-bit induction variables kept over the frame loop)
#pragma unroll 2
                            for (int j = lg; j < N; j += kWave) o[(long)j * A.channels] = dstg[k][j];
                        }
                        NYQ_POST_SYNC();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                NYQ_STAMP(1);                                  // slot 1: de-emphasis
#pragma unroll
                for (int h = 0; h < HALVES; h++) {
                    __builtin_amdgcn_sched_barrier(0);         // (phases stay apart: their registers are not live together)
                    vf4 sv[2 * QH];
                    const int lh = opaque(lane);
                    if (f >= 1) {
                        if (pairOut) {
                            // float4 v = {L[2v], R[2v], L[2v+1], R[2v+1]}: columns 2 QH h .. of the interleaved frame read
                            // floats [256 QH h, 256 QH (h+1)) of both stages -- what land(.., QH h, QH (h+1)) overwrites
#pragma unroll
                            for (int q = 0; q < 2 * QH; q++) {
                                const int v = lh + (2 * QH * h + q) * kWave, vv = v < 2 * NV ? v : 0;
                                const float2 l = *reinterpret_cast<const float2 *>(dstg[0] + 2 * vv);
                                const float2 r = *reinterpret_cast<const float2 *>(dstg[1] + 2 * vv);
                                sv[q] = vf4{l.x, r.x, l.y, r.y};
                            }
                        } else if (A.channels == 1) {
#pragma unroll
                            for (int k = 0; k < kPipeUnits; k++)
#pragma unroll
                                for (int q = 0; q < QH; q++) {
                                    const int v = lh + (QH * h + q) * kWave;
                                    sv[k * QH + q] = *reinterpret_cast<const vf4 *>(dstg[k] + 4 * (v < NV ? v : 0));
                                }
                        }
                        NYQ_POST_SYNC();
                    }
                    if (f + 1 < nfr) land(f + 1, nb, QH * h, QH * (h + 1));   // (waits for the prefetched loads)
                    if (f >= 1) {
                        const long t0 = (f - 1) * N;           // first sample of the frame
                        if (pairOut && D0.base) {
                            // into the file's interleaved layout: float4 v = samples 2v, 2v+1 of both channels
#pragma unroll
                            for (int q = 0; q < 2 * QH; q++) {
                                const int v = lh + (2 * QH * h + q) * kWave;
                                if (v < 2 * NV) {
                                    const long long ts = D0.t0 + t0 + 2 * v;
                                    mapped_put(D0, ts, D0.coff0, sv[q].x);
                                    mapped_put(D0, ts, D0.coff1, sv[q].y);
                                    mapped_put(D0, ts + 1, D0.coff0, sv[q].z);
                                    mapped_put(D0, ts + 1, D0.coff1, sv[q].w);
                                }
                            }
                        } else if (pairOut) {
                            vf4 *d4 = reinterpret_cast<vf4 *>(A.out + (sU[0] * pst * N + t0) * 2);
#pragma unroll
                            for (int q = 0; q < 2 * QH; q++) {
                                const int v = lh + (2 * QH * h + q) * kWave;
                                if (v < 2 * NV) pipe_st(d4 + v, sv[q]);
                            }
                        } else if (A.channels == 1) {
#pragma unroll
                            for (int k = 0; k < kPipeUnits; k++) {
                                if (!live[k]) continue;
                                const OutDesc &Dk = k ? D1 : D0;
                                vf4 *d4 = reinterpret_cast<vf4 *>(A.out + sU[k] * pst * N + t0);
#pragma unroll
                                for (int q = 0; q < QH; q++) {
                                    const int v = lh + (QH * h + q) * kWave;
                                    if (v >= NV) continue;
                                    const vf4 w = sv[k * QH + q];
                                    if (Dk.base) {
                                        const long long ts = Dk.t0 + t0 + 4 * v;
                                        mapped_put(Dk, ts, Dk.coff0, w.x);
                                        mapped_put(Dk, ts + 1, Dk.coff0, w.y);
                                        mapped_put(Dk, ts + 2, Dk.coff0, w.z);
                                        mapped_put(Dk, ts + 3, Dk.coff0, w.w);
                                    } else {
                                        pipe_st(d4 + v, w);
                                    }
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                NYQ_STAMP(2);                                  // slot 2: pick-up, frame f+1 -> LDS, stores of frame f-1
                // (issuing each half's loads right behind its landing, in front of its stores, was measured: 2.5-8 % slower,
                // profiles/r03_k_post_early_fetch_ab.jsonl)
                if (f + 2 < nfr) fetch(f + 2, 0, NLD);
                NYQ_STAMP(3);                                  // slot 3: fetch of frame f+2
                if (f == nfr) {
                    // state for the next call: cur = [history in front of the next frame | ...]
#pragma unroll
                    for (int k = 0; k < kPipeUnits; k++) {
                        if (!live[k]) continue;
                        // the last 1088 outputs: 1040 are cur's history region, the 48 before them sit in the other
                        // buffer's history region (what preceded the last frame), N - 48 samples in; no frames: unchanged
                        if (A.hist && nfr > 0)
#pragma unroll 1
                            for (int j = lane; j < kPostHist; j += kWave)
                                A.hist[(u0 + k) * kPostHist + j] = j < OLD ? bufs[k][nb][N - OLD + j] : bufs[k][cb][j - OLD];
                        if (A.deemph && lane == 0) A.deemph[u0 + k] = mem[k];
                    }
                }
                NYQ_STAMP(4);
                __syncthreads();
            }
            NYQ_STAMP_FLUSH(8);
        }
    }
#endif
    NYQ_WG_TRACE(3);
}

}  // namespace nyq
