// nyq_post_kernels.hpp -- what celt_decode_with_ec does between the IMDCT and the caller's PCM
// buffer, batched over streams: the pitch post-filter (comb_filter, celt.c:114-172 and
// comb_filter_const :87-110, applied as in celt_decoder_clean.c:658-683) and the de-emphasis +
// scaling + channel interleave of deemphasis() (celt_decoder_clean.c:192-256, float build,
// downsample 1), writing the interleaved [-1,1) float layout of nqr::AudioData::samples
// (include/libnyquist/Common.h:350-364).
//
// Both filters are recursive along time, so one wavefront owns one (stream, channel) -- or, for stereo, BOTH
// channels of a stream, whose filters share every parameter and therefore run in lock step: each step then
// carries two independent dependency chains (twice the work per LDS round trip) and the interleaved output
// leaves in full 16-byte stores -- and walks its frames in order; parallelism inside the wave comes from the
// filters' structure:
//   * comb filter: y[i] depends on y[i-T-2 .. i-T+2] with T >= 15 (COMBFILTER_MINPERIOD), so T-2
//     consecutive outputs are independent.  A wave step produces up to 256 of them, four adjacent
//     outputs per lane (their 8 taps come from three aligned 16-byte LDS reads), out of a 2048-sample LDS
//     buffer per channel (= DECODE_BUFFER_SIZE, the reference's decode_mem depth) laid out LINEARLY as
//     [1088 samples of filtered history | the frame]: every tap address is the output's address minus a
//     wave-uniform constant (immediate offsets, no wrap arithmetic); after a frame the last 1088 samples are
//     moved to the front (five 16-byte reads and writes per lane);
//   * de-emphasis: y[j] = x[j] + c*y[j-1] is a first-order linear recurrence: each lane runs it over
//     its own N/64 consecutive samples, one log-step wavefront scan (ratio c^(N/64)) links the lanes.
// Global memory is touched only at the frame boundaries, ordered so that no wait is for a young
// operation: filter frame f (LDS only) -> move frame f+1 from registers to the ring -> issue the loads
// of frame f+2 -> store frame f.
#pragma once
#include "nyq_post_common.hpp"

namespace nyq {

// The constant part of a comb_filter() call for one tap alignment AL: outputs [kOverlap, n) of the rings, w1 per
// step (four adjacent ones per lane, lanes 0 .. w1/4-1), taps from three aligned 16-byte reads per channel.
template <int NC, int AL>
__device__ __forceinline__ void comb_const_loop(float *ring, int lane, int r0, int n, int T1, float g10, float g11,
                                                float g12, int w1) {
    const int o = 4 * lane;
    int idx = r0 + kOverlap + o;                               // this lane's outputs of the current step
    int rb = idx - T1 - 2 - AL;                                // aligned start of their taps
    auto step = [&]() {
        f4 cen[NC], q0[NC], q1[NC], q2[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float *rc = ring + c * kPostRing;
            q0[c] = lds4(rc, rb);
            q1[c] = lds4(rc, rb + 4);
            q2[c] = lds4(rc, rb + 8);
            cen[c] = lds4(rc, idx);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float x[8];
            pick8<AL>(q0[c], q1[c], q2[c], x);
            float y[4] = {cen[c].x, cen[c].y, cen[c].z, cen[c].w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                y[u] += g10 * x[u + 2];
                y[u] += g11 * (x[u + 3] + x[u + 1]);
                y[u] += g12 * (x[u + 4] + x[u]);
            }
            sts4(ring + c * kPostRing, idx, f4{y[0], y[1], y[2], y[3]});
        }
    };
    // whole steps under one constant lane mask, then the remainder (n - kOverlap and w1 are multiples of 4)
    const int nfull = (n - kOverlap) / w1, rem = (n - kOverlap) - nfull * w1;
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

// One comb_filter() call (celt.c:114-172) on the n samples that start at index r0 (a multiple of 4, at least
// kCombMaxPeriod + 2) of each of the NC channel buffers (kPostRing floats apart): cross-fade from (T0,g0,tapset0) to
// (T1,g1,tapset1) over the first 120 samples, constant after.
template <int NC>
__device__ __forceinline__ void comb_call(float *ring, int lane, int r0, int n, int T0, int T1, float g0, float g1,
                                          int ts0, int ts1, const float *win2) {
    if (g0 == 0.f && g1 == 0.f) return;                       // celt.c:126-132 (in place: nothing to do)
    float g00, g01, g02, g10, g11, g12;
    comb_gains(g0, ts0, g00, g01, g02);
    comb_gains(g1, ts1, g10, g11, g12);
    // Outputs i .. i+w-1 are independent when w <= T-2 for every ACTIVE tap set.  A switched-off side
    // (gain 0) may carry any period, even 0 (postfilter_pitch of a frame without post-filter); the
    // reference multiplies those taps by zero, here they are skipped and do not bound w.
    const int o = 4 * lane;
    int tmin = 4 * kWave + 2;
    if (g0 != 0.f && T0 < tmin) tmin = T0;
    if (g1 != 0.f && T1 < tmin) tmin = T1;
    if (tmin - 2 <= kWave) {
        // short periods: one output per lane, scalar tap reads, no alignment cases
        const int w = tmin - 2;                                // T >= 15: w >= 13
        const int nfull = kOverlap / w, rem = kOverlap - nfull * w;
        float *rc0 = ring + (r0 + lane);
        const float *wp = win2 + lane;
        auto step = [&](float *rc, float f) {
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
        };
        if (lane < w) {
            for (int sidx = 0; sidx < nfull; sidx++) {
                const float f = *wp;
#pragma unroll
                for (int c = 0; c < NC; c++) step(rc0 + c * kPostRing, f);
                rc0 += w;
                wp += w;
                NYQ_POST_SYNC();
            }
        }
        if (lane < rem) {
            const float f = *wp;
#pragma unroll
            for (int c = 0; c < NC; c++) step(rc0 + c * kPostRing, f);
        }
        NYQ_POST_SYNC();
    } else {
    const int w = (tmin - 2) & ~3;
    for (int base = 0; base < kOverlap; base += w) {
        if (o < w && base + o < kOverlap) {
            const int idx = r0 + base + o;
            const f4 fw = lds4(win2, base + o);
            const float f[4] = {fw.x, fw.y, fw.z, fw.w};
            float y[NC][4];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const f4 cen = lds4(ring + c * kPostRing, idx);
                y[c][0] = cen.x; y[c][1] = cen.y; y[c][2] = cen.z; y[c][3] = cen.w;
            }
            if (g0 != 0.f) {
                float x[NC][8];
                taps8<NC>(ring, idx - T0 - 2, x);
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float nf = 1.0f - f[u];
                        y[c][u] += (nf * g00) * x[c][u + 2];
                        y[c][u] += (nf * g01) * (x[c][u + 3] + x[c][u + 1]);
                        y[c][u] += (nf * g02) * (x[c][u + 4] + x[c][u]);
                    }
            }
            if (g1 != 0.f) {
                float x[NC][8];
                taps8<NC>(ring, idx - T1 - 2, x);
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        y[c][u] += (f[u] * g10) * x[c][u + 2];
                        y[c][u] += (f[u] * g11) * (x[c][u + 3] + x[c][u + 1]);
                        y[c][u] += (f[u] * g12) * (x[c][u + 4] + x[c][u]);
                    }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) sts4(ring + c * kPostRing, idx, f4{y[c][0], y[c][1], y[c][2], y[c][3]});
        }
        NYQ_POST_SYNC();
    }
    }
    if (g1 == 0.f) return;                                    // celt.c:163-169
    // comb_filter_const (celt.c:87-110): the tap alignment (r0 + base + 4 lane - T1 - 2) & 3 is the same for every
    // step (base and the step width are multiples of 4), so it selects one of four branch-free loops
    if (T1 - 2 <= kWave) {
        // short periods (the common case in real streams): T1-2 <= 64 outputs per step anyway, so one output
        // per lane with five scalar tap reads is the leaner step
        const int w = T1 - 2;
        // whole steps under one constant lane mask (no per-step bound checks: the recursion is bound by VALU issue,
        // measured with SQ_INSTS_VALU / SQ_WAVE_CYCLES), then the remainder
        const int nfull = (n - kOverlap) / w, rem = (n - kOverlap) - nfull * w;
        float *rc0 = ring + (r0 + kOverlap + lane);            // this lane's output of the current step
        auto step = [&](float *rc) {
            const float *tp = rc - T1 - 2;
            const float x0 = tp[0], x1 = tp[1], x2 = tp[2], x3 = tp[3], x4 = tp[4];
            float y = rc[0];
            y += g10 * x2;
            y += g11 * (x3 + x1);
            y += g12 * (x4 + x0);
            rc[0] = y;
        };
        if (lane < w) {
            for (int sidx = 0; sidx < nfull; sidx++) {
#pragma unroll
                for (int c = 0; c < NC; c++) step(rc0 + c * kPostRing);
                rc0 += w;
                NYQ_POST_SYNC();
            }
        }
        if (lane < rem) {
#pragma unroll
            for (int c = 0; c < NC; c++) step(rc0 + c * kPostRing);
        }
        NYQ_POST_SYNC();
        return;
    }
    const int w1 = (T1 - 2 < 4 * kWave ? T1 - 2 : 4 * kWave) & ~3;
    switch ((r0 - T1 - 2) & 3) {
        case 0: comb_const_loop<NC, 0>(ring, lane, r0, n, T1, g10, g11, g12, w1); break;
        case 1: comb_const_loop<NC, 1>(ring, lane, r0, n, T1, g10, g11, g12, w1); break;
        case 2: comb_const_loop<NC, 2>(ring, lane, r0, n, T1, g10, g11, g12, w1); break;
        default: comb_const_loop<NC, 3>(ring, lane, r0, n, T1, g10, g11, g12, w1); break;
    }
}

// NC = channels a wave owns: 2 for stereo streams (A.channels == 2), 1 otherwise (one wave per (stream, channel)).
template <int LM, int WPB, int NC>
__global__ __launch_bounds__(kWave *WPB) void celt_post_kernel(PostArgs A, const float *__restrict__ window) {
    constexpr int N = 120 << LM;                    // samples per frame and channel
    constexpr int NV = N / 4;                       // float4 per frame and channel
    constexpr int NLD = (NV + kWave - 1) / kWave;   // float4 loads per lane, frame and channel
    constexpr int NST = (N + kWave - 1) / kWave;    // scalar stores per lane and frame (NC == 1, interleaved output)
    constexpr int NV2 = N * NC / 4;                 // float4 of one interleaved stereo frame (NC == 2)
    constexpr int NLD2 = (NV2 + kWave - 1) / kWave;
    constexpr int CH = (N + kWave - 1) / kWave;     // de-emphasis: consecutive samples per lane (15, 8, 4, 2)
    constexpr int NL = N / CH;                      // lanes that own samples (64, 60, 60, 60)
    static_assert(NL * CH == N, "frame splits evenly over the lanes");
    static_assert(NLD <= 4, "prefetch registers");
    constexpr int kSlice = NC * (kPostRing + 960);
    __shared__ __attribute__((aligned(16))) float rings[WPB * kSlice];
    __shared__ __attribute__((aligned(16))) float win2[kOverlap];   // window^2 of the cross-fade (celt.c:147-158)
    for (int i = threadIdx.x; i < kOverlap; i += kWave * WPB) win2[i] = window[i] * window[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    float *ring = rings + (threadIdx.x >> 6) * kSlice;   // channel c: ring + c * kPostRing
    float *stage = ring + NC * kPostRing;                // de-emphasised frame, channel c at stage + c * 960
    const long nunits = A.nstreams * (A.channels / NC);  // waves' work items: streams (NC 2) or (stream, channel)s
    const long nwaves = (long)gridDim.x * WPB;

    // De-emphasis t[j] = a[j] + c t[j-1] over a frame of N = NL * CH samples: lane l < NL runs the recurrence
    // over its own CH consecutive samples with a zero carry-in, ONE log-step wavefront scan with ratio c^CH
    // turns the lane-end values into the true ones, and each lane then adds c^k times the value entering
    // its chunk.
    float cch = 1.f;                                                  // c^CH
#pragma unroll
    for (int k = 0; k < CH; k++) cch *= kPreemph;
    float cstep[6];                                                   // (c^CH)^(2^k)
    cstep[0] = cch;
#pragma unroll
    for (int k = 1; k < 6; k++) cstep[k] = cstep[k - 1] * cstep[k - 1];
    float pw = 1.f;                                                   // (c^CH)^lane
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (lane & (1 << k)) pw *= cstep[k];
    const float pwEnd = __shfl(pw, NL - 1) * cch;                     // (c^CH)^NL

    const long u0 = (long)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    for (long unit = u0; unit < nunits; unit += nwaves) {
        const long s = NC == 2 ? unit : unit / A.channels;
        const int c0 = NC == 2 ? 0 : (int)(unit - s * A.channels);
        const long sc0 = s * A.channels + c0;                  // first (stream, channel) of this wave
        // buffer [0, 1088) <- filtered history that precedes frame 0 (times -1088 .. -1); the frame sits at [1088, 1088 + N)
#pragma unroll
        for (int c = 0; c < NC; c++)
            for (int j = lane; j < kPostHist; j += kWave)
                ring[c * kPostRing + j] = A.hist ? A.hist[(sc0 + c) * kPostHist + j] : 0.f;
        int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
        float g_old = 0.f, g_cur = 0.f;
        if (A.pf_state) {
            const float *ps = A.pf_state + 6 * s;
            T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
        }
        float mem[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) mem[c] = A.deemph ? A.deemph[sc0 + c] : 0.f;
        NYQ_POST_SYNC();
        const vf4 *src = reinterpret_cast<const vf4 *>(A.pcm + sc0 * A.nframes * N);   // channel c: + c * nframes * NV
        const long cstride = A.nframes * NV;
        const int *ppitch = A.pf_pitch + s * A.ps();
        const float *pgain = A.pf_gain + s * A.ps();
        const int *ptap = A.pf_tapset + s * A.ps();
        // the prefetched frame lives in registers (native vector values: an array of 16-byte structs here ends
        // up in scratch memory)
        vf4 nx[NC][4];
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) nx[c][k] = vf4{0, 0, 0, 0};
        int T_nx = 0, ts_nx = 0;
        float g_nx = 0.f;
#define NYQ_POST_FETCH(fidx)                                                                  \
    do {                                                                                      \
        _Pragma("unroll") for (int c = 0; c < NC; c++) {                                      \
            const vf4 *fr = src + c * cstride + (fidx) * NV;                                  \
            _Pragma("unroll") for (int k = 0; k < NLD; k++) {                                 \
                const int v = lane + k * kWave;                                               \
                nx[c][k] = fr[v < NV ? v : NV - 1]; /* lanes past the frame re-read its last vector */ \
            }                                                                                 \
        }                                                                                     \
        T_nx = ppitch[fidx];                                                                  \
        g_nx = pgain[fidx];                                                                   \
        ts_nx = ptap[fidx];                                                                   \
    } while (0)
#define NYQ_POST_TO_RING()                                                                    \
    do {                                                                                      \
        _Pragma("unroll") for (int c = 0; c < NC; c++)                                        \
            _Pragma("unroll") for (int k = 0; k < NLD; k++) {                                 \
                const int v = lane + k * kWave;                                               \
                if (v < NV) sts4(ring + c * kPostRing, kPostHist + 4 * v, nx[c][k]);          \
            }                                                                                 \
    } while (0)
        int T_new = 0, ts_new = 0;
        float g_new = 0.f;
        if (A.nframes > 0) {
            NYQ_POST_FETCH(0);
            NYQ_POST_TO_RING();
            T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
            if (A.nframes > 1) NYQ_POST_FETCH(1);
        }
        NYQ_POST_SYNC();
        constexpr int r0 = kPostHist;                          // buffer index of the frame start
        float *dst = A.out + (s * A.ps() * N) * A.channels + c0;
        for (long f = 0; f < A.nframes; f++) {
            if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;   // celt_decoder_clean.c:661-662
            if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
            if (T_cur > kCombMaxPeriod) T_cur = kCombMaxPeriod;   // (a decoder never produces more: keeps taps in the buffer)
            if (T_old > kCombMaxPeriod) T_old = kCombMaxPeriod;
            const int T_nw = T_new < kCombMinPeriod ? kCombMinPeriod : T_new > kCombMaxPeriod ? kCombMaxPeriod : T_new;
            comb_call<NC>(ring, lane, r0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
            if (LM != 0)
                comb_call<NC>(ring, lane, r0 + kOverlap, N - kOverlap, T_cur, T_nw, g_cur, g_new, ts_cur, ts_new, win2);
            // de-emphasis (celt_decoder_clean.c:243-248): tmp = x + m + VERY_SMALL; m = coef0*tmp; y = tmp/32768
            {
                // all LDS reads first, the recurrence in registers, one write per sample
                float loc[NC][CH];
                const int li = lane < NL ? lane : 0;
                const int p0 = r0 + li * CH;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float *rc = ring + c * kPostRing;
                    if constexpr (CH % 4 == 0) {
#pragma unroll
                        for (int k = 0; k < CH; k += 4) {
                            const f4 q = lds4(rc, p0 + k);
                            loc[c][k] = q.x; loc[c][k + 1] = q.y; loc[c][k + 2] = q.z; loc[c][k + 3] = q.w;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < CH; k++) loc[c][k] = rc[p0 + k];
                    }
                }
                float e[NC];                                      // e[l] = sum_{i<=l} (c^CH)^(l-i) acc[i]
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        acc = (loc[c][k] + 1e-30f) + kPreemph * acc;
                        loc[c][k] = acc;
                    }
                    e[c] = lane < NL ? acc : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const float up = __shfl_up(e[c], 1 << k);
                        if (lane >= (1 << k)) e[c] += cstep[k] * up;
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    // value entering lane l's chunk: c t[l CH - 1] = c e[l-1] + (c^CH)^l mem, with mem = c t[-1]
                    const float prevEnd = __shfl_up(e[c], 1);
                    float cp = lane == 0 ? mem[c] : kPreemph * prevEnd + pw * mem[c];
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        loc[c][k] = (loc[c][k] + cp) * (1.f / 32768.f);     // + c^k * carry
                        cp *= kPreemph;
                    }
                    if (lane < NL) {
                        float *sg = stage + c * 960;
                        if constexpr (CH % 4 == 0) {
#pragma unroll
                            for (int k = 0; k < CH; k += 4)
                                sts4(sg, lane * CH + k, f4{loc[c][k], loc[c][k + 1], loc[c][k + 2], loc[c][k + 3]});
                        } else {
#pragma unroll
                            for (int k = 0; k < CH; k++) sg[lane * CH + k] = loc[c][k];
                        }
                    }
                    mem[c] = kPreemph * __shfl(e[c], NL - 1) + pwEnd * mem[c]; // c t[N-1]
                }
            }
            T_old = T_cur; g_old = g_cur; ts_old = ts_cur;       // :672-677
            T_cur = T_new; g_cur = g_new; ts_cur = ts_new;
            if (LM != 0) { T_old = T_cur; g_old = g_cur; ts_old = ts_cur; }   // :678-683
            NYQ_POST_SYNC();
            // the last 1088 samples of [history | frame] become the history of the next frame: [N, N + 1088) -> [0, 1088)
            // (the ranges overlap: every lane reads its share first, writes after the barrier)
            {
                constexpr int HV = kPostHist / 4;                 // 272 float4 per channel
                constexpr int HLD = (HV + kWave - 1) / kWave;     // 5 per lane
                vf4 hv[NC][HLD];
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int k = 0; k < HLD; k++) {
                        const int v = lane + k * kWave;
                        hv[c][k] = *reinterpret_cast<const vf4 *>(ring + c * kPostRing + N + 4 * (v < HV ? v : HV - 1));
                    }
                NYQ_POST_SYNC();
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int k = 0; k < HLD; k++) {
                        const int v = lane + k * kWave;
                        if (v < HV) sts4(ring + c * kPostRing, 4 * v, hv[c][k]);
                    }
            }
            if (f + 1 < A.nframes) {
                NYQ_POST_TO_RING();
                T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
                if (f + 2 < A.nframes) NYQ_POST_FETCH(f + 2);
            }
            if constexpr (NC == 2) {
                // interleave the two planes on the way out: float4 v = {L[2v], R[2v], L[2v+1], R[2v+1]}
                vf4 *d4 = reinterpret_cast<vf4 *>(dst);
#pragma unroll
                for (int k = 0; k < NLD2; k++) {
                    const int v = lane + k * kWave;
                    if (v < NV2) {
                        const float2 l = *reinterpret_cast<const float2 *>(stage + 2 * v);
                        const float2 r = *reinterpret_cast<const float2 *>(stage + 960 + 2 * v);
                        d4[v] = vf4{l.x, r.x, l.y, r.y};
                    }
                }
            } else if (A.channels == 1) {
                vf4 *d4 = reinterpret_cast<vf4 *>(dst);
#pragma unroll
                for (int k = 0; k < NLD; k++) {
                    const int v = lane + k * kWave;
                    if (v < NV) d4[v] = *reinterpret_cast<const vf4 *>(stage + 4 * v);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NST; k++) {
                    const int j = lane + k * kWave;
                    if (j < N) dst[(long)j * A.channels] = stage[j];
                }
            }
            dst += (long)N * A.channels;
            NYQ_POST_SYNC();
        }
        // hand the state to the next call: the history in front of the (next) frame
        if (A.hist) {
#pragma unroll
            for (int c = 0; c < NC; c++)
                for (int j = lane; j < kPostHist; j += kWave) A.hist[(sc0 + c) * kPostHist + j] = ring[c * kPostRing + j];
        }
        if (A.deemph && lane == 0) {
#pragma unroll
            for (int c = 0; c < NC; c++) A.deemph[sc0 + c] = mem[c];
        }
        if (A.pf_state_out && c0 == 0 && lane == 0) {
            float *ps = A.pf_state_out + 6 * s;
            ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
        }
        NYQ_POST_SYNC();
    }
}

#undef NYQ_POST_FETCH
#undef NYQ_POST_TO_RING

}  // namespace nyq
