// nyq_post_kernels.hpp -- what celt_decode_with_ec does between the IMDCT and the caller's PCM
// buffer, batched over streams: the pitch post-filter (comb_filter, celt.c:114-172 and
// comb_filter_const :87-110, applied as in celt_decoder_clean.c:658-683) and the de-emphasis +
// scaling + channel interleave of deemphasis() (celt_decoder_clean.c:192-256, float build,
// downsample 1), writing the interleaved [-1,1) float layout of nqr::AudioData::samples
// (include/libnyquist/Common.h:350-364).
//
// Both filters are recursive along time, so one wavefront owns one (stream, channel) and walks its
// frames in order; parallelism inside the wave comes from the filters' structure:
//   * comb filter: y[i] depends on y[i-T-2 .. i-T+2] with T >= 15 (COMBFILTER_MINPERIOD), so
//     min(64, T-2) consecutive outputs are independent and are produced by one wave step out of a
//     2048-sample LDS ring that holds the filtered history (= DECODE_BUFFER_SIZE, the reference's
//     decode_mem depth);
//   * de-emphasis: y[j] = x[j] + c*y[j-1] is a first-order linear recurrence, solved 64 samples at a
//     time by a log-step wavefront scan (lane shuffles with the powers c^1, c^2, c^4, ... ).
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_imdct_lanes.hpp"

namespace nyq {

constexpr int kPostRing = 2048;         // DECODE_BUFFER_SIZE, celt_decoder_clean.c:59
constexpr int kPostHist = 1088;         // history handed from call to call (>= COMBFILTER_MAXPERIOD + 2)
constexpr int kCombMinPeriod = 15;      // celt.h:188
constexpr float kPreemph = 0.85000610f; // mode->preemph[0], static_modes_float.h:581

struct PostArgs {
    const float *pcm;        // [nstreams*channels][nframes*N]  IMDCT output (read only)
    const int *pf_pitch;     // [nstreams][nframes]
    const float *pf_gain;    // [nstreams][nframes]
    const int *pf_tapset;    // [nstreams][nframes]
    const float *pf_state;   // [nstreams][6] {period_old, period, gain_old, gain, tapset_old, tapset} or null
    float *pf_state_out;     // same layout, must not alias pf_state (channels of a stream run in different waves)
    float *hist;             // [nstreams*channels][1088] filtered history in/out, or null (zeros, discarded)
    float *deemph;           // [nstreams*channels] preemph_memD in/out, or null
    float *out;              // [nstreams][nframes*N][channels]
    long nstreams, nframes;
    int channels;
};

__device__ __forceinline__ void comb_gains(float g, int tapset, float &a, float &b, float &c) {
    // celt.c:121-124 gains[tapset][0..2]
    const float t0 = tapset == 0 ? 0.3066406250f : tapset == 1 ? 0.4638671875f : 0.7998046875f;
    const float t1 = tapset == 0 ? 0.2170410156f : tapset == 1 ? 0.2680664062f : 0.1000976562f;
    const float t2 = tapset == 0 ? 0.1296386719f : 0.f;
    a = g * t0;
    b = g * t1;
    c = g * t2;
}

constexpr int kCombSub = 4;   // 64-output sub-chunks per comb step

#define NYQ_POST_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// One comb_filter() call (celt.c:114-172) on ring samples [t0, t0+n): cross-fade from
// (T0,g0,tapset0) to (T1,g1,tapset1) over the first 120 samples, constant filter after.
__device__ __forceinline__ void comb_call(float *ring, int lane, long t0, int n, int T0, int T1, float g0, float g1,
                                          int ts0, int ts1, const float *win2) {
    if (g0 == 0.f && g1 == 0.f) return;                       // celt.c:126-132 (in place: nothing to do)
    float g00, g01, g02, g10, g11, g12;
    comb_gains(g0, ts0, g00, g01, g02);
    comb_gains(g1, ts1, g10, g11, g12);
    // Outputs i .. i+w-1 are independent when w <= T-2 for every ACTIVE tap set.  A switched-off side
    // (gain 0) may carry any period, even 0 (postfilter_pitch of a frame without post-filter); the
    // reference multiplies those taps by zero, here they are skipped and do not bound w.
    // A step covers up to kCombSub * 64 outputs: every lane first reads the taps of its kCombSub outputs,
    // then writes them, so one LDS round trip is paid per step rather than per 64 outputs.
    int tmin = kCombSub * kWave + 2;
    if (g0 != 0.f && T0 < tmin) tmin = T0;
    if (g1 != 0.f && T1 < tmin) tmin = T1;
    const int w = tmin - 2;
    for (int base = 0; base < kOverlap; base += w) {
        float y[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {                          // 120 outputs at most: two sub-chunks
            const int o = u * kWave + lane, i = base + o;
            y[u] = 0.f;
            if (o < w && i < kOverlap) {
                const float f = win2[i], nf = 1.0f - f;
                const long t = t0 + i;
                float v = ring[t & (kPostRing - 1)];
                if (g0 != 0.f) {
                    v += (nf * g00) * ring[(t - T0) & (kPostRing - 1)];
                    v += (nf * g01) * (ring[(t - T0 + 1) & (kPostRing - 1)] + ring[(t - T0 - 1) & (kPostRing - 1)]);
                    v += (nf * g02) * (ring[(t - T0 + 2) & (kPostRing - 1)] + ring[(t - T0 - 2) & (kPostRing - 1)]);
                }
                if (g1 != 0.f) {
                    v += (f * g10) * ring[(t - T1) & (kPostRing - 1)];
                    v += (f * g11) * (ring[(t - T1 + 1) & (kPostRing - 1)] + ring[(t - T1 - 1) & (kPostRing - 1)]);
                    v += (f * g12) * (ring[(t - T1 + 2) & (kPostRing - 1)] + ring[(t - T1 - 2) & (kPostRing - 1)]);
                }
                y[u] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int o = u * kWave + lane, i = base + o;
            if (o < w && i < kOverlap) ring[(t0 + i) & (kPostRing - 1)] = y[u];
        }
        NYQ_POST_SYNC();
    }
    if (g1 == 0.f) return;                                    // celt.c:163-169
    const int w1 = T1 - 2 < kCombSub * kWave ? T1 - 2 : kCombSub * kWave;
    for (int base = kOverlap; base < n; base += w1) {          // comb_filter_const, celt.c:87-110
        float y[kCombSub];
#pragma unroll
        for (int u = 0; u < kCombSub; u++) {
            const int o = u * kWave + lane, i = base + o;
            y[u] = 0.f;
            if (o < w1 && i < n) {
                const long t = t0 + i;
                float v = ring[t & (kPostRing - 1)];
                v += g10 * ring[(t - T1) & (kPostRing - 1)];
                v += g11 * (ring[(t - T1 + 1) & (kPostRing - 1)] + ring[(t - T1 - 1) & (kPostRing - 1)]);
                v += g12 * (ring[(t - T1 + 2) & (kPostRing - 1)] + ring[(t - T1 - 2) & (kPostRing - 1)]);
                y[u] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < kCombSub; u++) {
            const int o = u * kWave + lane, i = base + o;
            if (o < w1 && i < n) ring[(t0 + i) & (kPostRing - 1)] = y[u];
        }
        NYQ_POST_SYNC();
    }
}

template <int WPB>
__global__ __launch_bounds__(kWave *WPB) void celt_post_kernel(PostArgs A, int LM, const float *__restrict__ window) {
    __shared__ float rings[WPB * (kPostRing + 960)];
    __shared__ float win2[kOverlap];                 // window^2 of the cross-fade (celt.c:147-158), block-shared
    for (int i = threadIdx.x; i < kOverlap; i += kWave * WPB) win2[i] = window[i] * window[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    float *ring = rings + (threadIdx.x >> 6) * (kPostRing + 960);
    float *stage = ring + kPostRing;                 // de-emphasised frame, staged for coalesced stores
    const long N = 120L << LM;
    const long nsc = A.nstreams * A.channels;
    const long nwaves = (long)gridDim.x * WPB;

    // De-emphasis t[j] = a[j] + c t[j-1] over a frame of N = NL * CH samples (CH = 15, 8, 4, 2 and NL = 64, 60,
    // 60, 60 for N = 960 .. 120): lane l < NL runs the recurrence over its own CH consecutive samples with a
    // zero carry-in, ONE log-step wavefront scan with ratio c^CH turns the lane-end values into the true ones,
    // and each lane then adds c^k times the value entering its chunk.
    const int CH = (int)((N + kWave - 1) / kWave);
    const int NL = (int)(N / CH);
    float cch = 1.f;                                                  // c^CH
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

    const long sc0 = (long)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    for (long sc = sc0; sc < nsc; sc += nwaves) {
        const long s = sc / A.channels;
        const int c = (int)(sc - s * A.channels);
        // ring positions [0, 1088) <- filtered history that precedes frame 0 (time -1088 .. -1)
        for (int j = lane; j < kPostRing; j += kWave) {
            float v = 0.f;
            if (j >= kPostRing - kPostHist && A.hist) v = A.hist[sc * kPostHist + (j - (kPostRing - kPostHist))];
            ring[j] = v;                                       // time index (j - 2048): ring slot j
        }
        int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
        float g_old = 0.f, g_cur = 0.f;
        if (A.pf_state) {
            const float *ps = A.pf_state + 6 * s;
            T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
        }
        float mem = A.deemph ? A.deemph[sc] : 0.f;
        NYQ_POST_SYNC();
        const float *src = A.pcm + sc * A.nframes * N;
        // Order of the memory operations of iteration f: filter frame f (LDS only) -> move frame f+1 from
        // registers into the ring (its loads were issued one iteration ago) -> issue the loads of frame f+2 ->
        // store frame f.  Every wait on memory is then for operations issued a whole filtering phase earlier.
        float nx[15];
        int T_nx = 0, ts_nx = 0;
        float g_nx = 0.f;
        auto fetch = [&](long f) {
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const long j = lane + (long)k * kWave;
                nx[k] = j < N ? src[f * N + j] : 0.f;
            }
            T_nx = A.pf_pitch[s * A.nframes + f];
            g_nx = A.pf_gain[s * A.nframes + f];
            ts_nx = A.pf_tapset[s * A.nframes + f];
        };
        auto to_ring = [&](long f) {
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const long j = lane + (long)k * kWave;
                if (j < N) ring[(f * N + j) & (kPostRing - 1)] = nx[k];
            }
        };
        int T_new = 0, ts_new = 0;
        float g_new = 0.f;
        if (A.nframes > 0) {
            fetch(0);
            to_ring(0);
            T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
            if (A.nframes > 1) fetch(1);
        }
        NYQ_POST_SYNC();
        for (long f = 0; f < A.nframes; f++) {
            const long t0 = f * N;                             // frame start, time 0 = ring slot 0 (mod 2048)
            if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;   // celt_decoder_clean.c:661-662
            if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
            comb_call(ring, lane, t0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
            if (LM != 0)
                comb_call(ring, lane, t0 + kOverlap, (int)N - kOverlap, T_cur, T_new, g_cur, g_new, ts_cur, ts_new, win2);
            // de-emphasis (celt_decoder_clean.c:243-248): tmp = x + m + VERY_SMALL; m = coef0*tmp; y = tmp/32768
            float *dst = A.out + ((s * A.nframes + f) * N) * A.channels + c;
            {
                // all LDS reads first, the recurrence in registers, one write per sample
                float loc[15];
#pragma unroll
                for (int k = 0; k < 15; k++)
                    loc[k] = (k < CH && lane < NL) ? ring[(t0 + lane * CH + k) & (kPostRing - 1)] : 0.f;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 15; k++)
                    if (k < CH) {
                        acc = (loc[k] + 1e-30f) + kPreemph * acc;
                        loc[k] = acc;
                    }
                if (lane >= NL) acc = 0.f;
                float e = acc;                                    // e[l] = sum_{i<=l} (c^CH)^(l-i) acc[i]
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const float up = __shfl_up(e, 1 << k);
                    if (lane >= (1 << k)) e += cstep[k] * up;
                }
                // value entering lane l's chunk: c t[l CH - 1] = c e[l-1] + (c^CH)^l mem, with mem = c t[-1]
                const float prevEnd = __shfl_up(e, 1);
                float cp = lane == 0 ? mem : kPreemph * prevEnd + pw * mem;
#pragma unroll
                for (int k = 0; k < 15; k++)
                    if (k < CH) {
                        if (lane < NL) stage[lane * CH + k] = (loc[k] + cp) * (1.f / 32768.f);   // + c^k * carry
                        cp *= kPreemph;
                    }
                mem = kPreemph * __shfl(e, NL - 1) + pwEnd * mem; // c t[N-1]
            }
            T_old = T_cur; g_old = g_cur; ts_old = ts_cur;       // :672-677
            T_cur = T_new; g_cur = g_new; ts_cur = ts_new;
            if (LM != 0) { T_old = T_cur; g_old = g_cur; ts_old = ts_cur; }   // :678-683
            NYQ_POST_SYNC();
            // frame f+1 overwrites ring times [t0-1088, t0-128): older than anything its comb filter reads
            if (f + 1 < A.nframes) {
                to_ring(f + 1);
                T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
                if (f + 2 < A.nframes) fetch(f + 2);
            }
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const long j = lane + (long)k * kWave;
                if (j < N) dst[j * A.channels] = stage[j];
            }
            NYQ_POST_SYNC();
        }
        // hand the state to the next call
        if (A.hist) {
            const long tend = A.nframes * N;
            for (int j = lane; j < kPostHist; j += kWave)
                A.hist[sc * kPostHist + j] = ring[(tend - kPostHist + j) & (kPostRing - 1)];
        }
        if (A.deemph && lane == 0) A.deemph[sc] = mem;
        if (A.pf_state_out && c == 0 && lane == 0) {
            float *ps = A.pf_state_out + 6 * s;
            ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
        }
        NYQ_POST_SYNC();
    }
}

}  // namespace nyq
