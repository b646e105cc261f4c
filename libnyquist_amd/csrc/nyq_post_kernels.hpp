// nyq_post_kernels.hpp -- what celt_decode_with_ec does between the IMDCT and the caller's PCM
// buffer, batched over streams: the pitch post-filter (comb_filter, celt.c:114-172 and
// comb_filter_const :87-110, applied as in celt_decoder_clean.c:658-683) and the de-emphasis +
// scaling + channel interleave of deemphasis() (celt_decoder_clean.c:192-256, float build,
// downsample 1), writing the interleaved [-1,1) float layout of nqr::AudioData::samples
// (include/libnyquist/Common.h:350-364).
//
// Both filters are recursive along time, so one wavefront owns one (stream, channel) and walks its
// frames in order; parallelism inside the wave comes from the filters' structure:
//   * comb filter: y[i] depends on y[i-T-2 .. i-T+2] with T >= 15 (COMBFILTER_MINPERIOD), so T-2
//     consecutive outputs are independent.  A wave step produces up to 256 of them, four adjacent
//     outputs per lane (their 8 taps come from three aligned 16-byte LDS reads), out of a 2048-sample
//     LDS ring that holds the filtered history (= DECODE_BUFFER_SIZE, the reference's decode_mem depth);
//   * de-emphasis: y[j] = x[j] + c*y[j-1] is a first-order linear recurrence: each lane runs it over
//     its own N/64 consecutive samples, one log-step wavefront scan (ratio c^(N/64)) links the lanes.
// Global memory is touched only at the frame boundaries, ordered so that no wait is for a young
// operation: filter frame f (LDS only) -> move frame f+1 from registers to the ring -> issue the loads
// of frame f+2 -> store frame f.
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

#define NYQ_POST_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// 16-byte moves go through a native vector value (a struct copy stays a memcpy and lands in scratch memory)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 lds4(const float *base, int idx) {
    const vf4 v = *reinterpret_cast<const vf4 *>(base + idx);
    return f4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void sts4(float *base, int idx, const f4 &v) {
    *reinterpret_cast<vf4 *>(base + idx) = vf4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void sts4(float *base, int idx, const vf4 &v) { *reinterpret_cast<vf4 *>(base + idx) = v; }

template <int A>
__device__ __forceinline__ void pick8(const f4 &q0, const f4 &q1, const f4 &q2, float (&x)[8]) {
    const float e[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = e[A + i];
}

// x[0..8) = ring[r .. r+8) (mod 2048) for an arbitrary r whose alignment r & 3 is the same in every lane
__device__ __forceinline__ void taps8(const float *ring, int r, float (&x)[8]) {
    const int a = __builtin_amdgcn_readfirstlane(r) & 3;
    const int rb = r - a;
    const f4 q0 = lds4(ring, rb & (kPostRing - 1));
    const f4 q1 = lds4(ring, (rb + 4) & (kPostRing - 1));
    const f4 q2 = lds4(ring, (rb + 8) & (kPostRing - 1));
    switch (a) {
        case 0: pick8<0>(q0, q1, q2, x); break;
        case 1: pick8<1>(q0, q1, q2, x); break;
        case 2: pick8<2>(q0, q1, q2, x); break;
        default: pick8<3>(q0, q1, q2, x); break;
    }
}

// One comb_filter() call (celt.c:114-172) on the n ring samples that start at ring index r0 (a multiple
// of 4): cross-fade from (T0,g0,tapset0) to (T1,g1,tapset1) over the first 120 samples, constant after.
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
    const int w = (tmin - 2) & ~3;                             // T >= 15: w >= 12
    for (int base = 0; base < kOverlap; base += w) {
        if (o < w && base + o < kOverlap) {
            const int idx = (r0 + base + o) & (kPostRing - 1);
            const f4 cen = lds4(ring, idx);
            const f4 fw = lds4(win2, base + o);
            float y[4] = {cen.x, cen.y, cen.z, cen.w};
            const float f[4] = {fw.x, fw.y, fw.z, fw.w};
            if (g0 != 0.f) {
                float x[8];
                taps8(ring, idx - T0 - 2, x);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float nf = 1.0f - f[u];
                    y[u] += (nf * g00) * x[u + 2];
                    y[u] += (nf * g01) * (x[u + 3] + x[u + 1]);
                    y[u] += (nf * g02) * (x[u + 4] + x[u]);
                }
            }
            if (g1 != 0.f) {
                float x[8];
                taps8(ring, idx - T1 - 2, x);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    y[u] += (f[u] * g10) * x[u + 2];
                    y[u] += (f[u] * g11) * (x[u + 3] + x[u + 1]);
                    y[u] += (f[u] * g12) * (x[u + 4] + x[u]);
                }
            }
            sts4(ring, idx, f4{y[0], y[1], y[2], y[3]});
        }
        NYQ_POST_SYNC();
    }
    if (g1 == 0.f) return;                                    // celt.c:163-169
    const int w1 = (T1 - 2 < 4 * kWave ? T1 - 2 : 4 * kWave) & ~3;
    for (int base = kOverlap; base < n; base += w1) {          // comb_filter_const, celt.c:87-110
        if (o < w1 && base + o < n) {
            const int idx = (r0 + base + o) & (kPostRing - 1);
            const f4 cen = lds4(ring, idx);
            float x[8];
            taps8(ring, idx - T1 - 2, x);
            float y[4] = {cen.x, cen.y, cen.z, cen.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                y[u] += g10 * x[u + 2];
                y[u] += g11 * (x[u + 3] + x[u + 1]);
                y[u] += g12 * (x[u + 4] + x[u]);
            }
            sts4(ring, idx, f4{y[0], y[1], y[2], y[3]});
        }
        NYQ_POST_SYNC();
    }
}

template <int LM, int WPB>
__global__ __launch_bounds__(kWave *WPB) void celt_post_kernel(PostArgs A, const float *__restrict__ window) {
    constexpr int N = 120 << LM;                    // samples per frame and channel
    constexpr int NV = N / 4;                       // float4 per frame
    constexpr int NLD = (NV + kWave - 1) / kWave;   // float4 loads per lane and frame
    constexpr int NST = (N + kWave - 1) / kWave;    // scalar stores per lane and frame
    constexpr int CH = (N + kWave - 1) / kWave;     // de-emphasis: consecutive samples per lane (15, 8, 4, 2)
    constexpr int NL = N / CH;                      // lanes that own samples (64, 60, 60, 60)
    static_assert(NL * CH == N, "frame splits evenly over the lanes");
    constexpr int kSlice = kPostRing + 960;
    __shared__ __attribute__((aligned(16))) float rings[WPB * kSlice];
    __shared__ __attribute__((aligned(16))) float win2[kOverlap];   // window^2 of the cross-fade (celt.c:147-158)
    for (int i = threadIdx.x; i < kOverlap; i += kWave * WPB) win2[i] = window[i] * window[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    float *ring = rings + (threadIdx.x >> 6) * kSlice;
    float *stage = ring + kPostRing;                 // de-emphasised frame, staged for coalesced stores
    const long nsc = A.nstreams * A.channels;
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

    const long sc0 = (long)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    for (long sc = sc0; sc < nsc; sc += nwaves) {
        const long s = sc / A.channels;
        const int c = (int)(sc - s * A.channels);
        // ring slots [960, 2048) <- filtered history that precedes frame 0 (times -1088 .. -1); time 0 = slot 0
        for (int j = lane; j < kPostRing; j += kWave) {
            float v = 0.f;
            if (j >= kPostRing - kPostHist && A.hist) v = A.hist[sc * kPostHist + (j - (kPostRing - kPostHist))];
            ring[j] = v;
        }
        int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
        float g_old = 0.f, g_cur = 0.f;
        if (A.pf_state) {
            const float *ps = A.pf_state + 6 * s;
            T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
        }
        float mem = A.deemph ? A.deemph[sc] : 0.f;
        NYQ_POST_SYNC();
        const vf4 *src = reinterpret_cast<const vf4 *>(A.pcm + sc * A.nframes * N);
        const int *ppitch = A.pf_pitch + s * A.nframes;
        const float *pgain = A.pf_gain + s * A.nframes;
        const int *ptap = A.pf_tapset + s * A.nframes;
        // the prefetched frame lives in named registers (an indexed array here ends up in scratch memory)
        vf4 nx0 = {0, 0, 0, 0}, nx1 = nx0, nx2 = nx0, nx3 = nx0;
        int T_nx = 0, ts_nx = 0;
        float g_nx = 0.f;
#define NYQ_POST_LD(k, reg)                                                                   \
    if (k < NLD) {                                                                            \
        const int v = lane + k * kWave;                                                       \
        reg = fr[v < NV ? v : NV - 1]; /* lanes past the frame re-read its last vector */     \
    }
#define NYQ_POST_ST(k, reg)                                                                   \
    if (k < NLD) {                                                                            \
        const int v = lane + k * kWave;                                                       \
        if (v < NV) sts4(ring, (rpos + 4 * v) & (kPostRing - 1), reg);                        \
    }
#define NYQ_POST_FETCH(fidx)                                                                  \
    do {                                                                                      \
        const vf4 *fr = src + (fidx) * NV;                                                    \
        NYQ_POST_LD(0, nx0) NYQ_POST_LD(1, nx1) NYQ_POST_LD(2, nx2) NYQ_POST_LD(3, nx3)       \
        T_nx = ppitch[fidx];                                                                  \
        g_nx = pgain[fidx];                                                                   \
        ts_nx = ptap[fidx];                                                                   \
    } while (0)
#define NYQ_POST_TO_RING(rp)                                                                  \
    do {                                                                                      \
        const int rpos = (rp);                                                                \
        NYQ_POST_ST(0, nx0) NYQ_POST_ST(1, nx1) NYQ_POST_ST(2, nx2) NYQ_POST_ST(3, nx3)       \
    } while (0)
        int T_new = 0, ts_new = 0;
        float g_new = 0.f;
        if (A.nframes > 0) {
            NYQ_POST_FETCH(0);
            NYQ_POST_TO_RING(0);
            T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
            if (A.nframes > 1) NYQ_POST_FETCH(1);
        }
        NYQ_POST_SYNC();
        int r0 = 0;                                            // ring index of the frame start
        float *dst = A.out + (s * A.nframes * N) * A.channels + c;
        for (long f = 0; f < A.nframes; f++) {
            if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;   // celt_decoder_clean.c:661-662
            if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
            comb_call(ring, lane, r0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
            if (LM != 0)
                comb_call(ring, lane, r0 + kOverlap, N - kOverlap, T_cur, T_new, g_cur, g_new, ts_cur, ts_new, win2);
            // de-emphasis (celt_decoder_clean.c:243-248): tmp = x + m + VERY_SMALL; m = coef0*tmp; y = tmp/32768
            {
                // all LDS reads first, the recurrence in registers, one write per sample
                float loc[CH];
                const int li = lane < NL ? lane : 0;
                const int p0 = r0 + li * CH;
                if constexpr (CH % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < CH; k += 4) {
                        const f4 q = lds4(ring, (p0 + k) & (kPostRing - 1));
                        loc[k] = q.x; loc[k + 1] = q.y; loc[k + 2] = q.z; loc[k + 3] = q.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CH; k++) loc[k] = ring[(p0 + k) & (kPostRing - 1)];
                }
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < CH; k++) {
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
                for (int k = 0; k < CH; k++) {
                    loc[k] = (loc[k] + cp) * (1.f / 32768.f);     // + c^k * carry
                    cp *= kPreemph;
                }
                if (lane < NL) {
                    if constexpr (CH % 4 == 0) {
#pragma unroll
                        for (int k = 0; k < CH; k += 4) sts4(stage, lane * CH + k, f4{loc[k], loc[k + 1], loc[k + 2], loc[k + 3]});
                    } else {
#pragma unroll
                        for (int k = 0; k < CH; k++) stage[lane * CH + k] = loc[k];
                    }
                }
                mem = kPreemph * __shfl(e, NL - 1) + pwEnd * mem; // c t[N-1]
            }
            T_old = T_cur; g_old = g_cur; ts_old = ts_cur;       // :672-677
            T_cur = T_new; g_cur = g_new; ts_cur = ts_new;
            if (LM != 0) { T_old = T_cur; g_old = g_cur; ts_old = ts_cur; }   // :678-683
            NYQ_POST_SYNC();
            // frame f+1 overwrites ring times [t0+N-2048, t0+2N-2048): older than anything its comb filter reads
            const int r1 = (r0 + N) & (kPostRing - 1);
            if (f + 1 < A.nframes) {
                NYQ_POST_TO_RING(r1);
                T_new = T_nx; g_new = g_nx; ts_new = ts_nx;
                if (f + 2 < A.nframes) NYQ_POST_FETCH(f + 2);
            }
            if (A.channels == 1) {
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
            r0 = r1;
            NYQ_POST_SYNC();
        }
        // hand the state to the next call: the last 1088 filtered samples end at ring index r0
        if (A.hist) {
            for (int j = lane; j < kPostHist; j += kWave)
                A.hist[sc * kPostHist + j] = ring[(r0 - kPostHist + j) & (kPostRing - 1)];
        }
        if (A.deemph && lane == 0) A.deemph[sc] = mem;
        if (A.pf_state_out && c == 0 && lane == 0) {
            float *ps = A.pf_state_out + 6 * s;
            ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
        }
        NYQ_POST_SYNC();
    }
}

#undef NYQ_POST_LD
#undef NYQ_POST_ST
#undef NYQ_POST_FETCH
#undef NYQ_POST_TO_RING

}  // namespace nyq
