// nyq_chain_fused_r2.hpp -- freq[] -> interleaved PCM in ONE kernel (round 2): everything celt_decode_with_ec does after
// denormalise_bands (celt_decoder_clean.c:620-723) for many stereo streams at once:
//   compute_inv_mdcts (:264-312, clt_mdct_backward mdct.c:267-379)  ->  comb_filter (celt.c:114-172, as applied
//   :658-683)  ->  deemphasis (:192-256) with scaling and channel interleave.
// The two-kernel chain (nyq_celt_synth_dev, nyq_celt_post_dev) writes the IMDCT output to HBM and reads it back:
// 2 x 7680 B per channel-frame.  Here the time-domain frame never leaves the CU: 3840 B in, 3840 B out.
//
// The post-filter is a recursion along time, one chain per (stream, channel), and a chain's own instruction stream
// bounds it (nyq_post_pipe.hpp), so every chain of the job has to be RESIDENT and its wave must do nothing else.
// Workgroup = 2 stereo streams = 4 chains, 6 waves:
//     wave 0..3   comb wave of chain 0..3 (the recursion of nyq_post_pipe.hpp, in place in a linear
//                 [1088 history | 960 frame] LDS buffer of 8 KB -- half of the pipeline kernel's, so that two
//                 workgroups = 8 chains fit a CU next to the IMDCT slices)
//     wave 4      I/O wave: de-emphasis + interleave + global stores of frame f-1, post-filter parameters
//     wave 5      IMDCT wave: the lane program of nyq_imdct_lanes.hpp on "frame f+1 of the 4 chains" (one group of four
//                 nfft-480 rows; a transient stream contributes one group of 2 x 8 nfft-60 rows instead), global
//                 loads of frame f+2 in flight
// Per frame three barriers:  [comb f | de-emphasis f-1 | IMDCT f+1 up to the FFT output]  B1  [comb waves pick up the
// 1088 samples that stay history]  B2  [they put them down in front; the IMDCT wave post-rotates, mirrors (TDAC
// against the chain's 60-float tail) and lands frame f+1 in the frame region]  B3.
// ~80 KB of LDS per workgroup, two workgroups (8 chains, 12 waves) per CU: 1024 stereo streams are resident at once.
#pragma once
#include "nyq_kernels.hpp"
#include "nyq_post_pipe.hpp"

namespace nyq {

#ifndef NYQ_FUSE_DBG_MASK
#define NYQ_FUSE_DBG_MASK 0
#endif
#define NYQ_FUSE_DBG_OFF(role) ((NYQ_FUSE_DBG_MASK >> (role)) & 1)   /* diagnostic builds: compile a role out */
#ifndef NYQ_FUSE_MINWAVES
#define NYQ_FUSE_MINWAVES 3
#endif
constexpr int kFuseChains = 4;                      // chains (stream, channel) per workgroup: two stereo streams
constexpr int kFuseWaves = kFuseChains + 2;         // + I/O wave + IMDCT wave
constexpr int kFuseN = 960;                         // LM 3 only: 20 ms frames

struct FusedR2Args {
    const float *freq;               // [nstreams][nframes][2][960]   as the decoder leaves freq[]
    const unsigned char *transient;  // [nstreams][nframes] or null
    float *ov_state;                 // [nstreams*2][60] overlap carry in/out, or null (zeros, discarded)
    const int *pf_pitch;             // [nstreams][nframes]
    const float *pf_gain;
    const int *pf_tapset;
    const float *pf_state;           // [nstreams][6] or null
    float *pf_state_out;             // must not alias pf_state
    float *hist;                     // [nstreams*2][1088] filtered history in/out, or null
    float *deemph;                   // [nstreams*2] in/out, or null
    float *out;                      // [nstreams][nframes*960][2]
    long nstreams, nframes;
};

// Rows of "frame f of the workgroup's chains" for the lane program: input in global memory; the finished samples go
// back INTO THE ROW'S OWN SLICE SLOTS, rotated by 60: out[60 + i] at float i of the row (i < N2 - 60), out[0..60) at
// floats N2-60.. -- task j of the stage-out reads complex points 2j, 2j+1, N4-2-2j, N4-1-2j and writes exactly those
// eight floats, so the post-rotation + TDAC mirror run in place right behind the FFT, while the registers of the
// next frame's global loads are not yet occupied; what is left for the moment the frame region becomes free is a copy.
struct FusedLongRows {
    static constexpr bool STRIDED = false;
    static constexpr bool CHAINS = false;
    const float *in0;      // freq of chain 0, this frame (chain k: + k * 960 inside a stream, + in_stream for the second stream)
    long in_stream;        // floats between the two streams' frames
    float *srow0;          // LDS: row 0 of the slice as floats; row g at + 2 * Geo<32>::S * g
    float *tails;          // LDS: [4][60]
    unsigned valid_mask;   // bit k: chain k exists and its frame is a long one
    __device__ bool valid(int g) const { return (valid_mask >> g) & 1u; }
    __device__ const float *in(int g) const { return in0 + (g >> 1) * in_stream + (g & 1) * (long)kFuseN; }
    __device__ int stride() const { return 1; }
    __device__ float *fin(int g) const { return srow0 + g * (2 * Geo<32>::S) - kHalfOv; }
    __device__ float *head(int g) const { return srow0 + g * (2 * Geo<32>::S) + (kFuseN - kHalfOv); }
    __device__ float *tail(int g) const { return tails + g * kHalfOv; }
    __device__ const float *carry(int g) const { return tails + g * kHalfOv; }
    __device__ bool chain(int) const { return false; }
};

// the 2 x 8 interleaved short blocks of ONE transient stream-frame: row g = (channel g / 8, block g % 8)
struct FusedShortRows {
    static constexpr bool STRIDED = true;
    static constexpr bool CHAINS = true;
    const float *in0;      // freq of the stream's channel 0, this frame
    float *srow0;          // LDS: row 0 of the slice as floats; row g at + 2 * Geo<4>::S * g
    float *tails;          // LDS: tails of the stream's channel 0
    bool on;
    __device__ bool valid(int) const { return on; }
    __device__ const float *in(int g) const { return in0 + (g >> 3) * (long)kFuseN + (g & 7); }
    __device__ int stride() const { return 8; }
    __device__ float *fin(int g) const { return srow0 + g * (2 * Geo<4>::S) - kHalfOv; }
    __device__ float *head(int g) const { return srow0 + g * (2 * Geo<4>::S) + (120 - kHalfOv); }
    __device__ bool chain(int g) const { return (g & 7) != 0; }
    __device__ const float *carry(int g) const { return (g & 7) == 0 ? tails + (g >> 3) * kHalfOv : nullptr; }
    __device__ float *tail(int g) const { return (g & 7) == 7 ? tails + (g >> 3) * kHalfOv : nullptr; }
};

__global__ __launch_bounds__(kWave *kFuseWaves, NYQ_FUSE_MINWAVES) void celt_chain_fused_kernel(FusedR2Args A, const float *__restrict__ trig,
                                                                                 const float *__restrict__ window) {
    constexpr int N = kFuseN, NV = N / 4, NLD = 4;
    constexpr int R0 = kPostHist;
    constexpr int SLICE_A = 2 * Geo<32>::LDS_CPX;   // floats: four nfft-480 rows
    constexpr int SLICE_B = 2 * Geo<4>::LDS_CPX;    // floats: sixteen nfft-60 rows
    static_assert(SLICE_A >= SLICE_B, "a short group fits the long slice");
    __shared__ __attribute__((aligned(16))) float bufs[kFuseChains][kPostRing];   // [1088 history | frame]
    __shared__ __attribute__((aligned(16))) float slice[SLICE_A + SLICE_B];
    __shared__ __attribute__((aligned(16))) float stage[kFuseChains][N];
    __shared__ __attribute__((aligned(16))) float tails[kFuseChains][kHalfOv];
    __shared__ __attribute__((aligned(16))) float sring[Geo<4>::RING_FLOATS + 4];
    __shared__ __attribute__((aligned(16))) float win2[kOverlap];
    __shared__ __attribute__((aligned(16))) PipeParams pslot[2][2];               // [stream][frame parity]
    static_assert(kHalfOv % 4 == 0, "tails rows stay 16-byte aligned");
    for (int i = threadIdx.x; i < kOverlap; i += kWave * kFuseWaves) win2[i] = window[i] * window[i];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long nfr = A.nframes;
    const long ngroups = (A.nstreams + 1) / 2;      // workgroup units: pairs of streams
    __syncthreads();

    // Every role runs the same unit loop and the same barriers per unit: P (prologue done), then per frame B1, B2, B3.
    if (wave < kFuseChains && !NYQ_FUSE_DBG_OFF(0)) {
        // ------------------------------- comb wave of chain `wave` -------------------------------
        __builtin_amdgcn_s_setprio(3);
        float *buf = bufs[wave];
        for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            const long s = 2 * grp + (wave >> 1);
            const bool live = s < A.nstreams;
            int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
            float g_old = 0.f, g_cur = 0.f;
            if (live && A.pf_state) {
                const float *ps = A.pf_state + 6 * s;
                T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
            }
            __syncthreads();                                                   // P
            for (long f = 0; f < nfr; f++) {
                if (live) {
                    const PipeParams P = pslot[wave >> 1][f & 1];
                    const int T_new = __builtin_amdgcn_readfirstlane(P.T), ts_new = __builtin_amdgcn_readfirstlane(P.ts);
                    const float g_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P.g)));
                    if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;       // celt_decoder_clean.c:661-662
                    if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
                    if (T_cur > kCombMaxPeriod) T_cur = kCombMaxPeriod;
                    if (T_old > kCombMaxPeriod) T_old = kCombMaxPeriod;
                    const int T_nw = T_new < kCombMinPeriod ? kCombMinPeriod : T_new > kCombMaxPeriod ? kCombMaxPeriod : T_new;
                    pipe_comb_call<false>(buf, nullptr, lane, R0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
                    pipe_comb_call<false>(buf, nullptr, lane, R0 + kOverlap, N - kOverlap, T_cur, T_nw, g_cur, g_new, ts_cur, ts_new, win2);
                    T_old = T_cur = T_new; g_old = g_cur = g_new; ts_old = ts_cur = ts_new;   // :672-683 (LM != 0)
                }
                __syncthreads();                                               // B1: frame f filtered
                // the last 1088 samples of [history | frame] become the history of the next frame
                constexpr int HV = kPostHist / 4, HLD = (HV + kWave - 1) / kWave;   // 272 float4, 5 per lane
                vf4 hv[HLD];
#pragma unroll
                for (int k = 0; k < HLD; k++) {
                    const int v = lane + k * kWave;
                    hv[k] = *reinterpret_cast<const vf4 *>(buf + N + 4 * (v < HV ? v : HV - 1));
                }
                __syncthreads();                                               // B2: the frame region may be overwritten
#pragma unroll
                for (int k = 0; k < HLD; k++) {
                    const int v = lane + k * kWave;
                    if (v < HV) sts4(buf, 4 * v, hv[k]);
                }
                __syncthreads();                                               // B3: frame f+1 landed
            }
            if (live && A.pf_state_out && (wave & 1) == 0 && lane == 0) {
                float *ps = A.pf_state_out + 6 * s;
                ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
            }
            __syncthreads();                                                   // E: state written out, LDS free for the next unit
        }
    } else if (wave == kFuseChains && !NYQ_FUSE_DBG_OFF(1)) {
        // ------------------------------- I/O wave -------------------------------
        DeConst D;
        deemph_init<N>(D, lane);
        for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            bool live2[2][2];
            long sS[2];
            float mem[2][2];
            int pT[2] = {0, 0}, pS[2] = {0, 0};
            float pG[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                sS[q] = 2 * grp + q;
                const bool on = sS[q] < A.nstreams;
                if (!on) sS[q] = 0;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    live2[q][c] = on;
                    mem[q][c] = (on && A.deemph) ? A.deemph[2 * sS[q] + c] : 0.f;
                }
            }
            auto fetch_params = [&](long fidx) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (!live2[q][0]) continue;
                    const long pi = sS[q] * nfr + fidx;
                    pT[q] = A.pf_pitch[pi];
                    pG[q] = A.pf_gain[pi];
                    pS[q] = A.pf_tapset[pi];
                }
            };
            auto put_params = [&](long fidx) {
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        PipeParams *ps = &pslot[q][fidx & 1];
                        ps->T = pT[q];
                        ps->g = pG[q];
                        ps->ts = pS[q];
                    }
                }
            };
            // prologue: history in front of frame 0
#pragma unroll
            for (int k = 0; k < kFuseChains; k++) {
                if (!live2[k >> 1][0]) continue;
                const long u = 2 * sS[k >> 1] + (k & 1);
#pragma unroll 1
                for (int j = lane; j < kPostHist; j += kWave) bufs[k][j] = A.hist ? A.hist[u * kPostHist + j] : 0.f;
            }
            fetch_params(0);
            put_params(0);
            if (nfr > 1) fetch_params(1);
            __syncthreads();                                                   // P
            auto emit = [&](long fidx) {
                // frame fidx is final in [1088 - N, 1088) of every buffer: de-emphasis, interleave, store
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (!live2[q][0]) continue;
                    float *const cb[2] = {bufs[2 * q] + R0 - N, bufs[2 * q + 1] + R0 - N};
                    float *const sg[2] = {stage[2 * q], stage[2 * q + 1]};
                    // (deemph_frames of the pipeline kernel, on linear buffers)
                    constexpr int CH = DeGeo<N>::CH;
                    float loc[2][CH], e[2];
#pragma unroll
                    for (int c = 0; c < 2; c++)
#pragma unroll
                        for (int j = 0; j < CH; j++) loc[c][j] = cb[c][lane * CH + j];
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        float acc = 0.f;
#pragma unroll
                        for (int j = 0; j < CH; j++) {
                            acc = (loc[c][j] + 1e-30f) + kPreemph * acc;
                            loc[c][j] = acc;
                        }
                        e[c] = acc;
                    }
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        e[c] += D.cstep[0] * dpp_zero<0x111, 0xf>(e[c]);
                        e[c] += D.cstep[1] * dpp_zero<0x112, 0xf>(e[c]);
                        e[c] += D.cstep[2] * dpp_zero<0x114, 0xf>(e[c]);
                        e[c] += D.cstep[3] * dpp_zero<0x118, 0xf>(e[c]);
                        e[c] += D.wA * dpp_zero<0x142, 0xa>(e[c]);
                        e[c] += D.wB * dpp_zero<0x143, 0xc>(e[c]);
                    }
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const float prevEnd = dpp_shr1(0.f, e[c]);
                        float cp = kPreemph * prevEnd + D.pw * mem[q][c];
#pragma unroll
                        for (int j = 0; j < CH; j++) {
                            sg[c][lane * CH + j] = (loc[c][j] + cp) * (1.f / 32768.f);
                            cp *= kPreemph;
                        }
                        mem[q][c] = kPreemph * __shfl(e[c], kWave - 1) + D.pwEnd * mem[q][c];
                    }
                    NYQ_POST_SYNC();
                    vf4 *d4 = reinterpret_cast<vf4 *>(A.out + (sS[q] * nfr * N + fidx * N) * 2);
#pragma unroll
                    for (int k = 0; k < 2 * NLD; k++) {
                        const int v = lane + k * kWave;
                        if (v < 2 * NV) {
                            const float2 l = *reinterpret_cast<const float2 *>(sg[0] + 2 * v);
                            const float2 r = *reinterpret_cast<const float2 *>(sg[1] + 2 * v);
                            d4[v] = vf4{l.x, r.x, l.y, r.y};
                        }
                    }
                    NYQ_POST_SYNC();
                }
            };
            for (long f = 0; f < nfr; f++) {
                if (f >= 1) emit(f - 1);
                if (f + 1 < nfr) {
                    put_params(f + 1);
                    if (f + 2 < nfr) fetch_params(f + 2);
                }
                __syncthreads();                                               // B1
                __syncthreads();                                               // B2
                __syncthreads();                                               // B3
            }
            if (nfr > 0) emit(nfr - 1);                                        // (the last shift made it history too)
#pragma unroll
            for (int k = 0; k < kFuseChains; k++) {
                if (!live2[k >> 1][0]) continue;
                const long u = 2 * sS[k >> 1] + (k & 1);
                if (A.hist)
#pragma unroll 1
                    for (int j = lane; j < kPostHist; j += kWave) A.hist[u * kPostHist + j] = bufs[k][j];
                if (A.deemph && lane == 0) A.deemph[u] = mem[k >> 1][k & 1];
            }
            __syncthreads();                                                   // E
        }
    } else if (!NYQ_FUSE_DBG_OFF(2)) {
        // ------------------------------- IMDCT wave -------------------------------
        LaneConst<32> K32;
        lane_init<32>(K32, lane, trig, window);
        // (the lane constants of the short-block program are fetched where a transient frame needs them: they would
        // cost 18 registers for the life of the wave, and transient frames are a few per cent)
        cpx *sliceA = reinterpret_cast<cpx *>(slice), *sliceB = reinterpret_cast<cpx *>(slice + SLICE_A);
        for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            const long s0 = 2 * grp;
            const bool on1 = s0 + 1 < A.nstreams;
            const long in_stream = nfr * 2 * (long)N;                          // floats from stream s0's frame to stream s0+1's
            // overlap carry of the four chains
            for (int i = lane; i < kFuseChains * kHalfOv; i += kWave) {
                const int k = i / kHalfOv, j = i - k * kHalfOv;
                const bool on = k < 2 || on1;
                tails[k][j] = (on && A.ov_state) ? A.ov_state[(2 * s0 + k) * kHalfOv + j] : 0.f;
            }
            NYQ_WAVE_SYNC();
            // The next frame's coefficients are not held in registers across the barriers (64 VGPRs that the allocator
            // then spills): each lane TOUCHES two of the frame's 120 cache lines one frame ahead (the lines are on their
            // way into L2 while the current frame is transformed), the real 16-byte loads at the start of the frame hit there.
            float touch0 = 0.f, touch1 = 0.f;
            unsigned tmask = 0, tnext = 0;                                     // bit q: stream q's frame is transient
            auto flags = [&](long fidx) {
                unsigned m = 0;
                if (A.transient) {
                    if (A.transient[s0 * nfr + fidx]) m |= 1u;
                    if (on1 && A.transient[(s0 + 1) * nfr + fidx]) m |= 2u;
                }
                return m;
            };
            auto touch = [&](long fidx, unsigned tm) {
                // line l of the 4 x 30 lines of 128 B of the frame's long rows: row l / 30, line l % 30
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int l = lane + 64 * h;
                    const int row = l / 30, line = l - row * 30;
                    const bool ok = l < 120 && (row < 2 || on1) && !((tm >> (row >> 1)) & 1u);
                    const float *p = A.freq + ((s0 + (row >> 1)) * nfr + fidx) * 2 * (long)N + (row & 1) * (long)N + line * 32;
                    const float t = ok ? *p : 0.f;
                    if (h == 0) touch0 = t; else touch1 = t;
                }
            };
            auto long_rows = [&](long fidx, unsigned tm) {
                FusedLongRows rows;
                rows.in0 = A.freq + (s0 * nfr + fidx) * 2 * (long)N;
                rows.in_stream = in_stream;
                rows.srow0 = slice;
                rows.tails = tails[0];
                rows.valid_mask = ((tm & 1u) ? 0u : 3u) | ((!on1 || (tm & 2u)) ? 0u : 12u);
                return rows;
            };
            auto short_rows = [&](long fidx, int q, float *sl) {
                FusedShortRows rows;
                rows.in0 = A.freq + ((s0 + q) * nfr + fidx) * 2 * (long)N;
                rows.srow0 = sl;
                rows.tails = tails[2 * q];
                rows.on = true;
                return rows;
            };
            // frame fidx (its long rows are in R) up to the finished samples, left in the slices in the rotated layout
            auto front = [&](long fidx, unsigned tm) {
                // (an opaque copy of the lane id per call: otherwise every per-lane LDS / global offset of the lane program is
                // hoisted out of the frame loop and kept alive -- hundreds of registers, spilled to scratch)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                asm volatile("" ::"v"(touch0), "v"(touch1));                  // (the touches have arrived)
                if ((tm & 1u) == 0 || (on1 && (tm & 2u) == 0)) {
                    // (a fresh opaque lane id and a scheduling barrier per phase: the per-lane offsets of one phase are
                    // neither computed ahead of it nor kept alive behind it)
                    StageRegs<32> R;
                    stage_in_load<32, 0>(R, ln, long_rows(fidx, tm));
                    NYQ_WAVE_SYNC();
                    stage_in_store<32>(R, K32, ln, sliceA);
                    NYQ_WAVE_SYNC();
                    __builtin_amdgcn_sched_barrier(0);
                    fft_passes<32>(opaque(lane), sliceA);
                    __builtin_amdgcn_sched_barrier(0);
                    const FusedLongRows rows = long_rows(fidx, tm);
                    HeadRegs<32> H;
                    stage_out<32, 0>(K32, opaque(lane), sliceA, nullptr, rows, H);
                    NYQ_WAVE_SYNC();
                    __builtin_amdgcn_sched_barrier(0);
                }
                int nshort = 0;
#pragma unroll 1
                for (int q = 0; q < 2; q++) {
                    if (NYQ_FUSE_DBG_OFF(3) || !((tm >> q) & 1u)) continue;
                    float *sl = (nshort == 0) ? slice + SLICE_A : slice;       // (two transient streams: no long rows, slice A is free)
                    const FusedShortRows rows = short_rows(fidx, q, sl);
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        LaneConst<4> K4;
                        lane_init<4>(K4, ln, trig, window);
                        StageRegs<4> R4;
                        stage_in_load<4, 0>(R4, ln, rows);
                        NYQ_WAVE_SYNC();
                        stage_in_store<4>(R4, K4, ln, reinterpret_cast<cpx *>(sl));
                        NYQ_WAVE_SYNC();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    fft_passes<4>(opaque(lane), reinterpret_cast<cpx *>(sl));
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const int l3 = opaque(lane);
                        LaneConst<4> K4;
                        lane_init<4>(K4, l3, trig, window);
                        HeadRegs<4> H;
                        stage_out<4, 0>(K4, l3, reinterpret_cast<cpx *>(sl), sring, rows, H);
                        NYQ_WAVE_SYNC();
                        stage_out_heads<4, 0>(K4, l3, sring, rows, H);
                        NYQ_WAVE_SYNC();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    nshort++;
                }
            };
            // the finished frame from the slices into the chains' frame regions: out[0..60) sits behind out[60..N2)
            auto back = [&](unsigned tm) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                constexpr int NV4 = N / 4;                                      // 240 float4 per chain-frame
#pragma unroll
                for (int k = 0; k < kFuseChains; k++) {
                    const bool kl = k < 2 || on1;
                    if (!kl || ((tm >> (k >> 1)) & 1u)) continue;
                    const float *srow = slice + k * (2 * Geo<32>::S);
                    float *dst = bufs[k] + R0;
#pragma unroll
                    for (int it = 0; it < (NV4 + kWave - 1) / kWave; it++) {
                        const int v = ln + it * kWave;                       // float4 v of the frame
                        if (v < NV4) sts4(dst, 4 * v, *reinterpret_cast<const vf4 *>(srow + (v < 15 ? N - kHalfOv + 4 * v : 4 * v - kHalfOv)));
                    }
                }
                int nshort = 0;
#pragma unroll 1
                for (int q = 0; q < 2; q++) {
                    if (NYQ_FUSE_DBG_OFF(3) || !((tm >> q) & 1u)) continue;
                    const float *sl = (nshort == 0) ? slice + SLICE_A : slice;
                    // row g = (channel g / 8, block g % 8): 30 float4 each, 16 rows
                    for (int v = ln; v < 16 * 30; v += kWave) {
                        const int g = v / 30, w = v - g * 30;
                        const float *srow = sl + g * (2 * Geo<4>::S);
                        sts4(bufs[2 * q + (g >> 3)] + R0 + 120 * (g & 7), 4 * w,
                             *reinterpret_cast<const vf4 *>(srow + (w < 15 ? 120 - kHalfOv + 4 * w : 4 * w - kHalfOv)));
                    }
                    nshort++;
                }
                NYQ_WAVE_SYNC();
            };
            // prologue: frame 0 complete, frame 1 prepared up to the FFT output, loads of frame 2 in flight
            if (nfr > 0) {
                tmask = flags(0);
                front(0, tmask);
                back(tmask);
                if (nfr > 1) {
                    tnext = flags(1);
                    touch(1, tnext);
                }
            }
            __syncthreads();                                                   // P
            for (long f = 0; f < nfr; f++) {
                // during comb(f): frame f+1 up to the FFT output, then the loads of frame f+2
                const unsigned tm1 = tnext;
                if (f + 1 < nfr) {
                    front(f + 1, tm1);
                    if (f + 2 < nfr) {
                        tnext = flags(f + 2);
                        touch(f + 2, tnext);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                               // B1
                __syncthreads();                                               // B2: the comb waves have picked up what stays
                if (f + 1 < nfr) back(tm1);
                __syncthreads();                                               // B3
            }
            if (A.ov_state)
                for (int i = lane; i < kFuseChains * kHalfOv; i += kWave) {
                    const int k = i / kHalfOv, j = i - k * kHalfOv;
                    if (k < 2 || on1) A.ov_state[(2 * s0 + k) * kHalfOv + j] = tails[k][j];
                }
            __syncthreads();                                                   // E
        }
    }
}

}  // namespace nyq
