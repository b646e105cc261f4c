// nyq_chain_kernel.hpp -- freq[] -> interleaved PCM in ONE launch (round 4: the third fusion, the one that ships).
//
// Everything celt_decode_with_ec does after denormalise_bands (celt_decoder_clean.c:620-723) for many stereo 20 ms
// streams at once:  compute_inv_mdcts (:264-312, clt_mdct_backward mdct.c:267-379)  ->  comb_filter (celt.c:114-172 as
// applied :658-683)  ->  deemphasis (:192-256) with scaling and channel interleave.  The two-kernel chain
// (nyq_celt_synth_dev + nyq_celt_post_dev) writes the time-domain frame to HBM and reads it back: 2 x 7680 B per
// channel-frame.  Here it never leaves the CU: 3840 B in, 3840 B out.
//
// Workgroup = the post-filter pipeline of nyq_post_pipe.hpp (one stereo stream = 2 chains) plus ONE transform wave:
//     wave 0, 1   comb wave of chain 0 / 1: post-rotation + TDAC mirror of ITS frame f (the transform's last phase, in place,
//                 the carry in four of its registers), then the recursion of frame f
//     wave 2      I/O wave: de-emphasis + interleave + 16-byte global stores of frame f-1 STRAIGHT out of the filtered
//                 history (two adjacent samples of both channels per lane: the output's own order, so there is no staging
//                 pass and the frame region of `nxt` is free), history carry-over, post-filter parameters of frame f+1
//     wave 3      transform wave: pre-rotation and the inverse FFT of frame f+1 of both channels IN PLACE in the frame regions
//                 of `nxt` (nyq_fuse_lanes.hpp: no LDS of its own), the coefficients of frame f+2 prefetched into registers
//   one s_barrier per frame, 32 544 B of LDS as before: four workgroups (8 chains, 16 waves) per CU, 128 VGPRs.
// What made the earlier fusions lose was the transform's 16 KB LDS slice per wave against the 8
// resident chains x 16 KB the recursion needs; this transform needs none.
#pragma once
#include "nyq_fuse_lanes.hpp"
#include "nyq_kernels.hpp"
#include "nyq_post_pipe.hpp"

namespace nyq {

constexpr int kChainWaves = 4;     // 2 comb + I/O + transform

// Diagnostic build only (-DNYQ_PIPE_STAMPS, tools/chain_stamps.py): s_memtime accounting per role and phase, summed over
// the workgroups into a buffer nothing else reads.  The product build compiles none of it.
#ifdef NYQ_PIPE_STAMPS
__device__ unsigned long long g_chain_stamps[32];
#define NYQ_CSTAMP_DECL() unsigned long long cs_t = __builtin_amdgcn_s_memtime(), cs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define NYQ_CSTAMP(slot)                                               \
    do {                                                               \
        const unsigned long long cs_n = __builtin_amdgcn_s_memtime();  \
        cs_acc[slot] += cs_n - cs_t;                                   \
        cs_t = cs_n;                                                   \
    } while (0)
#define NYQ_CSTAMP_FLUSH(base)                                                                                      \
    do {                                                                                                            \
        if (lane == 0)                                                                                              \
            for (int cs_i = 0; cs_i < 8; cs_i++) atomicAdd(&g_chain_stamps[(base) + cs_i], cs_acc[cs_i]);           \
    } while (0)
#define NYQ_CSTAMP_ARGS , unsigned long long &cs_t, unsigned long long (&cs_acc)[8]
#define NYQ_CSTAMP_PASS , cs_t, cs_acc
#else
#define NYQ_CSTAMP_DECL() do { } while (0)
#define NYQ_CSTAMP(slot) do { } while (0)
#define NYQ_CSTAMP_FLUSH(base) do { } while (0)
#define NYQ_CSTAMP_ARGS
#define NYQ_CSTAMP_PASS
#endif

struct ChainArgs {
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
    // freq / transient, and pf_* / out, may be WINDOWS into longer per-stream arrays: consecutive streams are `fstride` /
    // `pstride` frames apart (0 = dense: nframes)
    long fstride, pstride;
    const OutDesc *desc;             // [nstreams] or null: per-stream destinations in a file's interleaved layout (OutDesc)
    __host__ __device__ long fs() const { return fstride ? fstride : nframes; }
    __host__ __device__ long ps() const { return pstride ? pstride : nframes; }
};

// weighted inclusive scan over the 64 lanes, ratio q per lane (the DPP steps of deemph_frames)
__device__ __forceinline__ float deemph_scan(float e, const DeConst &D) {
    e += D.cstep[0] * dpp_zero<0x111, 0xf>(e);      // row_shr:1
    e += D.cstep[1] * dpp_zero<0x112, 0xf>(e);      // row_shr:2
    e += D.cstep[2] * dpp_zero<0x114, 0xf>(e);      // row_shr:4
    e += D.cstep[3] * dpp_zero<0x118, 0xf>(e);      // row_shr:8
    e += D.wA * dpp_zero<0x142, 0xa>(e);            // row_bcast:15 into rows 1, 3
    e += D.wB * dpp_zero<0x143, 0xc>(e);            // row_bcast:31 into rows 2, 3
    return e;
}

// scan constants for CH consecutive samples per lane, all 64 lanes (ratio q = c^CH)
template <int CH>
__device__ __forceinline__ void deemph_init_lanes(DeConst &D, float &pwHalf, int lane) {
    float q = 1.f;
#pragma unroll
    for (int k = 0; k < CH; k++) q *= kPreemph;
    float st[7];
    st[0] = q;
#pragma unroll
    for (int k = 1; k < 7; k++) st[k] = st[k - 1] * st[k - 1];
    auto powq = [&](int e) {
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
    D.pwEnd = st[6];      // q^64
    pwHalf = st[5];       // q^32
}

// K weighted scans side by side: their dependent chains (a DPP move and a multiply-add per step, wait states between them)
// interleave, which is what the I/O wave's time is made of
template <int K>
__device__ __forceinline__ void deemph_scan_n(float (&e)[K], const DeConst &D) {
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.cstep[0] * dpp_zero<0x111, 0xf>(e[k]);      // row_shr:1
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.cstep[1] * dpp_zero<0x112, 0xf>(e[k]);      // row_shr:2
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.cstep[2] * dpp_zero<0x114, 0xf>(e[k]);      // row_shr:4
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.cstep[3] * dpp_zero<0x118, 0xf>(e[k]);      // row_shr:8
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.wA * dpp_zero<0x142, 0xa>(e[k]);            // row_bcast:15 into rows 1, 3
#pragma unroll
    for (int k = 0; k < K; k++) e[k] += D.wB * dpp_zero<0x143, 0xc>(e[k]);            // row_bcast:31 into rows 2, 3
}
__device__ __forceinline__ float lane_value(float v, int l) {   // v of lane l (a compile-time lane), through an SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// deemphasis() (celt_decoder_clean.c:243-248) + 1/32768 + stereo interleave of ONE frame, from the filtered samples in
// LDS straight to global memory: lane l of block b owns samples 2 (l + 64 b), + 1 of both channels = float4 l + 64 b of
// the interleaved frame.  Per block: two-sample recurrence, one weighted scan over the lanes, carry from the block before.
// Blocks go two at a time (four scans side by side); only the carry (one multiply-add per block and channel) is serial.
// Dm.base != null: the samples go through the stream's descriptor (ts0 = stream sample index of the frame's first sample)
template <int N>
__device__ __forceinline__ void deemph_store_pair(const float *sL, const float *sR, vf4 *d4, float &memL, float &memR,
                                                  int lane, const DeConst &D, float pwHalf, const OutDesc &Dm, long long ts0) {
    constexpr int NV2 = N / 2;                       // float4 of the interleaved frame
    constexpr int NB = (NV2 + kWave - 1) / kWave;    // blocks of 64
    static_assert(NV2 % kWave == 0 || NV2 % kWave == 32, "the last block is whole or half");
    static_assert(NB % 2 == 0, "blocks go in pairs");
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += 2) {
        float x0[4], x1[4], e[4];                    // [2 * block + channel]
        bool on[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int v = lane + (b0 + k) * kWave;
            on[k] = v < NV2;
            const int vv = on[k] ? v : 0;
            const float2 l = *reinterpret_cast<const float2 *>(sL + 2 * vv);
            const float2 r = *reinterpret_cast<const float2 *>(sR + 2 * vv);
            x0[2 * k] = l.x + 1e-30f;                // + VERY_SMALL
            x0[2 * k + 1] = r.x + 1e-30f;
            x1[2 * k] = (l.y + 1e-30f) + kPreemph * x0[2 * k];
            x1[2 * k + 1] = (r.y + 1e-30f) + kPreemph * x0[2 * k + 1];
            e[2 * k] = on[k] ? x1[2 * k] : 0.f;
            e[2 * k + 1] = on[k] ? x1[2 * k + 1] : 0.f;
        }
        deemph_scan_n<4>(e, D);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int v = lane + (b0 + k) * kWave;
            // value entering the lane's pair: c t[2l-1] = c e[l-1] + q^l mem, with mem = c t[-1]
            const float cL = kPreemph * dpp_shr1(0.f, e[2 * k]) + D.pw * memL, cR = kPreemph * dpp_shr1(0.f, e[2 * k + 1]) + D.pw * memR;
            const vf4 o = {(x0[2 * k] + cL) * (1.f / 32768.f), (x0[2 * k + 1] + cR) * (1.f / 32768.f),
                           (x1[2 * k] + kPreemph * cL) * (1.f / 32768.f), (x1[2 * k + 1] + kPreemph * cR) * (1.f / 32768.f)};
            if (on[k]) {
                if (Dm.base) {
                    const long long ts = ts0 + 2 * v;
                    mapped_put(Dm, ts, Dm.coff0, o.x);
                    mapped_put(Dm, ts, Dm.coff1, o.y);
                    mapped_put(Dm, ts + 1, Dm.coff0, o.z);
                    mapped_put(Dm, ts + 1, Dm.coff1, o.w);
                } else {
                    pipe_st(d4 + v, o);
                }
            }
            const bool half = (b0 + k + 1) * kWave > NV2;     // the last block ends at lane 31
            const float pe = half ? pwHalf : D.pwEnd;
            memL = kPreemph * lane_value(e[2 * k], half ? 31 : 63) + pe * memL;
            memR = kPreemph * lane_value(e[2 * k + 1], half ? 31 : 63) + pe * memR;
        }
    }
}

// ---- the transform wave's frame: coefficients in R -> FFT output (natural order) in reg[0], reg[1] ----
// `next`: the coefficients of the frame after this one, or null.  Their loads are issued at the END of the frame's transform
// (NYQ_CHAIN_EARLY_LOAD = 0: they have the rest of the iteration and the barrier to arrive in -- the comb waves, not this wave,
// are what a frame waits for).  Measured in one process (profiles/r04_*): issued behind S2 (= 2) 1-3 % SLOWER -- the eight
// 1 KB loads take 2400 cycles to ISSUE in the middle of the iteration against 400-800 at its end --, behind S0 (= 1) 128
// VGPRs and a spilled lane id inside the frame loop.
#ifndef NYQ_CHAIN_EARLY_LOAD
#define NYQ_CHAIN_EARLY_LOAD 0
#endif
// NYQ_CHAIN_TOUCH: right behind S0 / T0 every lane reads one dword of one of the next frame's 60 cache lines (default cache
// policy), so that the 16-byte loads issued later find the lines in L2.  Measured: +-1 %, not used.
#ifndef NYQ_CHAIN_TOUCH
#define NYQ_CHAIN_TOUCH 0
#endif
__device__ __forceinline__ float xf_touch(const float *next, int lane) {
    return (NYQ_CHAIN_TOUCH && next && lane < 60) ? next[32 * lane] : 0.f;
}
__device__ __forceinline__ void xf_long_front(fx::XfRegs &R, const fx::XfRot &K, const fx::XfTw &W, int lane, float *const (&reg)[2],
                                              const float *next, float &touch NYQ_CSTAMP_ARGS) {
    using namespace fx;
#ifdef NYQ_PIPE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NYQ_CSTAMP(1);                                                 // slot 1: wait for the prefetched coefficients
#endif
#pragma unroll
    for (int r = 0; r < 2; r++) xf_long_s0(R, K, W, opaque(lane), r, reg[r]);
    NYQ_WAVE_SYNC();
    touch = xf_touch(next, opaque(lane));
    if (NYQ_CHAIN_EARLY_LOAD == 1 && next) xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), next);
    NYQ_CSTAMP(2);                                                 // slot 2: S0
    {
        cpx u[16];
        const int ln = opaque(lane);
        xf_long_s2_load(ln, reg, u);
        xf_long_s2_store(ln, reg, u);
    }
    NYQ_WAVE_SYNC();
    NYQ_CSTAMP(3);                                                 // slot 3: S2
    if (NYQ_CHAIN_EARLY_LOAD == 2 && next) xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), next);
    {
        cpx v[15];
        const int ln = opaque(lane);
        xf_long_s3_load(ln, reg, v);
        NYQ_WAVE_SYNC();
        xf_long_s3_store(ln, reg, v);
    }
    NYQ_WAVE_SYNC();
    NYQ_CSTAMP(4);                                                 // slot 4: S3
}

// transient frame: de-interleave, pre-rotation, the two passes -> natural-order FFT output of the 16 rows
__device__ __forceinline__ void xf_short_front(fx::XfRegs &R, int lane, float *const (&reg)[2], const float *__restrict__ trig,
                                               const float *next, float &touch) {
    using namespace fx;
    // (the short program's six rotation values are fetched where a transient frame needs them: a few per cent of the frames)
    XfShortConst S;
    xf_short_init(S, opaque(lane), trig);
    xf_short_t0(R, opaque(lane), reg);
    NYQ_WAVE_SYNC();
    touch = xf_touch(next, opaque(lane));
    if (NYQ_CHAIN_EARLY_LOAD == 1 && next) xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), next);
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
        XfShortIn I;
        const int ln = opaque(lane);
        xf_short_t1_load(ln, s, reg, I);
        NYQ_WAVE_SYNC();
        xf_short_t1_store(S, ln, s, reg, I);
        NYQ_WAVE_SYNC();
    }
#pragma unroll 1
    for (int it = 0; it < 4; it++) xf_short_t2(opaque(lane), it, reg);
    NYQ_WAVE_SYNC();
    {
        cpx v[15];
        const int ln = opaque(lane);
        xf_short_t3_load(ln, reg, v);
        NYQ_WAVE_SYNC();
        xf_short_t3_store(ln, reg, v);
    }
    NYQ_WAVE_SYNC();
    if (NYQ_CHAIN_EARLY_LOAD >= 2 && next) xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), next);
}

// ---- the comb wave's share: FFT output of ITS chain's frame -> finished samples, in place, right before it filters them ----
// (the tail travels as a native vector: an array of 16-byte structs is copied with memcpy and stays in scratch memory)
__device__ __forceinline__ void xf_long_back(const fx::XfRot &K, const fx::XfWin &Wn, int lane, float *region, vf4 &tail) {
    using namespace fx;
    XfOut O;
    const int ln = opaque(lane);
    xf_long_s4_load(K, ln, region, O);
    NYQ_WAVE_SYNC();
    f4 t = {tail.x, tail.y, tail.z, tail.w};
    xf_long_s4_store(Wn, ln, region, O, t);
    tail = vf4{t.x, t.y, t.z, t.w};
    NYQ_WAVE_SYNC();
}
__device__ __forceinline__ void xf_short_back(const fx::XfWin &Wn, int lane, int c, float *region, vf4 &tail, const float *__restrict__ trig) {
    using namespace fx;
    XfShortConst S;
    xf_short_init(S, opaque(lane), trig);
    float *const reg[2] = {region, region};                        // (both slots: the lane functions select by channel)
    vf4 bk7 = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 2; s++) {                                  // post-rotation of the channel's 8 rows, in place
        const f4 bk = xf_short_t4(S, opaque(lane), 2 * c + s, reg);
        if (s == 1) bk7 = vf4{bk.x, bk.y, bk.z, bk.w};
    }
    NYQ_WAVE_SYNC();
#pragma unroll
    for (int h = 1; h >= 0; h--) {
        XfMirror M;
        const int ln = opaque(lane);
        xf_short_t5_load(ln, c, h, reg, f4{tail.x, tail.y, tail.z, tail.w}, M);
        NYQ_WAVE_SYNC();
        xf_short_t5_store(Wn, ln, c, h, reg, M);
        NYQ_WAVE_SYNC();
    }
    // the frame's tail: raw second half of block 7, from the lanes that post-rotated it
    const int src = short_tail_src(lane);
    tail = vf4{__shfl(bk7.x, src), __shfl(bk7.y, src), __shfl(bk7.z, src), __shfl(bk7.w, src)};
}

#ifndef NYQ_CHAIN_MINWAVES
#define NYQ_CHAIN_MINWAVES 4
#endif

// LM 3 (20 ms frames), stereo streams.  Workgroup unit = one stream.
__global__ __launch_bounds__(kWave *kChainWaves, NYQ_CHAIN_MINWAVES) void celt_chain_kernel(ChainArgs A, const float *__restrict__ trig,
                                                                                     const float *__restrict__ window) {
    constexpr int N = fx::kN;
    constexpr int R0 = kPipeHist;
    constexpr int KEEP = kPipeHist - N;
    constexpr int OLD = kPostHist - kPipeHist;
    static_assert(N >= OLD, "the previous buffer still holds the history the state hand-over needs");
    __shared__ __attribute__((aligned(16))) float bufs[kPipeUnits][2][kPipeHist + N];
    __shared__ __attribute__((aligned(16))) float win2[kOverlap];
    __shared__ __attribute__((aligned(16))) PipeParams pslot[2];           // [frame parity]: both chains share the stream's parameters
    for (int i = threadIdx.x; i < kOverlap; i += kWave * kChainWaves) win2[i] = window[i] * window[i];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long nfr = A.nframes;
    const long pst = A.ps(), fst = A.fs();
    __syncthreads();

    // Every role runs the same unit loop and the same barriers per unit: P (prologues done), then nfr + 1 frame barriers
    // (the last also separates this unit's LDS use from the next unit's prologue).
    if (wave < kPipeUnits) {
        // ------------------------------- comb wave of chain `wave` (as nyq_post_pipe.hpp) -------------------------------
#ifndef NYQ_CHAIN_COMB_PRIO
#define NYQ_CHAIN_COMB_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(NYQ_CHAIN_COMB_PRIO);
        fx::XfRot Kr;
        fx::XfWin Kw;
        fx::xf_init_rot(Kr, lane, trig);
        fx::xf_init_win(Kw, lane, window);
        for (long s = blockIdx.x; s < A.nstreams; s += gridDim.x) {
            float *bA = bufs[wave][0], *bB = bufs[wave][1];
            int T_old = 0, T_cur = 0, ts_old = 0, ts_cur = 0;
            float g_old = 0.f, g_cur = 0.f;
            // overlap carry of this chain (lane j < 15 holds state[56-4j .. 59-4j]) and the stream's transient flags, 64 frames
            // at a time (one byte per lane + a ballot)
            vf4 tail = {0, 0, 0, 0};
            if (A.ov_state && fx::tail_lane(lane)) tail = *reinterpret_cast<const vf4 *>(A.ov_state + (2 * s + wave) * kHalfOv + fx::tail_offset(lane, 0));
            const unsigned char *tbase = A.transient ? A.transient + s * fst : nullptr;
            unsigned long long tmask = 0;
            if (A.pf_state) {
                const float *ps = A.pf_state + 6 * s;
                T_old = (int)ps[0]; T_cur = (int)ps[1]; g_old = ps[2]; g_cur = ps[3]; ts_old = (int)ps[4]; ts_cur = (int)ps[5];
            }
            __syncthreads();                                                   // P
            NYQ_CSTAMP_DECL();
            for (long f = 0; f <= nfr; f++) {
                NYQ_CSTAMP(0);                                                 // slot 0: at the barrier
                if (f < nfr) {
                    float *cur = (f & 1) ? bB : bA, *nxt = (f & 1) ? bA : bB;
                    float *mir = nxt - N;                                      // mir[idx] = nxt[idx - N]
                    const PipeParams P = pslot[f & 1];
                    const int T_new = __builtin_amdgcn_readfirstlane(P.T), ts_new = __builtin_amdgcn_readfirstlane(P.ts);
                    const float g_new = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P.g)));
                    if (T_cur < kCombMinPeriod) T_cur = kCombMinPeriod;       // celt_decoder_clean.c:661-662
                    if (T_old < kCombMinPeriod) T_old = kCombMinPeriod;
                    if (T_cur > kCombMaxPeriod) T_cur = kCombMaxPeriod;
                    if (T_old > kCombMaxPeriod) T_old = kCombMaxPeriod;
                    const int T_nw = T_new < kCombMinPeriod ? kCombMinPeriod : T_new > kCombMaxPeriod ? kCombMaxPeriod : T_new;
                    // the frame arrives as the transform wave left it: natural-order FFT output.  Post-rotation + TDAC mirror
                    // (mdct.c:322-377) here, in place, then the filter.
                    if ((f & 63) == 0) {
                        bool t = false;
                        if (tbase && f + lane < nfr) t = tbase[f + lane] != 0;
                        tmask = __ballot(t);
                    }
#ifndef NYQ_CHAIN_DBG_NO_BACK
                    if ((tmask >> (f & 63)) & 1ull) xf_short_back(Kw, lane, wave, cur + R0, tail, trig);
                    else xf_long_back(Kr, Kw, lane, cur + R0, tail);
#endif
                    NYQ_CSTAMP(2);                                             // slot 2: post-rotation + mirror
#ifndef NYQ_CHAIN_DBG_NO_COMB
                    pipe_comb_call<true>(cur, mir, lane, R0, kOverlap, T_old, T_cur, g_old, g_cur, ts_old, ts_cur, win2);
                    pipe_comb_call<true>(cur, mir, lane, R0 + kOverlap, N - kOverlap, T_cur, T_nw, g_cur, g_new, ts_cur, ts_new, win2);
#else
                    (void)T_nw; (void)mir; (void)cur;
#endif
                    T_old = T_cur = T_new; g_old = g_cur = g_new; ts_old = ts_cur = ts_new;   // :672-683 (LM != 0)
                }
                NYQ_CSTAMP(1);                                                 // slot 1: the frame's comb steps
                __syncthreads();
            }
            if (wave == 0) NYQ_CSTAMP_FLUSH(0);
            if (A.ov_state && fx::tail_lane(lane)) *reinterpret_cast<vf4 *>(A.ov_state + (2 * s + wave) * kHalfOv + fx::tail_offset(lane, 0)) = tail;
            if (A.pf_state_out && wave == 0 && lane == 0) {
                float *ps = A.pf_state_out + 6 * s;
                ps[0] = (float)T_old; ps[1] = (float)T_cur; ps[2] = g_old; ps[3] = g_cur; ps[4] = (float)ts_old; ps[5] = (float)ts_cur;
            }
        }
    } else if (wave == kPipeUnits) {
        // ------------------------------- I/O wave -------------------------------
        DeConst D;
        float pwHalf;
        deemph_init_lanes<2>(D, pwHalf, lane);
        for (long s = blockIdx.x; s < A.nstreams; s += gridDim.x) {
            const long u0 = 2 * s;
            const OutDesc Dm = load_desc(A.desc, s);
            float memL = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A.deemph ? A.deemph[u0] : 0.f)));
            float memR = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A.deemph ? A.deemph[u0 + 1] : 0.f)));
            int pT = 0, pS = 0;
            float pG = 0.f;
            auto fetch_params = [&](long fidx) {
                const long pi = s * pst + fidx;
                pT = A.pf_pitch[pi];
                pG = A.pf_gain[pi];
                pS = A.pf_tapset[pi];
            };
            auto put_params = [&](long fidx) {
                if (lane == 0) {
                    PipeParams *ps = &pslot[fidx & 1];
                    ps->T = pT;
                    ps->g = pG;
                    ps->ts = pS;
                }
            };
            // prologue: buffer A's history = what precedes frame 0
#pragma unroll
            for (int k = 0; k < kPipeUnits; k++) {
#pragma unroll 1
                for (int j = lane; j < kPipeHist; j += kWave) bufs[k][0][j] = A.hist ? A.hist[(u0 + k) * kPostHist + OLD + j] : 0.f;
            }
            if (nfr > 0) {
                fetch_params(0);
                put_params(0);
                if (nfr > 1) fetch_params(1);
            }
            __syncthreads();                                                   // P
            NYQ_CSTAMP_DECL();
            for (long f = 0; f <= nfr; f++) {
                NYQ_CSTAMP(0);                                                 // slot 0: at the barrier
                const int cb = (int)(f & 1), nb = cb ^ 1;
                if (f < nfr && KEEP > 0) {
#pragma unroll
                    for (int k = 0; k < kPipeUnits; k++) pipe_copy<true>(bufs[k][cb] + N, bufs[k][nb], lane, 0, KEEP);
                }
                if (f + 1 < nfr) {
                    put_params(f + 1);
                    if (f + 2 < nfr) fetch_params(f + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
                NYQ_CSTAMP(1);                                                 // slot 1: carry-over, parameters
                // frame f-1 is final in cur[R0 - N, R0)
#ifndef NYQ_CHAIN_DBG_NO_IO
                if (f >= 1)
#else
                if (f >= 1 && nfr < 0)
#endif
                    deemph_store_pair<N>(bufs[0][cb] + R0 - N, bufs[1][cb] + R0 - N,
                                         reinterpret_cast<vf4 *>(A.out + (s * pst * N + (f - 1) * N) * 2), memL, memR, opaque(lane), D, pwHalf,
                                         Dm, Dm.t0 + (f - 1) * (long long)N);
                if (f == nfr) {
                    // state for the next call: the last 1088 outputs (1040 in cur's history region, the 48 before them in nxt's)
#pragma unroll
                    for (int k = 0; k < kPipeUnits; k++) {
                        if (A.hist && nfr > 0)
#pragma unroll 1
                            for (int j = lane; j < kPostHist; j += kWave)
                                A.hist[(u0 + k) * kPostHist + j] = j < OLD ? bufs[k][nb][N - OLD + j] : bufs[k][cb][j - OLD];
                    }
                    if (A.deemph && lane == 0) {
                        A.deemph[u0] = memL;
                        A.deemph[u0 + 1] = memR;
                    }
                }
                NYQ_CSTAMP(2);                                                 // slot 2: de-emphasis + stores
                __syncthreads();
            }
            NYQ_CSTAMP_FLUSH(8);
        }
    } else {
        // ------------------------------- transform wave -------------------------------
#ifdef NYQ_CHAIN_XF_PRIO
        __builtin_amdgcn_s_setprio(NYQ_CHAIN_XF_PRIO);
#endif
        fx::XfRot K;
        fx::XfTw W;
        fx::xf_init_rot(K, lane, trig);
        fx::xf_init_tw(W, lane);
        for (long s = blockIdx.x; s < A.nstreams; s += gridDim.x) {
            const float *fbase = A.freq + s * fst * 2 * (long)N;
            const unsigned char *tbase = A.transient ? A.transient + s * fst : nullptr;
            // transient flags of 64 frames at a time: one byte per lane + a ballot
            unsigned long long tmask = 0;
            auto flags = [&](long f0) {
                bool t = false;
                if (tbase && f0 + lane < nfr) t = tbase[f0 + lane] != 0;
                return __ballot(t);
            };
            fx::XfRegs R;
            float touch = 0.f;
            if (nfr > 0) fx::xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), fbase);
            // iteration g transforms frame g in front of barrier g: frame 0 in front of P, frame f + 1 beside the filtering of
            // frame f; the last two barriers have no transform beside them (one call site for each of the two programs)
            NYQ_CSTAMP_DECL();
            for (long g = 0; g < nfr + 2; g++) {
                NYQ_CSTAMP(0);                                     // slot 0: at the barrier
                if (g < nfr) {
                    if ((g & 63) == 0) tmask = flags(g);
                    float *const reg[2] = {bufs[0][g & 1] + R0, bufs[1][g & 1] + R0};
                    const float *next = g + 1 < nfr ? fbase + (g + 1) * 2 * (long)N : nullptr;
#ifndef NYQ_CHAIN_DBG_NO_XF
                    if (NYQ_CHAIN_TOUCH) asm volatile("" ::"v"(touch));          // (the touch of this frame's lines has come back)
                    if ((tmask >> (g & 63)) & 1ull) xf_short_front(R, lane, reg, trig, next, touch);
                    else xf_long_front(R, K, W, lane, reg, next, touch NYQ_CSTAMP_PASS);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if (!NYQ_CHAIN_EARLY_LOAD && next) fx::xf_load<NYQ_PIPE_NT & 1>(R, opaque(lane), next);
                    NYQ_CSTAMP(6);                                 // slot 6: transient frames whole; issue of the next frame's loads
                }
                __syncthreads();
            }
            NYQ_CSTAMP_FLUSH(16);
        }
    }
}

}  // namespace nyq
