// nyq_post_common.hpp -- what every form of the post-filter stage shares: the operator's arguments, the constants of
// comb_filter (celt.c:114-172) and deemphasis() (celt_decoder_clean.c:192-256), LDS access helpers.  The shipped form is
// the workgroup pipeline of nyq_post_pipe.hpp; nyq_post_kernels.hpp (round 1: one wave per channel / stereo pair) and
// nyq_chain_fused.hpp are A/B forms compiled only with -DNYQ_AB_FORMS.
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_imdct_lanes.hpp"

namespace nyq {

constexpr int kPostRing = 2048;         // per-channel LDS buffer: DECODE_BUFFER_SIZE, celt_decoder_clean.c:59
constexpr int kPostHist = 1088;         // history in front of the frame (>= COMBFILTER_MAXPERIOD + 2); also handed from call to call
constexpr int kCombMinPeriod = 15;      // celt.h:188
constexpr int kCombMaxPeriod = 1024;    // celt.h:187 (keeps every tap inside the buffer whatever the caller passes)
constexpr float kPreemph = 0.85000610f; // mode->preemph[0], static_modes_float.h:581

// Where an elementary stream's samples go when they do not go to the dense [stream][sample][channel] output: straight into
// the interleaved layout of the FILE the stream belongs to (row f3 of SURVEY.md section 8: opus_copy_channel_out_float,
// opus_multistream_decoder.c:305-331, with the pre-skip / end trim of opusfile and the header gain of
// opus_decoder_clean.c:700-712 applied on the way).  Same layout as nyq_out_desc of include/nyq_imdct.h.
struct OutDesc {
    float *base;             // device memory: stream sample `first` of destination channel slot 0
    long long first, last;   // stream samples [first, last) are written, the rest is dropped (trimmed)
    long long t0;            // stream sample index of the call's first sample
    int cstride;             // floats from one sample to the next in the destination (the file's channel count)
    int coff0, coff1;        // destination slot of the stream's channel 0 / 1; -1 = not written
    float gain;              // multiplied in (1 = none)
};
__device__ __forceinline__ void mapped_put(const OutDesc &D, long long ts, int coff, float v) {
    if (coff >= 0 && ts >= D.first && ts < D.last) D.base[(ts - D.first) * D.cstride + coff] = v * D.gain;
}
// the descriptor of stream s into wave-uniform registers (base == null: this stream goes to the dense output)
__device__ __forceinline__ OutDesc load_desc(const OutDesc *desc, long s) {
    OutDesc D;
    D.base = nullptr;
    D.first = D.last = D.t0 = 0;
    D.cstride = 0;
    D.coff0 = D.coff1 = -1;
    D.gain = 1.f;
    if (desc) D = desc[s];
    return D;
}

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
    // pf_* and out may be WINDOWS into longer per-stream arrays: consecutive streams are `pstride` frames apart
    // (0 = dense: nframes).  pcm and the states are always dense.
    long pstride;
    const OutDesc *desc;     // [nstreams] or null: per-stream destinations (channels <= 2), see OutDesc
    __host__ __device__ long ps() const { return pstride ? pstride : nframes; }
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

// the value, but not a loop invariant to the optimiser: per-lane offsets derived from an opaque copy of the lane id are
// recomputed where they are used (a few VALU operations) instead of being hoisted out of the frame loop and kept alive
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
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

// x[c][0..8) = buf_c[r .. r+8) for an arbitrary r >= 0 whose alignment r & 3 is the same in every lane
// (and in every channel: the channels of a stream share the pitch period)
template <int NC>
__device__ __forceinline__ void taps8(const float *ring, int r, float (&x)[NC][8]) {
    const int a = __builtin_amdgcn_readfirstlane(r) & 3;
    const int rb = r - a;
    f4 q0[NC], q1[NC], q2[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float *rc = ring + c * kPostRing;
        q0[c] = lds4(rc, rb);
        q1[c] = lds4(rc, rb + 4);
        q2[c] = lds4(rc, rb + 8);
    }
    switch (a) {
        case 0:
#pragma unroll
            for (int c = 0; c < NC; c++) pick8<0>(q0[c], q1[c], q2[c], x[c]);
            break;
        case 1:
#pragma unroll
            for (int c = 0; c < NC; c++) pick8<1>(q0[c], q1[c], q2[c], x[c]);
            break;
        case 2:
#pragma unroll
            for (int c = 0; c < NC; c++) pick8<2>(q0[c], q1[c], q2[c], x[c]);
            break;
        default:
#pragma unroll
            for (int c = 0; c < NC; c++) pick8<3>(q0[c], q1[c], q2[c], x[c]);
            break;
    }
}

}  // namespace nyq
