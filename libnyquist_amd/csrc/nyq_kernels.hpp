// nyq_kernels.hpp -- the gfx950 __global__ kernels of the batched CELT IMDCT.
// Included by nyq_imdct.hip (the product library) and by tools/kbench.hip (the tuning
// harness that times kernel configurations against each other in one process).
//
// Kernels (all wave-autonomous: 64 lanes own 4 rows, meet only in their LDS slice):
//   imdct_rows_kernel<N2R,Cfg>  clt_mdct_backward  (mdct.c:267-379)  on independent rows
//   ifft_rows_kernel<N2R,WPB>   opus_ifft          (kiss_fft.c:696-747)
//   chain_fixup_kernel          adds the carry terms of the TDAC mirror (mdct.c:362-377)
//                               for rows chained to their predecessor's tail
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_imdct_lanes.hpp"

namespace nyq {

// Lanes of one wave exchange data through LDS without a workgroup barrier.  The LDS
// unit executes one wave's accesses in issue order, so only the COMPILER must be told
// not to move LDS accesses across a phase boundary.
#define NYQ_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// kernel configuration (compile time)
//   WPB       waves per workgroup (each wave is autonomous; WPB only sets the LDS granule)
//   PREFETCH  issue the next group's global loads right after pre-rotation of the current one
//   NT        bit 0: non-temporal loads, bit 1: non-temporal stores
template <int WPB_, bool PREFETCH_, int NT_, bool CHUNKED_ = false>
struct KCfg {
    static constexpr bool CHUNKED = CHUNKED_;   // contiguous run of groups per wave instead of grid-stride
    static constexpr int WPB = WPB_;
    static constexpr bool PREFETCH = PREFETCH_;
    static constexpr int NT_LD = NT_ & 1;
    static constexpr int NT_ST = (NT_ >> 1) & 1;
};

#ifndef NYQ_DEFAULT_WPB
#define NYQ_DEFAULT_WPB 1
#endif
// Waves kept resident per CU.  Measured on MI355X with tools/kbench (2^20 nfft-480 rows,
// interleaved A/B): 3 -> 2.02 ms, 4 -> 1.67, 5 -> 1.52, 6 -> 1.49, 7 -> 1.57, 8 -> 1.59,
// 10 (the LDS limit) -> 2.05.  Six autonomous waves already keep ~90 KB of loads in flight per
// CU; beyond that the memory system loses efficiency (plain copies on the same box show the
// same shape), so the persistent grid is capped here rather than at the occupancy limit.
#ifndef NYQ_WAVES_PER_CU
#define NYQ_WAVES_PER_CU 6
#endif
using DefaultCfg = KCfg<NYQ_DEFAULT_WPB, false, 0>;

template <int N2R>
__device__ __forceinline__ void fft_passes(int lane, cpx *lds) {
    using Gm = Geo<N2R>;
    pass1<N2R>(lane, lds);
    NYQ_WAVE_SYNC();
#pragma unroll
    for (int it = 0; it < Gm::P2_ITERS; it++) {
        cpx v[15];
        int g, n2;
        bool ok = pass2_load<N2R>(lane, it, lds, v, g, n2);
        NYQ_WAVE_SYNC();
        if (ok) pass2_store<N2R>(g, n2, lds, v);
    }
    NYQ_WAVE_SYNC();
}

template <int N2R, typename Cfg>
__global__ __launch_bounds__(kWave *Cfg::WPB) void imdct_rows_kernel(
    const float *__restrict__ in, const float *__restrict__ carry, float *__restrict__ fin,
    float *__restrict__ tail, long nrows, const float *__restrict__ trig,
    const float *__restrict__ window) {
    using Gm = Geo<N2R>;
    __shared__ cpx lds_all[Cfg::WPB * Gm::LDS_CPX];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = lds_all + wv * Gm::LDS_CPX;

    LaneConst<N2R> K;
    lane_init<N2R>(K, lane, trig, window);

    const long ngroups_all = (nrows + kGroup - 1) / kGroup;
    const long nwaves_all = (long)gridDim.x * Cfg::WPB;
    const long wid = (long)blockIdx.x * Cfg::WPB + wv;
    // grid-stride: wave w takes groups w, w+W, ...; chunked: wave w takes one contiguous run
    const long per = (ngroups_all + nwaves_all - 1) / nwaves_all;
    const long nwaves = Cfg::CHUNKED ? 1 : nwaves_all;
    long gi = Cfg::CHUNKED ? wid * per : wid;
    const long ngroups = Cfg::CHUNKED ? (gi + per < ngroups_all ? gi + per : ngroups_all) : ngroups_all;
    StageRegs<N2R> R;
    if constexpr (Cfg::PREFETCH) {
        // Software pipeline over this wave's groups: the float4 loads of group g+1 are issued
        // as soon as group g's registers have been pre-rotated into LDS, and stay in flight
        // under both FFT passes and the stores of group g.
        if (gi < ngroups) stage_in_load<N2R, Cfg::NT_LD>(R, lane, in, gi * kGroup, nrows);
    }
    for (; gi < ngroups; gi += nwaves) {
        const long row0 = gi * kGroup;
        if constexpr (!Cfg::PREFETCH) stage_in_load<N2R, Cfg::NT_LD>(R, lane, in, row0, nrows);
        NYQ_WAVE_SYNC();
        stage_in_store<N2R>(R, K, lane, lds);
        if constexpr (Cfg::PREFETCH) {
            if (gi + nwaves < ngroups) stage_in_load<N2R, Cfg::NT_LD>(R, lane, in, (gi + nwaves) * kGroup, nrows);
        }
        NYQ_WAVE_SYNC();
        fft_passes<N2R>(lane, lds);
        stage_out<N2R, Cfg::NT_ST>(K, lane, lds, carry, fin, tail, row0, nrows);
    }
}

template <int N2R, int WPB>
__global__ __launch_bounds__(kWave *WPB) void ifft_rows_kernel(const float *__restrict__ in,
                                                                float *__restrict__ out, long nrows) {
    using Gm = Geo<N2R>;
    __shared__ cpx lds_all[WPB * Gm::LDS_CPX];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    cpx *lds = lds_all + wv * Gm::LDS_CPX;
    const long ngroups = (nrows + kGroup - 1) / kGroup;
    const long nwaves = (long)gridDim.x * WPB;
    for (long gi = (long)blockIdx.x * WPB + wv; gi < ngroups; gi += nwaves) {
        const long row0 = gi * kGroup;
        NYQ_WAVE_SYNC();
        ifft_stage_in<N2R>(lane, in, lds, row0, nrows);
        NYQ_WAVE_SYNC();
        fft_passes<N2R>(lane, lds);
        ifft_stage_out<N2R>(lane, lds, out, row0, nrows);
    }
}

// One 64-thread block per row: lanes 0..59 add the carry terms of mdct.c:371-372 to a
// head that was produced with zero carry:  out[i] += w[119-i]*c[i];  out[119-i] += w[i]*c[i].
// Row r of chain c takes c[] from tails[row-1] (r > 0) or carry0[c] (r == 0, may be NULL).
__global__ __launch_bounds__(64) void chain_fixup_kernel(float *__restrict__ pcm,
                                                          const float *__restrict__ tails,
                                                          const float *__restrict__ carry0,
                                                          float *__restrict__ tail_out, int n2,
                                                          long len, long nrows,
                                                          const float *__restrict__ window) {
    const int i = threadIdx.x;
    if (i >= kHalfOv) return;
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const long c = row / len, r = row - c * len;
        float cv = 0.f;
        if (r > 0) cv = tails[(row - 1) * kHalfOv + i];
        else if (carry0) cv = carry0[c * kHalfOv + i];
        float *o = pcm + row * (long)n2;
        o[i] += window[kOverlap - 1 - i] * cv;
        o[kOverlap - 1 - i] += window[i] * cv;
        if (tail_out && r == len - 1) tail_out[c * kHalfOv + i] = tails[row * kHalfOv + i];
    }
}

}  // namespace nyq
