// nyq_entropy_kernel.hpp -- the entropy stage on the device: nyq_entropy_core.hpp's decode_frame, a FRAME PER LANE, and the energy
// pass, a wave per stream.
//
// celt_entropy_kernel: lane u decodes frame u (stream-major: u = stream * nframes + frame) from its bytes into its record slot
// (the "spread" form: lists at fixed places named in the head), its EntInfo and its EntEnergy.  The lanes of a wave walk
// different bitstreams, so they diverge wherever the streams do; neighbouring lanes are consecutive frames of one stream, whose
// loop counts (bands, leaves) are close.  Every table is read through the scalar / L2 path from one block (EntropyTables).
//
// celt_energy_kernel: lane = (channel, band) of one stream, 42 of a wave's 64; frame after frame it folds the deltas into the
// band energies (quant_bands.c:427-540 as a recurrence), writes the record's log gains, finishes the anti-collapse levels
// (bands.c:284-306) and hands every record the previous frame's final range as its noise seed.  The next frame's operands are
// loaded while the current one is folded.
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_entropy_core.hpp"

namespace nyq {

__global__ __launch_bounds__(64) void celt_entropy_kernel(const nyq_ent::EntropyTables *__restrict__ T, const unsigned char *__restrict__ payload,
                                                          const nyq_ent::EntDesc *__restrict__ desc, long total, int LM,
                                                          unsigned char *__restrict__ records, long slot, nyq_ent::EntInfo *__restrict__ info,
                                                          nyq_ent::EntEnergy *__restrict__ energy) {
    const long u = (long)blockIdx.x * 64 + threadIdx.x;
    if (u >= total) return;
    const nyq_ent::EntDesc d = desc[u];
    nyq_ent::EntInfo fi;
    const int C = d.C == 2 ? 2 : 1, end = d.end > 21 ? 21 : d.end, start = d.start < end ? d.start : 0;
    const int len = d.len > 1275 ? 1275 : d.len;
    nyq_ent::decode_frame(*T, payload + d.offset, len, LM, C, start, end, records + u * slot, (int)slot, fi, energy[u]);
    info[u] = fi;
}

struct EnergyState {                   // what a stream carries from one call to the next
    float E[42], L1[42], L2[42];
    unsigned range, pad;
};

__global__ __launch_bounds__(64) void celt_energy_kernel(const nyq_ent::EntropyTables *__restrict__ T, const nyq_ent::EntInfo *__restrict__ info,
                                                         const nyq_ent::EntEnergy *__restrict__ energy, unsigned char *__restrict__ records, long slot,
                                                         long nstreams, long nframes, EnergyState *__restrict__ state, int fresh) {
#pragma clang fp contract(off)
    const long s = blockIdx.x;
    if (s >= nstreams) return;
    const int lane = threadIdx.x;
    const bool live = lane < 42;
    const int l = live ? lane : 0, band = l >= 21 ? l - 21 : l, partnerLane = l >= 21 ? l - 21 : l + 21;
    nyq_ent::EnergyLane st{0.f, -28.f, -28.f};
    unsigned range = 0;
    if (!fresh) {
        st.E = state[s].E[l];
        st.L1 = state[s].L1[l];
        st.L2 = state[s].L2[l];
        range = state[s].range;
    }
    const float eMean = T->eMeans[band];
    const long u0 = s * nframes;
    // (operands of frame 0)
    nyq_ent::EntInfo fi = nframes > 0 ? info[u0] : nyq_ent::EntInfo{};
    float prev = 0.f, q = 0.f, fine = 0.f, last = 0.f;
    if (nframes > 0) {
        prev = energy[u0].prev[l]; q = energy[u0].q[l]; fine = energy[u0].fine[l]; last = energy[u0].last[l];
    }
    for (long f = 0; f < nframes; f++) {
        const long u = u0 + f;
        // the next frame's operands: nothing of them depends on this frame's result
        nyq_ent::EntInfo nfi = fi;
        float nprev = 0.f, nq = 0.f, nfine = 0.f, nlast = 0.f;
        if (f + 1 < nframes) {
            nfi = info[u + 1];
            nprev = energy[u + 1].prev[l]; nq = energy[u + 1].q[l]; nfine = energy[u + 1].fine[l]; nlast = energy[u + 1].last[l];
        }
        unsigned char *r = records + u * slot;
        nyq_ent::RecHead *H = reinterpret_cast<nyq_ent::RecHead *>(r);
        const bool hasRecord = H->nops != 0;
        const bool collapse = hasRecord && (H->flags & 2);
        float *levelAt = collapse ? reinterpret_cast<float *>(r + (H->reserved[1] >> 16)) + l : nullptr;
        float lv = collapse && live ? *levelAt : 0.f;
        const int bins = T->alloc[fi.LM & 3][fi.C == 2 ? 1 : 0].bins[band];
        nyq_ent::EnergyLane partner;
        partner.E = __shfl(st.E, partnerLane);
        partner.L1 = __shfl(st.L1, partnerLane);
        partner.L2 = __shfl(st.L2, partnerLane);
        float gain = 0.f;
        const float E = nyq_ent::energy_begin(st, partner, l, fi, prev, q, fine, last, eMean, lv, bins, &gain, collapse ? &lv : nullptr);
        const float partnerE = __shfl(E, partnerLane);
        nyq_ent::energy_finish(st, E, partnerE, l, fi);
        if (hasRecord && live) {
            const int c = l >= 21;
            if (c < fi.C && band >= fi.start && band < fi.end) {
                reinterpret_cast<float *>(r + nyq_ent::kRecGainOff)[l] = gain;
                if (collapse) *levelAt = lv;
            }
        }
        if (lane == 0) H->seed = range;
        range = fi.rangeFinal;
        fi = nfi;
        prev = nprev; q = nq; fine = nfine; last = nlast;
    }
    if (live) {
        state[s].E[l] = st.E;
        state[s].L1[l] = st.L1;
        state[s].L2[l] = st.L2;
    }
    if (lane == 0) state[s].range = range;
}

}  // namespace nyq
