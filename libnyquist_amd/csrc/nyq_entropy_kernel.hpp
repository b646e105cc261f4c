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
// (bands.c:284-306) and hands every record the previous frame's final range as its noise seed.  Operands are staged a chunk of
// frames at a time through LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "nyq_entropy_core.hpp"

namespace nyq {

__global__ __launch_bounds__(64) void celt_entropy_kernel(const nyq_ent::EntropyTables *__restrict__ T, const unsigned char *__restrict__ payload,
                                                          long payload_bytes, const nyq_ent::EntDesc *__restrict__ desc, long total, int LM,
                                                          unsigned char *__restrict__ records, long slot, nyq_ent::EntInfo *__restrict__ info,
                                                          nyq_ent::EntEnergy *__restrict__ energy, long pslot) {
    const long u = (long)blockIdx.x * 64 + threadIdx.x;
    if (u >= total) return;
    nyq_ent::EntDesc d;
    if (pslot > 0) {                                                 // frame u's bytes in slot u of `pslot` bytes; desc = words: len | C << 16 | end << 24
        const unsigned w = reinterpret_cast<const unsigned *>(desc)[u];
        d.offset = 0;
        d.len = (uint16_t)(w & 0xffffu);
        d.C = (uint8_t)(w >> 16 & 0xff);
        d.start = 0;
        d.end = (uint8_t)(w >> 24);
        if ((long)d.len > pslot) d.len = 0;
        payload += u * pslot;
    } else {
        d = desc[u];
    }
    nyq_ent::EntInfo fi;
    const int C = d.C == 2 ? 2 : 1, end = d.end > 21 ? 21 : d.end, start = d.start < end ? d.start : 0;
    int len = d.len > 1275 ? 1275 : d.len;
    if (pslot == 0 && (long)d.offset + len > payload_bytes) len = 0;   // (a descriptor that points outside the payload: an empty frame)
    nyq_ent::decode_frame(*T, payload + d.offset, len, LM, C, start, end, records + u * slot, (int)slot, fi, energy[u]);
    info[u] = fi;
}

struct EnergyState {                   // what a stream carries from one call to the next
    float E[42], L1[42], L2[42];
    unsigned range;
    unsigned valid;                    // kEnergyValid once a call has written it (anything else: a stream that starts here)
    unsigned errors;                   // frames in error so far (NYQ_ENT_ERROR), counted over the calls
};
constexpr unsigned kEnergyValid = 0x0E17A7E5u;

constexpr int kEnergyChunk = 32;       // frames staged per round: 32 x (672 + 16 + 8) bytes of LDS

__global__ __launch_bounds__(64) void celt_energy_kernel(const nyq_ent::EntropyTables *__restrict__ T, const nyq_ent::EntInfo *__restrict__ info,
                                                         const nyq_ent::EntEnergy *__restrict__ energy, unsigned char *__restrict__ records, long slot,
                                                         long nstreams, long nframes, EnergyState *__restrict__ state, int fresh) {
#pragma clang fp contract(off)
    // The recurrence is one short step per frame; what a step waits for is its operands.  They are staged a chunk of frames at a
    // time -- coalesced loads, all in flight together -- and the steps then run out of LDS.
    __shared__ float sE[kEnergyChunk * 168];
    __shared__ nyq_ent::EntInfo sI[kEnergyChunk];
    __shared__ unsigned sH[kEnergyChunk][2];                         // nops | flags << 16, the level's offset
    const long s = blockIdx.x;
    if (s >= nstreams) return;
    const int lane = threadIdx.x;
    const bool live = lane < 42;
    const int l = live ? lane : 0, band = l >= 21 ? l - 21 : l, partnerLane = l >= 21 ? l - 21 : l + 21;
    nyq_ent::EnergyLane st{0.f, -28.f, -28.f};
    unsigned range = 0, errors = 0;
    if (!fresh && state[s].valid == kEnergyValid) {
        st.E = state[s].E[l];
        st.L1 = state[s].L1[l];
        st.L2 = state[s].L2[l];
        range = state[s].range;
        errors = state[s].errors;
    }
    __syncthreads();                                                 // (every lane has read the state before lane 0 rewrites it)
    const float eMean = T->eMeans[band];
    const long u0 = s * nframes;
    for (long base = 0; base < nframes; base += kEnergyChunk) {
        const int cnt = (int)(nframes - base < kEnergyChunk ? nframes - base : kEnergyChunk);
        __syncthreads();
        {
            const float *src = reinterpret_cast<const float *>(energy + u0 + base);
            for (int j = lane; j < cnt * 168; j += 64) sE[j] = src[j];
            if (lane < cnt) {
                sI[lane] = info[u0 + base + lane];
                const nyq_ent::RecHead *H = reinterpret_cast<const nyq_ent::RecHead *>(records + (u0 + base + lane) * slot);
                sH[lane][0] = (unsigned)H->nops | (unsigned)H->flags << 16;
                sH[lane][1] = H->reserved[1] >> 16;
            }
        }
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            const nyq_ent::EntInfo fi = sI[k];
            const float *e = sE + k * 168;
            const float prev = e[l], q = e[42 + l], fine = e[84 + l], last = e[126 + l];
            unsigned char *r = records + (u0 + base + k) * slot;
            const bool hasRecord = (sH[k][0] & 0xffffu) != 0;
            const bool collapse = hasRecord && ((sH[k][0] >> 16) & 2);
            float *levelAt = collapse ? reinterpret_cast<float *>(r + sH[k][1]) + l : nullptr;
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
            if (lane == 0) reinterpret_cast<nyq_ent::RecHead *>(r)->seed = range;
            range = fi.rangeFinal;
            errors += (fi.flags & nyq_ent::kEntError) ? 1u : 0u;
        }
    }
    if (live) {
        state[s].E[l] = st.E;
        state[s].L1[l] = st.L1;
        state[s].L2[l] = st.L2;
    }
    if (lane == 0) {
        state[s].range = range;
        state[s].valid = kEnergyValid;
        state[s].errors = errors;
    }
}

// what the synthesis and the post-filter take per frame (transient flag, post-filter period / gain / tapset), from the infos
__global__ void celt_entropy_split_kernel(const nyq_ent::EntInfo *__restrict__ info, long n, unsigned char *__restrict__ transient,
                                          int *__restrict__ pf_pitch, float *__restrict__ pf_gain, int *__restrict__ pf_tapset) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const nyq_ent::EntInfo f = info[u];
    transient[u] = (f.flags & nyq_ent::kEntTransient) ? 1 : 0;
    pf_pitch[u] = f.pfPitch;
    pf_gain[u] = .09375f * (float)f.pfGainIndex;
    pf_tapset[u] = f.pfTapset;
}

}  // namespace nyq
