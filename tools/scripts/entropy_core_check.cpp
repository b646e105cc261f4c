// tools/scripts/entropy_core_check.cpp -- the frame-per-lane entropy stage (csrc/nyq_entropy_core.hpp) compiled for the HOST, against
// the host decoder's symbol records (CeltDecoder::decodeSymbols) on every frame of the given Ogg Opus files: list by list, field
// by field, log gains and anti-collapse levels after the energy pass, final range, post-filter parameters.  Exit 0 = all equal.
//   g++ -O2 -std=c++17 -ffp-contract=off -Ilibnyquist_amd/host -Iinclude -o /tmp/ecc tools/scripts/entropy_core_check.cpp \
//       libnyquist_amd/host/{celt_mode,celt_decoder,opus_stream}.cpp && /tmp/ecc tests/golden/sb-reverie.opus ...
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../include/nyq_imdct.h"
#include "../../libnyquist_amd/csrc/nyq_entropy_core.hpp"
#include "celt_decoder.hpp"
#include "opus_stream.hpp"
using namespace nyq_host;
using namespace nyq_ent;

int main(int argc, char **argv) {
    std::vector<uint8_t> tabBytes(entropyTablesBytes());
    fillEntropyTables(tabBytes.data());
    const EntropyTables &T = *reinterpret_cast<const EntropyTables *>(tabBytes.data());
    long bad = 0, total = 0, hostBuilt = 0, tooLarge = 0;
    double coreSeconds = 0;
    for (int a = 1; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), {});
        OggOpusFile of;
        try {
            of = parseOggOpus(raw.data(), raw.size());
        } catch (const std::exception &e) {
            printf("%s: %s\n", argv[a], e.what());
            return 2;
        }
        if (of.head.mappingFamily != 0 || of.head.channels > 2) {
            printf("%s: skipped (multistream)\n", argv[a]);
            continue;
        }
        const int CC = of.head.channels;
        CeltDecoder dec(CC);
        EnergyLane lanes[42];
        for (auto &l : lanes) l = EnergyLane{0.f, -28.f, -28.f};
        uint32_t prevRange = 0;
        long fileBad = 0, fileFrames = 0;
        std::vector<uint8_t> hostRec(CeltDecoder::symbolBytes(2, 3)), coreRec((size_t)recFullSlot(2, 3));   // (the core: a slot that holds any frame)
        for (const auto &pkt : of.packets) {
            PacketFrames pf;
            if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return 3;
            if (pf.config < 16) {
                printf("%s: skipped (not CELT-only)\n", argv[a]);
                goto next;
            }
            {
                int LM = 0;
                while ((120 << LM) != pf.frameSize) LM++;
                const int C = pf.stereo ? 2 : 1;
                dec.setEndBand(pf.bandwidthEnd);
                dec.setStreamChannels(C);
                const size_t slot = (size_t)recFullSlot(CC, LM);
                for (const auto &fr : pf.frames) {
                    CeltFrame hi;
                    std::memset(hostRec.data(), 0, hostRec.size());
                    const int rc = dec.decodeSymbols(fr.first, fr.second, pf.frameSize, hostRec.data(), hi);
                    EntInfo ci;
                    EntEnergy ed;
                    const auto t0 = std::chrono::steady_clock::now();
                    decode_frame(T, fr.first, fr.second, LM, C, 0, pf.bandwidthEnd, coreRec.data(), (int)slot, ci, ed);
                    coreSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    // the energy pass, all 42 lanes
                    RecHead *H = reinterpret_cast<RecHead *>(coreRec.data());
                    float *gain = reinterpret_cast<float *>(coreRec.data() + kRecGainOff);
                    float *level = (H->flags & 2) ? reinterpret_cast<float *>(coreRec.data() + (H->reserved[1] >> 16)) : nullptr;
                    float E[42];
                    EnergyLane before[42];
                    std::memcpy(before, lanes, sizeof before);
                    for (int l = 0; l < 42; l++) {
                        const int c = l >= 21, i = l - 21 * c, p = c ? l - 21 : l + 21;
                        float lv = level ? level[l] : 0.f;
                        E[l] = energy_begin(lanes[l], before[p], l, ci, ed.prev[l], ed.q[l], ed.fine[l], ed.last[l], T.eMeans[i], lv,
                                            T.alloc[LM][C - 1].bins[i], &gain[l], level ? &lv : nullptr);
                        if (level) level[l] = lv;
                    }
                    for (int l = 0; l < 42; l++) {
                        const int c = l >= 21, p = c ? l - 21 : l + 21;
                        energy_finish(lanes[l], E[l], E[p], l, ci);
                    }
                    H->seed = prevRange;
                    prevRange = ci.rangeFinal;
                    total++;
                    fileFrames++;
                    // compare
                    const nyq_sym_head *hh = reinterpret_cast<const nyq_sym_head *>(hostRec.data());
                    bool diff = false;
                    std::string why;
                    if (ci.rangeFinal != hi.rangeFinal) { diff = true; why += " rangeFinal"; }
                    if (((ci.flags & kEntTransient) != 0) != hi.transient) { diff = true; why += " transient"; }
                    if (((ci.flags & kEntSilence) != 0) != hi.silence) { diff = true; why += " silence"; }
                    if (ci.pfPitch != hi.pfPitch || ci.pfTapset != hi.pfTapset || .09375f * ci.pfGainIndex != hi.pfGain) { diff = true; why += " postfilter"; }
                    if (((ci.flags & kEntError) != 0) != (rc < 0)) { diff = true; why += " error"; }
                    if (hh->flags & NYQ_SYM_HOST_FREQ) {
                        hostBuilt++;
                    } else if (ci.flags & kEntTooLarge) {
                        tooLarge++;
                    } else if (!hi.silence && rc >= 0) {
                        if (hh->seed != H->seed || hh->nleaves != H->nleaves || hh->nvecs != H->nvecs || hh->nops != H->nops || hh->flags != H->flags ||
                            hh->spread != H->spread || hh->start != H->start || hh->end != H->end || hh->channels != H->channels || hh->lm != H->lm) {
                            diff = true;
                            why += " head";
                        } else {
                            const uint8_t *ho = hostRec.data() + 200;
                            const uint8_t *hv = ho + 16 * hh->nops, *hl = hv + 24 * hh->nvecs, *hlev = hl + 40 * hh->nleaves;
                            if (std::memcmp(ho, coreRec.data() + kRecOpsOff, 16 * hh->nops)) { diff = true; why += " ops"; }
                            if (std::memcmp(hv, coreRec.data() + recVecsOff(C), 24 * hh->nvecs)) { diff = true; why += " vecs"; }
                            if (std::memcmp(hl, coreRec.data() + recLeavesOff(C), 40 * hh->nleaves)) { diff = true; why += " leaves"; }
                            const float *hg = reinterpret_cast<const float *>(hostRec.data() + 32);
                            for (int c = 0; c < C; c++)
                                for (int i = 0; i < hh->end; i++)
                                    if (std::memcmp(&hg[c * 21 + i], &gain[c * 21 + i], 4)) { diff = true; why += " gain"; c = 2; break; }
                            if (hh->flags & NYQ_SYM_ANTI_COLLAPSE) {
                                const float *hlv = reinterpret_cast<const float *>(hlev);
                                for (int c = 0; c < C; c++)
                                    for (int i = 0; i < hh->end; i++)
                                        if (std::memcmp(&hlv[c * 21 + i], &level[c * 21 + i], 4)) { diff = true; why += " level"; c = 2; break; }
                            }
                        }
                    }
                    if (diff) {
                        if (fileBad < 5) printf("%s frame %ld:%s\n", argv[a], fileFrames - 1, why.c_str());
                        fileBad++;
                    }
                }
            }
        }
    next:
        bad += fileBad;
        printf("%s: %ld frames, %ld differ\n", argv[a], fileFrames, fileBad);
    }
    printf("total %ld frames, %ld differ, %ld built by the host decoder, %ld too large for the spread record; core %.0f frames/s on one thread\n", total, bad,
           hostBuilt, tooLarge, total / (coreSeconds > 0 ? coreSeconds : 1));
    return bad ? 1 : 0;
}
