// tools/scripts/entropy_stage_bench.cpp -- frames/s/thread of the CPU entropy stage (CeltDecoder::decode) on one Ogg Opus file, best of N passes,
// and the chain of final range-coder states (one mis-decoded symbol anywhere changes it).  A third argument "sym" times
// CeltDecoder::decodeSymbols instead (20 ms frames: the entropy stage stops at the symbol record, the GPU builds the band shapes).
//   g++ -O3 -std=c++17 -ffp-contract=off -Ilibnyquist_amd/host -o /tmp/esb tools/scripts/entropy_stage_bench.cpp libnyquist_amd/host/{celt_mode,celt_decoder,opus_stream}.cpp && /tmp/esb tests/golden/sb-reverie.opus 5
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include "celt_decoder.hpp"
#include "opus_stream.hpp"
using namespace nyq_host;
int main(int argc, char **argv) {
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), {});
    int reps = argc > 2 ? atoi(argv[2]) : 3;
    OggOpusFile of = parseOggOpus(raw.data(), raw.size());
    const int CC = of.head.channels;
    const bool sym = argc > 3 && std::string(argv[3]) == "sym";
    std::vector<float> freq(2 * 960);
    // (records written round-robin over 256 slots: a staging buffer larger than L1, as in the batch decoder)
    std::vector<uint8_t> recs(256 * CeltDecoder::symbolBytes(CC));
    double best = 1e9; long nframes = 0; unsigned long long chk = 0;
    for (int r = 0; r < reps; r++) {
        CeltDecoder dec(CC);
        nframes = 0; chk = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (const auto &pkt : of.packets) {
            PacketFrames pf;
            if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return 1;
            dec.setEndBand(pf.bandwidthEnd);
            dec.setStreamChannels(pf.stereo ? 2 : 1);
            for (const auto &fr : pf.frames) {
                CeltFrame info;
                const int rc = sym && pf.frameSize == 960
                                   ? dec.decodeSymbols(fr.first, fr.second, pf.frameSize, recs.data() + (nframes & 255) * CeltDecoder::symbolBytes(CC), info)
                                   : dec.decode(fr.first, fr.second, pf.frameSize, freq.data(), info);
                if (rc < 0) return 2;
                chk = chk * 1315423911ull + info.rangeFinal;
                nframes++;
            }
        }
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (s < best) best = s;
    }
    if (sym) printf("symbol records: ");
    printf("%ld frames, best %.4f s = %.0f frames/s/thread, rangeFinal chain %016llx\n", nframes, best, nframes / best, chk);
    return 0;
}
