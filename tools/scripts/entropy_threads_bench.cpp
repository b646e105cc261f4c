// tools/scripts/entropy_threads_bench.cpp -- how the entropy stage scales over host threads, and what the destination memory costs:
// T threads each decode the whole file to symbol records, written round-robin into a per-thread window of (a) ordinary memory,
// (b) page-locked memory from nyq_host_alloc (what the batch decoder stages in).  Prints frames/s per thread and in total.
//   g++ -O3 -std=c++17 -ffp-contract=off -pthread -Ilibnyquist_amd/host -o /tmp/etb tools/scripts/entropy_threads_bench.cpp \
//       libnyquist_amd/host/{celt_mode,celt_decoder,opus_stream}.cpp -Llibnyquist_amd -lnyq_imdct -Wl,-rpath,$PWD/libnyquist_amd
//   /tmp/etb tests/golden/sb-reverie.opus 16 [window_frames]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "../../include/nyq_imdct.h"
#include "celt_decoder.hpp"
#include "opus_stream.hpp"
using namespace nyq_host;

static double run(const OggOpusFile &of, int T, uint8_t *base, size_t window, size_t rec) {
    std::vector<std::thread> th;
    std::vector<double> secs(T);
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            CeltDecoder dec(of.head.channels);
            uint8_t *mine = base + (size_t)t * window * rec;
            long n = 0;
            auto t0 = std::chrono::steady_clock::now();
            for (const auto &pkt : of.packets) {
                PacketFrames pf;
                if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return;
                dec.setEndBand(pf.bandwidthEnd);
                dec.setStreamChannels(pf.stereo ? 2 : 1);
                for (const auto &fr : pf.frames) {
                    CeltFrame info;
                    if (pf.frameSize == 960) dec.decodeSymbols(fr.first, fr.second, 960, mine + (size_t)(n % (long)window) * rec, info);
                    n++;
                }
            }
            secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    for (auto &x : th) x.join();
    double worst = 0;
    for (double s : secs) worst = std::max(worst, s);
    return worst;
}

int main(int argc, char **argv) {
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), {});
    const int T = argc > 2 ? std::atoi(argv[2]) : 16;
    const size_t window = argc > 3 ? (size_t)std::atol(argv[3]) : 11200;
    OggOpusFile of = parseOggOpus(raw.data(), raw.size());
    const size_t rec = CeltDecoder::symbolBytes(of.head.channels);
    long frames = 0;
    for (const auto &pkt : of.packets) {
        PacketFrames pf;
        parseOpusPacket(pkt.data(), (int)pkt.size(), pf);
        frames += (long)pf.frames.size();
    }
    std::vector<uint8_t> plain((size_t)T * window * rec);
    uint8_t *pinned = (uint8_t *)nyq_host_alloc((size_t)T * window * rec);
    for (int threads : {1, T / 2 > 0 ? T / 2 : 1, T}) {
        double a = 1e9, b = 1e9;
        for (int r = 0; r < 3; r++) a = std::min(a, run(of, threads, plain.data(), window, rec));
        if (pinned)
            for (int r = 0; r < 3; r++) b = std::min(b, run(of, threads, pinned, window, rec));
        std::printf("%2d threads: ordinary memory %.0f frames/s/thread (%.2e total), page-locked %.0f (%.2e total)\n", threads, frames / a,
                    threads * frames / a, pinned ? frames / b : 0.0, pinned ? threads * frames / b : 0.0);
    }
    return 0;
}
