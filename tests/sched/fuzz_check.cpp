// tests/sched/fuzz_check.cpp -- TEST ONLY.  Random damage of real files through nyq_host::BatchOpusDecoder against
// the fake GPU, built with AddressSanitizer + UBSan: whatever the damage, no out-of-bounds access anywhere in
// scan / layout / entropy stage / scheduler / hand-over / later segments / pass 3.
//   fuzz_check <iterations> <seed> file...
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "batch_decoder.hpp"

// re-seal the pages after the damage (the parser verifies the Ogg CRC) so that the damage reaches the layers below
static void resealPages(std::vector<uint8_t> &b) {
    static uint32_t table[256];
    if (!table[1])
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            table[i] = r;
        }
    size_t pos = 0;
    while (pos + 27 <= b.size()) {
        if (b[pos] != 'O' || b[pos + 1] != 'g' || b[pos + 2] != 'g' || b[pos + 3] != 'S') { pos++; continue; }
        const size_t nsegs = b[pos + 26];
        if (pos + 27 + nsegs > b.size()) break;
        size_t len = 27 + nsegs;
        for (size_t i = 0; i < nsegs; i++) len += b[pos + 27 + i];
        if (pos + len > b.size()) break;
        for (int i = 22; i < 26; i++) b[pos + i] = 0;
        uint32_t c = 0;
        for (size_t i = 0; i < len; i++) c = (c << 8) ^ table[((c >> 24) & 0xff) ^ b[pos + i]];
        for (int i = 0; i < 4; i++) b[pos + 22 + i] = (uint8_t)(c >> (8 * i));
        pos += len;
    }
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int iters = std::atoi(argv[1]);
    std::mt19937 rng((unsigned)std::atoi(argv[2]));
    std::vector<std::vector<uint8_t>> src;
    for (int a = 3; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        src.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    }
    nyq_host::BatchOpusDecoder dec(0);
    long decoded = 0, refused = 0;
    for (int it = 0; it < iters; it++) {
        const int cnt = 1 + (int)(rng() % 6);
        std::vector<std::vector<uint8_t>> files;
        for (int k = 0; k < cnt; k++) {
            std::vector<uint8_t> b = src[rng() % src.size()];
            switch (rng() % 4) {
                case 0:
                    for (int j = 0; j < 1 + (int)(rng() % 8); j++) b[rng() % b.size()] = (uint8_t)rng();
                    break;
                case 1:
                    b.resize(1 + rng() % b.size());
                    break;
                case 2:
                    for (int j = 0; j < 1 + (int)(rng() % 16); j++) {
                        const size_t lo = std::min<size_t>(b.size() - 1, 120);
                        b[lo + rng() % (b.size() - lo)] ^= (uint8_t)(1u << (rng() % 8));
                    }
                    break;
                default:
                    break;
            }
            if (rng() % 4) resealPages(b);
            files.push_back(std::move(b));
        }
        std::vector<const std::vector<uint8_t> *> ptrs;
        for (auto &f : files) ptrs.push_back(&f);
        std::vector<nyq_host::DecodedStream> res;
        dec.decode(ptrs, res, nullptr, 1 + (int)(rng() % 8));
        for (auto &r : res) (r.error.empty() ? decoded : refused)++;
    }
    std::printf("%d iterations: %ld decoded, %ld refused\n", iters, decoded, refused);
    return 0;
}
