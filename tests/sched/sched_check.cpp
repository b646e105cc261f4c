// tests/sched/sched_check.cpp -- TEST ONLY.  Drives nyq_host::BatchOpusDecoder (the real scheduler, the real
// entropy decoder) against tests/sched/fake_gpu.cpp: decodes a list of files alone and then as batches of mixed
// shapes with several thread counts and a small staging budget, and requires every file of every batch to equal
// its stand-alone result bit for bit.  Built with -fsanitize=thread by tests/sched/Makefile.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "batch_decoder.hpp"

using nyq_host::BatchOpusDecoder;
using nyq_host::DecodedStream;

int main(int argc, char **argv) {
    std::vector<std::vector<uint8_t>> files;
    for (int a = 1; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        files.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    }
    if (files.empty()) return 2;
    BatchOpusDecoder dec(0);
    std::vector<DecodedStream> alone(files.size());
    for (size_t i = 0; i < files.size(); i++) {
        std::vector<DecodedStream> r;
        dec.decode({&files[i]}, r, nullptr, 1);
        alone[i] = r[0];
    }
    auto listOf = [](const char *env, std::vector<int> dflt) {
        const char *e = std::getenv(env);
        if (!e) return dflt;
        std::vector<int> v;
        for (const char *p = e; *p;) {
            v.push_back(std::atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        return v;
    };
    int bad = 0, runs = 0;
    for (int threads : listOf("SCHED_THREADS", {8, 3, 1})) {
        for (int rep : listOf("SCHED_REPS", {1, 3})) {
            std::vector<const std::vector<uint8_t> *> batch;
            std::vector<size_t> which;
            for (int r = 0; r < rep; r++)
                for (size_t i = 0; i < files.size(); i++) {
                    batch.push_back(&files[(i * 7 + r) % files.size()]);
                    which.push_back((i * 7 + r) % files.size());
                }
            std::vector<DecodedStream> res;
            dec.decode(batch, res, nullptr, threads);
            runs++;
            for (size_t k = 0; k < batch.size(); k++) {
                const DecodedStream &a = alone[which[k]], &b = res[k];
                if (a.error.empty() != b.error.empty() || a.pcm != b.pcm || a.totalSamples != b.totalSamples) {
                    std::printf("MISMATCH file %zu in a batch of %zu on %d threads\n", which[k], batch.size(), threads);
                    bad++;
                }
            }
        }
    }
    std::printf("%d batches, %d mismatches\n", runs, bad);
    return bad ? 1 : 0;
}
