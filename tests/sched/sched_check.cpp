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

extern "C" long fake_gpu_calls(int device);   // tests/sched/fake_gpu.cpp

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
    // One decoder over TWO devices of the stand-in (and the sink form): elementary streams go to device s mod 2, each
    // with its own feeders and staging arena; every file must equal its stand-alone, one-device result bit for bit, and
    // both devices must have been given work.
    if (std::getenv("SCHED_DEVICES")) {
        BatchOpusDecoder two(listOf("SCHED_DEVICES", {0, 1}));
        const long before0 = fake_gpu_calls(0), before1 = fake_gpu_calls(1);
        std::vector<const std::vector<uint8_t> *> batch;
        std::vector<size_t> which;
        for (int r = 0; r < 3; r++)
            for (size_t i = 0; i < files.size(); i++) {
                batch.push_back(&files[(i * 5 + r) % files.size()]);
                which.push_back((i * 5 + r) % files.size());
            }
        size_t seen = 0;
        two.decode(batch, [&](size_t k, DecodedStream &b) {
            seen++;
            const DecodedStream &a = alone[which[k]];
            if (a.error.empty() != b.error.empty() || a.pcm != b.pcm || a.totalSamples != b.totalSamples) {
                std::printf("MISMATCH file %zu on two devices\n", which[k]);
                bad++;
            }
        }, nullptr, 4);
        runs++;
        const long d0 = fake_gpu_calls(0) - before0, d1 = fake_gpu_calls(1) - before1;
        std::printf("two devices: %zu of %zu files delivered, calls per device %ld / %ld\n", seen, batch.size(), d0, d1);
        if (seen != batch.size() || d0 <= 0 || d1 <= 0) bad++;
    }
    std::printf("%d batches, %d mismatches\n", runs, bad);
    return bad ? 1 : 0;
}
