// oracle/ref_decode_bench.cpp -- times the REAL reference decoder (NyquistIO::Load, built from the
// reference's own sources by oracle/Makefile, no recording tap) on `count` in-memory copies of one
// Opus file spread over `threads` host threads.  TEST / BENCH INFRASTRUCTURE ONLY: the CPU baseline
// of the file-level decode (bench.py's cpu_baseline leg); nothing in the product links it.
//
//   libref_decode.so: ref_decode_bench(bytes, size, count, threads, &samples, &checksum) -> seconds
//   ref_decode_bench <file.opus> <count> <threads>   (same, as a program; prints one JSON line)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "libnyquist/Decoders.h"

// In-process entry (ctypes from bench.py: no child process next to an initialised GPU runtime).
// Returns seconds; *samples_per_file and *checksum_file0 describe the decoded audio.
extern "C" double ref_decode_bench(const unsigned char *bytes, long size, long count, int threads, long *samples_per_file,
                                   double *checksum_file0) {
    const std::vector<uint8_t> file(bytes, bytes + size);
    threads = std::max(1, threads);
    std::vector<size_t> nsamples(threads, 0);
    std::vector<double> sums(threads, 0.0);
    std::atomic<long> next{0};
    auto work = [&](int t) {
        for (long i = next++; i < count; i = next++) {
            nqr::NyquistIO loader;              // src/Common.cpp: Load(AudioData*, ext, buffer) -> OpusDecoder::LoadFromBuffer
            nqr::AudioData data;
            loader.Load(&data, "opus", file);
            nsamples[t] = data.samples.size();
            if (i == 0) for (float v : data.samples) sums[t] += v;
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t ns = 0;
    double sum = 0;
    for (int t = 0; t < threads; t++) { ns = std::max(ns, nsamples[t]); sum += sums[t]; }
    if (samples_per_file) *samples_per_file = (long)ns;
    if (checksum_file0) *checksum_file0 = sum;
    return s;
}

// Whole-file decode through the reference's plugin surface, PCM handed back (tests compare against it).
// info = {channels, sample rate, samples (interleaved count)}; returns the sample count or -1.
extern "C" long ref_decode_pcm(const unsigned char *bytes, long size, float *out, long capacity, long *info) {
    try {
        const std::vector<uint8_t> file(bytes, bytes + size);
        nqr::NyquistIO loader;
        nqr::AudioData data;
        loader.Load(&data, "opus", file);
        if (info) { info[0] = data.channelCount; info[1] = data.sampleRate; info[2] = (long)data.samples.size(); }
        if (out && capacity >= (long)data.samples.size())
            for (size_t i = 0; i < data.samples.size(); i++) out[i] = data.samples[i];
        return (long)data.samples.size();
    } catch (...) {
        return -1;
    }
}

#ifdef REF_DECODE_MAIN
int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: ref_decode_bench file.opus count threads\n"); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    const std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (file.empty()) return 3;
    long ns = 0;
    double sum = 0;
    const long count = std::atol(argv[2]);
    const int threads = std::atoi(argv[3]);
    const double s = ref_decode_bench(file.data(), (long)file.size(), count, threads, &ns, &sum);
    std::printf("{\"files\": %ld, \"threads\": %d, \"seconds\": %.6f, \"samples_per_file\": %ld, \"checksum_file0\": %.6f}\n", count,
                threads, s, ns, sum);
    return 0;
}
#endif
