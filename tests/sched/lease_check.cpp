// tests/sched/lease_check.cpp -- TEST ONLY.  The situation of the one process abort this project has seen on a GPU box
// (DESIGN.md section 5a): several host threads inside nqr::NyquistIO::Load at the same time, each leasing a decoder from
// the pool of libnyquist_amd/host/nqr_surface.cpp.  Here the real plugin surface, the real lease pool, the real scheduler
// and the real entropy decoder run against tests/sched/fake_gpu.cpp, under ThreadSanitizer and AddressSanitizer
// (tests/sched/Makefile), in three modes:
//   plain          six threads x REPS rounds over the files: every Load equals the sequential result bit for bit, and
//                  the pool neither makes more decoders than there are threads nor tears one down
//   gpu-faults     every K-th GPU call / context creation of the stand-in fails: a Load then throws std::runtime_error
//                  (never anything else, never std::terminate), every Load that returns is still bit-exact
//   churn          more loading threads than the pool keeps decoders (16): teardown beside running decoders, still exact
//   thread-faults  thread starts fail on and off (-DNYQ_HOST_TEST_HOOKS): the batch runs on the threads it gets or
//                  fails with an exception; nothing is left joinable, nothing leaks a running thread
// usage: lease_check MODE THREADS REPS file.opus...
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "libnyquist/Decoders.h"

extern "C" void fake_gpu_counts(long *created, long *destroyed, long *live);
extern "C" long fake_gpu_destroyed_beside_calls(void);
extern "C" void fake_gpu_fail_every(long calls, long creates);
namespace nyq_host { extern std::atomic<long> g_failThreadStartIn; }

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string mode = argv[1];
    const int nthreads = std::atoi(argv[2]), reps = std::atoi(argv[3]);
    std::vector<std::vector<uint8_t>> files;
    for (int a = 4; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        files.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    }
    nqr::NyquistIO io;
    std::vector<nqr::AudioData> want(files.size());
    for (size_t i = 0; i < files.size(); i++) io.Load(&want[i], files[i]);

    long c0 = 0, d0 = 0;
    nqr::DecoderPoolCounts(&c0, &d0);
    if (mode == "gpu-faults") fake_gpu_fail_every(29, 17);
    std::atomic<long> okLoads{0}, thrown{0}, wrong{0}, foreign{0};
    std::atomic<bool> stopInjector{false};
    std::thread injector;
    if (mode == "thread-faults")
        injector = std::thread([&] {                       // arm a failing thread start again and again
            unsigned x = 12345;
            while (!stopInjector.load()) {
                x = x * 1664525u + 1013904223u;
                nyq_host::g_failThreadStartIn.store(1 + (x >> 16) % 9);
                std::this_thread::sleep_for(std::chrono::microseconds(300));
            }
            nyq_host::g_failThreadStartIn.store(0);
        });
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            nqr::NyquistIO mine;
            for (int r = 0; r < reps; r++)
                for (size_t k = (size_t)t % files.size(), n = 0; n < files.size(); n++, k = (k + 1) % files.size()) {
                    nqr::AudioData d;
                    try {
                        mine.Load(&d, files[k]);
                    } catch (const std::runtime_error &) {
                        thrown++;
                        continue;
                    } catch (...) {
                        foreign++;
                        continue;
                    }
                    okLoads++;
                    if (d.samples != want[k].samples || d.channelCount != want[k].channelCount) wrong++;
                }
        });
    for (auto &t : th) t.join();
    if (injector.joinable()) {
        stopInjector = true;
        injector.join();
    }
    fake_gpu_fail_every(0, 0);
    // and a batch over two devices of the stand-in with the same faults armed once more (BatchLoad's lease path)
    long c1 = 0, d1 = 0, fc = 0, fd = 0, fl = 0;
    nqr::DecoderPoolCounts(&c1, &d1);
    fake_gpu_counts(&fc, &fd, &fl);
    std::printf("mode %s: %ld loads ok, %ld threw runtime_error, %ld threw something else, %ld wrong; decoders made %ld, torn down %ld; "
                "contexts made %ld, destroyed %ld, live %ld\n",
                mode.c_str(), okLoads.load(), thrown.load(), foreign.load(), wrong.load(), c1 - c0, d1 - d0, fc, fd, fl);
    int bad = 0;
    if (wrong.load() || foreign.load()) bad = 1;
    if (mode == "plain") {
        if (thrown.load() || okLoads.load() != (long)nthreads * reps * (long)files.size()) bad = 1;
        if (c1 - c0 > nthreads || d1 - d0 != 0) bad = 1;   // pooled: at most one decoder per thread, none destroyed
    } else if (mode == "churn") {                          // more threads than the pool keeps: surplus decoders are retired and
        if (thrown.load() || okLoads.load() != (long)nthreads * reps * (long)files.size()) bad = 1;   // destroyed only while no lease is active
        if (c1 - d1 > 16 || d1 - d0 == 0) bad = 1;
        if (fake_gpu_destroyed_beside_calls() != 0) {
            std::printf("%ld contexts were destroyed while a GPU call was in flight\n", fake_gpu_destroyed_beside_calls());
            bad = 1;
        }
    } else {
        if (okLoads.load() == 0) bad = 1;                  // the faults are sparse: most loads get through
    }
    if (fl != (c1 - d1) * 6) bad = 1;                      // every live context belongs to a pooled decoder (6 per decoder): no leak
    std::printf("%s\n", bad ? "FAILED" : "OK");
    return bad;
}
