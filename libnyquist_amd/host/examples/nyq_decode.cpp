// nyq_decode.cpp -- the reference's example program (examples/src/Main.cpp:24-174) for the MI355X build:
// load an audio file through NyquistIO, print sample count and float checksum, and compare against the
// checksum pairs the reference hard-codes (Main.cpp:146-148).
//
//   nyq_decode <file.opus> [out.f32]
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

#include "libnyquist/Decoders.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: nyq_decode <file.opus> [out.f32]\n");
        return 2;
    }
    try {
        nqr::NyquistIO loader;
        nqr::AudioData data;
        loader.Load(&data, std::string(argv[1]));
        float sum = 0;                                   // float accumulation, channel after channel, as Main.cpp:136-142
        const size_t per = data.channelCount ? data.samples.size() / (size_t)data.channelCount : 0;
        for (int c = 0; c < data.channelCount; c++)
            for (size_t j = 0; j < per; j++) sum += data.samples[j * (size_t)data.channelCount + (size_t)c];
        std::printf("channels: %d rate: %d seconds: %.0f\n", data.channelCount, data.sampleRate, data.lengthSeconds);
        std::printf("len: %zu sum: %f\n", data.samples.size(), sum);
        struct { int sum; size_t size; const char *name; } known[] = {
            {403, 21472602, "sb-reverie.opus"}, {40, 127712488, "Rachel8ch.opus"}, {719, 21472602, "sb-reverie-60ms-frames.opus"}};
        bool ok = false;
        for (auto &k : known)
            if ((int)sum == k.sum && data.samples.size() == k.size) {
                std::printf("matches the reference checksum of %s\n", k.name);
                ok = true;
            }
        if (!ok) std::printf("not one of the reference's three test files (Main.cpp:144-148 would say \"wrong results!\")\n");
        if (argc > 2) {
            FILE *f = std::fopen(argv[2], "wb");
            if (f) {
                std::fwrite(data.samples.data(), sizeof(float), data.samples.size(), f);
                std::fclose(f);
            }
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "nyq_decode: %s\n", e.what());
        return 1;
    }
}
