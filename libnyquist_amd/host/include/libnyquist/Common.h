// libnyquist/Common.h -- the data types of the plugin surface, MI355X build.
//
// Only the types the Opus path hands to its caller exist here, under the reference's names so that code
// written against dafx/libnyquist compiles unchanged (PCMFormat: reference include/libnyquist/Common.h:316-327,
// AudioData: :350-364).  The reference's sample-format converters, dithering and WAV structures belong to
// other decoders and are not part of this build.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nqr {

// Sample formats, enumerators and order as in the reference (so the integer values agree too).
enum PCMFormat { PCM_U8, PCM_S8, PCM_16, PCM_24, PCM_32, PCM_64, PCM_FLT, PCM_DBL, PCM_END };

int GetFormatBitsPerSample(PCMFormat format);                               // 8 ... 64, 0 for PCM_END
PCMFormat MakeFormatForBits(int bits, bool floatingPoint, bool isSigned);   // PCM_END if there is none

// What a decoder fills in.  For Opus: 48 kHz, PCM_FLT, frameSize = channels * 32, lengthSeconds in whole
// seconds (src/OpusDecoder.cpp:75-79,160), samples.size() = samples per channel * channels.
struct AudioData {
    int channelCount = 0;
    int sampleRate = 0;
    double lengthSeconds = 0.0;
    size_t frameSize = 0;                      // channels * bits per sample
    std::vector<float> samples;                // interleaved, in [-1, 1]
    PCMFormat sourceFormat = PCM_END;
};

// A whole file in memory (ReadFile throws std::runtime_error when the file cannot be read).
struct NyquistFileBuffer {
    std::vector<uint8_t> buffer;
    size_t size = 0;
};
NyquistFileBuffer ReadFile(const std::string &pathToFile);

}  // namespace nqr
