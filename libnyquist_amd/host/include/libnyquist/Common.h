// libnyquist/Common.h -- the public data types of the plugin surface, for the MI355X build.
//
// Same names, fields and meaning as the reference's include/libnyquist/Common.h:316-327 (PCMFormat)
// and :350-364 (AudioData), so code written against dafx/libnyquist's NyquistIO::Load() compiles
// unchanged.  Only what the Opus path touches is provided; the reference's PCM conversion helpers,
// dithering and WAV structures are outside the accelerated path.
#ifndef LIBNYQUIST_COMMON_H
#define LIBNYQUIST_COMMON_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#define NO_COPY(C) C(const C &) = delete; C & operator = (const C &) = delete
#define NO_MOVE(C) NO_COPY(C); C(C &&) = delete; C & operator = (const C &&) = delete

namespace nqr
{

enum PCMFormat
{
    PCM_U8,
    PCM_S8,
    PCM_16,
    PCM_24,
    PCM_32,
    PCM_64,
    PCM_FLT,
    PCM_DBL,
    PCM_END
};

int GetFormatBitsPerSample(PCMFormat f);
PCMFormat MakeFormatForBits(int bits, bool floatingPt, bool isSigned);

struct AudioData
{
    int channelCount;
    int sampleRate;
    double lengthSeconds;
    size_t frameSize;              // channels * bits per sample
    std::vector<float> samples;    // interleaved, [-1, 1]
    PCMFormat sourceFormat;
};

struct NyquistFileBuffer
{
    std::vector<uint8_t> buffer;
    size_t size;
};

NyquistFileBuffer ReadFile(const std::string & pathToFile);

} // end namespace nqr

#endif
