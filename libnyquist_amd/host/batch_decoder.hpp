// batch_decoder.hpp -- the batched two-pass Opus frame loop:
//   pass 1 (CPU, bit-serial, one thread per stream): Ogg demux -> Opus packets -> CeltDecoder -> freq[]
//   pass 2 (MI355X, one call per group of equally shaped streams): nyq_celt_frames_to_pcm
//           = inverse MDCTs + TDAC chaining + post-filter + de-emphasis + interleave
//   pass 3 (CPU): pre-skip / end trimming (RFC 7845 section 4), header gain.
// Replaces the per-packet loop of src/OpusDecoder.cpp:95-122 (op_read_float).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nyq_host {

struct DecodedStream {
    int channels = 0;
    int preSkip = 0;
    int64_t totalSamples = 0;          // per channel, after trimming (op_pcm_total)
    std::vector<float> pcm;            // interleaved, totalSamples * channels
    std::string error;                 // non-empty: this stream failed (others are unaffected)
    // statistics of the run
    long frames = 0;
    long transientFrames = 0;
};

struct BatchStats {
    double cpuSeconds = 0;             // wall time of pass 1
    double gpuSeconds = 0;             // wall time of pass 2 incl. PCIe copies
    long frames = 0;                   // channel-independent frame count over all streams
    int threads = 0;
};

class BatchOpusDecoder {
public:
    explicit BatchOpusDecoder(int device = 0);
    ~BatchOpusDecoder();
    BatchOpusDecoder(const BatchOpusDecoder &) = delete;
    BatchOpusDecoder &operator=(const BatchOpusDecoder &) = delete;
    // files[i] = a whole Ogg Opus file in memory
    void decode(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out,
                BatchStats *stats = nullptr, int threads = 0);

private:
    void *ctx_ = nullptr;              // nyq_ctx*
};

}  // namespace nyq_host
