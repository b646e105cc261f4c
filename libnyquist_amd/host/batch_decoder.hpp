// batch_decoder.hpp -- the batched two-pass Opus frame loop:
//   pass 0 (CPU): Ogg demux + packet framing -> how many frames of which size every stream holds; streams
//           of one shape form a group with page-locked freq/PCM buffers, cut into pieces of ~24 MB
//   pass 1 (CPU, bit-serial, one thread per file): CeltDecoder -> freq[] written in place into the group
//   pass 2 (MI355X, overlapping pass 1): as soon as the streams of a piece are decoded, one of six feeder
//           threads (one GPU context each: pieces upload, compute and download side by side -- the post-filter
//           of a piece with few long streams is a handful of sequential waves, so several must be in flight) runs
//           nyq_celt_frames_to_pcm = inverse MDCTs + TDAC chaining + post-filter + de-emphasis + interleave
//   pass 3 (CPU threads): channel mapping, pre-skip / end trimming (RFC 7845 section 4), header gain.
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
    double gpuSeconds = 0;             // wall time after pass 1: GPU work not hidden behind it, and pass 3
    double gpuBusySeconds = 0;         // time the feeder threads spent inside GPU calls (PCIe included), summed
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

    // give the page-locked staging memory back if it has grown beyond `keepBytes` (a pooled decoder should not sit
    // on gigabytes of pinned memory after one big job)
    void trim(size_t keepBytes);

private:
    void *arena(size_t bytes);
    static constexpr int kFeeders = 6;        // GPU contexts / feeder threads: pieces in flight at once
    void *ctx_[kFeeders] = {nullptr};         // nyq_ctx*
    void *arena_ = nullptr;            // staging memory of the groups, kept between calls
    size_t arenaBytes_ = 0;
    bool pinned_ = false;
};

}  // namespace nyq_host
