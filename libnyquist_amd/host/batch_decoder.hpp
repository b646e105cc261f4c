// batch_decoder.hpp -- the batched two-pass Opus frame loop:
//   pass 0 (CPU): Ogg demux + packet framing -> how many frames of which size every stream holds; streams
//           of one shape form a group with page-locked freq/PCM buffers, cut into pieces of ~24 MB
//   pass 1 (CPU, bit-serial, one thread per file): CeltDecoder -> freq[] written in place into the group
//   pass 2 (MI355X, overlapping pass 1): as soon as the streams of a piece are decoded, one of six feeder
//           threads (one GPU context each: pieces upload, compute and download side by side -- the post-filter
//           of a piece with few long streams is a handful of sequential waves, so several must be in flight) runs
//           nyq_celt_frames_to_pcm = inverse MDCTs + TDAC chaining + post-filter + de-emphasis + interleave
//   pass 3: channel mapping, pre-skip / end trimming (RFC 7845 section 4), header gain -- on the DEVICE (round 4): every
//           elementary stream of a file that needs them writes through an output record (nyq_out_desc) straight into the
//           file's interleaved layout in device memory, one download per file; files that are one identity-mapped stream at
//           unit gain take their samples from the dense output slice by slice.  No per-sample work on the host.
// Replaces the per-packet loop of src/OpusDecoder.cpp:95-122 (op_read_float).
#pragma once
#include <cstdint>
#include <functional>
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
    double wallSeconds = 0;            // the whole decode() call, first instruction to return (the figure to quote)
    double cpuSeconds = 0;             // breakdown: wall time of pass 0 + pass 1
    double gpuSeconds = 0;             // breakdown: wall time after pass 1: GPU work not hidden behind it, and pass 3
    double gpuBusySeconds = 0;         // time the feeder threads spent inside GPU calls (PCIe included), summed
    long frames = 0;                   // channel-independent frame count over all streams
    int threads = 0;
};

// Receives every file of a batch once, complete (or with its error set), in an unspecified order but never concurrently;
// the stream's storage goes back to the decoder's pool when the callback returns (move `pcm` out to keep it).
using StreamSink = std::function<void(size_t index, DecodedStream &stream)>;

class BatchOpusDecoder {
public:
    explicit BatchOpusDecoder(int device = 0);
    // One decoder over several GPUs of the node (BASELINE config 4 "batch-sharded 1 -> 8 GPUs"):
    // elementary stream number s of a (channels, frame size) class goes to devices[s mod G]; every device has its own
    // feeder threads, contexts and page-locked staging arena; no data moves between devices, results stay on the host
    // side of each device's PCIe link until they are handed over.
    explicit BatchOpusDecoder(const std::vector<int> &devices);
    ~BatchOpusDecoder();
    BatchOpusDecoder(const BatchOpusDecoder &) = delete;
    BatchOpusDecoder &operator=(const BatchOpusDecoder &) = delete;
    // files[i] = a whole Ogg Opus file in memory
    void decode(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out,
                BatchStats *stats = nullptr, int threads = 0);
    // The same, results streamed to `sink` sub-batch by sub-batch out of POOLED sample buffers (kept by the decoder from
    // call to call: a big job does not allocate, first-touch and unmap gigabytes of output per call -- the shape of the
    // reference's own loop, src/OpusDecoder.cpp:82-87 inside a caller that drops each AudioData after use).
    void decode(const std::vector<const std::vector<uint8_t> *> &files, const StreamSink &sink, BatchStats *stats = nullptr,
                int threads = 0);
    int deviceCount() const { return (int)devices_.size(); }
    static constexpr int kFeedersPerDevice = 6;   // GPU contexts / feeder threads per device: pieces in flight at once
    // page-locked staging memory a sub-batch may use (default 12 GiB -- fewer, larger sub-batches: -5 % on a 512-stream job against 6, profiles/r04_ac_* --, or NYQ_BATCH_BYTES as it stood at construction)
    void setStagingBudget(size_t bytes) { if (bytes) stagingBudget_ = bytes; }
    // 20 ms mono / stereo streams: the entropy stage stops at the symbols and the GPU builds the band shapes
    // (nyq_celt_symbols_to_pcm_mapped); off = freq[] built on the host as for every other frame size.  Default on
    // (NYQ_HOST_SYMBOLS=0 at construction turns it off: the A/B switch of bench.py's file legs).
    void setSymbolRecords(bool on) { symbolRecords_ = on; }

    // give the page-locked staging memory back if it has grown beyond `keepBytes` (a pooled decoder should not sit
    // on gigabytes of pinned memory after one big job)
    void trim(size_t keepBytes);

private:
    void decodeImpl(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out, const StreamSink *sink,
                    BatchStats *stats, int threads);
    void *arena(int dev, size_t bytes);
    void *deviceArena(int dev, size_t bytes);   // device memory of one device for the files' interleaved output (grow only)
    static constexpr int kFeeders = kFeedersPerDevice;
    struct Arena {                            // page-locked staging memory of one device's groups, kept between calls
        void *p = nullptr;
        size_t bytes = 0;
        bool pinned = false;
    };
    std::vector<int> devices_;
    size_t stagingBudget_ = (size_t)12 << 30;
    bool symbolRecords_ = true;
    bool deviceEntropy_ = false;              // NYQ_DEVICE_ENTROPY=1: the entropy stage of one-frame-size mono / stereo streams runs on the GPU
    size_t longPieceStreams_ = 0;             // NYQ_LONG_PIECE_STREAMS: streams per piece of time-sliced streams (0: as many as threads)
    size_t pieceBytes_ = (size_t)24 << 20;    // of GPU input per piece of short streams (NYQ_PIECE_BYTES at construction)
    bool packedRecords_ = false;              // NYQ_HOST_PACKED=1 at construction: symbol records packed back to back
    bool trace_ = false;                      // NYQ_BATCH_TRACE=1 at construction: per-sub-batch timing on stderr
    std::vector<void *> ctx_;                 // nyq_ctx*, device d's feeders at [d * kFeeders, (d + 1) * kFeeders)
    std::vector<Arena> arenas_;
    std::vector<Arena> devArenas_;            // per device: device memory (Arena::pinned unused)
    std::vector<std::vector<float>> pool_;    // sample buffers of the sink form, capacity kept
};

// frames whose entropy stage ran on the GPU so far in this process (NYQ_DEVICE_ENTROPY=1; a test hook)
long deviceEntropyFrames();

}  // namespace nyq_host
