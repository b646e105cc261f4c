// opus_stream.hpp -- the container and packet layers in front of the CELT frame decoder:
//   * Ogg page / packet demultiplexing and the OpusHead header (RFC 3533, RFC 7845) -- what the
//     reference gets from third_party/libogg and third_party/opus/opusfile (op_test_memory,
//     op_head, op_pcm_total, pre-skip and end trimming; used from src/OpusDecoder.cpp:44-122);
//   * Opus packet framing: TOC byte, frame count codes 0-3, frame lengths (RFC 6716 section 3;
//     reference: libopus/src/opus.c opus_packet_parse_impl, opus_decoder_clean.c:608-722).
// Only what the MI355X decode path needs is here: CELT-only packets (TOC configurations 16-31) in
// channel mapping family 0 (mono/stereo) or family 1/255 (multistream, e.g. 5.1 or 8 channels).  SILK and hybrid packets are reported as
// unsupported, exactly because the hot path this repository accelerates is CELT's.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace nyq_host {

struct OpusHead {
    int version = 0;
    int channels = 0;
    int preSkip = 0;
    uint32_t inputRate = 0;
    int outputGainQ8 = 0;
    int mappingFamily = 0;
    // channel mapping (RFC 7845 section 5.1.1); family 0 is filled in as 1 stream, coupled iff stereo
    int streamCount = 1;
    int coupledCount = 0;
    uint8_t mapping[255] = {0};
};

// The audio packets of a file, in order: ONE byte arena and a span per packet (a 224 s file has 11184 of them: a vector each
// was 11184 allocations per scan).  Iterates as objects with data() / size(), like the vectors it replaces.
class PacketList {
public:
    struct Ref {
        const uint8_t *p;
        size_t n;
        const uint8_t *data() const { return p; }
        size_t size() const { return n; }
    };
    class Iter {
    public:
        Iter(const PacketList *l, size_t i) : l_(l), i_(i) {}
        Ref operator*() const { return (*l_)[i_]; }
        Iter &operator++() { ++i_; return *this; }
        bool operator!=(const Iter &o) const { return i_ != o.i_; }
    private:
        const PacketList *l_;
        size_t i_;
    };
    void push_back(const std::vector<uint8_t> &pkt) {
        spans_.emplace_back(bytes_.size(), pkt.size());
        bytes_.insert(bytes_.end(), pkt.begin(), pkt.end());
    }
    void reserveBytes(size_t n) { bytes_.reserve(n); }
    size_t size() const { return spans_.size(); }
    bool empty() const { return spans_.empty(); }
    Ref operator[](size_t i) const { return Ref{bytes_.data() + spans_[i].first, spans_[i].second}; }
    Iter begin() const { return Iter(this, 0); }
    Iter end() const { return Iter(this, spans_.size()); }
private:
    std::vector<uint8_t> bytes_;
    std::vector<std::pair<size_t, size_t>> spans_;
};

struct OggOpusFile {
    OpusHead head;
    PacketList packets;                          // audio packets in order
    int64_t lastGranule = -1;                    // granule position of the last page (samples at 48 kHz incl. pre-skip)
};

// Parse a whole Ogg Opus file held in memory; throws std::runtime_error on malformed input.
OggOpusFile parseOggOpus(const uint8_t *data, size_t size);

struct PacketFrames {
    int config = 0;            // TOC >> 3
    bool stereo = false;       // TOC bit 2
    int frameSize = 0;         // samples per frame at 48 kHz
    int bandwidthEnd = 21;     // CELT end band for the packet's audio bandwidth
    std::vector<std::pair<const uint8_t *, int>> frames;   // (pointer, length) of each compressed frame
};

// Split one Opus packet into its frames.  Returns false if the packet is malformed.
// selfDelimited: the framing of RFC 6716 appendix B used for all but the last stream of a
// multistream packet; *consumed receives the number of bytes the packet occupies.
bool parseOpusPacket(const uint8_t *data, int len, PacketFrames &out, bool selfDelimited = false, int *consumed = nullptr);

}  // namespace nyq_host
