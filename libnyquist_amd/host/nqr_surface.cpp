// nqr_surface.cpp -- NyquistIO / OpusDecoder of the plugin surface (reference: src/Common.cpp:33-219,
// src/OpusDecoder.cpp:39-183), backed by the batched MI355X decode path.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <iostream>

#include "batch_decoder.hpp"
#include "libnyquist/Decoders.h"

using namespace nqr;

namespace {
const int OPUS_SAMPLE_RATE = 48000;               // src/OpusDecoder.cpp:37

void fill(AudioData *d, nyq_host::DecodedStream &s) {
    if (!s.error.empty()) throw std::runtime_error(s.error);
    d->sampleRate = OPUS_SAMPLE_RATE;              // src/OpusDecoder.cpp:75-79
    d->channelCount = s.channels;
    d->sourceFormat = MakeFormatForBits(32, true, false);
    d->lengthSeconds = double(uint64_t(s.totalSamples / OPUS_SAMPLE_RATE));   // integer division, :160
    d->frameSize = (size_t)s.channels * GetFormatBitsPerSample(d->sourceFormat);
    d->samples = std::move(s.pcm);                 // samples.size() == op_pcm_total * channels
    if (d->samples.empty()) throw std::runtime_error("could not read any data");
}

int deviceFromEnv() {
    const char *e = std::getenv("NYQ_DEVICE");
    return e ? std::atoi(e) : 0;
}
}  // namespace

int nqr::GetFormatBitsPerSample(PCMFormat f) {
    switch (f) {
    case PCM_U8: case PCM_S8: return 8;
    case PCM_16: return 16;
    case PCM_24: return 24;
    case PCM_32: case PCM_FLT: return 32;
    case PCM_64: case PCM_DBL: return 64;
    default: return 0;
    }
}

PCMFormat nqr::MakeFormatForBits(int bits, bool floatingPt, bool isSigned) {
    switch (bits) {
    case 8: return isSigned ? PCM_S8 : PCM_U8;
    case 16: return PCM_16;
    case 24: return PCM_24;
    case 32: return floatingPt ? PCM_FLT : PCM_32;
    case 64: return floatingPt ? PCM_DBL : PCM_64;
    default: return PCM_END;
    }
}

NyquistFileBuffer nqr::ReadFile(const std::string &pathToFile) {
    FILE *f = std::fopen(pathToFile.c_str(), "rb");
    if (!f) throw std::runtime_error("file not found");
    std::fseek(f, 0, SEEK_END);
    const long len = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    NyquistFileBuffer b;
    b.buffer.resize(len > 0 ? (size_t)len : 0);
    const size_t got = b.buffer.empty() ? 0 : std::fread(b.buffer.data(), 1, b.buffer.size(), f);
    std::fclose(f);
    if (got != b.buffer.size()) throw std::runtime_error("error reading file or file too small");
    b.size = got;
    return b;
}

void nqr::OpusDecoder::LoadFromPath(AudioData *data, const std::string &path) {
    auto fileBuffer = nqr::ReadFile(path);
    LoadFromBuffer(data, fileBuffer.buffer);
}

// Decoders (two GPU contexts and the page-locked staging memory each) are kept between calls: a lease takes
// an idle one of the device or makes one, and hands it back afterwards.  Concurrent Loads from several
// threads therefore each work on their own decoder; at most two idle ones are kept per device.  The pool is
// never destroyed (no HIP call from a static destructor after the runtime has shut down).
namespace {
struct DecoderPool {
    std::mutex mu;
    std::map<std::vector<int>, std::vector<nyq_host::BatchOpusDecoder *>> idle;   // by device list
};
DecoderPool &decoderPool() {
    static DecoderPool *p = new DecoderPool;
    return *p;
}
struct DecoderLease {
    std::vector<int> device;
    nyq_host::BatchOpusDecoder *dec = nullptr;
    explicit DecoderLease(int d) : DecoderLease(std::vector<int>{d}) {}
    explicit DecoderLease(const std::vector<int> &d) : device(d) {
        {
            std::lock_guard<std::mutex> lk(decoderPool().mu);
            auto &v = decoderPool().idle[d];
            if (!v.empty()) {
                dec = v.back();
                v.pop_back();
            }
        }
        if (!dec) dec = new nyq_host::BatchOpusDecoder(d);
    }
    ~DecoderLease() {
        dec->trim((size_t)512 << 20);
        {
            std::lock_guard<std::mutex> lk(decoderPool().mu);
            auto &v = decoderPool().idle[device];
            if (v.size() < 2) {
                v.push_back(dec);
                dec = nullptr;
            }
        }
        delete dec;
    }
    DecoderLease(const DecoderLease &) = delete;
    DecoderLease &operator=(const DecoderLease &) = delete;
};
}  // namespace

void nqr::OpusDecoder::LoadFromBuffer(AudioData *data, const std::vector<uint8_t> &memory) {
    DecoderLease lease(deviceFromEnv());
    std::vector<nyq_host::DecodedStream> out;
    lease.dec->decode({&memory}, out, nullptr, 1);
    fill(data, out[0]);
}

std::vector<std::string> nqr::OpusDecoder::GetSupportedFileExtensions() { return {"opus"}; }

void nqr::BatchLoad(std::vector<AudioData> &out, const std::vector<std::vector<uint8_t>> &buffers, int device) {
    BatchLoad(out, buffers, std::vector<int>{device});
}

void nqr::BatchLoad(std::vector<AudioData> &out, const std::vector<std::vector<uint8_t>> &buffers, const std::vector<int> &devices) {
    DecoderLease lease(devices);
    std::vector<const std::vector<uint8_t> *> files;
    for (const auto &b : buffers) files.push_back(&b);
    std::vector<nyq_host::DecodedStream> dec_out;
    lease.dec->decode(files, dec_out);
    out.resize(buffers.size());
    for (size_t i = 0; i < buffers.size(); i++) fill(&out[i], dec_out[i]);
}

NyquistIO::NyquistIO() { registerDecoder(std::make_shared<OpusDecoder>()); }
NyquistIO::~NyquistIO() {}

std::string NyquistIO::extensionOf(const std::string &path) {
    const size_t dot = path.rfind('.');
    return dot == std::string::npos ? std::string() : path.substr(dot + 1);
}

void NyquistIO::registerDecoder(const std::shared_ptr<BaseDecoder> &decoder) {
    for (const std::string &ext : decoder->GetSupportedFileExtensions())
        if (!byExtension_.emplace(ext, decoder).second) throw std::runtime_error("decoder already exists for extension");
}

std::shared_ptr<BaseDecoder> NyquistIO::decoderFor(const std::string &extension) const {
    const auto it = byExtension_.find(extension);
    return it == byExtension_.end() ? nullptr : it->second;
}

bool NyquistIO::IsFileSupported(const std::string &path) const { return decoderFor(extensionOf(path)) != nullptr; }

void NyquistIO::Load(AudioData *data, const std::string &path) {
    const auto decoder = decoderFor(extensionOf(path));
    if (!decoder) throw UnsupportedExtensionEx();           // src/Common.cpp:49
    try {
        decoder->LoadFromPath(data, path);
    } catch (const std::exception &e) {
        std::cerr << "NyquistIO::Load(" << path << ") caught internal exception: " << e.what() << std::endl;
        throw;
    }
}

void NyquistIO::Load(AudioData *data, const std::vector<uint8_t> &buffer) {
    // magic sniffing (src/Common.cpp:66-141): of the formats the reference recognises only Ogg Opus is served here
    if (buffer.size() >= 36 && std::memcmp(buffer.data(), "OggS", 4) == 0) {
        const size_t lim = std::min<size_t>(buffer.size(), 4096);
        for (size_t i = 0; i + 8 <= lim; i++)
            if (std::memcmp(buffer.data() + i, "OpusHead", 8) == 0) return Load(data, "opus", buffer);
    }
    throw UnsupportedExtensionEx();
}

void NyquistIO::Load(AudioData *data, const std::string &extension, const std::vector<uint8_t> &buffer) {
    const auto decoder = decoderFor(extension);
    if (!decoder) throw UnsupportedExtensionEx();
    try {
        decoder->LoadFromBuffer(data, buffer);
    } catch (const std::exception &e) {
        std::cerr << "caught internal loading exception: " << e.what() << std::endl;
        throw;
    }
}
