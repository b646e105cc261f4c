// nqr_surface.cpp -- NyquistIO / OpusDecoder of the plugin surface (reference: src/Common.cpp:33-219,
// src/OpusDecoder.cpp:39-183), backed by the batched MI355X decode path.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <condition_variable>
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

// The device of the single-file Load path (the reference has no device argument: its offload uses device 0): set with
// nqr::SetDefaultDevice, otherwise NYQ_DEVICE as it stood when the FIRST Load ran -- the environment is read once, never
// again on a decoding path (a host program may change its environment from other threads).
std::atomic<int> g_defaultDevice{-1};
int defaultDevice() {
    int d = g_defaultDevice.load(std::memory_order_acquire);
    if (d >= 0) return d;
    static const int fromEnv = [] {
        const char *e = std::getenv("NYQ_DEVICE");
        char *end = nullptr;
        const long v = e ? std::strtol(e, &end, 10) : 0;
        return (e && end != e && *end == 0 && v >= 0 && v < 1024) ? (int)v : 0;
    }();
    return fromEnv;
}
}  // namespace

void nqr::SetDefaultDevice(int device) { g_defaultDevice.store(device < 0 ? -1 : device, std::memory_order_release); }

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

// Decoders (GPU contexts and page-locked staging memory) are kept between calls: a lease takes an idle one of the device
// list or makes one, and hands it back afterwards, so concurrent Loads from several threads each work on their own decoder
// and repeated Loads do not re-create contexts.  (The reference makes a fresh decoder per Load and shares nothing,
// src/OpusDecoder.cpp:169-178.)
//
// A decoder that is handed back is KEPT, up to kMaxIdle per device list -- as many as any sensible number of loading
// threads -- and only its staging memory is trimmed: the first two keep up to 512 MB of page-locked memory, the others 32 MB.
// Round 2 kept two and destroyed every further one on return, i.e. with six loading threads four decoders (24 contexts,
// 48+ HIP streams, pinned arenas) were torn down and rebuilt per Load while the other threads were inside their GPU calls;
// the one process abort this project has seen happened in exactly that test.  A decoder beyond
// kMaxIdle is RETIRED, not destroyed on the spot: it waits on a list that is drained only by the lease that brings the
// number of active leases to zero -- no context, stream or pinned arena is ever torn down while another thread may be inside
// a GPU call (round 4; before, teardown was merely serialised with construction).
// The pool itself is never destroyed (no HIP call from a static destructor after the runtime has shut down).
namespace {
constexpr size_t kMaxIdle = 16;
struct DecoderPool {
    std::mutex mu;
    std::map<std::vector<int>, std::vector<nyq_host::BatchOpusDecoder *>> idle;   // by device list
    std::vector<nyq_host::BatchOpusDecoder *> retired;   // surplus decoders waiting for a moment without active leases
    long active = 0;                         // leases alive (guarded by mu)
    bool draining = false;                   // the retired decoders are being destroyed: new leases wait (guarded by mu)
    std::condition_variable drained;
    std::mutex lifecycle;                    // context construction / destruction, one at a time
    std::atomic<long> created{0}, destroyed{0};
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
        DecoderPool &P = decoderPool();
        {
            std::unique_lock<std::mutex> lk(P.mu);
            P.drained.wait(lk, [&] { return !P.draining; });
            P.active++;
            auto &v = P.idle[d];
            if (!v.empty()) {
                dec = v.back();
                v.pop_back();
            }
        }
        if (!dec) {
            try {
                std::lock_guard<std::mutex> lk(P.lifecycle);
                dec = new nyq_host::BatchOpusDecoder(d);   // (throws: nothing leased, nothing to hand back)
                P.created++;
            } catch (...) {
                release(nullptr);
                throw;
            }
        }
    }
    // one lease less; the lease that leaves nobody behind destroys what was retired meanwhile
    static void release(nyq_host::BatchOpusDecoder *surplus) {
        DecoderPool &P = decoderPool();
        std::vector<nyq_host::BatchOpusDecoder *> doomed;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (surplus) P.retired.push_back(surplus);
            if (--P.active == 0 && !P.retired.empty()) {
                doomed.swap(P.retired);
                P.draining = true;                         // leases that start from here on wait until the teardown is over
            }
        }
        if (doomed.empty()) return;
        {
            std::lock_guard<std::mutex> lk(P.lifecycle);
            for (nyq_host::BatchOpusDecoder *d : doomed) {
                delete d;
                P.destroyed++;
            }
        }
        {
            std::lock_guard<std::mutex> lk(P.mu);
            P.draining = false;
        }
        P.drained.notify_all();
    }
    ~DecoderLease() {                                      // never throws: trim() and the destructor only free
        DecoderPool &P = decoderPool();
        size_t ahead;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            ahead = P.idle[device].size();
        }
        if (ahead < kMaxIdle) {
            dec->trim(ahead < 2 ? (size_t)512 << 20 : (size_t)32 << 20);
            std::lock_guard<std::mutex> lk(P.mu);
            auto &v = P.idle[device];
            if (v.size() < kMaxIdle) {
                v.push_back(dec);
                dec = nullptr;
            }
        }
        release(dec);                                      // (dec != null: beyond kMaxIdle, retired)
    }
    DecoderLease(const DecoderLease &) = delete;
    DecoderLease &operator=(const DecoderLease &) = delete;
};
}  // namespace

void nqr::OpusDecoder::LoadFromBuffer(AudioData *data, const std::vector<uint8_t> &memory) {
    DecoderLease lease(defaultDevice());
    std::vector<nyq_host::DecodedStream> out;
    lease.dec->decode({&memory}, out, nullptr, 1);
    fill(data, out[0]);
}

// decoders made / torn down by the Load paths so far (tests: a steady state of concurrent Loads makes none)
void nqr::DecoderPoolCounts(long *created, long *destroyed) {
    if (created) *created = decoderPool().created.load();
    if (destroyed) *destroyed = decoderPool().destroyed.load();
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
