// libnyquist/Decoders.h -- the decoder plugin surface of the MI355X build.
//
// A program written against dafx/libnyquist includes this header under the same name and keeps compiling:
// the public names, signatures and exception types are the reference's (include/libnyquist/Decoders.h:37-91).
// Everything else -- layout, private members, how decoders are registered -- is this build's own.  Behind
// nqr::OpusDecoder the packets of the WHOLE file are entropy-decoded on the CPU (nyq_host::CeltDecoder) and
// the inverse MDCT, post-filter and de-emphasis of all frames run as batches on the GPU (libnyq_imdct.so):
// the frame loop that replaces src/OpusDecoder.cpp:95-122.  nqr::BatchLoad() is new.
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "Common.h"

namespace nqr {

// ---- what the loaders throw ------------------------------------------------------------------------
struct UnsupportedExtensionEx : public std::runtime_error {
    UnsupportedExtensionEx() : std::runtime_error("Unsupported file extension") {}
};
struct LoadPathNotImplEx : public std::runtime_error {
    LoadPathNotImplEx() : std::runtime_error("Loading from path not implemented") {}
};
struct LoadBufferNotImplEx : public std::runtime_error {
    LoadBufferNotImplEx() : std::runtime_error("Loading from buffer not implemented") {}
};

// ---- one decoder per container / codec ---------------------------------------------------------------
struct BaseDecoder {
    virtual ~BaseDecoder() = default;
    virtual std::vector<std::string> GetSupportedFileExtensions() = 0;
    virtual void LoadFromPath(AudioData *data, const std::string &path) = 0;
    virtual void LoadFromBuffer(AudioData *data, const std::vector<uint8_t> &memory) = 0;
};

using DecoderPair = std::pair<std::string, std::shared_ptr<BaseDecoder>>;

// Ogg Opus (CELT-only packets; mono, stereo and multistream files): CPU entropy stage + GPU synthesis
struct OpusDecoder final : public BaseDecoder {
    std::vector<std::string> GetSupportedFileExtensions() override;
    void LoadFromPath(AudioData *data, const std::string &path) override;
    void LoadFromBuffer(AudioData *data, const std::vector<uint8_t> &memory) override;
};

// ---- the front door ------------------------------------------------------------------------------
class NyquistIO {
public:
    NyquistIO();
    ~NyquistIO();
    NyquistIO(const NyquistIO &) = delete;
    NyquistIO &operator=(const NyquistIO &) = delete;

    bool IsFileSupported(const std::string &path) const;
    void Load(AudioData *data, const std::string &path);                         // by file name extension
    void Load(AudioData *data, const std::vector<uint8_t> &buffer);              // by content sniffing
    void Load(AudioData *data, const std::string &extension, const std::vector<uint8_t> &buffer);

private:
    static std::string extensionOf(const std::string &path);
    void registerDecoder(const std::shared_ptr<BaseDecoder> &decoder);
    std::shared_ptr<BaseDecoder> decoderFor(const std::string &extension) const;   // null if none

    std::map<std::string, std::shared_ptr<BaseDecoder>> byExtension_;
};

// Not in the reference: many Ogg Opus files decoded as ONE batch -- one entropy-decoding thread per host core
// feeding GPU pieces as they complete.  out[i] is filled exactly as NyquistIO::Load would fill it for
// buffers[i]; a file that fails makes the call throw after the others have been decoded.
void BatchLoad(std::vector<AudioData> &out, const std::vector<std::vector<uint8_t>> &buffers, int device = 0);
// The same over several GPUs of the node: elementary stream s of a shape class is decoded on devices[s mod G]
// (independent streams, no exchange between devices); the results are identical to the one-device call.
void BatchLoad(std::vector<AudioData> &out, const std::vector<std::vector<uint8_t>> &buffers, const std::vector<int> &devices);

// GPU that NyquistIO::Load / OpusDecoder use (default: NYQ_DEVICE as it stood at the first Load, else 0); < 0 = back to that
void SetDefaultDevice(int device);

// Diagnostics: decoders (GPU contexts + staging memory) the Load / BatchLoad paths have made and torn down so far.  They
// are pooled per device list, so a steady state of concurrent Loads makes and destroys none.
void DecoderPoolCounts(long *created, long *destroyed);

}  // namespace nqr
