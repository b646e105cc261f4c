// libnyquist/Decoders.h -- the decoder plugin surface (reference: include/libnyquist/Decoders.h:37-91).
//
// BaseDecoder, NyquistIO and OpusDecoder keep the reference's signatures and exception types.  What
// differs is behind OpusDecoder: packets are entropy-decoded on the CPU into freq[] for the WHOLE
// file (nyq_host::CeltDecoder), then the inverse MDCT, post-filter and de-emphasis of every frame run
// as batches on the MI355X through libnyq_imdct.so -- the two-pass frame loop that replaces
// src/OpusDecoder.cpp:95-122.  nqr::BatchLoad() decodes many files in one GPU batch.
#ifndef AUDIO_DECODER_H
#define AUDIO_DECODER_H

#include "Common.h"
#include <utility>
#include <map>
#include <memory>
#include <exception>
#include <stdexcept>

namespace nqr
{
    struct BaseDecoder
    {
        virtual void LoadFromPath(nqr::AudioData * data, const std::string & path) = 0;
        virtual void LoadFromBuffer(nqr::AudioData * data, const std::vector<uint8_t> & memory) = 0;
        virtual std::vector<std::string> GetSupportedFileExtensions() = 0;
        virtual ~BaseDecoder() {}
    };

    typedef std::pair< std::string, std::shared_ptr<nqr::BaseDecoder> > DecoderPair;

    class NyquistIO
    {
        std::string ParsePathForExtension(const std::string & path) const;
        std::shared_ptr<nqr::BaseDecoder> GetDecoderForExtension(const std::string & ext);
        void BuildDecoderTable();
        void AddDecoderToTable(std::shared_ptr<nqr::BaseDecoder> decoder);
        std::map< std::string, std::shared_ptr<BaseDecoder> > decoderTable;

        NO_MOVE(NyquistIO);

    public:

        NyquistIO();
        ~NyquistIO();
        void Load(AudioData * data, const std::string & path);
        void Load(AudioData * data, const std::vector<uint8_t> & buffer);
        void Load(AudioData * data, const std::string & extension, const std::vector<uint8_t> & buffer);
        bool IsFileSupported(const std::string & path) const;
    };

    struct UnsupportedExtensionEx : public std::runtime_error { UnsupportedExtensionEx() : std::runtime_error("Unsupported file extension") {} };
    struct LoadPathNotImplEx : public std::runtime_error { LoadPathNotImplEx() : std::runtime_error("Loading from path not implemented") {} };
    struct LoadBufferNotImplEx : public std::runtime_error { LoadBufferNotImplEx() : std::runtime_error("Loading from buffer not implemented") {} };

    struct OpusDecoder final : public nqr::BaseDecoder
    {
        OpusDecoder() = default;
        virtual ~OpusDecoder() override {}
        virtual void LoadFromPath(nqr::AudioData * data, const std::string & path) override final;
        virtual void LoadFromBuffer(nqr::AudioData * data, const std::vector<uint8_t> & memory) override final;
        virtual std::vector<std::string> GetSupportedFileExtensions() override final;
    };

    // Not in the reference: decode many Ogg Opus files as ONE GPU batch (one entropy-decoding thread per
    // host core, then a single nyq_celt_frames_to_pcm call per group of equally shaped streams).
    // out[i] is filled exactly as NyquistIO::Load would fill it for buffers[i].
    void BatchLoad(std::vector<AudioData> & out, const std::vector< std::vector<uint8_t> > & buffers, int device = 0);

} // end namespace nqr

#endif
