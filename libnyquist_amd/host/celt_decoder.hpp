// celt_decoder.hpp -- the bit-serial half of a CELT frame: range decoding, band energies, bit
// allocation, PVQ shape decoding, folding/anti-collapse and denormalisation -- everything
// celt_decode_with_ec (third_party/opus/celt/celt_decoder_clean.c:353-731) does BEFORE
// compute_inv_mdcts.  The output is the frame's freq[] plus the few scalars the GPU stages need
// (transient flag, post-filter parameters); the IMDCT, post-filter and de-emphasis then run batched
// on the MI355X through libnyq_imdct.so.  One CeltDecoder per Opus stream (it carries the inter-frame
// energy prediction state).
#pragma once
#include <cstddef>
#include <cstdint>

#include "celt_mode.hpp"
#include "range_decoder.hpp"

namespace nyq_host {

struct CeltFrame {
    int LM = 3;                 // frame size = 120 << LM samples
    int channels = 2;           // CC: channels of the decoder / of freq[]
    bool transient = false;     // isTransient: 2^LM interleaved short blocks in freq[]
    bool silence = false;
    int pfPitch = 0;            // postfilter_pitch   (celt_decoder_clean.c:492-509)
    float pfGain = 0.f;         // postfilter_gain
    int pfTapset = 0;           // postfilter_tapset
    uint32_t rangeFinal = 0;    // dec->rng after the frame (st->rng), the spec's conformance hook
    size_t recordBytes = 0;     // decodeSymbols(): length of the (compact) record written, a multiple of 16
    // freq[channels][120 << LM] is written to the caller's buffer by CeltDecoder::decode()
};

class CeltDecoder {
public:
    explicit CeltDecoder(int channels);
    void reset();                                   // OPUS_RESET_STATE, celt_decoder_clean.c:846-859
    void setEndBand(int end) { end_ = end; }        // CELT_SET_END_BAND (bandwidth from the TOC)
    void setStreamChannels(int c) { streamChannels_ = c; }   // CELT_SET_CHANNELS (stereo flag of the TOC)
    // Decode one CELT frame of `frameSize` samples per channel from data[0..len).
    // freq: channels() * frameSize floats, channel-major (celt_decoder_clean.c:620-652).
    // Returns 0, or a negative OPUS_* style code (-1 bad arg, -3 internal, -4 corrupt).
    int decode(const uint8_t *data, int len, int frameSize, float *freq, CeltFrame &info);
    // The same frame as a SYMBOL record for the GPU (include/nyq_imdct.h: at most symbolBytes(channels(), LM) bytes): the
    // bit-serial work only -- the band shapes are built from the record by nyq_celt_shape_dev.  A frame whose record would
    // not fit its slot (an overlong leaf list) is finished here and travels as freq[] inside the record (NYQ_SYM_HOST_FREQ).
    int decodeSymbols(const uint8_t *data, int len, int frameSize, void *record, CeltFrame &info);
    static size_t symbolBytes(int channels, int LM = 3);   // = nyq_celt_symbol_bytes_lm (computed here: the entropy stage needs no GPU library)
    int channels() const { return channels_; }

    // Working memory of a frame, owned by the decoder (no allocation on the decode path): the normalised coefficients of
    // both channels, the pulse vectors of the band in hand, the copy later bands fold from, and the leaf list of the band's
    // split tree (celt_decoder.cpp: BandShaper).
    struct Scratch {
        float X[2 * 960];
        float norm[2 * 960];
        float foldWork[192], regroupTmp[192];
        // what phase 1 (symbols) leaves behind: the leaves of every split tree (with their pulse vectors' codewords), a record
        // per band vector, and the operations in execution order; phase 1b adds the pulse vectors themselves, at the
        // offsets of their coefficients
        int16_t pulses[2 * 960];
        // (the three records are laid out like include/nyq_imdct.h's nyq_sym_leaf / nyq_sym_vec / nyq_sym_op: decodeSymbols()
        // hands them to the GPU as they are)
        struct LeafSlot {
            int16_t off, n, k; uint8_t blocks, kind; float gain; int16_t foldOff; uint8_t shift, pad;
            uint32_t index; int16_t abs, pad2; uint16_t img[8];
        } leaves[1024];
        int32_t leafEnergy[1024];       // |y|^2 of a leaf's pulse vector (an exact small integer)
        struct VecSlot {
            int16_t x, n, fold, out, nbTree, leaf0, leaf1;
            uint8_t sel, recombine, timeDivide, Btree, Bin, band, cmCh, fillMode, fillLo, fillHi;
        } vecs[2 * 21 + 2];
        struct OpSlot { uint8_t kind, band; int16_t a, b, n; float f0, f1; } ops[5 * 21 + 8];
        int nleaves = 0, nvecs = 0, nops = 0;
    };

private:
    struct BandShaper;
    int decodeFrame(const uint8_t *data, int len, int frameSize, float *freq, uint8_t *record, CeltFrame &info);
    const CeltMode &m_;
    int channels_;            // CC
    int streamChannels_;      // C
    int start_ = 0, end_ = kBands;
    uint32_t rng_ = 0;
    float oldBandE_[2 * kBands];
    float oldLogE_[2 * kBands];
    float oldLogE2_[2 * kBands];
    float backgroundLogE_[2 * kBands];
    float wide_[2 * 960];     // both channels of a stereo-coded packet handed to a mono decoder
    Scratch scratch_;
};

// the tables of the frame-per-lane entropy stage (csrc/nyq_entropy_core.hpp: nyq_ent::EntropyTables), built from this decoder's
size_t entropyTablesBytes();
void fillEntropyTables(void *out);

}  // namespace nyq_host
