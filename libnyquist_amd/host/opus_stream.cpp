// opus_stream.cpp -- see opus_stream.hpp.
#include "opus_stream.hpp"

#include <cstring>

namespace nyq_host {

namespace {
uint32_t rd32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
int64_t rd64(const uint8_t *p) { return (int64_t)((uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32); }
}  // namespace

namespace {
// CRC-32 of RFC 3533 (polynomial 0x04c11db7, most significant bit first, no reflection), eight bytes per step ("slicing by
// 8": T[k][i] is the remainder of byte i followed by k zero bytes): the scan of a file checks every page, and at one byte per
// step that was 5-10 % of the host's time per file next to the entropy stage's 2e5 frames/s.
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            t[0][i] = r;
        }
        for (int k = 1; k < 8; k++)
            for (uint32_t i = 0; i < 256; i++) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
    }
};
inline uint32_t crcBytes(const CrcTables &T, uint32_t c, const uint8_t *p, size_t n) {
    while (n >= 8) {
        c ^= (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3];
        c = T.t[7][c >> 24] ^ T.t[6][(c >> 16) & 0xff] ^ T.t[5][(c >> 8) & 0xff] ^ T.t[4][c & 0xff] ^ T.t[3][p[4]] ^ T.t[2][p[5]] ^
            T.t[1][p[6]] ^ T.t[0][p[7]];
        p += 8;
        n -= 8;
    }
    for (; n > 0; n--, p++) c = (c << 8) ^ T.t[0][((c >> 24) & 0xff) ^ *p];
    return c;
}
uint32_t oggCrc(const uint8_t *header, size_t headerLen, const uint8_t *body, size_t bodyLen) {
    static const CrcTables T;
    uint8_t h[27 + 255];
    std::memcpy(h, header, headerLen);                              // (27 + segments <= 282 bytes)
    h[22] = h[23] = h[24] = h[25] = 0;                               // the checksum field counts as zero
    return crcBytes(T, crcBytes(T, 0, h, headerLen), body, bodyLen);
}
}  // namespace

OggOpusFile parseOggOpus(const uint8_t *data, size_t size) {
    OggOpusFile f;
    f.packets.reserveBytes(size);
    std::vector<uint8_t> pending;     // packet continued across pages
    bool havePending = false;
    int packetIndex = 0;
    size_t pos = 0;
    uint32_t serial = 0, nextSeq = 0;
    bool haveSerial = false;
    while (pos + 27 <= size) {
        if (std::memcmp(data + pos, "OggS", 4) != 0) {   // resynchronise on garbage
            pos++;
            continue;
        }
        const uint8_t *h = data + pos;
        const int headerType = h[5];
        const int64_t granule = rd64(h + 6);
        const uint32_t ser = rd32(h + 14);
        const int nsegs = h[26];
        if (pos + 27 + nsegs > size) break;
        size_t bodyLen = 0;
        for (int i = 0; i < nsegs; i++) bodyLen += h[27 + i];
        const uint8_t *body = h + 27 + nsegs;
        if (body + bodyLen > data + size) break;
        // page checksum (RFC 3533 section 6: CRC-32, polynomial 0x04c11db7, over the page with the field zeroed).
        // As libogg does (ogg_sync_pageseek), a capture whose checksum fails is NOT a page: skip one byte and look for
        // the next capture pattern -- stray "OggS" bytes in trailing junk and damaged pages of other logical streams
        // then do no harm.  A page of the selected stream that is lost this way shows as a gap in the page sequence
        // numbers below, which is where the reference gives up too (opusfile reports OP_HOLE, OpusDecoder.cpp:108-112
        // returns 0 and the load throws): there is no concealment in either build.
        if (oggCrc(h, 27 + (size_t)nsegs, body, bodyLen) != rd32(h + 22)) {
            pos++;
            continue;
        }
        if (!haveSerial) { serial = ser; haveSerial = true; nextSeq = rd32(h + 18); }
        if (ser == serial) {                               // first logical stream only
            if (rd32(h + 18) != nextSeq) throw std::runtime_error("Ogg page missing from the stream (damaged file)");
            nextSeq++;
            if (!(headerType & 1) && havePending) {        // a fresh packet starts: drop the dangling one
                pending.clear();
                havePending = false;
            }
            size_t off = 0;
            for (int i = 0; i < nsegs; i++) {
                const int seg = h[27 + i];
                pending.insert(pending.end(), body + off, body + off + seg);
                havePending = true;
                off += seg;
                if (seg < 255) {                           // packet complete
                    if (packetIndex == 0) {
                        if (pending.size() < 19 || std::memcmp(pending.data(), "OpusHead", 8) != 0)
                            throw std::runtime_error("not an Ogg Opus stream (OpusHead missing)");
                        f.head.version = pending[8];
                        f.head.channels = pending[9];
                        f.head.preSkip = pending[10] | pending[11] << 8;
                        f.head.inputRate = rd32(&pending[12]);
                        f.head.outputGainQ8 = (int16_t)(pending[16] | pending[17] << 8);
                        f.head.mappingFamily = pending[18];
                        if (f.head.channels < 1) throw std::runtime_error("OpusHead: zero channels");
                        if (f.head.mappingFamily == 0) {
                            if (f.head.channels > 2) throw std::runtime_error("OpusHead: family 0 with more than 2 channels");
                            f.head.streamCount = 1;
                            f.head.coupledCount = f.head.channels == 2;
                            f.head.mapping[0] = 0;
                            f.head.mapping[1] = 1;
                        } else {
                            if (pending.size() < (size_t)21 + f.head.channels) throw std::runtime_error("OpusHead: truncated mapping table");
                            f.head.streamCount = pending[19];
                            f.head.coupledCount = pending[20];
                            if (f.head.streamCount < 1 || f.head.coupledCount > f.head.streamCount)
                                throw std::runtime_error("OpusHead: bad stream counts");
                            for (int c = 0; c < f.head.channels; c++) {
                                f.head.mapping[c] = pending[21 + c];
                                if (f.head.mapping[c] != 255 && f.head.mapping[c] >= f.head.streamCount + f.head.coupledCount)
                                    throw std::runtime_error("OpusHead: mapping index out of range");
                            }
                        }
                    } else if (packetIndex == 1) {
                        if (pending.size() < 8 || std::memcmp(pending.data(), "OpusTags", 8) != 0)
                            throw std::runtime_error("OpusTags packet missing");
                    } else {
                        f.packets.push_back(pending);
                    }
                    packetIndex++;
                    pending.clear();
                    havePending = false;
                }
            }
            if (granule >= 0) f.lastGranule = granule;
        }
        pos += 27 + nsegs + bodyLen;
    }
    if (packetIndex < 2) throw std::runtime_error("truncated Ogg Opus stream");
    return f;
}

bool parseOpusPacket(const uint8_t *data, int len, PacketFrames &out, bool selfDelimited, int *consumed) {
    out.frames.clear();
    if (len < 1) return false;
    const int toc = data[0];
    out.config = toc >> 3;
    out.stereo = (toc >> 2) & 1;
    // RFC 6716 table 2: 16-19 CELT NB, 20-23 WB, 24-27 SWB, 28-31 FB; 2.5/5/10/20 ms
    if (out.config >= 16) {
        out.frameSize = 120 << (out.config & 3);
        static const int endBand[4] = {13, 17, 19, 21};    // opus_decoder_clean.c:463-483 (CELT_SET_END_BAND)
        out.bandwidthEnd = endBand[(out.config - 16) >> 2];
    } else if (out.config >= 12) {
        out.frameSize = (out.config & 1) ? 960 : 480;      // hybrid 10 / 20 ms
    } else {
        static const int silkSize[4] = {480, 960, 1920, 2880};
        out.frameSize = silkSize[out.config & 3];
    }
    const uint8_t *p = data + 1;
    int rem = len - 1;
    auto readLen = [&](int &val) -> bool {                 // 1- or 2-byte frame length
        if (rem < 1) return false;
        if (p[0] < 252) { val = p[0]; p += 1; rem -= 1; return true; }
        if (rem < 2) return false;
        val = 4 * p[1] + p[0];
        p += 2; rem -= 2;
        return true;
    };
    std::vector<int> lens;
    int padding = 0;
    switch (toc & 3) {
    case 0:
        lens.assign(1, -1);
        break;
    case 1:
        lens.assign(2, -1);                                // two frames of equal size
        break;
    case 2: {
        int l0;
        if (!readLen(l0)) return false;
        lens.assign(2, -1);
        lens[0] = l0;
        break;
    }
    default: {
        if (rem < 1) return false;
        const int ch = *p++;
        rem--;
        const int count = ch & 0x3F;
        if (count == 0 || out.frameSize * count > 5760) return false;
        if (ch & 0x40) {                                   // padding length, 255 = 254 more + continue
            int b;
            do {
                if (rem < 1) return false;
                b = *p++;
                rem--;
                padding += b == 255 ? 254 : b;
            } while (b == 255);
        }
        lens.assign(count, -1);
        if (ch & 0x80) {                                   // VBR: count-1 explicit lengths
            for (int i = 0; i < count - 1; i++)
                if (!readLen(lens[i])) return false;
        } else {
            lens.assign(count, -2);                        // CBR: all equal
        }
        break;
    }
    }
    // the last (or, for CBR, the common) frame length: explicit in self-delimited framing
    const int count = (int)lens.size();
    const bool cbr = lens[count - 1] == -2 || ((toc & 3) == 1);
    int known = 0;
    for (int i = 0; i < count - 1; i++)
        if (lens[i] >= 0) known += lens[i];
    if (selfDelimited) {
        int l;
        if (!readLen(l)) return false;
        if (cbr) lens.assign(count, l);
        else lens[count - 1] = l;
    } else {
        const int avail = rem - padding;
        if (avail < 0) return false;
        if (cbr) {
            if (avail % count) return false;
            lens.assign(count, avail / count);
        } else {
            if (known > avail) return false;
            lens[count - 1] = avail - known;
        }
    }
    int total = 0;
    for (int l : lens) {
        if (l < 0 || l > 1275) return false;
        total += l;
    }
    if (total + padding > rem) return false;
    for (int l : lens) {
        out.frames.push_back({p, l});
        p += l;
    }
    if (consumed) *consumed = (int)(p - data) + padding;
    return true;
}

}  // namespace nyq_host
