// range_decoder.hpp -- the Opus range decoder (RFC 6716 section 4.1; reference: celt/entdec.c,
// entcode.c:69-93).  32-bit code word, 8-bit symbols, raw bits read from the end of the frame.
#pragma once
#include <cstdint>

#include "celt_mode.hpp"

namespace nyq_host {

class RangeDecoder {
public:
    void init(const uint8_t *data, uint32_t len) {
        buf_ = data; storage_ = len; endOffs_ = 0; endWindow_ = 0; nEndBits_ = 0; offs_ = 0; error_ = 0;
        nbitsTotal_ = 32 + 1 - ((32 - kExtra) / 8) * 8;          // entdec.c:128-130
        rng_ = 1u << kExtra;
        rem_ = readByte();
        val_ = rng_ - 1 - (rem_ >> (8 - kExtra));
        normalize();
    }
    // cumulative frequency of the next symbol, total ft (entdec.c:138-144)
    unsigned decode(unsigned ft) {
        ext_ = rng_ / ft;
        const unsigned s = (unsigned)(val_ / ext_);
        return ft - (s + 1 < ft ? s + 1 : ft);
    }
    unsigned decodeBin(unsigned bits) {
        ext_ = rng_ >> bits;
        const unsigned s = (unsigned)(val_ / ext_);
        const unsigned top = 1u << bits;
        return top - (s + 1 < top ? s + 1 : top);
    }
    void update(unsigned fl, unsigned fh, unsigned ft) {
        const uint32_t s = ext_ * (ft - fh);
        val_ -= s;
        rng_ = fl > 0 ? ext_ * (fh - fl) : rng_ - s;
        normalize();
    }
    int bitLogp(unsigned logp) {                                  // entdec.c:162-176
        const uint32_t r = rng_, d = val_, s = r >> logp;
        const int ret = d < s;
        if (!ret) val_ = d - s;
        rng_ = ret ? s : r - s;
        normalize();
        return ret;
    }
    int icdf(const uint8_t *table, unsigned ftb) {                // entdec.c:178-199
        uint32_t s = rng_, t;
        const uint32_t d = val_, r = s >> ftb;
        int ret = -1;
        do {
            t = s;
            s = r * table[++ret];
        } while (d < s);
        val_ = d - s;
        rng_ = t - s;
        normalize();
        return ret;
    }
    uint32_t uint(uint32_t ft) {                                   // entdec.c:201-226
        ft--;
        int ftb = ilog(ft);
        if (ftb > 8) {
            ftb -= 8;
            const unsigned top = (unsigned)(ft >> ftb) + 1;
            const unsigned s = decode(top);
            update(s, s + 1, top);
            const uint32_t t = (uint32_t)s << ftb | bits(ftb);
            if (t <= ft) return t;
            error_ = 1;
            return ft;
        }
        ft++;
        const unsigned s = decode((unsigned)ft);
        update(s, s + 1, (unsigned)ft);
        return s;
    }
    uint32_t bits(unsigned n) {                                    // entdec.c:228-245
        uint32_t window = endWindow_;
        int available = nEndBits_;
        if ((unsigned)available < n) {
            do {
                window |= (uint32_t)readByteFromEnd() << available;
                available += 8;
            } while (available <= 32 - 8);
        }
        const uint32_t ret = window & (((uint32_t)1 << n) - 1u);
        window >>= n;
        available -= n;
        endWindow_ = window;
        nEndBits_ = available;
        nbitsTotal_ += n;
        return ret;
    }
    int tell() const { return nbitsTotal_ - ilog(rng_); }           // entcode.h ec_tell
    // Bits used so far in 1/8 bit (entcode.c:69-93).  The reference squares the top 16 bits of the range three times, a bit of
    // the fraction per squaring; at 1/8-bit resolution the result is a step function of those 16 bits with eight steps, so a
    // linear guess plus one table comparison gives the same integer for every range value (all 32768 cases compared:
    // tests/test_host_decoder.py::test_tell_frac_table_equals_the_squaring_loop) -- the band loop asks this twice per
    // angle and once per band.
    uint32_t tellFrac() const {
        static constexpr uint32_t kStep[8] = {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535};
        const uint32_t nbits = (uint32_t)nbitsTotal_ << kBitRes;
        const int l = ilog(rng_);
        const uint32_t r = rng_ >> (l - 16);
        uint32_t b = (r >> 12) - 8;
        b += r > kStep[b];
        return nbits - (((uint32_t)l << 3) + b);
    }
    // every value of the range's top 16 bits at every normalised width: mismatches between the two forms (test hook)
    static long tellFracSelfCheck() {
        long bad = 0;
        RangeDecoder d;
        d.nbitsTotal_ = 400;
        for (int width = 24; width <= 32; width++)
            for (uint32_t top = 32768; top < 65536; top++) {
                d.rng_ = width == 32 ? top << 16 | 0xFFFFu : (top << (width - 16)) | ((1u << (width - 16)) - 1);
                bad += d.tellFrac() != d.tellFracBySquaring();
                d.rng_ = top << (width - 16);
                bad += d.tellFrac() != d.tellFracBySquaring();
            }
        return bad;
    }
    uint32_t tellFracBySquaring() const {                           // (the reference's form, kept for the comparison test)
        const uint32_t nbits = (uint32_t)nbitsTotal_ << kBitRes;
        int l = ilog(rng_);
        uint32_t r = rng_ >> (l - 16);
        for (int i = kBitRes; i-- > 0;) {
            r = r * r >> 15;
            const int b = (int)(r >> 16);
            l = l << 1 | b;
            r >>= b;
        }
        return nbits - l;
    }
    void skipTo(int totalBits) { nbitsTotal_ += totalBits - tell(); }   // silence frames
    uint32_t storageBytes() const { return storage_; }
    uint32_t range() const { return rng_; }
    int error() const { return error_; }

private:
    static constexpr int kExtra = (32 - 2) % 8 + 1;   // EC_CODE_EXTRA = 7
    int readByte() { return offs_ < storage_ ? buf_[offs_++] : 0; }
    int readByteFromEnd() { return endOffs_ < storage_ ? buf_[storage_ - ++endOffs_] : 0; }
    void normalize() {                                              // entdec.c:111-121
        while (rng_ <= (1u << 23)) {
            nbitsTotal_ += 8;
            rng_ <<= 8;
            int sym = rem_;
            rem_ = readByte();
            sym = (sym << 8 | rem_) >> (8 - kExtra);
            val_ = ((val_ << 8) + (255 & ~sym)) & ((1u << 31) - 1);
        }
    }
    const uint8_t *buf_ = nullptr;
    uint32_t storage_ = 0, endOffs_ = 0, endWindow_ = 0, offs_ = 0, rng_ = 0, val_ = 0, ext_ = 0;
    int nEndBits_ = 0, nbitsTotal_ = 0, rem_ = 0, error_ = 0;
};

// Laplace-distributed integer (coarse energy), laplace.c:92-134
int laplaceDecode(RangeDecoder &dec, unsigned fs, int decay);

}  // namespace nyq_host
