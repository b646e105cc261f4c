// nyq_entropy_core.hpp -- the entropy stage of a CELT frame (RFC 6716 section 4.3: range decoder, coarse / fine energy, tf flags,
// dynalloc, the bit allocation, split trees with their angles and pulse-vector CODEWORDS) as one function of ONE frame's bytes,
// written so that a GPU lane can run it: no state between frames, no heap, tables passed in as one plain block.
//
// Why this is possible: nothing a CELT frame's symbols decide depends on another frame.  What does cross frames is values only --
// the band energies (a predictor over the previous frame's), the noise generator's seed (the previous frame's final range) and the
// anti-collapse levels (the two previous frames' energies).  Those are left as per-frame DELTAS (EntEnergy) which a second,
// sequential-per-stream pass (energy_lane_*) folds into the records: one lane per (channel, band), a wave per stream.
//
// The same text compiles for the host (g++: tools/scripts/entropy_core_check.cpp compares every record with the host decoder's,
// CeltDecoder::decodeSymbols, on the CPU tier) and for the device (hipcc: csrc/nyq_entropy_kernel.hpp, a frame per lane).  The
// bitstream semantics are those of libnyquist_amd/host/celt_decoder.cpp (reference: third_party/opus/celt/celt_decoder_clean.c:
// 462-720, quant_bands.c:427-540, rate.c:247-638, bands.c:661-1518, entdec.c, laplace.c:92-134); records: include/nyq_imdct.h.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define NYQ_ED __host__ __device__ inline
#else
#define NYQ_ED inline
#endif

namespace nyq_ent {

constexpr int kBands = 21, kBitRes = 3, kOneBit = 8, kAllocRows = 11, kMaxFineBits = 8, kFineOffset = 21;
constexpr int kPvqInfo = 180, kPvqWords = 2595;            // the compact U(n, k) table of nyq_shape_kernel.hpp
constexpr int kLutBytes = 28672;

struct EntAlloc {                      // what of the allocation depends on (LM, C) alone (host: AllocConst)
    int16_t width[kBands], bins[kBands], first[kBands + 1], floorBits[kBands];
    int32_t trimUnit[kBands];
    int16_t row[kAllocRows][kBands];
    int16_t cap[kBands], pulseCap[kBands], boostQuantum[kBands];
    int32_t minBits;
};
struct EntropyTables {                 // filled by the host library (nyqh_entropy_tables), uploaded once
    EntAlloc alloc[4][2];
    int16_t eBands[kBands + 1], logN[kBands];
    float eMeans[kBands];
    int16_t cacheIndex[5 * kBands];
    uint8_t cacheBits[512];
    uint16_t lutOff[5 * kBands], lutLen[5 * kBands];   // bits -> pseudo-pulses per (LM + 1, band) (len 0: no such cache)
    uint8_t lut[kLutBytes];
    uint32_t pvq[kPvqInfo + kPvqWords];
};

// what the entropy stage says about a frame beside its record
struct EntInfo {
    uint32_t rangeFinal;
    int16_t pfPitch;
    uint8_t pfTapset, pfGainIndex;     // gain = .09375 * pfGainIndex (0: no post-filter parameters in the frame)
    uint8_t flags;                     // EntFlags
    uint8_t LM, C, start, end, pad[3];
};
enum EntFlags : uint8_t { kEntTransient = 1, kEntSilence = 2, kEntIntra = 4, kEntAntiCollapse = 8, kEntError = 16, kEntTooLarge = 32 };
// a frame's contribution to the energy recurrence, per (channel, band): E = coef * max(-9, E) + prev + q, then + fine, then + last
struct EntEnergy {
    float prev[2 * kBands], q[2 * kBands], fine[2 * kBands], last[2 * kBands];
};
// where a frame's bytes are and what its packet says
struct EntDesc {
    uint32_t offset;
    uint16_t len;
    uint8_t C, start, end, pad[3];
};

// the "spread" record the entropy stage writes: the three lists at fixed places, named in the head's reserved words
// (reserved[0] = ops offset | vecs offset << 16, reserved[1] = leaves offset | level offset << 16; zero = the compact form)
// A packet that codes one channel has at most one operation and one vector per band, one that codes two at most five and two
// (host: Scratch::ops / vecs): the places of the lists follow the packet's channel count.
constexpr int kRecGainOff = 32, kRecOpsOff = 200, kRecMaxLeaves = 416;   // (= the shape kernel's kSymMaxLeaves)
NYQ_ED int recMaxOps(int C) { return C == 2 ? 113 : 24; }
NYQ_ED int recMaxVecs(int C) { return C == 2 ? 44 : 22; }
NYQ_ED int recVecsOff(int C) { return kRecOpsOff + 16 * recMaxOps(C); }
NYQ_ED int recLeavesOff(int C) { return recVecsOff(C) + 24 * recMaxVecs(C); }                                // 3064 (two channels), 1112 (one)
// the slot that holds ANY frame of 120 << LM samples in `channels` channels: a vector of n bins has at most min(2^(LM+1), n/2)
// leaves -- 42, 84, 104, 208 per channel for LM 0..3 (the 48 kHz mode's band widths)
NYQ_ED int recFullSlot(int channels, int LM) {
    const int perChannel = LM == 0 ? 42 : LM == 1 ? 84 : LM == 2 ? 104 : 208;
    return (recLeavesOff(2) + 40 * perChannel * (channels == 2 ? 2 : 1) + 2 * kBands * 4 + 15) & ~15;
}

struct RecHead {
    uint32_t seed;
    uint16_t nleaves, nvecs, nops;
    uint8_t flags, spread, start, end, channels, lm;
    uint32_t reserved[4];
};
struct RecLeaf {
    int16_t off, n, k;
    uint8_t blocks, kind;
    float gain;
    int16_t foldOff;
    uint8_t shift, pad;
    uint32_t index;
    int16_t abs, pad2;
    uint16_t img[8];
};
struct RecVec {
    int16_t x, n, fold, out, nbTree, leaf0, leaf1;
    uint8_t sel, recombine, timeDivide, Btree, Bin, band, cmCh, fillMode, fillLo, fillHi;
};
struct RecOp {
    uint8_t kind, band;
    int16_t a, b, n;
    float f0, f1;
};
static_assert(sizeof(RecHead) == 32 && sizeof(RecLeaf) == 40 && sizeof(RecVec) == 24 && sizeof(RecOp) == 16, "record layout");

NYQ_ED int imin(int a, int b) { return a < b ? a : b; }
NYQ_ED int imax(int a, int b) { return a > b ? a : b; }
NYQ_ED int ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }

// ---- the range decoder (entdec.c) ----------------------------------------------------------------------------------------
struct Range {
    const uint8_t *buf;
    uint32_t storage, endOffs, endWindow, offs, rng, val, ext;
    int nEndBits, nbitsTotal, rem, error;
    NYQ_ED int readByte() { return offs < storage ? buf[offs++] : 0; }
    NYQ_ED int readByteFromEnd() { return endOffs < storage ? buf[storage - ++endOffs] : 0; }
    NYQ_ED void normalize() {
        while (rng <= (1u << 23)) {
            nbitsTotal += 8;
            rng <<= 8;
            int sym = rem;
            rem = readByte();
            sym = (sym << 8 | rem) >> 1;
            val = ((val << 8) + (255 & ~sym)) & ((1u << 31) - 1);
        }
    }
    NYQ_ED void init(const uint8_t *data, uint32_t len) {
        buf = data; storage = len; endOffs = 0; endWindow = 0; nEndBits = 0; offs = 0; error = 0; ext = 0;
        nbitsTotal = 9;
        rng = 1u << 7;
        rem = readByte();
        val = rng - 1 - (rem >> 1);
        normalize();
    }
    NYQ_ED unsigned decode(unsigned ft) {
        ext = rng / ft;
        const unsigned s = (unsigned)(val / ext);
        return ft - (s + 1 < ft ? s + 1 : ft);
    }
    NYQ_ED unsigned decodeBin(unsigned bits) {
        ext = rng >> bits;
        const unsigned s = (unsigned)(val / ext);
        const unsigned top = 1u << bits;
        return top - (s + 1 < top ? s + 1 : top);
    }
    NYQ_ED void update(unsigned fl, unsigned fh, unsigned ft) {
        const uint32_t s = ext * (ft - fh);
        val -= s;
        rng = fl > 0 ? ext * (fh - fl) : rng - s;
        normalize();
    }
    NYQ_ED int bitLogp(unsigned logp) {
        const uint32_t r = rng, d = val, s = r >> logp;
        const int ret = d < s;
        if (!ret) val = d - s;
        rng = ret ? s : r - s;
        normalize();
        return ret;
    }
    NYQ_ED int icdf(const uint8_t *table, unsigned ftb) {
        uint32_t s = rng, t;
        const uint32_t d = val, r = s >> ftb;
        int ret = -1;
        do {
            t = s;
            s = r * table[++ret];
        } while (d < s);
        val = d - s;
        rng = t - s;
        normalize();
        return ret;
    }
    NYQ_ED uint32_t bits(unsigned n) {
        uint32_t window = endWindow;
        int available = nEndBits;
        if ((unsigned)available < n) {
            do {
                window |= (uint32_t)readByteFromEnd() << available;
                available += 8;
            } while (available <= 32 - 8);
        }
        const uint32_t ret = window & (((uint32_t)1 << n) - 1u);
        window >>= n;
        available -= (int)n;
        endWindow = window;
        nEndBits = available;
        nbitsTotal += (int)n;
        return ret;
    }
    NYQ_ED uint32_t uint(uint32_t ft) {
        ft--;
        int ftb = ilog(ft);
        if (ftb > 8) {
            ftb -= 8;
            const unsigned top = (unsigned)(ft >> ftb) + 1;
            const unsigned s = decode(top);
            update(s, s + 1, top);
            const uint32_t t = (uint32_t)s << ftb | bits((unsigned)ftb);
            if (t <= ft) return t;
            error = 1;
            return ft;
        }
        ft++;
        const unsigned s = decode((unsigned)ft);
        update(s, s + 1, (unsigned)ft);
        return s;
    }
    NYQ_ED int tell() const { return nbitsTotal - ilog(rng); }
    NYQ_ED uint32_t tellFrac() const {
        // (the table form of libnyquist_amd/host/range_decoder.hpp: the eight steps of the top 16 bits at 1/8-bit resolution)
        const uint32_t nbits = (uint32_t)nbitsTotal << kBitRes;
        const int l = ilog(rng);
        const uint32_t r = rng >> (l - 16);
        uint32_t b = (r >> 12) - 8;
        const uint32_t step = b == 0 ? 35733u : b == 1 ? 38967u : b == 2 ? 42495u : b == 3 ? 46340u : b == 4 ? 50535u : b == 5 ? 55109u : b == 6 ? 60097u : 65535u;
        b += r > step;
        return nbits - (((uint32_t)l << 3) + b);
    }
    NYQ_ED void skipTo(int totalBits) { nbitsTotal += totalBits - tell(); }
};

NYQ_ED int laplace(Range &dec, unsigned fs, int decay) {
    int val = 0;
    unsigned fl = 0;
    const unsigned fm = dec.decodeBin(15);
    if (fm >= fs) {
        val++;
        fl = fs;
        fs = ((32768u - 32u - fs) * (uint32_t)(16384 - decay) >> 15) + 1;
        while (fs > 1 && fm >= fl + 2 * fs) {
            fs *= 2;
            fl += fs;
            fs = ((fs - 2) * (uint32_t)decay) >> 15;
            fs += 1;
            val++;
        }
        if (fs <= 1) {
            const int di = (int)(fm - fl) >> 1;
            val += di;
            fl += 2 * (unsigned)di;
        }
        if (fm < fl + fs) val = -val;
        else fl += fs;
    }
    const unsigned fh = fl + fs < 32768u ? fl + fs : 32768u;
    dec.update(fl, fh, 32768);
    return val;
}

// ---- constants of the specification (as in host/celt_decoder.cpp) -----------------------------------------------------------
NYQ_ED int energyModel(int LM, int intra, int k) {
    const uint8_t t[4][2][42] = {
        {{72, 127, 65, 129, 66, 128, 65, 128, 64, 128, 62, 128, 64, 128, 64, 128, 92, 78, 92, 79, 92, 78, 90, 79, 116, 41, 115, 40, 114, 40, 132, 26, 132, 26, 145, 17, 161, 12, 176, 10, 177, 11},
         {24, 179, 48, 138, 54, 135, 54, 132, 53, 134, 56, 133, 55, 132, 55, 132, 61, 114, 70, 96, 74, 88, 75, 88, 87, 74, 89, 66, 91, 67, 100, 59, 108, 50, 120, 40, 122, 37, 97, 43, 78, 50}},
        {{83, 78, 84, 81, 88, 75, 86, 74, 87, 71, 90, 73, 93, 74, 93, 74, 109, 40, 114, 36, 117, 34, 117, 34, 143, 17, 145, 18, 146, 19, 162, 12, 165, 10, 178, 7, 189, 6, 190, 8, 177, 9},
         {23, 178, 54, 115, 63, 102, 66, 98, 69, 99, 74, 89, 71, 91, 73, 91, 78, 89, 86, 80, 92, 66, 93, 64, 102, 59, 103, 60, 104, 60, 117, 52, 123, 44, 138, 35, 133, 31, 97, 38, 77, 45}},
        {{61, 90, 93, 60, 105, 42, 107, 41, 110, 45, 116, 38, 113, 38, 112, 38, 124, 26, 132, 27, 136, 19, 140, 20, 155, 14, 159, 16, 158, 18, 170, 13, 177, 10, 187, 8, 192, 6, 175, 9, 159, 10},
         {21, 178, 59, 110, 71, 86, 75, 85, 84, 83, 91, 66, 88, 73, 87, 72, 92, 75, 98, 72, 105, 58, 107, 54, 115, 52, 114, 55, 112, 56, 129, 51, 132, 40, 150, 33, 140, 29, 98, 35, 77, 42}},
        {{42, 121, 96, 66, 108, 43, 111, 40, 117, 44, 123, 32, 120, 36, 119, 33, 127, 33, 134, 34, 139, 21, 147, 23, 152, 20, 158, 25, 154, 26, 166, 21, 173, 16, 184, 13, 184, 10, 150, 13, 139, 15},
         {22, 178, 63, 114, 74, 82, 84, 83, 92, 82, 103, 62, 96, 72, 96, 67, 101, 73, 107, 72, 113, 55, 118, 52, 125, 52, 118, 52, 117, 55, 135, 49, 137, 39, 157, 32, 145, 29, 97, 33, 77, 40}}};
    return t[LM][intra][k];
}
NYQ_ED float predCoef(int LM) { return LM == 0 ? 29440 / 32768.f : LM == 1 ? 26112 / 32768.f : LM == 2 ? 21248 / 32768.f : 16384 / 32768.f; }
NYQ_ED float betaCoef(int LM) { return LM == 0 ? 30147 / 32768.f : LM == 1 ? 22282 / 32768.f : LM == 2 ? 12124 / 32768.f : 6554 / 32768.f; }
NYQ_ED int tfSelect(int LM, int k) {
    const int8_t t[4][8] = {{0, -1, 0, -1, 0, -1, 0, -1}, {0, -1, 0, -2, 1, 0, 1, -1}, {0, -2, 0, -3, 2, 0, 1, -1}, {0, -2, 0, -3, 3, 0, 1, -1}};
    return t[LM][k];
}
NYQ_ED int log2FracTab(int k) {
    const uint8_t t[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};
    return t[k];
}
NYQ_ED int fracMul16(int a, int b) { return (16384 + (int32_t)(int16_t)a * (int16_t)b) >> 15; }
NYQ_ED int bitexactCos(int x) {
    const int32_t t = (4096 + (int32_t)x * x) >> 13;
    int x2 = (int16_t)t;
    x2 = (32767 - x2) + fracMul16(x2, (-7651 + fracMul16(x2, (8277 + fracMul16(-626, x2)))));
    return 1 + x2;
}
NYQ_ED int bitexactLog2tan(int isin, int icos) {
    const int lc = ilog((uint32_t)icos), ls = ilog((uint32_t)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + fracMul16(isin, fracMul16(isin, -2597) + 7932) - fracMul16(icos, fracMul16(icos, -2597) + 7932);
}
NYQ_ED uint32_t floorSqrt(uint32_t v) {                         // v < 2^18 here: exact in float
    uint32_t r = (uint32_t)__builtin_sqrtf((float)v);
    if (r * r > v) r--;
    else if ((r + 1) * (r + 1) <= v) r++;
    return r;
}
NYQ_ED int pulsesOf(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }

// ---- fill-mask images: eight 16-bit lanes in two words (host: BandShaper::Img) -------------------------------------------------
struct Img {
    uint64_t a, b;
};
NYQ_ED uint64_t rep16(unsigned m) { return (uint64_t)(m & 0xFFFFu) * 0x0001000100010001ull; }
NYQ_ED Img imgAnd(Img x, unsigned m) { return Img{x.a & rep16(m), x.b & rep16(m)}; }
NYQ_ED Img imgOr(Img x, Img y) { return Img{x.a | y.a, x.b | y.b}; }
NYQ_ED Img imgShl(Img x, int s) {
    const uint64_t m = rep16(0xFFFFu << s);
    return Img{(x.a << s) & m, (x.b << s) & m};
}
NYQ_ED Img imgShr(Img x, int s) {
    const uint64_t m = rep16(0xFFFFu >> s);
    return Img{(x.a >> s) & m, (x.b >> s) & m};
}

struct BitPlan {
    int shape[kBands], fine[kBands], finePrio[kBands];
    int codedBands, intensity, dualStereo;
    int32_t balance;
};

// ---- one frame ---------------------------------------------------------------------------------------------------------------
struct Frame {
    const EntropyTables &T;
    const EntAlloc &K;
    Range rc;
    const int LM, C, N;
    uint8_t *rec;                      // the record slot
    int leafCap;
    int spread, intensity;
    int32_t remaining;
    int band, tfChange, fillLo, fillHi;
    int nops, nvecs, nleaves;
    int overflow;                      // which list outgrew its place (1 tree stack, 2 leaves, 4 operations, 8 vectors)
    NYQ_ED Frame(const EntropyTables &t, const EntAlloc &k, int lm, int c, int n, uint8_t *r, int cap)
        : T(t), K(k), LM(lm), C(c), N(n), rec(r), leafCap(cap), spread(0), intensity(0), remaining(0), band(0), tfChange(0), fillLo(0), fillHi(0),
          nops(0), nvecs(0), nleaves(0), overflow(0) {}

    NYQ_ED RecOp *ops() const { return reinterpret_cast<RecOp *>(rec + kRecOpsOff); }
    NYQ_ED RecVec *vecs() const { return reinterpret_cast<RecVec *>(rec + recVecsOff(C)); }
    NYQ_ED RecLeaf *leaves() const { return reinterpret_cast<RecLeaf *>(rec + recLeavesOff(C)); }
    NYQ_ED const uint8_t *cacheFor(int b, int lm) const { return T.cacheBits + T.cacheIndex[(lm + 1) * kBands + b]; }
    NYQ_ED int lut(int b, int lm, int bits) const {
        const int k = (lm + 1) * kBands + b, len = T.lutLen[k];
        if (len == 0) return 0;
        const int at = imin(imax(bits, 0), len - 1);
        return T.lut[T.lutOff[k] + at];
    }
    NYQ_ED uint32_t pvqU(int n, int k) const {
        const uint32_t info = T.pvq[k];
        return n < (int)(info >> 16) ? T.pvq[kPvqInfo + (info & 0xffffu) + n] : 0xFFFFFFFFu;
    }

    // the bit allocation (rate.c:247-638; host: Allocator::run)
    NYQ_ED int rowBits(int r, int j, const int *trimOff) const {
        int v = K.row[r][j];
        if (v > 0) v = imax(0, v + trimOff[j]);
        return v;
    }
    NYQ_ED void allocate(int start, int end, const int *boost, int trim, int32_t total, BitPlan &P) {
        total = total > 0 ? total : 0;
        const int skipRsv = total >= kOneBit ? kOneBit : 0;
        total -= skipRsv;
        int intensityRsv = 0, dualRsv = 0;
        if (C == 2) {
            intensityRsv = log2FracTab(end - start);
            if (intensityRsv > total) {
                intensityRsv = 0;
            } else {
                total -= intensityRsv;
                dualRsv = total >= kOneBit ? kOneBit : 0;
                total -= dualRsv;
            }
        }
        int trimOff[kBands];
        for (int j = start; j < end; j++) {
            trimOff[j] = K.trimUnit[j] * (trim - 5 - LM) * (end - j - 1) >> 6;
            if (K.bins[j] == 1) trimOff[j] -= K.minBits;
        }
        int lo = 1, hi = kAllocRows - 1;
        do {
            const int mid = (lo + hi) >> 1;
            int32_t sum = 0;
            bool reached = false;
            for (int j = end; j-- > start;) {
                const int v = rowBits(mid, j, trimOff) + boost[j];
                if (v >= K.floorBits[j] || reached) {
                    reached = true;
                    sum += imin(v, K.cap[j]);
                } else if (v >= K.minBits) {
                    sum += K.minBits;
                }
            }
            if (sum > total) hi = mid - 1;
            else lo = mid + 1;
        } while (lo <= hi);
        hi = lo--;
        int base[kBands], span[kBands];
        int skipStart = start;
        for (int j = start; j < end; j++) {
            int b1 = rowBits(lo, j, trimOff);
            int b2 = hi >= kAllocRows ? K.cap[j] : rowBits(hi, j, trimOff);
            if (lo > 0) b1 += boost[j];
            b2 += boost[j];
            if (boost[j] > 0) skipStart = j;
            base[j] = b1;
            span[j] = imax(0, b2 - b1);
        }
        int wlo = 0, whi = 1 << 6;
        for (int step = 0; step < 6; step++) {
            const int mid = (wlo + whi) >> 1;
            int32_t sum = 0;
            bool reached = false;
            for (int j = end; j-- > start;) {
                const int v = base[j] + (int)(mid * (int32_t)span[j] >> 6);
                if (v >= K.floorBits[j] || reached) {
                    reached = true;
                    sum += imin(v, K.cap[j]);
                } else if (v >= K.minBits) {
                    sum += K.minBits;
                }
            }
            if (sum > total) whi = mid;
            else wlo = mid;
        }
        int *bits = P.shape;
        int32_t psum = 0;
        {
            bool reached = false;
            for (int j = end; j-- > start;) {
                int v = base[j] + (wlo * span[j] >> 6);
                if (v < K.floorBits[j] && !reached) v = v >= K.minBits ? K.minBits : 0;
                else reached = true;
                v = imin(v, K.cap[j]);
                bits[j] = v;
                psum += v;
            }
        }
        int coded = end;
        for (;; coded--) {
            const int j = coded - 1;
            if (j <= skipStart) {
                total += skipRsv;
                break;
            }
            int32_t left = total - psum;
            const int span0 = T.eBands[coded] - T.eBands[start];
            const int32_t perBin = left / span0;
            left -= span0 * perBin;
            const int rem = imax((int)left - (T.eBands[j] - T.eBands[start]), 0);
            int bandBits = (int)(bits[j] + perBin * K.width[j] + rem);
            if (bandBits >= imax(K.floorBits[j], K.minBits + kOneBit)) {
                if (rc.bitLogp(1)) break;
                psum += kOneBit;
                bandBits -= kOneBit;
            }
            psum -= bits[j] + intensityRsv;
            if (intensityRsv > 0) intensityRsv = log2FracTab(j - start);
            psum += intensityRsv;
            if (bandBits >= K.minBits) {
                psum += K.minBits;
                bits[j] = K.minBits;
            } else {
                bits[j] = 0;
            }
        }
        P.codedBands = coded;
        P.intensity = intensityRsv > 0 ? start + (int)rc.uint((uint32_t)(coded + 1 - start)) : 0;
        if (P.intensity <= start) {
            total += dualRsv;
            dualRsv = 0;
        }
        P.dualStereo = dualRsv > 0 ? rc.bitLogp(1) : 0;
        {
            int32_t left = total - psum;
            const int span0 = T.eBands[coded] - T.eBands[start];
            const int32_t perBin = left / span0;
            left -= span0 * perBin;
            for (int j = start; j < coded; j++) bits[j] += (int)perBin * K.width[j];
            for (int j = start; j < coded; j++) {
                const int t = (int)(left < K.width[j] ? left : K.width[j]);
                bits[j] += t;
                left -= t;
            }
        }
        const int stereo = C > 1;
        int32_t carry = 0;
        int j = start;
        for (; j < coded; j++) {
            const int Nb = K.bins[j];
            const int32_t have = (int32_t)bits[j] + carry;
            int32_t excess;
            if (Nb > 1) {
                excess = have - K.cap[j] > 0 ? have - K.cap[j] : 0;
                bits[j] = (int)(have - excess);
                const int den = C * Nb + ((C == 2 && Nb > 2 && !P.dualStereo && j < P.intensity) ? 1 : 0);
                const int nLogN = den * K.pulseCap[j];
                int offset = (nLogN >> 1) - den * kFineOffset;
                if (Nb == 2) offset += den << kBitRes >> 2;
                if (bits[j] + offset < den * 2 << kBitRes) offset += nLogN >> 2;
                else if (bits[j] + offset < den * 3 << kBitRes) offset += nLogN >> 3;
                int e = imax(0, (bits[j] + offset + (den << (kBitRes - 1))) / (den << kBitRes));
                if (C * e > (bits[j] >> kBitRes)) e = bits[j] >> stereo >> kBitRes;
                e = imin(e, kMaxFineBits);
                P.fine[j] = e;
                P.finePrio[j] = e * (den << kBitRes) >= bits[j] + offset;
                bits[j] -= C * e << kBitRes;
            } else {
                excess = have - (C << kBitRes) > 0 ? have - (C << kBitRes) : 0;
                bits[j] = (int)(have - excess);
                P.fine[j] = 0;
                P.finePrio[j] = 1;
            }
            if (excess > 0) {
                const int extra = imin((int)(excess >> (stereo + kBitRes)), kMaxFineBits - P.fine[j]);
                P.fine[j] += extra;
                const int extraBits = extra * C << kBitRes;
                P.finePrio[j] = extraBits >= excess - carry;
                excess -= extraBits;
            }
            carry = excess;
        }
        P.balance = carry;
        for (; j < end; j++) {
            P.fine[j] = bits[j] >> stereo >> kBitRes;
            bits[j] = 0;
            P.finePrio[j] = P.fine[j] < 1;
        }
    }

    // ---- angles (bands.c:661-832) ----
    struct Angle {
        int inv, imid, iside, delta, itheta, qalloc;
    };
    NYQ_ED int angleResolution(int n, int b, int offset, int pulseCap, bool stereo) const {
        const int16_t exp2Table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
        int N2 = 2 * n - 1;
        if (stereo && n == 2) N2--;
        int qb = imin(b - pulseCap - (4 << kBitRes), (b + N2 * offset) / N2);
        qb = imin(8 << kBitRes, qb);
        if (qb < (1 << kBitRes >> 1)) return 1;
        const int qn = exp2Table8[qb & 0x7] >> (14 - (qb >> kBitRes));
        return (qn + 1) >> 1 << 1;
    }
    NYQ_ED Angle readAngle(int n, int &b, int B, int B0, int lm, bool stereo, Img &fill) {
        const int pulseCap = T.logN[band] + lm * kOneBit;
        const int offset = (pulseCap >> 1) - (stereo && n == 2 ? 16 : 4);
        int qn = angleResolution(n, b, offset, pulseCap, stereo);
        if (stereo && band >= intensity) qn = 1;
        const int32_t before = (int32_t)rc.tellFrac();
        int itheta = 0, inv = 0;
        if (qn != 1) {
            if (stereo && n > 2) {
                const int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
                const int fs = (int)rc.decode((unsigned)ft);
                const int x = fs < (x0 + 1) * p0 ? fs / p0 : x0 + 1 + (fs - (x0 + 1) * p0);
                rc.update((unsigned)(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0), (unsigned)(x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0), (unsigned)ft);
                itheta = x;
            } else if (B0 > 1 || stereo) {
                itheta = (int)rc.uint((uint32_t)qn + 1);
            } else {
                const int h = qn >> 1, ft = (h + 1) * (h + 1);
                const int fm = (int)rc.decode((unsigned)ft);
                int fs, fl;
                if (fm < (h * (h + 1) >> 1)) {
                    itheta = (int)(floorSqrt(8 * (uint32_t)fm + 1) - 1) >> 1;
                    fs = itheta + 1;
                    fl = itheta * (itheta + 1) >> 1;
                } else {
                    itheta = (int)(2 * (qn + 1) - floorSqrt(8 * (uint32_t)(ft - fm - 1) + 1)) >> 1;
                    fs = qn + 1 - itheta;
                    fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
                }
                rc.update((unsigned)fl, (unsigned)(fl + fs), (unsigned)ft);
            }
            itheta = (int)((int32_t)itheta * 16384 / qn);
        } else if (stereo) {
            inv = (b > 2 << kBitRes && remaining > 2 << kBitRes) ? rc.bitLogp(2) : 0;
        }
        const int qalloc = (int)rc.tellFrac() - before;
        b -= qalloc;
        Angle a{inv, 0, 0, 0, itheta, qalloc};
        if (itheta == 0) {
            a.imid = 32767; a.iside = 0; a.delta = -16384;
            fill = imgAnd(fill, (1u << B) - 1);
        } else if (itheta == 16384) {
            a.imid = 0; a.iside = 32767; a.delta = 16384;
            fill = imgAnd(fill, ((1u << B) - 1) << B);
        } else {
            a.imid = bitexactCos((int16_t)itheta);
            a.iside = bitexactCos((int16_t)(16384 - itheta));
            a.delta = fracMul16((n - 1) << 7, bitexactLog2tan(a.iside, a.imid));
        }
        return a;
    }

    struct Node {
        int16_t off, n, foldOff;
        int b, B, lm;
        Img fill;
        int shift;
        float gain;
        bool deferred, mayGrow;
        int firstBits;
        int32_t remainingBefore;
    };

    // the split tree of one vector (bands.c:879-1055 as an explicit stack; host: BandShaper::readTree)
    NYQ_ED void readTree(int n0, int b0, int B0, bool hasFold, int lm0, float gain0, Img fill0, int x0) {
        Node stack[10];
        int sp = 0;
        Node nd{0, (int16_t)n0, (int16_t)(hasFold ? 0 : -1), b0, B0, lm0, fill0, 0, gain0, false, false, 0, 0};
        for (;;) {
            if (nd.deferred) {
                const int32_t surplus = nd.firstBits - (nd.remainingBefore - remaining);
                if (surplus > 3 << kBitRes && nd.mayGrow) nd.b += surplus - (3 << kBitRes);
            }
            const uint8_t *cache = cacheFor(band, nd.lm);
            if (nd.lm != -1 && nd.b > cache[cache[0]] + 12 && nd.n > 2) {
                const int half = nd.n >> 1, lm = nd.lm - 1, Bbefore = nd.B;
                Img fill = nd.fill;
                if (nd.B == 1) fill = imgOr(imgAnd(fill, 1), imgShl(fill, 1));
                const int B = (nd.B + 1) >> 1;
                int b = nd.b;
                const Angle a = readAngle(half, b, B, Bbefore, lm, false, fill);
                const float mid = (1.f / 32768) * a.imid, side = (1.f / 32768) * a.iside;
                int delta = a.delta;
                if (Bbefore > 1 && (a.itheta & 0x3fff)) {
                    if (a.itheta > 8192) delta -= delta >> (4 - lm);
                    else delta = imin(0, delta + (half << kBitRes >> (5 - lm)));
                }
                const int mbits = imax(0, imin(b, (b - delta) / 2)), sbits = b - mbits;
                remaining -= a.qalloc;
                Node lo{nd.off, (int16_t)half, nd.foldOff, mbits, B, lm, fill, nd.shift, nd.gain * mid, false, false, 0, 0};
                Node hi{(int16_t)(nd.off + half), (int16_t)half, (int16_t)(nd.foldOff >= 0 ? nd.foldOff + half : -1), sbits, B, lm, imgShr(fill, B),
                        nd.shift + (Bbefore >> 1), nd.gain * side, false, false, 0, 0};
                const bool midFirst = mbits >= sbits;
                Node &first = midFirst ? lo : hi, &second = midFirst ? hi : lo;
                second.deferred = true;
                second.firstBits = first.b;
                second.remainingBefore = remaining;
                second.mayGrow = midFirst ? a.itheta != 0 : a.itheta != 16384;
                if (sp < 10) stack[sp++] = second;
                else overflow |= 1;
                nd = first;
                continue;
            }
            int q = lut(band, nd.lm, nd.b);
            int cost = q ? cache[q] + 1 : 0;
            remaining -= cost;
            while (remaining < 0 && q > 0) {
                remaining += cost;
                q--;
                cost = q ? cache[q] + 1 : 0;
                remaining -= cost;
            }
            uint32_t index = 0;
            int Kp = 0;
            if (q != 0) {
                Kp = pulsesOf(q);
                index = rc.uint(pvqU(nd.n, Kp) + pvqU(nd.n, Kp + 1));
            }
            if (nleaves < leafCap) {
                RecLeaf &lf = leaves()[nleaves];
                lf.pad = 0;
                lf.off = nd.off;
                lf.n = nd.n;
                lf.blocks = (uint8_t)nd.B;
                lf.gain = nd.gain;
                lf.foldOff = nd.foldOff;
                lf.shift = (uint8_t)nd.shift;
                lf.abs = (int16_t)(x0 + nd.off);
                lf.pad2 = 0;
                lf.k = (int16_t)Kp;
                lf.index = index;
                Img img{0, 0};
                if (q != 0) {
                    lf.kind = 0;
                } else {
                    lf.kind = 1;
                    img = imgAnd(nd.fill, (1u << nd.B) - 1);
                }
                for (int i = 0; i < 4; i++) {
                    lf.img[i] = (uint16_t)(img.a >> (16 * i));
                    lf.img[4 + i] = (uint16_t)(img.b >> (16 * i));
                }
            } else {
                overflow |= 2;
            }
            nleaves++;
            if (sp == 0) break;
            nd = stack[--sp];
        }
    }

    NYQ_ED void emit(int kind, int a, int b, int n, float f0, float f1, int bnd = 0) {
        if (nops >= recMaxOps(C)) {
            overflow |= 4;
            return;
        }
        RecOp &o = ops()[nops++];
        o.kind = (uint8_t)kind; o.band = (uint8_t)bnd; o.a = (int16_t)a; o.b = (int16_t)b; o.n = (int16_t)n; o.f0 = f0; o.f1 = f1;
    }
    NYQ_ED void planSingles(int x, int y, int out, int sel) {
        for (int c = 0; c < 1 + (y >= 0); c++) {
            int sign = 0;
            if (remaining >= kOneBit) {
                sign = (int)rc.bits(1);
                remaining -= kOneBit;
            }
            emit(1, c ? y : x, c == 0 ? out : -1, sel, sign ? -1.f : 1.f, 0.f, band);
        }
    }
    NYQ_ED void planVector(int x, int n, int b, int B, int fold, int out, int sel, float gain, Img fill, int fillMode, int cmCh) {
        if (n == 1) {
            planSingles(x, -1, out, sel);
            return;
        }
        const int Bin = B;
        int recombine = tfChange > 0 ? tfChange : 0, tf = tfChange, timeDivide = 0, nb = n / B;
        for (int k = 0; k < recombine; k++) {
            const Img g = imgOr(fill, imgShr(fill, 1));
            fill = imgOr(imgOr(imgAnd(g, 1), imgAnd(imgShr(g, 1), 2)), imgOr(imgAnd(imgShr(g, 2), 4), imgAnd(imgShr(g, 3), 8)));
        }
        B >>= recombine;
        nb <<= recombine;
        while ((nb & 1) == 0 && tf < 0) {
            fill = imgOr(fill, imgShl(fill, B));
            B <<= 1;
            nb >>= 1;
            timeDivide++;
            tf++;
        }
        if (nvecs >= recMaxVecs(C)) {
            overflow |= 8;
            return;
        }
        RecVec &v = vecs()[nvecs];
        v.x = (int16_t)x; v.n = (int16_t)n; v.fold = (int16_t)fold; v.out = (int16_t)out; v.sel = (uint8_t)sel;
        v.recombine = (uint8_t)recombine; v.timeDivide = (uint8_t)timeDivide; v.Btree = (uint8_t)B; v.Bin = (uint8_t)Bin;
        v.nbTree = (int16_t)nb;
        v.band = (uint8_t)band; v.cmCh = (uint8_t)cmCh; v.fillMode = (uint8_t)fillMode; v.fillLo = (uint8_t)fillLo; v.fillHi = (uint8_t)fillHi;
        v.leaf0 = (int16_t)nleaves;
        readTree(n, b, B, fold >= 0, LM, gain, fill, x);
        v.leaf1 = (int16_t)nleaves;
        emit(0, nvecs++, 0, 0, 0.f, 0.f);
    }
    NYQ_ED void planStereo(int x, int y, int n, int b, int B, int fold, int out, Img fill, int fillMode) {
        if (n == 1) {
            planSingles(x, y, out, 0);
            return;
        }
        const Img fill0 = fill;
        const Angle a = readAngle(n, b, B, B, LM, true, fill);
        const float mid = (1.f / 32768) * a.imid, side = (1.f / 32768) * a.iside;
        const Img sideFill = imgShr(fill, B);
        if (n == 2) {
            int mbits = b, sbits = 0;
            if (a.itheta != 0 && a.itheta != 16384) sbits = kOneBit;
            mbits -= sbits;
            const int swap = a.itheta > 8192;
            remaining -= a.qalloc + sbits;
            const int sign = 1 - 2 * (sbits ? (int)rc.bits(1) : 0);
            planVector(swap ? y : x, n, mbits, B, fold, out, 0, 1.0f, fill0, fillMode, 3);
            emit(2, x, y, (sign < 0) | swap << 1, mid, side);
        } else {
            int mbits = imax(0, imin(b, (b - a.delta) / 2)), sbits = b - mbits;
            remaining -= a.qalloc;
            const int32_t before = remaining;
            if (mbits >= sbits) {
                planVector(x, n, mbits, B, fold, out, 0, 1.0f, fill, fillMode, 3);
                const int32_t surplus = mbits - (before - remaining);
                if (surplus > 3 << kBitRes && a.itheta != 0) sbits += surplus - (3 << kBitRes);
                planVector(y, n, sbits, B, -1, -1, 0, side, sideFill, fillMode, 3);
            } else {
                planVector(y, n, sbits, B, -1, -1, 0, side, sideFill, fillMode, 3);
                const int32_t surplus = sbits - (before - remaining);
                if (surplus > 3 << kBitRes && a.itheta != 16384) mbits += surplus - (3 << kBitRes);
                planVector(x, n, mbits, B, fold, out, 0, 1.0f, fill, fillMode, 3);
            }
            emit(3, x, y, n, mid, 0.f);
        }
        if (a.inv) emit(4, y, 0, n, 0.f, 0.f);
    }
    // the band loop (bands.c:1355-1518; host: BandShaper::plan)
    NYQ_ED void plan(int start, int end, const BitPlan &P, int shortBlocks, const int *tfRes, int32_t totalBits) {
        const int M = 1 << LM, B = shortBlocks ? M : 1;
        const int16_t *edge = K.first;
        const int normOffset = edge[start];
        const bool stereoFrame = C == 2;
        int foldBand = 0;
        bool refresh = true;
        int32_t balance = P.balance;
        int dual = P.dualStereo;
        intensity = P.intensity;
        for (int i = start; i < end; i++) {
            band = i;
            const bool last = i == end - 1;
            const int x = edge[i], y = stereoFrame ? N + edge[i] : -1;
            const int n = edge[i + 1] - edge[i];
            const int32_t tell = (int32_t)rc.tellFrac();
            if (i != start) balance -= tell;
            remaining = totalBits - tell - 1;
            int b = 0;
            if (i <= P.codedBands - 1) {
                const int32_t share = balance / imin(3, P.codedBands - i);
                const int32_t want = P.shape[i] + share;
                b = imax(0, imin(16383, (int)(remaining + 1 < want ? remaining + 1 : want)));
            }
            if (edge[i] - n >= edge[start] && (refresh || foldBand == 0)) foldBand = i;
            tfChange = tfRes[i];
            int foldAt = -1;
            bool fromMasks = false;
            fillLo = fillHi = 0;
            if (foldBand != 0 && (spread != 3 || B > 1 || tfChange < 0)) {
                foldAt = imax(0, edge[foldBand] - normOffset - n);
                int f0 = foldBand;
                while (edge[--f0] > foldAt + normOffset) {}
                int f1 = foldBand - 1;
                while (edge[++f1] < foldAt + normOffset + n) {}
                fillLo = f0;
                fillHi = f1;
                fromMasks = true;
            }
            const Img fill = imgAnd(Img{0x0008000400020001ull, 0x0080004000200010ull}, (1u << B) - 1);
            if (dual && i == P.intensity) {
                dual = 0;
                emit(5, edge[i] - normOffset, 0, 0, 0.f, 0.f);
            }
            const int outAt = last ? -1 : edge[i] - normOffset;
            if (dual) {
                planVector(x, n, b / 2, B, foldAt, outAt, 0, 1.0f, fill, fromMasks ? 1 : 3, 1);
                planVector(y, n, b / 2, B, foldAt, outAt, 1, 1.0f, fill, fromMasks ? 2 : 3, 2);
            } else if (y >= 0) {
                planStereo(x, y, n, b, B, foldAt, outAt, fill, fromMasks ? 0 : 3);
            } else {
                planVector(x, n, b, B, foldAt, outAt, 0, 1.0f, fill, fromMasks ? 0 : 3, 3);
            }
            balance += P.shape[i] + tell;
            refresh = b > (n << kBitRes);
        }
    }
};

// One frame: its bytes -> a spread record in `rec` (slotBytes long), its info and its energy deltas.  start / end: the coded
// bands (the packet's bandwidth), C: channels the packet codes.  Nothing else is read or kept.
NYQ_ED void decode_frame(const EntropyTables &T, const uint8_t *data, int len, int LM, int C, int start, int end, uint8_t *rec,
                         int slotBytes, EntInfo &info, EntEnergy &ed) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
    const int M = 1 << LM, N = M * 120;
    const int effEnd = imin(end, kBands);
    int cap = (slotBytes - recLeavesOff(C) - 2 * kBands * 4) / (int)sizeof(RecLeaf);
    cap = imax(0, imin(cap, kRecMaxLeaves));
    Frame F(T, T.alloc[LM][C - 1], LM, C, N, rec, cap);
    Range &dec = F.rc;
    dec.init(data, (uint32_t)len);
    const EntAlloc &K = F.K;

    int32_t totalBits = len * 8, tell = dec.tell();
    const bool silence = tell >= totalBits ? true : tell == 1 ? dec.bitLogp(15) != 0 : false;
    if (silence) {
        tell = len * 8;
        dec.skipTo(tell);
    }
    info.pfPitch = 0;
    info.pfTapset = 0;
    info.pfGainIndex = 0;
    if (start == 0 && tell + 16 <= totalBits) {
        if (dec.bitLogp(1)) {
            const int octave = (int)dec.uint(6);
            info.pfPitch = (int16_t)((16 << octave) + (int)dec.bits(4 + (unsigned)octave) - 1);
            const int qg = (int)dec.bits(3);
            const uint8_t tapsetIcdf[3] = {2, 1, 0};
            if (dec.tell() + 2 <= totalBits) info.pfTapset = (uint8_t)dec.icdf(tapsetIcdf, 2);
            info.pfGainIndex = (uint8_t)(qg + 1);
        }
        tell = dec.tell();
    }
    int transient = 0;
    if (LM > 0 && tell + 3 <= totalBits) {
        transient = dec.bitLogp(3);
        tell = dec.tell();
    }
    const int intra = tell + 3 <= totalBits ? dec.bitLogp(3) : 0;

    // coarse energies: the residuals and the in-frame half of the predictor (quant_bands.c:427-489)
    for (int k = 0; k < 2 * kBands; k++) ed.prev[k] = ed.q[k] = ed.fine[k] = ed.last[k] = 0.f;
    {
        const float beta = intra ? 4915 / 32768.f : betaCoef(LM);
        const int32_t budget = (int32_t)len * 8;
        float prev[2] = {0.f, 0.f};
        for (int i = start; i < end; i++)
            for (int c = 0; c < C; c++) {
                const int32_t room = budget - dec.tell();
                int qi;
                if (room >= 15) {
                    const int pi = 2 * imin(i, 20);
                    qi = laplace(dec, (unsigned)energyModel(LM, intra, pi) << 7, energyModel(LM, intra, pi + 1) << 6);
                } else if (room >= 2) {
                    const uint8_t smallIcdf[3] = {2, 1, 0};
                    qi = dec.icdf(smallIcdf, 2);
                    qi = (qi >> 1) ^ -(qi & 1);
                } else {
                    qi = room >= 1 ? -dec.bitLogp(1) : -1;
                }
                const float q = (float)qi;
                ed.prev[c * kBands + i] = prev[c];
                ed.q[c * kBands + i] = q;
                prev[c] = prev[c] + q - beta * q;
            }
    }

    int tfRes[kBands];
    {
        uint32_t budget = (uint32_t)len * 8, t = (uint32_t)dec.tell();
        int logp = transient ? 2 : 4;
        const int selectRsv = LM > 0 && t + (uint32_t)logp + 1 <= budget;
        budget -= (uint32_t)selectRsv;
        int changed = 0, cur = 0;
        for (int i = start; i < end; i++) {
            if (t + (uint32_t)logp <= budget) {
                cur ^= dec.bitLogp((unsigned)logp);
                t = (uint32_t)dec.tell();
                changed |= cur;
            }
            tfRes[i] = cur;
            logp = transient ? 4 : 5;
        }
        int select = 0;
        if (selectRsv && tfSelect(LM, 4 * transient + changed) != tfSelect(LM, 4 * transient + 2 + changed)) select = dec.bitLogp(1);
        for (int i = start; i < end; i++) tfRes[i] = tfSelect(LM, 4 * transient + 2 * select + tfRes[i]);
    }
    tell = dec.tell();
    const uint8_t spreadIcdf[4] = {25, 23, 2, 0};
    const int spread = tell + 4 <= totalBits ? dec.icdf(spreadIcdf, 5) : 2;
    F.spread = spread;

    int boost[kBands];
    int32_t total8 = totalBits << kBitRes;
    {
        int logp = 6;
        int32_t t8 = (int32_t)dec.tellFrac();
        for (int i = start; i < end; i++) {
            const int quanta = K.boostQuantum[i];
            int loopLogp = logp, bst = 0;
            while (t8 + (loopLogp << kBitRes) < total8 && bst < K.cap[i]) {
                const int flag = dec.bitLogp((unsigned)loopLogp);
                t8 = (int32_t)dec.tellFrac();
                if (!flag) break;
                bst += quanta;
                total8 -= quanta;
                loopLogp = 1;
            }
            boost[i] = bst;
            if (bst > 0) logp = imax(2, logp - 1);
        }
        tell = t8;
    }
    const uint8_t trimIcdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
    const int trim = tell + (6 << kBitRes) <= total8 ? dec.icdf(trimIcdf, 7) : 5;
    int32_t bits = (((int32_t)len * 8) << kBitRes) - (int32_t)dec.tellFrac() - 1;
    const int antiCollapseRsv = transient && LM >= 2 && bits >= ((LM + 2) << kBitRes) ? kOneBit : 0;
    bits -= antiCollapseRsv;
    BitPlan plan;
    F.allocate(start, end, boost, trim, bits, plan);

    for (int i = start; i < end; i++) {                            // fine energy (quant_bands.c:491-510)
        const int fb = plan.fine[i];
        if (fb <= 0) continue;
        for (int c = 0; c < C; c++) {
            const int q2 = (int)dec.bits((unsigned)fb);
            ed.fine[c * kBands + i] = (q2 + .5f) * (float)(1 << (14 - fb)) * (1.f / 16384) - .5f;
        }
    }

    F.plan(start, end, plan, transient ? M : 0, tfRes, len * (8 << kBitRes) - antiCollapseRsv);
    const int antiCollapseOn = antiCollapseRsv > 0 ? (int)dec.bits(1) : 0;
    {                                                              // the bits that are left (quant_bands.c:512-540)
        int left = len * 8 - dec.tell();
        for (int prio = 0; prio < 2; prio++)
            for (int i = start; i < end && left >= C; i++) {
                if (plan.fine[i] >= kMaxFineBits || plan.finePrio[i] != prio) continue;
                for (int c = 0; c < C; c++) {
                    const int q2 = (int)dec.bits(1);
                    ed.last[c * kBands + i] = (q2 - .5f) * (float)(1 << (14 - plan.fine[i] - 1)) * (1.f / 16384);
                    left--;
                }
            }
    }

    RecHead *H = reinterpret_cast<RecHead *>(rec);
    const int levelOff = recLeavesOff(C) + (int)sizeof(RecLeaf) * imin(F.nleaves, cap);
    const bool tooLarge = F.overflow != 0 || F.nleaves > cap;
    H->seed = 0;                                                   // (the energy pass sets it: the previous frame's final range)
    H->nleaves = (uint16_t)(silence || tooLarge ? 0 : F.nleaves);
    H->nvecs = (uint16_t)(silence || tooLarge ? 0 : F.nvecs);
    H->nops = (uint16_t)(silence || tooLarge ? 0 : F.nops);
    H->flags = (uint8_t)(antiCollapseOn && !silence && !tooLarge ? 2 : 0);
    H->spread = (uint8_t)spread;
    H->start = (uint8_t)start;
    H->end = (uint8_t)effEnd;
    H->channels = (uint8_t)C;
    H->lm = (uint8_t)LM;
    H->reserved[0] = (uint32_t)kRecOpsOff | (uint32_t)recVecsOff(C) << 16;
    H->reserved[1] = (uint32_t)recLeavesOff(C) | (uint32_t)levelOff << 16;
    H->reserved[2] = H->reserved[3] = 0;
    if (antiCollapseOn && !silence && !tooLarge) {
        // what of the anti-collapse level the frame itself decides: the threshold per band (the energy pass adds the rest)
        float *level = reinterpret_cast<float *>(rec + levelOff);
        for (int c = 0; c < 2; c++)
            for (int i = 0; i < kBands; i++) {
                float th = 0.f;
                if (i >= start && i < end) {
                    const int depth = (1 + plan.shape[i]) / K.bins[i];
                    th = -.125f * (float)depth;                    // (its exponential is taken where the level is finished)
                }
                level[c * kBands + i] = th;
            }
    }
    info.rangeFinal = dec.rng;
    info.flags = (uint8_t)((transient ? kEntTransient : 0) | (silence ? kEntSilence : 0) | (intra ? kEntIntra : 0) | (antiCollapseOn ? kEntAntiCollapse : 0) |
                           ((dec.tell() > 8 * len || dec.error) ? kEntError : 0) | (tooLarge && !silence ? kEntTooLarge : 0));
    info.LM = (uint8_t)LM;
    info.C = (uint8_t)C;
    info.start = (uint8_t)start;
    info.end = (uint8_t)end;
    info.pad[0] = (uint8_t)F.overflow;
    info.pad[1] = info.pad[2] = 0;
}

// ---- the energy pass: one (channel, band) of one stream, frame after frame (host: decodeFrame's energy clauses) ----------------
// lane = c * 21 + band.  Two steps per frame because a mono-coded frame couples the two channels' lanes: `begin` needs the
// partner lane's state from before the frame, `finish` the partner's energy from after `begin`.
struct EnergyLane {
    float E, L1, L2;                   // oldBandE, oldLogE, oldLogE2 of the host decoder
};
#if defined(__HIPCC__)
NYQ_ED float exp2Ref(float x) { return (float)::exp(0.6931471805599453094 * (double)x); }
#else
NYQ_ED float exp2Ref(float x) { return (float)__builtin_exp(0.6931471805599453094 * (double)x); }
#endif

// returns the lane's energy after the frame's own contributions; *logGain / *level: what goes into the record (if it has them)
NYQ_ED float energy_begin(EnergyLane &s, const EnergyLane &partner, int lane, const EntInfo &f, float prev, float q, float fine, float last,
                          float eMean, float threshExp, int bins, float *logGain, float *level) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
    const int c = lane >= kBands, i = lane - c * kBands;
    const int C = f.C, LM = f.LM;
    const bool coded = c < C && i >= f.start && i < f.end;
    float E = s.E;
    if (C == 1 && c == 0) E = E > partner.E ? E : partner.E;
    if (coded) {
        const float coef = (f.flags & kEntIntra) ? 0.f : predCoef(LM);
        float e = E > -9.f ? E : -9.f;
        e = coef * e + prev + q;
        e += fine;
        e += last;
        E = e;
    }
    if (level && coded && (f.flags & kEntAntiCollapse)) {
        const float thresh = .5f * exp2Ref(threshExp);
        const float sqrt1 = 1.f / __builtin_sqrtf((float)bins);
        float p1 = s.L1, p2 = s.L2;
        if (C == 1) {
            p1 = p1 > partner.L1 ? p1 : partner.L1;
            p2 = p2 > partner.L2 ? p2 : partner.L2;
        }
        const float pm = p1 < p2 ? p1 : p2;
        const float ediff = E - pm > 0.f ? E - pm : 0.f;
        float r = 2.f * exp2Ref(-ediff);
        if (LM == 3) r *= 1.41421356f;
        *level = (thresh < r ? thresh : r) * sqrt1;
    }
    if ((f.flags & kEntSilence) && c < C) E = -28.f;
    if (logGain && coded) *logGain = E + eMean;
    return E;
}
NYQ_ED void energy_finish(EnergyLane &s, float E, float partnerE, int lane, const EntInfo &f) {
    const int c = lane >= kBands, i = lane - c * kBands;
    if (f.C == 1 && c == 1) E = partnerE;
    if (!(f.flags & kEntTransient)) {
        s.L2 = s.L1;
        s.L1 = E;
    } else {
        s.L1 = s.L1 < E ? s.L1 : E;
    }
    if (i < f.start || i >= f.end) {
        E = 0.f;
        s.L1 = s.L2 = -28.f;
    }
    s.E = E;
}

}  // namespace nyq_ent
