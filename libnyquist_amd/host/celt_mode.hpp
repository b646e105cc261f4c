// celt_mode.hpp -- the static 48 kHz / 960 CELT mode as the CPU entropy stage needs it.
//
// Host side of the batched decode path: everything up to freq[] stays on the
// CPU because it is bit-serial; this header holds the mode constants that stage reads.  The
// reference keeps them in generated tables (third_party/opus/celt/static_modes_float.h, modes.c);
// here the normative constants of the Opus specification (band edges, allocation matrix) are
// stated once and everything derived (logN, the pulse cache, the caps, the PVQ codebook sizes)
// is computed at start-up and checked against the reference's tables by tests/test_host_decoder.py.
#pragma once
#include <cstdint>
#include <vector>

namespace nyq_host {

constexpr int kBands = 21;            // nbEBands, static_modes_float.h:580
constexpr int kBitRes = 3;            // BITRES, entcode.h
constexpr int kMaxLM = 3;
constexpr int kShortMdct = 120;
constexpr int kAllocVectors = 11;     // modes.c:47
constexpr int kMaxFineBits = 8;       // rate.h:37
constexpr int kFineOffset = 21;       // rate.h:39
constexpr int kQThetaOffset = 4;      // rate.h:40
constexpr int kQThetaOffsetTwoPhase = 16;
constexpr int kMaxPseudo = 40;        // rate.h:32
constexpr int kLogMaxPseudo = 6;

struct CeltMode {
    int16_t eBands[kBands + 1];                 // band edges in units of 2.5 ms bins (modes.c:41-44)
    int16_t logN[kBands];                       // log2 of the band width in 1/8 bit (modes.c:384-389)
    uint8_t alloc[kAllocVectors * kBands];      // bit allocation matrix, 1/32 bit per sample (modes.c:49-62)
    std::vector<int16_t> cacheIndex;            // [(LM+2)][kBands]  (rate.c:73-130)
    std::vector<uint8_t> cacheBits;
    std::vector<uint8_t> cacheCaps;             // [(LM+1)*2][kBands]
    float eMeans[25];                           // quant_bands.c:53-60

    // number of pulses that pseudo-pulse index i stands for (rate.h:48-51)
    static int pulsesOf(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }
    const uint8_t *cacheFor(int band, int LM) const { return cacheBits.data() + cacheIndex[(LM + 1) * kBands + band]; }
    int bits2pulses(int band, int LM, int bits) const;     // rate.h:53-79
    int pulses2bits(int band, int LM, int pulses) const;   // rate.h:81-88
    void initCaps(int *cap, int LM, int C) const;          // celt.c:182-191
};

const CeltMode &mode48k();

// PVQ codebook combinatorics (cwrs.c): U(n,k) with U(0,0)=1, U(n,k)=U(n-1,k)+U(n,k-1)+U(n-1,k-1);
// V(n,k) = U(n,k) + U(n,k+1) codewords of k pulses in n dimensions.  Saturating 64-bit.
uint64_t pvqU(int n, int k);
uint64_t pvqV(int n, int k);
constexpr int kPvqTableDim = 178;      // widest band at LM 3 is 176 bins
const uint64_t *pvqTable();            // U(n,k) at [n * kPvqTableDim + k], symmetric in (n,k)
// The same numbers clamped to 32 bits (0xFFFFFFFF = "2^32 or more"): every codebook a CELT frame can name has fewer than
// 2^32 codewords (ec_dec_uint's limit), so an index never reaches a clamped entry's value and the decoder's comparisons
// against this table come out as against the exact one -- at half the cache footprint and with 32-bit arithmetic.
const uint32_t *pvqTable32();
// conservative integer log2 with `frac` fractional bits (cwrs.c:45-71)
int log2Frac(uint32_t val, int frac);
inline int ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }   // EC_ILOG
unsigned isqrt32(uint32_t v);                                            // mathops.c:42-65

}  // namespace nyq_host
