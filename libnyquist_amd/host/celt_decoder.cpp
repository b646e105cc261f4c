// celt_decoder.cpp -- see celt_decoder.hpp.  The CPU stage of the batched decode path: one CELT frame's bitstream in,
// freq[] and the GPU stages' scalars out (RFC 6716 section 4.3; float build semantics of the reference).
//
// Organisation (this decoder's own, built for the batch pipeline -- the bitstream semantics are the specification's):
//   * AllocConst -- everything of the bit allocation that depends only on (frame size, channels) is tabulated once:
//     band widths, the eleven allocation rows already scaled to 1/8 bits, thresholds, trim slopes, caps;
//   * BitPlan    -- the frame's bit allocation from those tables (rows bisection, interpolation, band skipping,
//     PVQ / fine-energy split), no per-band recomputation of mode arithmetic;
//   * BandShaper -- a frame's band shapes in THREE phases.  1, symbols: every split tree is walked with an explicit stack (no
//     recursion) reading angles and the CODEWORDS of the pulse vectors into a flat program (vectors / leaves / operations,
//     integers only); the fill masks, which depend on pulse vectors not yet unranked, are carried as their images under each
//     input bit.  1b, resolve(): pulse vectors from their codewords, collapse masks, what becomes of the leaves without pulses.
//     2, build(): the program's floats by the vector kernels of celt_synth.hpp (spreading rotations of sibling leaves in
//     lockstep; the fold source is prepared lazily).  decodeSymbols() stops after phase 1 and packs the program into a symbol
//     record (include/nyq_imdct.h): phases 1b and 2 then run on the GPU (csrc/nyq_shape_kernel.hpp);
//   * pulse vectors are unranked on 32-bit rows of the U(n, k) table;
//   * all working memory lives in the decoder object (no heap traffic per frame).
// Reference line numbers (third_party/opus/celt/) mark the clauses of the reference a block answers to.
#include "celt_decoder.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <vector>

#include "../../include/nyq_imdct.h"
#include "../csrc/nyq_entropy_core.hpp"
#include "celt_synth.hpp"

namespace nyq_host {

// ---- entropy-coded side information: constants of the Opus specification ----------------------
namespace {

// Laplace parameters of the coarse energy, {P(0), decay} in Q8 per band, [LM][intra] (RFC 6716 4.3.2.1; quant_bands.c:77-138)
const uint8_t kEnergyModel[4][2][42] = {
    {{72, 127, 65, 129, 66, 128, 65, 128, 64, 128, 62, 128, 64, 128, 64, 128, 92, 78, 92, 79, 92, 78, 90, 79, 116, 41, 115, 40, 114, 40, 132, 26, 132, 26, 145, 17, 161, 12, 176, 10, 177, 11},
     {24, 179, 48, 138, 54, 135, 54, 132, 53, 134, 56, 133, 55, 132, 55, 132, 61, 114, 70, 96, 74, 88, 75, 88, 87, 74, 89, 66, 91, 67, 100, 59, 108, 50, 120, 40, 122, 37, 97, 43, 78, 50}},
    {{83, 78, 84, 81, 88, 75, 86, 74, 87, 71, 90, 73, 93, 74, 93, 74, 109, 40, 114, 36, 117, 34, 117, 34, 143, 17, 145, 18, 146, 19, 162, 12, 165, 10, 178, 7, 189, 6, 190, 8, 177, 9},
     {23, 178, 54, 115, 63, 102, 66, 98, 69, 99, 74, 89, 71, 91, 73, 91, 78, 89, 86, 80, 92, 66, 93, 64, 102, 59, 103, 60, 104, 60, 117, 52, 123, 44, 138, 35, 133, 31, 97, 38, 77, 45}},
    {{61, 90, 93, 60, 105, 42, 107, 41, 110, 45, 116, 38, 113, 38, 112, 38, 124, 26, 132, 27, 136, 19, 140, 20, 155, 14, 159, 16, 158, 18, 170, 13, 177, 10, 187, 8, 192, 6, 175, 9, 159, 10},
     {21, 178, 59, 110, 71, 86, 75, 85, 84, 83, 91, 66, 88, 73, 87, 72, 92, 75, 98, 72, 105, 58, 107, 54, 115, 52, 114, 55, 112, 56, 129, 51, 132, 40, 150, 33, 140, 29, 98, 35, 77, 42}},
    {{42, 121, 96, 66, 108, 43, 111, 40, 117, 44, 123, 32, 120, 36, 119, 33, 127, 33, 134, 34, 139, 21, 147, 23, 152, 20, 158, 25, 154, 26, 166, 21, 173, 16, 184, 13, 184, 10, 150, 13, 139, 15},
     {22, 178, 63, 114, 74, 82, 84, 83, 92, 82, 103, 62, 96, 72, 96, 67, 101, 73, 107, 72, 113, 55, 118, 52, 125, 52, 118, 52, 117, 55, 135, 49, 137, 39, 157, 32, 145, 29, 97, 33, 77, 40}}};
const uint8_t kSmallEnergyIcdf[3] = {2, 1, 0};
const float kPredCoef[4] = {29440 / 32768.f, 26112 / 32768.f, 21248 / 32768.f, 16384 / 32768.f};
const float kBetaCoef[4] = {30147 / 32768.f, 22282 / 32768.f, 12124 / 32768.f, 6554 / 32768.f};
const float kBetaIntra = 4915 / 32768.f;
const int8_t kTfSelect[4][8] = {{0, -1, 0, -1, 0, -1, 0, -1}, {0, -1, 0, -2, 1, 0, 1, -1}, {0, -2, 0, -3, 2, 0, 1, -1}, {0, -2, 0, -3, 3, 0, 1, -1}};
const uint8_t kTrimIcdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
const uint8_t kSpreadIcdf[4] = {25, 23, 2, 0};
const uint8_t kTapsetIcdf[3] = {2, 1, 0};
const uint8_t kLog2Frac[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};
constexpr int kSpreadNone = 0, kSpreadNormal = 2, kSpreadAggressive = 3;
constexpr int kOneBit = 1 << kBitRes;                       // one bit in the allocator's 1/8-bit units

inline float exp2Ref(float x) { return (float)std::exp(0.6931471805599453094 * (double)x); }   // celt_exp2, float build
inline uint32_t lcg(uint32_t s) { return 1664525u * s + 1013904223u; }
inline int fracMul16(int a, int b) { return (16384 + (int32_t)(int16_t)a * (int16_t)b) >> 15; }

}  // namespace

int laplaceDecode(RangeDecoder &dec, unsigned fs, int decay) {
    // RFC 6716 4.3.2.1 (laplace.c:92-134): geometric tails around a centre of probability fs, a flat floor of 1/32768
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
            fl += 2 * di;
        }
        if (fm < fl + fs) val = -val;
        else fl += fs;
    }
    dec.update(fl, std::min(fl + fs, 32768u), 32768);
    return val;
}

namespace {

// ---- what of the allocation depends on (LM, C) alone ------------------------------------------------------------------
struct AllocConst {
    int16_t width[kBands];                  // bins of the band in a 2.5 ms block
    int16_t bins[kBands];                   // ... in this frame size
    int16_t first[kBands + 1];              // first bin of the band in this frame size
    int16_t floorBits[kBands];              // below this a band gets nothing (or the bare minimum)
    int32_t trimUnit[kBands];               // trim offset = trimUnit * (trim - 5 - LM) * (end - 1 - band) >> 6
    int16_t row[kAllocVectors][kBands];     // the static allocation rows in 1/8 bits for this (LM, C)
    int16_t cap[kBands];
    int16_t pulseCap[kBands];               // logN + 8 LM
    int16_t boostQuantum[kBands];           // what one dynalloc boost step adds
    int minBits;                            // C << 3
};

const AllocConst &allocConst(int LM, int C) {
    static const AllocConst *tab = [] {
        static AllocConst t[4][2];
        const CeltMode &m = mode48k();
        for (int lm = 0; lm < 4; lm++)
            for (int c = 1; c <= 2; c++) {
                AllocConst &a = t[lm][c - 1];
                int caps[kBands];
                m.initCaps(caps, lm, c);
                a.minBits = c << kBitRes;
                for (int j = 0; j < kBands; j++) {
                    const int w = m.eBands[j + 1] - m.eBands[j];
                    a.width[j] = (int16_t)w;
                    a.bins[j] = (int16_t)(w << lm);
                    a.first[j] = (int16_t)(m.eBands[j] << lm);
                    a.floorBits[j] = (int16_t)std::max(c << kBitRes, (3 * w << lm << kBitRes) >> 4);
                    a.trimUnit[j] = c * w * (1 << (lm + kBitRes));
                    for (int r = 0; r < kAllocVectors; r++) a.row[r][j] = (int16_t)(c * w * m.alloc[r * kBands + j] << lm >> 2);
                    a.cap[j] = (int16_t)caps[j];
                    a.pulseCap[j] = (int16_t)(m.logN[j] + lm * kOneBit);
                    const int width = c * w << lm;
                    a.boostQuantum[j] = (int16_t)std::min(width << kBitRes, std::max(6 << kBitRes, width));
                }
                a.first[kBands] = (int16_t)(m.eBands[kBands] << lm);
            }
        return &t[0][0];
    }();
    return tab[LM * 2 + (C - 1)];
}

// ---- the frame's bit allocation (RFC 6716 4.3.3; rate.c:247-638) --------------------------------------------------------
struct BitPlan {
    int shape[kBands];          // 1/8 bits for the band's PVQ shape
    int fine[kBands];           // whole bits per channel of fine energy
    int finePrio[kBands];       // which bands get the left-over bits first
    int codedBands = 0;
    int intensity = 0;          // first band coded with intensity stereo
    int dualStereo = 0;
    int32_t balance = 0;        // bits carried into the band loop
};

struct Allocator {
    const AllocConst &K;
    const CeltMode &m;
    const int start, end, C, LM;
    RangeDecoder &rc;

    // the bits band j would get from `raw` (a row value, already trimmed and boosted), scanning from the top band down:
    // once a band reaches its floor every lower band is taken whole; a band under its floor gets the minimum or nothing
    template <class RawFn>
    int32_t demand(RawFn raw) const {
        int32_t sum = 0;
        bool reached = false;
        for (int j = end; j-- > start;) {
            const int v = raw(j);
            if (v >= K.floorBits[j] || reached) {
                reached = true;
                sum += std::min<int>(v, K.cap[j]);
            } else if (v >= K.minBits) {
                sum += K.minBits;
            }
        }
        return sum;
    }

    void run(const int *boost, int trim, int32_t total, BitPlan &P) const {
        total = std::max<int32_t>(total, 0);
        // reservations: the skip flag, the intensity parameter, the dual-stereo flag
        const int skipRsv = total >= kOneBit ? kOneBit : 0;
        total -= skipRsv;
        int intensityRsv = 0, dualRsv = 0;
        if (C == 2) {
            intensityRsv = kLog2Frac[end - start];
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
        auto rowBits = [&](int r, int j) {
            int v = K.row[r][j];
            if (v > 0) v = std::max(0, v + trimOff[j]);
            return v;
        };
        // the two neighbouring rows of the static table between which the budget falls
        int lo = 1, hi = kAllocVectors - 1;
        do {
            const int mid = (lo + hi) >> 1;
            if (demand([&](int j) { return rowBits(mid, j) + boost[j]; }) > total) hi = mid - 1;
            else lo = mid + 1;
        } while (lo <= hi);
        hi = lo--;
        int base[kBands], span[kBands];
        int skipStart = start;
        for (int j = start; j < end; j++) {
            int b1 = rowBits(lo, j);
            int b2 = hi >= kAllocVectors ? K.cap[j] : rowBits(hi, j);
            if (lo > 0) b1 += boost[j];
            b2 += boost[j];
            if (boost[j] > 0) skipStart = j;
            base[j] = b1;
            span[j] = std::max(0, b2 - b1);
        }
        // six bisection steps on the interpolation weight between the two rows
        int wlo = 0, whi = 1 << 6;
        for (int step = 0; step < 6; step++) {
            const int mid = (wlo + whi) >> 1;
            if (demand([&](int j) { return base[j] + (int)(mid * (int32_t)span[j] >> 6); }) > total) whi = mid;
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
                v = std::min<int>(v, K.cap[j]);
                bits[j] = v;
                psum += v;
            }
        }
        // band skipping from the top: a band that could be coded carries a flag saying whether it is
        int coded = end;
        for (;; coded--) {
            const int j = coded - 1;
            if (j <= skipStart) {
                total += skipRsv;
                break;
            }
            int32_t left = total - psum;
            const int span0 = m.eBands[coded] - m.eBands[start];
            const int32_t perBin = left / span0;
            left -= span0 * perBin;
            const int rem = std::max((int)left - (m.eBands[j] - m.eBands[start]), 0);
            int bandBits = (int)(bits[j] + perBin * K.width[j] + rem);
            if (bandBits >= std::max<int>(K.floorBits[j], K.minBits + kOneBit)) {
                if (rc.bitLogp(1)) break;
                psum += kOneBit;
                bandBits -= kOneBit;
            }
            psum -= bits[j] + intensityRsv;
            if (intensityRsv > 0) intensityRsv = kLog2Frac[j - start];
            psum += intensityRsv;
            if (bandBits >= K.minBits) {
                psum += K.minBits;
                bits[j] = K.minBits;
            } else {
                bits[j] = 0;
            }
        }
        P.codedBands = coded;
        P.intensity = intensityRsv > 0 ? start + (int)rc.uint(coded + 1 - start) : 0;
        if (P.intensity <= start) {
            total += dualRsv;
            dualRsv = 0;
        }
        P.dualStereo = dualRsv > 0 ? rc.bitLogp(1) : 0;
        // what is left goes to the coded bands in proportion to their width
        {
            int32_t left = total - psum;
            const int span0 = m.eBands[coded] - m.eBands[start];
            const int32_t perBin = left / span0;
            left -= span0 * perBin;
            for (int j = start; j < coded; j++) bits[j] += (int)perBin * K.width[j];
            for (int j = start; j < coded; j++) {
                const int t = (int)std::min<int32_t>(left, K.width[j]);
                bits[j] += t;
                left -= t;
            }
        }
        // every band's bits between its shape and the fine energy of its channels
        const int stereo = C > 1;
        int32_t carry = 0;
        int j = start;
        for (; j < coded; j++) {
            const int N = K.bins[j];
            const int32_t have = (int32_t)bits[j] + carry;
            int32_t excess;
            if (N > 1) {
                excess = std::max<int32_t>(have - K.cap[j], 0);
                bits[j] = (int)(have - excess);
                const int den = C * N + ((C == 2 && N > 2 && !P.dualStereo && j < P.intensity) ? 1 : 0);
                const int nLogN = den * K.pulseCap[j];
                int offset = (nLogN >> 1) - den * kFineOffset;
                if (N == 2) offset += den << kBitRes >> 2;
                if (bits[j] + offset < den * 2 << kBitRes) offset += nLogN >> 2;
                else if (bits[j] + offset < den * 3 << kBitRes) offset += nLogN >> 3;
                int e = std::max(0, (bits[j] + offset + (den << (kBitRes - 1))) / (den << kBitRes));
                if (C * e > (bits[j] >> kBitRes)) e = bits[j] >> stereo >> kBitRes;
                e = std::min(e, kMaxFineBits);
                P.fine[j] = e;
                P.finePrio[j] = e * (den << kBitRes) >= bits[j] + offset;
                bits[j] -= C * e << kBitRes;
            } else {
                excess = std::max<int32_t>(0, have - (C << kBitRes));
                bits[j] = (int)(have - excess);
                P.fine[j] = 0;
                P.finePrio[j] = 1;
            }
            if (excess > 0) {
                const int extra = (int)std::min<int32_t>(excess >> (stereo + kBitRes), kMaxFineBits - P.fine[j]);
                P.fine[j] += extra;
                const int extraBits = extra * C << kBitRes;
                P.finePrio[j] = extraBits >= excess - carry;
                excess -= extraBits;
            }
            carry = excess;
        }
        P.balance = carry;
        for (; j < end; j++) {                                    // skipped bands keep fine energy only
            P.fine[j] = bits[j] >> stereo >> kBitRes;
            bits[j] = 0;
            P.finePrio[j] = P.fine[j] < 1;
        }
    }
};

// ---- pulse vectors (RFC 6716 4.3.4.2; cwrs.c) -----------------------------------------------------------------------------
// Codeword `idx` of the k-pulse codebook in n dimensions -> y[0 .. n).  U(n, k) counts the vectors whose first coordinate is
// not negative ... the index space of one coordinate is [ positive values | zero | negative values ] in terms of row n of U:
// idx >= U(n, k + 1) means "negative" (and that many codewords are skipped), then the largest k' with U(n, k') <= idx is
// what is left for the remaining coordinates.  32-bit rows (pvqTable32: clamped entries are never reached by a valid index).
// Returns |y|^2 (what the gain normalisation needs, so that no second pass over y computes it).
int32_t unrankPulses(int n, int k, uint32_t idx, int16_t *y) {
    // U is symmetric, so U(n, k) is read as T[k][n]: while the pulse count stands still the two values a step needs --
    // U(n, k) and U(n, k + 1) -- come from two rows walked along n (sequential reads); a row changes only where a pulse is
    // found.  (Measured on the pulse vectors of a real stream, tools/scripts/pvq_unrank_bench.cpp: a branch-free lattice walk
    // and one with both possible next cells loaded ahead are 10-25 % SLOWER than this loop -- its branches predict well.)
    const uint32_t *T = pvqTable32();
    constexpr long D = kPvqTableDim;
    const uint32_t *here = T + (size_t)k * D;
    int32_t yy = 0;
    while (n > 2) {
        uint32_t a = here[D + n];
        const int neg = idx >= a;
        if (neg) idx -= a;
        a = here[n];
        if (a <= idx) {                                         // no pulse at this coordinate
            idx -= a;
            *y++ = 0;
        } else {
            int kk = k;
            const uint32_t *p = here + n;                       // p[-j * D] = U(n, k - j)
            if (kk > n && T[(size_t)n * D + n] > idx) {         // (many pulses: skip what cannot match)
                kk = n;
                p = T + (size_t)n * D + n;
            }
            do {
                kk--;
                p -= D;
            } while (*p > idx);
            idx -= *p;
            const int v = k - kk;
            *y++ = (int16_t)(neg ? -v : v);
            yy += v * v;
            k = kk;
            here = T + (size_t)k * D;
        }
        n--;
    }
    {                                                           // two coordinates left: closed forms
        const uint32_t a = 2 * (uint32_t)k + 1;
        const int neg = idx >= a;
        if (neg) idx -= a;
        const int kk = (int)((idx + 1) >> 1);
        if (kk) idx -= 2 * (uint32_t)kk - 1;
        const int v = k - kk;
        *y++ = (int16_t)(neg ? -v : v);
        *y = (int16_t)(idx ? -kk : kk);
        yy += v * v + kk * kk;
    }
    return yy;
}

// bits -> pseudo-pulse count of a (band, size) cache (rate.h bits2pulses: a six-step bisection and a rounding rule) as a direct
// table per cache, filled from that very function: the band loop asks this for every leaf of every frame
struct PulseLut {
    std::vector<uint8_t> q[(kMaxLM + 2) * kBands];
    PulseLut() {
        const CeltMode &m = mode48k();
        for (int l = -1; l <= kMaxLM; l++)
            for (int j = 0; j < kBands; j++) {
                if (m.cacheIndex[(size_t)((l + 1) * kBands + j)] < 0) continue;   // (a band too narrow to be split to this size)
                const uint8_t *cache = m.cacheFor(j, l);
                const int top = cache[cache[0]] + 3;             // beyond the costliest entry the answer no longer changes
                std::vector<uint8_t> &t = q[(l + 1) * kBands + j];
                t.resize((size_t)top + 1);
                for (int b = 0; b <= top; b++) t[(size_t)b] = (uint8_t)m.bits2pulses(j, l, b);
            }
    }
    int operator()(int band, int lm, int bits) const {
        const std::vector<uint8_t> &t = q[(lm + 1) * kBands + band];
        return t[(size_t)std::min<int>(std::max(bits, 0), (int)t.size() - 1)];
    }
};
const PulseLut &pulseLut() {
    static const PulseLut lut;
    return lut;
}

// ---- angles between two halves of a band (RFC 6716 4.3.4.3; bands.c:661-832) ---------------------------------------------------
int bitexactCos(int x) {
    const int32_t t = (4096 + (int32_t)x * x) >> 13;
    int x2 = (int16_t)t;
    x2 = (32767 - x2) + fracMul16(x2, (-7651 + fracMul16(x2, (8277 + fracMul16(-626, x2)))));
    return 1 + x2;
}
int bitexactLog2tan(int isin, int icos) {
    const int lc = ilog((uint32_t)icos), ls = ilog((uint32_t)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + fracMul16(isin, fracMul16(isin, -2597) + 7932) - fracMul16(icos, fracMul16(icos, -2597) + 7932);
}
// floor(sqrt(v)) for the triangular angle density (v < 2^18 there): the hardware square root and one correction step in place
// of the bit-serial isqrt32 (mathops.c:42-65) -- the same integer
inline uint32_t floorSqrt(uint32_t v) {
    uint32_t r = (uint32_t)std::sqrt((float)v);
    if (r * r > v) r--;
    else if ((r + 1) * (r + 1) <= v) r++;
    return r;
}

// itheta * 16384 / qn (bands.c:779) without the division: qn <= 256, itheta <= qn, so the product is below 2^22 and a 40-bit
// reciprocal rounded up gives the exact quotient (error < 2^22 / 2^40 < 1 / qn)
inline int scaleAngle(int itheta, int qn) {
    static const struct Recip {
        uint64_t m[257];
        Recip() {
            m[0] = 0;
            for (int q = 1; q <= 256; q++) m[q] = ((1ull << 40) + (uint64_t)q - 1) / (uint64_t)q;
        }
    } R;
    return (int)(((uint64_t)itheta * 16384u * R.m[qn]) >> 40);
}

int angleResolution(int N, int b, int offset, int pulseCap, bool stereo) {
    static const int16_t exp2Table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
    int N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    int qb = std::min(b - pulseCap - (4 << kBitRes), (b + N2 * offset) / N2);
    qb = std::min(8 << kBitRes, qb);
    if (qb < (1 << kBitRes >> 1)) return 1;
    const int qn = exp2Table8[qb & 0x7] >> (14 - (qb >> kBitRes));
    return (qn + 1) >> 1 << 1;
}

const int kHadamardOrder[] = {1, 0, 3, 0, 2, 1, 7, 0, 4, 3, 6, 1, 5, 2, 15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5};

}  // namespace

// ---- a frame's band shapes ---------------------------------------------------------------------------------------------------------
// One object per decode() call: the range decoder, the allocation and the running state of the band loop (bits left, the
// noise generator) plus views of the decoder's scratch memory.
struct CeltDecoder::BandShaper {
    const CeltMode &m;
    const AllocConst &K;
    RangeDecoder &rc;
    Scratch &S;
    const int LM, C, N;              // N: bins per channel
    const int spread, intensity;
    int32_t remaining = 0;           // bits the current band may still spend (1/8 bits)
    uint32_t seed = 0;
    int band = 0, tfChange = 0;

    // The collapse masks ("which short blocks of a band carry energy") steer the folding of later bands but never the
    // bitstream, and they depend on the pulse vectors -- which phase 1 does not unrank.  Phase 1 therefore carries, in place
    // of a band's `fill` mask, its IMAGE under every input bit: lane i of an Img is what the mask becomes when only bit i of
    // the band's initial mask is set (16 bits: a transient band whose time resolution is raised once more has 16 blocks).
    // Every step of the reference on `fill` (bands.c:1106-1123, 905-1003) is a shift, an AND with a constant or an OR, so
    // the images are transformed lane-wise by the same steps (one vector register) and a leaf's fill is the OR of the images
    // of the bits that are set -- evaluated later, by whoever knows the masks (resolve() below, or the GPU).
    typedef uint16_t Img __attribute__((vector_size(16)));
    static Img rep(unsigned m) {
        const uint16_t v = (uint16_t)m;
        return Img{v, v, v, v, v, v, v, v};
    }

    // ---- phase 1: symbols ------------------------------------------------------------------------------------------
    struct Angle {
        int inv, imid, iside, delta, itheta, qalloc;
    };
    Angle readAngle(int n, int &b, int B, int B0, int lm, bool stereo, Img &fill) {
        const int pulseCap = m.logN[band] + lm * kOneBit;
        const int offset = (pulseCap >> 1) - (stereo && n == 2 ? kQThetaOffsetTwoPhase : kQThetaOffset);
        int qn = angleResolution(n, b, offset, pulseCap, stereo);
        if (stereo && band >= intensity) qn = 1;
        const int32_t before = (int32_t)rc.tellFrac();
        int itheta = 0, inv = 0;
        if (qn != 1) {
            if (stereo && n > 2) {                                        // a step density: the low angles three times as likely
                const int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
                const int fs = (int)rc.decode(ft);
                const int x = fs < (x0 + 1) * p0 ? fs / p0 : x0 + 1 + (fs - (x0 + 1) * p0);
                rc.update(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
                itheta = x;
            } else if (B0 > 1 || stereo) {                                // uniform
                itheta = (int)rc.uint(qn + 1);
            } else {                                                      // triangular
                const int h = qn >> 1, ft = (h + 1) * (h + 1);
                const int fm = (int)rc.decode(ft);
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
                rc.update(fl, fl + fs, ft);
            }
            itheta = scaleAngle(itheta, qn);
        } else if (stereo) {
            inv = (b > 2 << kBitRes && remaining > 2 << kBitRes) ? rc.bitLogp(2) : 0;
        }
        const int qalloc = (int)rc.tellFrac() - before;
        b -= qalloc;
        Angle a{inv, 0, 0, 0, itheta, qalloc};
        if (itheta == 0) {
            a.imid = 32767; a.iside = 0; a.delta = -16384;
            fill &= rep((1u << B) - 1);
        } else if (itheta == 16384) {
            a.imid = 0; a.iside = 32767; a.delta = 16384;
            fill &= rep(((1u << B) - 1) << B);
        } else {
            a.imid = bitexactCos((int16_t)itheta);
            a.iside = bitexactCos((int16_t)(16384 - itheta));
            a.delta = fracMul16((n - 1) << 7, bitexactLog2tan(a.iside, a.imid));
        }
        return a;
    }

    // a leaf of the split tree (Scratch::LeafSlot): off / n = where in the vector and how many bins, k = pulses, blocks =
    // interleaved short blocks inside it, kind, gain, foldOff = offset of its source inside the band's fold source (-1: none),
    // index = the pulse vector's codeword, img = the images of its fill mask, shift = where its blocks sit in the band's mask.
    // Phase 1 knows two kinds -- kPulses and "no pulses" (kZero) --; resolve() turns the latter into kZero / kNoise / kFold.
    enum LeafKind : uint8_t { kPulses, kZero, kNoise, kFold };
    using Leaf = Scratch::LeafSlot;
    struct Node {                    // a pending partition of the split tree
        int16_t off, n, foldOff;
        int b, B, lm;
        Img fill;
        int shift;
        float gain;
        // second halves wait for what the first half really spent: b += max(0, surplus - 3 bits) unless the angle is degenerate
        bool deferred, mayGrow;
        int firstBits;
        int32_t remainingBefore;
    };

    // Walk the split tree of one vector (quant_partition's recursion, bands.c:879-1055, as an explicit stack) reading every
    // angle and every pulse vector's CODEWORD; leaves go to S.leaves[nleaves..] (x0: where the vector starts in X).
    void readTree(int n0, int b0, int B0, bool hasFold, int lm0, float gain0, Img fill0, int x0, int &nleaves) {
        Node stack[2 * (kMaxLM + 2)];
        const PulseLut &lut = pulseLut();
        const uint32_t *U32 = pvqTable32();
        int sp = 0;
        // (the node in hand stays in `nd`: of a split's halves only the second one goes through the stack)
        Node nd{0, (int16_t)n0, (int16_t)(hasFold ? 0 : -1), b0, B0, lm0, fill0, 0, gain0, false, false, 0, 0};
        for (;;) {
            if (nd.deferred) {
                const int32_t surplus = nd.firstBits - (nd.remainingBefore - remaining);
                if (surplus > 3 << kBitRes && nd.mayGrow) nd.b += surplus - (3 << kBitRes);
            }
            const uint8_t *cache = m.cacheFor(band, nd.lm);
            if (nd.lm != -1 && nd.b > cache[cache[0]] + 12 && nd.n > 2) {
                // split in two halves and an angle
                const int half = nd.n >> 1, lm = nd.lm - 1, Bbefore = nd.B;
                Img fill = nd.fill;
                if (nd.B == 1) fill = (fill & rep(1)) | (fill << 1);
                const int B = (nd.B + 1) >> 1;
                int b = nd.b;
                const Angle a = readAngle(half, b, B, Bbefore, lm, false, fill);
                const float mid = (1.f / 32768) * a.imid, side = (1.f / 32768) * a.iside;
                int delta = a.delta;
                if (Bbefore > 1 && (a.itheta & 0x3fff)) {
                    if (a.itheta > 8192) delta -= delta >> (4 - lm);
                    else delta = std::min(0, delta + (half << kBitRes >> (5 - lm)));
                }
                const int mbits = std::max(0, std::min(b, (b - delta) / 2)), sbits = b - mbits;
                remaining -= a.qalloc;
                Node lo{nd.off, (int16_t)half, nd.foldOff, mbits, B, lm, fill, nd.shift, nd.gain * mid, false, false, 0, 0};
                Node hi{(int16_t)(nd.off + half), (int16_t)half, (int16_t)(nd.foldOff >= 0 ? nd.foldOff + half : -1), sbits, B, lm, fill >> B,
                        nd.shift + (Bbefore >> 1), nd.gain * side, false, false, 0, 0};
                // the half with more bits goes first; the other one is pushed first (popped second) and inherits the surplus
                Node &first = mbits >= sbits ? lo : hi, &second = mbits >= sbits ? hi : lo;
                second.deferred = true;
                second.firstBits = first.b;
                second.remainingBefore = remaining;
                second.mayGrow = mbits >= sbits ? a.itheta != 0 : a.itheta != 16384;
                stack[sp++] = second;
                nd = first;
                continue;
            }
            // a leaf: as many pulses as its bits buy (fewer if the frame runs out)
            int q = lut(band, nd.lm, nd.b);
            int cost = q ? cache[q] + 1 : 0;
            remaining -= cost;
            while (remaining < 0 && q > 0) {
                remaining += cost;
                q--;
                cost = q ? cache[q] + 1 : 0;
                remaining -= cost;
            }
            const int leafIndex = nleaves++;
            Leaf &lf = S.leaves[leafIndex];
            lf.pad = 0;
            lf.off = nd.off;
            lf.n = nd.n;
            lf.blocks = (uint8_t)nd.B;
            lf.gain = nd.gain;
            lf.foldOff = nd.foldOff;
            lf.shift = (uint8_t)nd.shift;
            lf.abs = (int16_t)(x0 + nd.off);
            lf.pad2 = 0;
            if (q != 0) {
                const int K = CeltMode::pulsesOf(q);
                lf.kind = kPulses;
                lf.k = (int16_t)K;
                const uint32_t *urow = U32 + (size_t)nd.n * kPvqTableDim + K;          // codebook size V = U(n, K) + U(n, K + 1)
                lf.index = rc.uint(urow[0] + urow[1]);
                std::memset(lf.img, 0, sizeof lf.img);                // (a pulse leaf's fill is never asked for)
            } else {
                lf.kind = kZero;
                lf.k = 0;
                lf.index = 0;
                const Img img = nd.fill & rep((1u << nd.B) - 1);
                std::memcpy(lf.img, &img, sizeof img);                // (lane i = the image of input bit i)
            }
            if (sp == 0) break;
            nd = stack[--sp];
        }
    }

    // ---- what phase 1 leaves behind: a flat program for phase 2 (Scratch::vecs, ops, leaves, pulses) ----------------------
    // Phase 1 reads the WHOLE frame's symbols first (nothing it decides depends on a coefficient's value: fold sources are
    // named by offset, collapse masks come from the pulse vectors); phase 2 then builds every band in order.  The program is
    // plain data -- the same records could be executed by a GPU kernel instead of the loops below.
    using VecRec = Scratch::VecSlot;
    using Op = Scratch::OpSlot;
    enum OpKind : uint8_t { kOpVector, kOpSingle, kOpPair2, kOpMerge, kOpNegate, kOpAverage };
    // kOpVector: a = index of the vector | kOpSingle: X[a] = f0, copied to fold memory b of channel n if b >= 0 |
    // kOpPair2: bins a, b, mid f0, side f1, n = sign | swap << 1 | kOpMerge: a, b over n bins, mid f0 | kOpNegate: a over n |
    // kOpAverage: the two channels' fold memories over a bins
    void emit(OpKind kind, int a, int b, int n, float f0, float f1) {
        Op &o = S.ops[S.nops++];
        o.kind = kind; o.band = 0; o.a = (int16_t)a; o.b = (int16_t)b; o.n = (int16_t)n; o.f0 = f0; o.f1 = f1;
    }

    // One vector of one band (quant_band, bands.c:1060-1191), phase 1: the resolution changes' effect on the masks, then the
    // split tree.  x: offset of the vector in X; fold / out: offsets into the fold memory (or -1), sel: which channel's.
    // fillMode: where the vector's initial fill mask comes from (FillSource), cmCh: which channels' masks of the band its own
    // collapse mask is ORed into (1 = first, 2 = second, 3 = both)
    enum FillSource : uint8_t { kFillBoth, kFillFirst, kFillSecond, kFillAll };
    int fillLo = 0, fillHi = 0;      // the bands whose masks make up the current band's initial fill
    void planVector(int x, int n, int b, int B, int fold, int out, int sel, float gain, Img fill, int fillMode, int cmCh) {
        if (n == 1) {
            planSingles(x, -1, out, sel);
            return;
        }
        const int Bin = B;
        int recombine = tfChange > 0 ? tfChange : 0, tf = tfChange, timeDivide = 0, nb = n / B;
        // (bit_interleave_table: output bit j = input bits 2j | 2j + 1)
        for (int k = 0; k < recombine; k++) {
            const Img g = fill | (fill >> 1);
            fill = (g & rep(1)) | ((g >> 1) & rep(2)) | ((g >> 2) & rep(4)) | ((g >> 3) & rep(8));
        }
        B >>= recombine;
        nb <<= recombine;
        while ((nb & 1) == 0 && tf < 0) {
            fill |= fill << B;
            B <<= 1;
            nb >>= 1;
            timeDivide++;
            tf++;
        }
        VecRec &v = S.vecs[S.nvecs];
        v.x = (int16_t)x; v.n = (int16_t)n; v.fold = (int16_t)fold; v.out = (int16_t)out; v.sel = (uint8_t)sel;
        v.recombine = (uint8_t)recombine; v.timeDivide = (uint8_t)timeDivide; v.Btree = (uint8_t)B; v.Bin = (uint8_t)Bin;
        v.nbTree = (int16_t)nb;
        v.band = (uint8_t)band; v.cmCh = (uint8_t)cmCh; v.fillMode = (uint8_t)fillMode; v.fillLo = (uint8_t)fillLo; v.fillHi = (uint8_t)fillHi;
        v.leaf0 = (int16_t)S.nleaves;
        readTree(n, b, B, fold >= 0, LM, gain, fill, x, S.nleaves);
        v.leaf1 = (int16_t)S.nleaves;
        emit(kOpVector, S.nvecs++, 0, 0, 0.f, 0.f);
    }

    // bands of one bin: a sign per channel while bits last (bands.c:834-872)
    void planSingles(int x, int y, int out, int sel) {
        for (int c = 0; c < 1 + (y >= 0); c++) {
            int sign = 0;
            if (remaining >= kOneBit) {
                sign = (int)rc.bits(1);
                remaining -= kOneBit;
            }
            emit(kOpSingle, c ? y : x, c == 0 ? out : -1, sel, sign ? -1.f : 1.f, 0.f);
            S.ops[S.nops - 1].band = (uint8_t)band;                    // (its collapse mask is 1: resolve() / the GPU set it)
        }
    }

    // both channels of a band that is coded as mid / side around an angle (quant_band_stereo, bands.c:1194-1353)
    void planStereo(int x, int y, int n, int b, int B, int fold, int out, Img fill, int fillMode) {
        if (n == 1) {
            planSingles(x, y, out, 0);
            return;
        }
        const Img fill0 = fill;
        const Angle a = readAngle(n, b, B, B, LM, true, fill);
        const float mid = (1.f / 32768) * a.imid, side = (1.f / 32768) * a.iside;
        const Img sideFill = fill >> B;
        if (n == 2) {
            // two bins: the side is the mid turned by a quarter, its sign one raw bit
            int mbits = b, sbits = 0;
            if (a.itheta != 0 && a.itheta != 16384) sbits = kOneBit;
            mbits -= sbits;
            const int swap = a.itheta > 8192;
            remaining -= a.qalloc + sbits;
            const int sign = 1 - 2 * (sbits ? (int)rc.bits(1) : 0);
            planVector(swap ? y : x, n, mbits, B, fold, out, 0, 1.0f, fill0, fillMode, 3);
            emit(kOpPair2, x, y, (sign < 0) | swap << 1, mid, side);
        } else {
            int mbits = std::max(0, std::min(b, (b - a.delta) / 2)), sbits = b - mbits;
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
            emit(kOpMerge, x, y, n, mid, 0.f);
        }
        if (a.inv) emit(kOpNegate, y, 0, n, 0.f, 0.f);
    }

    // the band loop (quant_all_bands, bands.c:1355-1518), phase 1
    void plan(int start, int end, bool stereoFrame, const BitPlan &P, int shortBlocks, const int *tfRes, int32_t totalBits) {
        const int M = 1 << LM, B = shortBlocks ? M : 1;
        const int16_t *edge = K.first;
        const int normOffset = edge[start];
        int foldBand = 0;                        // the lowest band whose copy may serve as fold source ("lowband_offset")
        bool refresh = true;
        int32_t balance = P.balance;
        int dual = P.dualStereo;
        S.nops = S.nvecs = S.nleaves = 0;
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
                const int32_t share = balance / std::min(3, P.codedBands - i);
                b = std::max(0, std::min(16383, (int)std::min<int32_t>(remaining + 1, P.shape[i] + share)));
            }
            if (edge[i] - n >= edge[start] && (refresh || foldBand == 0)) foldBand = i;
            tfChange = tfRes[i];
            // where a band without pulses copies from: the n bins below `foldBand`, and which blocks of them carry energy
            int foldAt = -1;
            bool fromMasks = false;
            fillLo = fillHi = 0;
            if (foldBand != 0 && (spread != kSpreadAggressive || B > 1 || tfChange < 0)) {
                foldAt = std::max(0, edge[foldBand] - normOffset - n);
                int f0 = foldBand;
                while (edge[--f0] > foldAt + normOffset) {}
                int f1 = foldBand - 1;
                while (edge[++f1] < foldAt + normOffset + n) {}
                fillLo = f0;                                              // initial fill = the masks of bands [f0, f1) ORed
                fillHi = f1;
                fromMasks = true;
            }
            const Img fill = Img{1, 2, 4, 8, 16, 32, 64, 128} & rep((1u << B) - 1);
            if (dual && i == P.intensity) {                            // from here on the two channels fold from their mean
                dual = 0;
                emit(kOpAverage, edge[i] - normOffset, 0, 0, 0.f, 0.f);
            }
            const int outAt = last ? -1 : edge[i] - normOffset;
            if (dual) {
                planVector(x, n, b / 2, B, foldAt, outAt, 0, 1.0f, fill, fromMasks ? kFillFirst : kFillAll, 1);
                planVector(y, n, b / 2, B, foldAt, outAt, 1, 1.0f, fill, fromMasks ? kFillSecond : kFillAll, 2);
            } else if (y >= 0) {
                planStereo(x, y, n, b, B, foldAt, outAt, fill, fromMasks ? kFillBoth : kFillAll);
            } else {
                planVector(x, n, b, B, foldAt, outAt, 0, 1.0f, fill, fromMasks ? kFillBoth : kFillAll, 3);
            }
            balance += P.shape[i] + tell;
            refresh = b > (n << kBitRes);
        }
    }

    // ---- phase 1b: what the symbols imply without reading another bit -------------------------------------------------
    // Pulse vectors from their codewords, the collapse mask of every band (bands.c:1483-1511) and, from the masks of the
    // bands below, what becomes of the leaves without pulses (zeros / noise / a folded copy: bands.c:1005-1047).  The GPU
    // does the same from the record (nyq_shape_kernel.hpp); masks[c][band].
    void resolve(uint8_t masks[2][kBands]) {
        std::memset(masks, 0, 2 * kBands);
        for (int q = 0; q < S.nops; q++) {
            const Op &o = S.ops[q];
            if (o.kind == kOpSingle) {
                masks[0][o.band] |= 1;
                masks[1][o.band] |= 1;
                continue;
            }
            if (o.kind != kOpVector) continue;
            const VecRec &v = S.vecs[o.a];
            unsigned fill0 = 0;
            if (v.fillMode == kFillAll) {
                fill0 = (1u << v.Bin) - 1;
            } else {
                for (int f = v.fillLo; f < v.fillHi; f++)
                    fill0 |= v.fillMode == kFillFirst ? masks[0][f] : v.fillMode == kFillSecond ? masks[1][f] : (unsigned)(masks[0][f] | masks[1][f]);
            }
            unsigned cm = 0;
            for (int l = v.leaf0; l < v.leaf1; l++) {
                Leaf &lf = S.leaves[l];
                unsigned lcm;
                if (lf.kind == kPulses) {
                    int16_t *y = S.pulses + lf.abs;
                    S.leafEnergy[l] = unrankPulses(lf.n, lf.k, lf.index, y);
                    if (lf.blocks <= 1) {
                        lcm = 1;
                    } else {                                          // which of the interleaved blocks received a pulse
                        const int per = lf.n / lf.blocks;
                        lcm = 0;
                        for (int i = 0; i < lf.blocks; i++) {
                            int any = 0;
                            for (int j = 0; j < per; j++) any |= y[i * per + j];
                            lcm |= (unsigned)(any != 0) << i;
                        }
                    }
                } else {
                    unsigned fill = 0;
                    for (int i = 0; i < 8; i++)
                        if (fill0 >> i & 1) fill |= lf.img[i];
                    if (!fill) {
                        lf.kind = kZero;
                        lcm = 0;
                    } else if (lf.foldOff < 0) {
                        lf.kind = kNoise;
                        lcm = (1u << lf.blocks) - 1;
                    } else {
                        lf.kind = kFold;
                        lcm = fill;
                    }
                }
                cm |= lcm << lf.shift;
            }
            // the mask follows the band back through the resolution changes
            int B = v.Btree;
            for (int k = 0; k < v.timeDivide; k++) {
                B >>= 1;
                cm |= cm >> B;
            }
            static const uint8_t bitDeinterleave[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F, 0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
            for (int k = 0; k < v.recombine; k++) cm = bitDeinterleave[cm & 0xF];
            B <<= v.recombine;
            cm &= (1u << B) - 1;
            if (v.cmCh & 1) masks[0][v.band] |= (uint8_t)cm;
            if (v.cmCh & 2) masks[1][v.band] |= (uint8_t)cm;
        }
    }

    // ---- phase 2: coefficients ---------------------------------------------------------------------------------------
    // the spreading rotation's angle for (len, K): two cosines per distinct triple, remembered per thread
    static void spreadAngle(int len, int K, int spreadMode, float &c, float &s) {
        static const int factorOf[3] = {15, 10, 5};
        struct Memo { uint32_t key; float c, s; };
        static thread_local Memo memo[1024];
        const uint32_t key = ((uint32_t)len << 12) | ((uint32_t)K << 2) | (uint32_t)spreadMode;
        Memo &mm = memo[(key * 2654435761u) >> 22];
        if (mm.key != key) {
            const float gain = (float)(1.0f * len) / (float)(len + factorOf[spreadMode - 1] * K);
            const float theta = .5f * (gain * gain);
            mm.c = (float)std::cos((double)((.5f * 3.141592653f) * theta));
            mm.s = (float)std::cos((double)((.5f * 3.141592653f) * (1.0f - theta)));
            mm.key = key;
        }
        c = mm.c;
        s = mm.s;
    }

    // leaves [l0, l1) of vector x: pulses -> coefficients, then the spreading rotations (leaves of one shape side by side:
    // a rotation pass is a chain of dependent steps), then the leaves without pulses (zero / noise / folded copy of `fold`)
    void buildLeaves(float *x, const int16_t *pulses, const float *fold, int l0, int l1) {
        float *rx[32];
        float rc_[32], rs_[32];
        for (int l = l0; l < l1; l++) {
            const Leaf &lf = S.leaves[l];
            if (lf.kind == kPulses) synth::fromPulses(x + lf.off, pulses + lf.off, lf.n, lf.gain, S.leafEnergy[l]);
        }
        if (spread != kSpreadNone) {
            uint64_t done = 0;
            for (int l = l0; l < l1; l++) {
                const Leaf &a = S.leaves[l];
                if ((done >> (l - l0) & 1) || a.kind != kPulses || 2 * a.k >= a.n) continue;
                int cnt = 0;
                const int stride = a.blocks, len = a.n / stride;
                for (int q = l; q < l1 && cnt + stride <= 32; q++) {
                    const Leaf &bq = S.leaves[q];
                    if ((done >> (q - l0) & 1) || bq.kind != kPulses || bq.n != a.n || bq.blocks != a.blocks || 2 * bq.k >= bq.n) continue;
                    done |= 1ull << (q - l0);
                    float c, s;
                    spreadAngle(bq.n, bq.k, spread, c, s);
                    for (int i = 0; i < stride; i++) {                 // the leaf's interleaved blocks are chains of their own
                        rx[cnt] = x + bq.off + i * len;
                        rc_[cnt] = c;
                        rs_[cnt] = s;
                        cnt++;
                    }
                }
                int stride2 = 0;
                if (a.n >= 8 * stride) {
                    stride2 = 1;
                    while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < a.n) stride2++;
                }
                // decoder direction (vq.c:65-111 with dir < 0): the long-stride pass first, with (s, c); then stride 1 with (c, s)
                if (stride2) synth::rotateChains(rx, rs_, rc_, cnt, len, stride2);
                synth::rotateChains(rx, rc_, rs_, cnt, len, 1);
            }
        }
        for (int l = l0; l < l1; l++) {
            const Leaf &lf = S.leaves[l];
            float *o = x + lf.off;
            switch (lf.kind) {
            case kPulses: break;
            case kZero:
                std::memset(o, 0, sizeof(float) * (size_t)lf.n);
                break;
            case kNoise:
                for (int j = 0; j < lf.n; j++) {
                    seed = lcg(seed);
                    o[j] = (float)((int32_t)seed >> 20);
                }
                synth::renormalise(o, lf.n, lf.gain);
                break;
            case kFold:
                for (int j = 0; j < lf.n; j++) {
                    seed = lcg(seed);
                    o[j] = fold[lf.foldOff + j] + ((seed & 0x8000) ? 1.0f / 256 : -1.0f / 256);
                }
                synth::renormalise(o, lf.n, lf.gain);
                break;
            }
        }
    }

    static void regroup(float *X, float *tmp, int n0, int stride, bool hadamard, bool toBlocks) {
        // between "bin-major, blocks interleaved" and "block after block" (bands.c (de)interleave_hadamard)
        const int n = n0 * stride;
        const int *order = kHadamardOrder + stride - 2;
        for (int i = 0; i < stride; i++) {
            const int blk = hadamard ? order[i] : i;
            if (toBlocks)
                for (int j = 0; j < n0; j++) tmp[blk * n0 + j] = X[j * stride + i];
            else
                for (int j = 0; j < n0; j++) tmp[j * stride + i] = X[blk * n0 + j];
        }
        std::memcpy(X, tmp, sizeof(float) * (size_t)n);
    }

    // one vector: its fold source through the band's resolution changes (only if a leaf copies from it), the leaves, the
    // resolution changes undone on the band, the scaled copy for later bands
    void buildVector(const VecRec &v, float *X, float *norm, float *norm2) {
        float *x = X + v.x;
        const int n = v.n, recombine = v.recombine, timeDivide = v.timeDivide, Btree = v.Btree;
        const bool longBlocks = v.Bin == 1;
        const float *src = v.fold >= 0 ? (v.sel ? norm2 : norm) + v.fold : nullptr;
        bool folds = false;
        for (int l = v.leaf0; l < v.leaf1; l++) folds = folds || S.leaves[l].kind == kFold;
        if (folds && (recombine || timeDivide || Btree > 1)) {   // (the last band transforms its source in place in the reference: same values)
            float *w = S.foldWork;
            std::memcpy(w, src, sizeof(float) * (size_t)n);
            for (int k = 0; k < recombine; k++) synth::haar(w, n >> k, 1 << k);
            int bb = v.Bin >> recombine, nn = (n / v.Bin) << recombine;
            for (int k = 0; k < timeDivide; k++) {
                synth::haar(w, nn, bb);
                bb <<= 1;
                nn >>= 1;
            }
            if (Btree > 1) regroup(w, S.regroupTmp, v.nbTree >> recombine, Btree << recombine, longBlocks, true);
            src = w;
        }
        buildLeaves(x, S.pulses + v.x, src, v.leaf0, v.leaf1);
        if (Btree > 1) regroup(x, S.regroupTmp, v.nbTree >> recombine, Btree << recombine, longBlocks, false);
        int B = Btree, nb = v.nbTree;
        for (int k = 0; k < timeDivide; k++) {
            B >>= 1;
            nb <<= 1;
            synth::haar(x, nb, B);
        }
        for (int k = 0; k < recombine; k++) synth::haar(x, n >> k, 1 << k);
        if (v.out >= 0) synth::scaleTo((v.sel ? norm2 : norm) + v.out, x, n, std::sqrt((float)n));   // what later bands fold from
    }

    void build(float *X, int start, uint32_t *seedInOut) {
        float *norm = S.norm, *norm2 = S.norm + (K.first[kBands - 1] - K.first[start]);
        seed = *seedInOut;
        for (int q = 0; q < S.nops; q++) {
            const Op &o = S.ops[q];
            switch ((OpKind)o.kind) {
            case kOpVector: buildVector(S.vecs[o.a], X, norm, norm2); break;
            case kOpSingle:
                X[o.a] = o.f0;
                if (o.b >= 0) (o.n ? norm2 : norm)[o.b] = o.f0;
                break;
            case kOpPair2: {
                float *x = X + o.a, *y = X + o.b;
                const int sign = (o.n & 1) ? -1 : 1;
                float *x2 = (o.n & 2) ? y : x, *y2 = (o.n & 2) ? x : y;
                y2[0] = -sign * x2[1];
                y2[1] = sign * x2[0];
                const float x0 = o.f0 * x[0], x1 = o.f0 * x[1], y0 = o.f1 * y[0], y1 = o.f1 * y[1];
                x[0] = x0 - y0;
                y[0] = x0 + y0;
                x[1] = x1 - y1;
                y[1] = x1 + y1;
                break;
            }
            case kOpMerge: synth::stereoMerge(X + o.a, X + o.b, o.f0, o.n); break;
            case kOpNegate: synth::negate(X + o.a, o.n); break;
            case kOpAverage:
                for (int j = 0; j < o.a; j++) norm[j] = .5f * (norm[j] + norm2[j]);
                break;
            }
        }
        *seedInOut = seed;
    }
};

// ---- the frame decoder --------------------------------------------------------------------------
CeltDecoder::CeltDecoder(int channels) : m_(mode48k()), channels_(channels), streamChannels_(channels) { reset(); }

void CeltDecoder::reset() {
    rng_ = 0;
    for (int i = 0; i < 2 * kBands; i++) {
        oldBandE_[i] = 0.f;
        oldLogE_[i] = oldLogE2_[i] = -28.f;
        backgroundLogE_[i] = 0.f;
    }
}

int CeltDecoder::decode(const uint8_t *data, int len, int frameSize, float *freqOut, CeltFrame &info) {
    return decodeFrame(data, len, frameSize, freqOut, nullptr, info);
}

// layout of a symbol record (include/nyq_imdct.h), COMPACT: head | log_gain[42] | ops[nops] | vecs[nvecs] | leaves[nleaves] |
// (anti-collapse) level[42], rounded up to 16 bytes -- or head | freq[channels * 960] for a frame built on the host.
// A record never exceeds symbolBytes(channels) = the slot of the fixed-stride form.
namespace {
constexpr size_t kSymGainOff = sizeof(nyq_sym_head), kSymOpsOff = kSymGainOff + 2 * kBands * sizeof(float),
                 kSymFreqOff = sizeof(nyq_sym_head);
constexpr size_t kSymSlotFixed = 3072;                                    // slot = 3072 + channels * 3840 bytes
static_assert(sizeof(nyq_sym_head) == 32 && sizeof(nyq_sym_leaf) == 40 && sizeof(nyq_sym_vec) == 24 && sizeof(nyq_sym_op) == 16, "record layout");
static_assert(sizeof(CeltDecoder::Scratch::LeafSlot) == sizeof(nyq_sym_leaf) && sizeof(CeltDecoder::Scratch::VecSlot) == sizeof(nyq_sym_vec) &&
                  sizeof(CeltDecoder::Scratch::OpSlot) == sizeof(nyq_sym_op),
              "the decoder's scratch records travel to the GPU as they are");
static_assert(sizeof(CeltDecoder::Scratch::ops) / sizeof(nyq_sym_op) == NYQ_SYM_MAX_OPS && sizeof(CeltDecoder::Scratch::vecs) / sizeof(nyq_sym_vec) == NYQ_SYM_MAX_VECS, "record capacity");
}  // namespace

// (= nyq_celt_symbol_bytes_lm: an eighth of the 20 ms slot per LM step, never less than the fixed parts plus a leaf per vector)
size_t CeltDecoder::symbolBytes(int channels, int LM) {
    const size_t full = kSymSlotFixed + (size_t)channels * 960 * sizeof(float);
    const size_t floor_ = LM == 3 ? 0 : (size_t)2048 * (size_t)channels + 512;
    const size_t scaled = LM == 3 ? full : LM == 2 ? full * 5 / 8 : full >> (3 - LM);
    return (std::max(scaled, floor_) + 15) & ~(size_t)15;
}

int CeltDecoder::decodeSymbols(const uint8_t *data, int len, int frameSize, void *record, CeltFrame &info) {
    if (!record) return -1;
    uint8_t *r = static_cast<uint8_t *>(record);
    return decodeFrame(data, len, frameSize, reinterpret_cast<float *>(r + kSymFreqOff), r, info);
}

int CeltDecoder::decodeFrame(const uint8_t *data, int len, int frameSize, float *freqOut, uint8_t *record, CeltFrame &info) {
    const CeltMode &m = m_;
    const int CC = channels_, C = streamChannels_;
    // A mono decoder may be handed stereo-coded packets (the TOC's stereo flag is per packet): both channels are decoded and
    // mixed down (celt_decoder_clean.c:648-652), so the work buffer is max(CC, C) channels wide while the caller's holds CC.
    float *freq = C > CC ? wide_ : freqOut;
    int LM;
    for (LM = 0; LM <= kMaxLM; LM++)
        if ((kShortMdct << LM) == frameSize) break;
    if (LM > kMaxLM || len < 0 || len > 1275 || !data || !freqOut) return -1;
    const int M = 1 << LM, N = M * kShortMdct;
    const int start = start_, end = end_, effEnd = std::min(end, kBands);
    const AllocConst &K = allocConst(LM, C);
    float *E = oldBandE_;

    RangeDecoder dec;
    dec.init(data, (uint32_t)len);
    if (C == 1)
        for (int i = 0; i < kBands; i++) E[i] = std::max(E[i], E[kBands + i]);

    // ---- frame header: silence, post-filter, transient, intra (celt_decoder_clean.c:462-520) ----
    int32_t totalBits = len * 8, tell = dec.tell();
    const bool silence = tell >= totalBits ? true : tell == 1 ? dec.bitLogp(15) != 0 : false;
    if (silence) {
        tell = len * 8;
        dec.skipTo(tell);
    }
    info.pfGain = 0.f;
    info.pfPitch = info.pfTapset = 0;
    if (start == 0 && tell + 16 <= totalBits) {
        if (dec.bitLogp(1)) {
            const int octave = (int)dec.uint(6);
            info.pfPitch = (16 << octave) + (int)dec.bits(4 + octave) - 1;
            const int qg = (int)dec.bits(3);
            if (dec.tell() + 2 <= totalBits) info.pfTapset = dec.icdf(kTapsetIcdf, 2);
            info.pfGain = .09375f * (qg + 1);
        }
        tell = dec.tell();
    }
    int transient = 0;
    if (LM > 0 && tell + 3 <= totalBits) {
        transient = dec.bitLogp(3);
        tell = dec.tell();
    }
    const int intra = tell + 3 <= totalBits ? dec.bitLogp(3) : 0;

    // ---- coarse band energies: Laplace-coded residual of a two-way predictor (quant_bands.c:427-489) ----
    {
        const uint8_t *model = kEnergyModel[LM][intra];
        const float coef = intra ? 0.f : kPredCoef[LM], beta = intra ? kBetaIntra : kBetaCoef[LM];
        const int32_t budget = (int32_t)len * 8;
        float prev[2] = {0.f, 0.f};
        for (int i = start; i < end; i++)
            for (int c = 0; c < C; c++) {
                const int32_t room = budget - dec.tell();
                int qi;
                if (room >= 15) {
                    const int pi = 2 * std::min(i, 20);
                    qi = laplaceDecode(dec, model[pi] << 7, model[pi + 1] << 6);
                } else if (room >= 2) {
                    qi = dec.icdf(kSmallEnergyIcdf, 2);
                    qi = (qi >> 1) ^ -(qi & 1);
                } else {
                    qi = room >= 1 ? -dec.bitLogp(1) : -1;
                }
                const float q = (float)qi;
                float &e = E[i + c * kBands];
                e = std::max(-9.f, e);
                e = coef * e + prev[c] + q;
                prev[c] = prev[c] + q - beta * q;
            }
    }

    // ---- time-frequency resolution per band (celt_decoder_clean.c:314-351) ----
    int tfRes[kBands];
    {
        uint32_t budget = (uint32_t)len * 8, t = (uint32_t)dec.tell();
        int logp = transient ? 2 : 4;
        const int selectRsv = LM > 0 && t + logp + 1 <= budget;
        budget -= selectRsv;
        int changed = 0, cur = 0;
        for (int i = start; i < end; i++) {
            if (t + logp <= budget) {
                cur ^= dec.bitLogp(logp);
                t = (uint32_t)dec.tell();
                changed |= cur;
            }
            tfRes[i] = cur;
            logp = transient ? 4 : 5;
        }
        int select = 0;
        if (selectRsv && kTfSelect[LM][4 * transient + changed] != kTfSelect[LM][4 * transient + 2 + changed]) select = dec.bitLogp(1);
        for (int i = start; i < end; i++) tfRes[i] = kTfSelect[LM][4 * transient + 2 * select + tfRes[i]];
    }
    tell = dec.tell();
    const int spread = tell + 4 <= totalBits ? dec.icdf(kSpreadIcdf, 5) : kSpreadNormal;

    // ---- dynamic allocation boosts, trim, and the allocation itself ----
    int boost[kBands];
    int32_t total8 = totalBits << kBitRes;
    {
        int logp = 6;
        int32_t t8 = (int32_t)dec.tellFrac();
        for (int i = start; i < end; i++) {
            const int quanta = K.boostQuantum[i];
            int loopLogp = logp, bst = 0;
            while (t8 + (loopLogp << kBitRes) < total8 && bst < K.cap[i]) {
                const int flag = dec.bitLogp(loopLogp);
                t8 = (int32_t)dec.tellFrac();
                if (!flag) break;
                bst += quanta;
                total8 -= quanta;
                loopLogp = 1;
            }
            boost[i] = bst;
            if (bst > 0) logp = std::max(2, logp - 1);
        }
        tell = t8;
    }
    const int trim = tell + (6 << kBitRes) <= total8 ? dec.icdf(kTrimIcdf, 7) : 5;
    int32_t bits = (((int32_t)len * 8) << kBitRes) - (int32_t)dec.tellFrac() - 1;
    const int antiCollapseRsv = transient && LM >= 2 && bits >= ((LM + 2) << kBitRes) ? kOneBit : 0;
    bits -= antiCollapseRsv;
    BitPlan plan;
    Allocator{K, m, start, end, C, LM, dec}.run(boost, trim, bits, plan);

    // ---- fine energy (quant_bands.c:491-510) ----
    for (int i = start; i < end; i++) {
        const int fb = plan.fine[i];
        if (fb <= 0) continue;
        for (int c = 0; c < C; c++) {
            const int q2 = (int)dec.bits(fb);
            E[i + c * kBands] += (q2 + .5f) * (1 << (14 - fb)) * (1.f / 16384) - .5f;
        }
    }

    // ---- band shapes ----
    uint8_t masks[2][kBands];
    float *X = scratch_.X;
    BandShaper shaper{m, K, dec, scratch_, LM, C, N, spread, plan.intensity};
    shaper.plan(start, end, C == 2, plan, transient ? M : 0, tfRes, len * (8 << kBitRes) - antiCollapseRsv);   // phase 1: symbols
    const int antiCollapseOn = antiCollapseRsv > 0 ? (int)dec.bits(1) : 0;
    // phase 2 (floats) here, or -- symbol records -- on the GPU, unless the frame needs what the record does not carry
    // (room in the slot; 192 leaves is what the device stages)
    const size_t symBytes = kSymOpsOff + sizeof(nyq_sym_op) * (size_t)scratch_.nops + sizeof(nyq_sym_vec) * (size_t)scratch_.nvecs +
                            sizeof(nyq_sym_leaf) * (size_t)scratch_.nleaves + (antiCollapseOn ? 2 * kBands * sizeof(float) : 0);
    const bool asSymbols = record && !silence && scratch_.nleaves <= 192 && symBytes <= symbolBytes(CC, LM);
    if (!asSymbols) {
        std::memset(X, 0, sizeof(float) * (size_t)C * N);
        shaper.resolve(masks);                                            // phase 1b: pulse vectors, collapse masks, fill decisions
        shaper.build(X, start, &rng_);                                    // phase 2: floats
    }
    // ---- the bits that are left refine the energies once more, by priority (quant_bands.c:512-540) ----
    {
        int left = len * 8 - dec.tell();
        for (int prio = 0; prio < 2; prio++)
            for (int i = start; i < end && left >= C; i++) {
                if (plan.fine[i] >= kMaxFineBits || plan.finePrio[i] != prio) continue;
                for (int c = 0; c < C; c++) {
                    const int q2 = (int)dec.bits(1);
                    E[i + c * kBands] += (q2 - .5f) * (1 << (14 - plan.fine[i] - 1)) * (1.f / 16384);
                    left--;
                }
            }
    }
    // ---- anti-collapse: short blocks that received nothing get noise at the level of the quieter of the two previous
    // frames (bands.c:258-351) ----
    // the level of that noise per (channel, band): from the energies alone -- the GPU fills the blocks (it has the masks)
    float collapseLevel[2 * kBands];
    if (antiCollapseOn)
        for (int i = start; i < end; i++) {
            const int depth = (1 + plan.shape[i]) / K.bins[i];
            const float thresh = .5f * exp2Ref(-.125f * depth);
            const float sqrt1 = 1.f / std::sqrt((float)K.bins[i]);
            for (int c = 0; c < C; c++) {
                float p1 = oldLogE_[c * kBands + i], p2 = oldLogE2_[c * kBands + i];
                if (C == 1) {
                    p1 = std::max(p1, oldLogE_[kBands + i]);
                    p2 = std::max(p2, oldLogE2_[kBands + i]);
                }
                const float ediff = std::max(0.f, E[c * kBands + i] - std::min(p1, p2));
                float r = 2.f * exp2Ref(-ediff);
                if (LM == 3) r *= 1.41421356f;
                collapseLevel[c * kBands + i] = std::min(thresh, r) * sqrt1;
            }
        }
    if (antiCollapseOn && !asSymbols) {
        uint32_t seed = rng_;
        for (int i = start; i < end; i++) {
            const int n0 = K.width[i];
            for (int c = 0; c < C; c++) {
                const float r = collapseLevel[c * kBands + i];
                float *x = X + c * N + K.first[i];
                bool touched = false;
                for (int k = 0; k < M; k++) {
                    if (masks[c][i] & 1 << k) continue;
                    for (int j = 0; j < n0; j++) {
                        seed = lcg(seed);
                        x[(j << LM) + k] = (seed & 0x8000) ? r : -r;
                    }
                    touched = true;
                }
                if (touched) synth::renormalise(x, K.bins[i], 1.0f);
            }
        }
    }

    // ---- denormalisation: every band times 2^(energy + mean) (bands.c:192-256), silence, band limits, channel layout ----
    if (silence) {
        for (int i = 0; i < C * kBands; i++) E[i] = -28.f;
        if (record) {
            std::memset(record, 0, sizeof(nyq_sym_head));                 // (a record without operations is a silent frame)
            info.recordBytes = sizeof(nyq_sym_head);
        } else {
            std::memset(freq, 0, sizeof(float) * (size_t)std::max(CC, C) * N);
        }
    } else if (asSymbols) {
        nyq_sym_head *H = reinterpret_cast<nyq_sym_head *>(record);
        std::memset(H, 0, sizeof *H);
        H->seed = rng_;                                                   // the band loop starts from the previous frame's final range
        H->nleaves = (unsigned short)scratch_.nleaves;
        H->nvecs = (unsigned short)scratch_.nvecs;
        H->nops = (unsigned short)scratch_.nops;
        H->spread = (unsigned char)spread;
        H->start = (unsigned char)start;
        H->end = (unsigned char)effEnd;
        H->channels = (unsigned char)C;
        H->lm = (unsigned char)LM;
        float *gain = reinterpret_cast<float *>(record + kSymGainOff);
        for (int c = 0; c < C; c++)
            for (int i = start; i < effEnd; i++) gain[c * kBands + i] = E[i + c * kBands] + m.eMeans[i];   // (log2: the device raises 2 to it)
        uint8_t *w = record + kSymOpsOff;
        std::memcpy(w, scratch_.ops, sizeof(nyq_sym_op) * (size_t)scratch_.nops);
        w += sizeof(nyq_sym_op) * (size_t)scratch_.nops;
        std::memcpy(w, scratch_.vecs, sizeof(nyq_sym_vec) * (size_t)scratch_.nvecs);
        w += sizeof(nyq_sym_vec) * (size_t)scratch_.nvecs;
        std::memcpy(w, scratch_.leaves, sizeof(nyq_sym_leaf) * (size_t)scratch_.nleaves);
        w += sizeof(nyq_sym_leaf) * (size_t)scratch_.nleaves;
        if (antiCollapseOn) {
            H->flags |= NYQ_SYM_ANTI_COLLAPSE;
            std::memcpy(w, collapseLevel, sizeof collapseLevel);
            w += sizeof collapseLevel;
        }
        info.recordBytes = ((size_t)(w - record) + 15) & ~(size_t)15;
    } else {
        if (record) {
            nyq_sym_head *H = reinterpret_cast<nyq_sym_head *>(record);
            std::memset(H, 0, sizeof *H);
            H->flags = NYQ_SYM_HOST_FREQ;
            H->channels = (unsigned char)CC;
            H->lm = (unsigned char)LM;
            info.recordBytes = kSymFreqOff + sizeof(float) * (size_t)CC * N;
        }
        for (int c = 0; c < C; c++) {
            float *f = freq + c * N;
            std::memset(f, 0, sizeof(float) * (size_t)K.first[start]);
            for (int i = start; i < effEnd; i++)
                synth::scaleTo(f + K.first[i], X + c * N + K.first[i], K.bins[i], exp2Ref(E[i + c * kBands] + m.eMeans[i]));
            std::memset(f + K.first[effEnd], 0, sizeof(float) * (size_t)(N - K.first[effEnd]));
        }
    }
    if (!asSymbols && (!silence || !record)) {                           // (a symbol record leaves this to the device)
        if (CC == 2 && C == 1) std::memcpy(freq + N, freq, sizeof(float) * N);
        if (CC == 1 && C == 2)
            for (int i = 0; i < N; i++) freqOut[i] = .5f * (freq[i] + freq[N + i]);
    }

    // ---- what the next frame predicts from (celt_decoder_clean.c:685-718) ----
    if (C == 1) std::memcpy(E + kBands, E, sizeof(float) * kBands);
    if (!transient) {
        std::memcpy(oldLogE2_, oldLogE_, sizeof oldLogE_);
        std::memcpy(oldLogE_, E, sizeof oldBandE_);
        for (int i = 0; i < 2 * kBands; i++) backgroundLogE_[i] = std::min(backgroundLogE_[i] + M * 0.001f, E[i]);
    } else {
        for (int i = 0; i < 2 * kBands; i++) oldLogE_[i] = std::min(oldLogE_[i], E[i]);
    }
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < kBands; i++)
            if (i < start || i >= end) {
                E[c * kBands + i] = 0;
                oldLogE_[c * kBands + i] = oldLogE2_[c * kBands + i] = -28.f;
            }
    rng_ = dec.range();
    info.LM = LM;
    info.channels = CC;
    info.transient = transient != 0;
    info.silence = silence;
    info.rangeFinal = rng_;
    if (dec.tell() > 8 * len) return -3;
    if (dec.error()) return -4;
    return 0;
}

// ---- the tables of the frame-per-lane entropy stage (csrc/nyq_entropy_core.hpp), from this decoder's own ------------------------
size_t entropyTablesBytes() { return sizeof(nyq_ent::EntropyTables); }
void fillEntropyTables(void *out) {
    nyq_ent::EntropyTables &T = *new (out) nyq_ent::EntropyTables();
    std::memset(&T, 0, sizeof T);
    const CeltMode &m = mode48k();
    for (int lm = 0; lm < 4; lm++)
        for (int c = 1; c <= 2; c++) {
            const AllocConst &a = allocConst(lm, c);
            nyq_ent::EntAlloc &e = T.alloc[lm][c - 1];
            for (int j = 0; j < kBands; j++) {
                e.width[j] = a.width[j]; e.bins[j] = a.bins[j]; e.first[j] = a.first[j]; e.floorBits[j] = a.floorBits[j];
                e.trimUnit[j] = a.trimUnit[j]; e.cap[j] = a.cap[j]; e.pulseCap[j] = a.pulseCap[j]; e.boostQuantum[j] = a.boostQuantum[j];
                for (int r = 0; r < kAllocVectors; r++) e.row[r][j] = a.row[r][j];
            }
            e.first[kBands] = a.first[kBands];
            e.minBits = a.minBits;
        }
    for (int j = 0; j <= kBands; j++) T.eBands[j] = m.eBands[j];
    for (int j = 0; j < kBands; j++) {
        T.logN[j] = m.logN[j];
        T.eMeans[j] = m.eMeans[j];
    }
    if (m.cacheBits.size() > sizeof T.cacheBits) throw std::runtime_error("entropy tables: pulse cache larger than its place");
    std::memcpy(T.cacheBits, m.cacheBits.data(), m.cacheBits.size());
    const PulseLut &lut = pulseLut();
    size_t at = 0;
    for (int k = 0; k < 5 * kBands; k++) {
        T.cacheIndex[k] = m.cacheIndex[(size_t)k] < 0 ? 0 : m.cacheIndex[(size_t)k];
        const std::vector<uint8_t> &q = lut.q[k];
        if (at + q.size() > sizeof T.lut) throw std::runtime_error("entropy tables: pulse look-up larger than its place");
        T.lutOff[k] = (uint16_t)at;
        T.lutLen[k] = (uint16_t)q.size();
        if (!q.empty()) std::memcpy(T.lut + at, q.data(), q.size());
        at += q.size();
    }
    // the compact U(n, k) rows (csrc/nyq_shape_kernel.hpp pvq_table_build): row k = U(0 .. len - 1, k), every entry below 2^32
    const uint64_t *U = pvqTable();
    size_t off = 0;
    for (int k = 0; k < nyq_ent::kPvqInfo; k++) {
        int len = 0;
        while (k < kPvqTableDim && len < kPvqTableDim && U[(size_t)len * kPvqTableDim + k] < 0x100000000ull) len++;
        T.pvq[k] = (uint32_t)off | (uint32_t)len << 16;
        for (int n = 0; n < len; n++) {
            if (off >= (size_t)nyq_ent::kPvqWords) throw std::runtime_error("entropy tables: codebook table larger than its place");
            T.pvq[nyq_ent::kPvqInfo + off++] = (uint32_t)U[(size_t)n * kPvqTableDim + k];
        }
    }
}

}  // namespace nyq_host
