// celt_decoder.cpp -- see celt_decoder.hpp.  Decoder-only restatement, float build semantics
// (MULT16_16 = *, SHR/SHL = identity, Q15ONE = 1.0f, NORM_SCALING = 1.0f).  Reference line numbers
// refer to third_party/opus/celt/ of the reference tree.
#include "celt_decoder.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace nyq_host {

// ---- entropy-coded side information: constants of the Opus specification ----------------------
namespace {

// Laplace parameters of the coarse energy, {P(0), decay} in Q8 per band, [LM][intra] (quant_bands.c:77-138)
const uint8_t kEnergyModel[4][2][42] = {
    {{72, 127, 65, 129, 66, 128, 65, 128, 64, 128, 62, 128, 64, 128, 64, 128, 92, 78, 92, 79, 92, 78, 90, 79, 116, 41, 115, 40, 114, 40, 132, 26, 132, 26, 145, 17, 161, 12, 176, 10, 177, 11},
     {24, 179, 48, 138, 54, 135, 54, 132, 53, 134, 56, 133, 55, 132, 55, 132, 61, 114, 70, 96, 74, 88, 75, 88, 87, 74, 89, 66, 91, 67, 100, 59, 108, 50, 120, 40, 122, 37, 97, 43, 78, 50}},
    {{83, 78, 84, 81, 88, 75, 86, 74, 87, 71, 90, 73, 93, 74, 93, 74, 109, 40, 114, 36, 117, 34, 117, 34, 143, 17, 145, 18, 146, 19, 162, 12, 165, 10, 178, 7, 189, 6, 190, 8, 177, 9},
     {23, 178, 54, 115, 63, 102, 66, 98, 69, 99, 74, 89, 71, 91, 73, 91, 78, 89, 86, 80, 92, 66, 93, 64, 102, 59, 103, 60, 104, 60, 117, 52, 123, 44, 138, 35, 133, 31, 97, 38, 77, 45}},
    {{61, 90, 93, 60, 105, 42, 107, 41, 110, 45, 116, 38, 113, 38, 112, 38, 124, 26, 132, 27, 136, 19, 140, 20, 155, 14, 159, 16, 158, 18, 170, 13, 177, 10, 187, 8, 192, 6, 175, 9, 159, 10},
     {21, 178, 59, 110, 71, 86, 75, 85, 84, 83, 91, 66, 88, 73, 87, 72, 92, 75, 98, 72, 105, 58, 107, 54, 115, 52, 114, 55, 112, 56, 129, 51, 132, 40, 150, 33, 140, 29, 98, 35, 77, 42}},
    {{42, 121, 96, 66, 108, 43, 111, 40, 117, 44, 123, 32, 120, 36, 119, 33, 127, 33, 134, 34, 139, 21, 147, 23, 152, 20, 158, 25, 154, 26, 166, 21, 173, 16, 184, 13, 184, 10, 150, 13, 139, 15},
     {22, 178, 63, 114, 74, 82, 84, 83, 92, 82, 103, 62, 96, 72, 96, 67, 101, 73, 107, 72, 113, 55, 118, 52, 125, 52, 118, 52, 117, 55, 135, 49, 137, 39, 157, 32, 145, 29, 97, 33, 77, 40}}};
const uint8_t kSmallEnergyIcdf[3] = {2, 1, 0};                                   // quant_bands.c:140
const float kPredCoef[4] = {29440 / 32768.f, 26112 / 32768.f, 21248 / 32768.f, 16384 / 32768.f};   // :67
const float kBetaCoef[4] = {30147 / 32768.f, 22282 / 32768.f, 12124 / 32768.f, 6554 / 32768.f};    // :68
const float kBetaIntra = 4915 / 32768.f;                                         // :69
const int8_t kTfSelect[4][8] = {{0, -1, 0, -1, 0, -1, 0, -1},                    // celt.c:174-179
                                {0, -1, 0, -2, 1, 0, 1, -1},
                                {0, -2, 0, -3, 2, 0, 1, -1},
                                {0, -2, 0, -3, 3, 0, 1, -1}};
const uint8_t kTrimIcdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};      // celt.h:145
const uint8_t kSpreadIcdf[4] = {25, 23, 2, 0};                                   // celt.h:147
const uint8_t kTapsetIcdf[3] = {2, 1, 0};                                        // celt.h:149
const uint8_t kLog2Frac[24] = {0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37};   // rate.c:42-48
constexpr int kSpreadNone = 0, kSpreadNormal = 2, kSpreadAggressive = 3;

inline float exp2f_ref(float x) { return (float)std::exp(0.6931471805599453094 * (double)x); }   // mathops.h celt_exp2 (float)
inline uint32_t lcg(uint32_t s) { return 1664525u * s + 1013904223u; }                           // bands.c:61-64
inline int fracMul16(int a, int b) { return (16384 + (int32_t)(int16_t)a * (int16_t)b) >> 15; } // mathops.h:42

}  // namespace

int laplaceDecode(RangeDecoder &dec, unsigned fs, int decay) {
    int val = 0;
    unsigned fl = 0;
    const unsigned fm = dec.decodeBin(15);
    if (fm >= fs) {
        val++;
        fl = fs;
        fs = ((32768u - 32u - fs) * (uint32_t)(16384 - decay) >> 15) + 1;     // ec_laplace_get_freq1 + MINP
        while (fs > 1 && fm >= fl + 2 * fs) {
            fs *= 2;
            fl += fs;
            fs = ((fs - 2) * (uint32_t)decay) >> 15;
            fs += 1;
            val++;
        }
        if (fs <= 1) {                                                          // the flat tail
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

// ---- band energies (quant_bands.c:427-540) -----------------------------------------------------
void unquantCoarse(const CeltMode &m, int start, int end, float *oldE, int intra, RangeDecoder &dec, int C, int LM) {
    (void)m;
    const uint8_t *model = kEnergyModel[LM][intra];
    float prev[2] = {0.f, 0.f};
    const float coef = intra ? 0.f : kPredCoef[LM];
    const float beta = intra ? kBetaIntra : kBetaCoef[LM];
    const int32_t budget = (int32_t)dec.storageBytes() * 8;
    for (int i = start; i < end; i++)
        for (int c = 0; c < C; c++) {
            int qi;
            const int32_t tell = dec.tell();
            if (budget - tell >= 15) {
                const int pi = 2 * std::min(i, 20);
                qi = laplaceDecode(dec, model[pi] << 7, model[pi + 1] << 6);
            } else if (budget - tell >= 2) {
                qi = dec.icdf(kSmallEnergyIcdf, 2);
                qi = (qi >> 1) ^ -(qi & 1);
            } else if (budget - tell >= 1) {
                qi = -dec.bitLogp(1);
            } else {
                qi = -1;
            }
            const float q = (float)qi;
            float &e = oldE[i + c * kBands];
            e = std::max(-9.f, e);
            const float tmp = coef * e + prev[c] + q;
            e = tmp;
            prev[c] = prev[c] + q - beta * q;
        }
}

void unquantFine(int start, int end, float *oldE, const int *fineQuant, RangeDecoder &dec, int C) {
    for (int i = start; i < end; i++) {
        if (fineQuant[i] <= 0) continue;
        for (int c = 0; c < C; c++) {
            const int q2 = (int)dec.bits(fineQuant[i]);
            const float offset = (q2 + .5f) * (1 << (14 - fineQuant[i])) * (1.f / 16384) - .5f;
            oldE[i + c * kBands] += offset;
        }
    }
}

void unquantFinalise(int start, int end, float *oldE, const int *fineQuant, const int *finePriority, int bitsLeft,
                     RangeDecoder &dec, int C) {
    for (int prio = 0; prio < 2; prio++)
        for (int i = start; i < end && bitsLeft >= C; i++) {
            if (fineQuant[i] >= kMaxFineBits || finePriority[i] != prio) continue;
            for (int c = 0; c < C; c++) {
                const int q2 = (int)dec.bits(1);
                const float offset = (q2 - .5f) * (1 << (14 - fineQuant[i] - 1)) * (1.f / 16384);
                oldE[i + c * kBands] += offset;
                bitsLeft--;
            }
        }
}

// ---- time-frequency resolution flags (celt_decoder_clean.c:314-351) ------------------------------
void tfDecode(int start, int end, int isTransient, int *tfRes, int LM, RangeDecoder &dec) {
    uint32_t budget = dec.storageBytes() * 8;
    uint32_t tell = (uint32_t)dec.tell();
    int logp = isTransient ? 2 : 4;
    const int tfSelectRsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= tfSelectRsv;
    int tfChanged = 0, curr = 0;
    for (int i = start; i < end; i++) {
        if (tell + logp <= budget) {
            curr ^= dec.bitLogp(logp);
            tell = (uint32_t)dec.tell();
            tfChanged |= curr;
        }
        tfRes[i] = curr;
        logp = isTransient ? 4 : 5;
    }
    int tfSelect = 0;
    if (tfSelectRsv && kTfSelect[LM][4 * isTransient + 0 + tfChanged] != kTfSelect[LM][4 * isTransient + 2 + tfChanged])
        tfSelect = dec.bitLogp(1);
    for (int i = start; i < end; i++) tfRes[i] = kTfSelect[LM][4 * isTransient + 2 * tfSelect + tfRes[i]];
}

// ---- bit allocation (rate.c:247-638) ----------------------------------------------------------
int interpBits2Pulses(const CeltMode &m, int start, int end, int skipStart, const int *bits1, const int *bits2,
                      const int *thresh, const int *cap, int32_t total, int32_t *balanceOut, int skipRsv, int *intensity,
                      int intensityRsv, int *dualStereo, int dualStereoRsv, int *bits, int *ebits, int *finePriority,
                      int C, int LM, RangeDecoder &ec) {
    const int allocFloor = C << kBitRes;
    const int stereo = C > 1;
    const int logM = LM << kBitRes;
    int lo = 0, hi = 1 << 6;
    for (int i = 0; i < 6; i++) {                                   // ALLOC_STEPS
        const int mid = (lo + hi) >> 1;
        int32_t psum = 0;
        int done = 0;
        for (int j = end; j-- > start;) {
            const int tmp = bits1[j] + (int)(mid * (int32_t)bits2[j] >> 6);
            if (tmp >= thresh[j] || done) {
                done = 1;
                psum += std::min(tmp, cap[j]);
            } else if (tmp >= allocFloor) {
                psum += allocFloor;
            }
        }
        if (psum > total) hi = mid;
        else lo = mid;
    }
    int32_t psum = 0;
    int done = 0;
    for (int j = end; j-- > start;) {
        int tmp = bits1[j] + (lo * bits2[j] >> 6);
        if (tmp < thresh[j] && !done) tmp = tmp >= allocFloor ? allocFloor : 0;
        else done = 1;
        tmp = std::min(tmp, cap[j]);
        bits[j] = tmp;
        psum += tmp;
    }
    int codedBands;
    for (codedBands = end;; codedBands--) {                          // band skipping, from the top
        const int j = codedBands - 1;
        if (j <= skipStart) {
            total += skipRsv;
            break;
        }
        int32_t left = total - psum;
        const int32_t percoeff = left / (m.eBands[codedBands] - m.eBands[start]);
        left -= (m.eBands[codedBands] - m.eBands[start]) * percoeff;
        const int rem = std::max((int)left - (m.eBands[j] - m.eBands[start]), 0);
        const int bandWidth = m.eBands[codedBands] - m.eBands[j];
        int bandBits = (int)(bits[j] + percoeff * bandWidth + rem);
        if (bandBits >= std::max(thresh[j], allocFloor + (1 << kBitRes))) {
            if (ec.bitLogp(1)) break;
            psum += 1 << kBitRes;
            bandBits -= 1 << kBitRes;
        }
        psum -= bits[j] + intensityRsv;
        if (intensityRsv > 0) intensityRsv = kLog2Frac[j - start];
        psum += intensityRsv;
        if (bandBits >= allocFloor) {
            psum += allocFloor;
            bits[j] = allocFloor;
        } else {
            bits[j] = 0;
        }
    }
    if (intensityRsv > 0) *intensity = start + (int)ec.uint(codedBands + 1 - start);
    else *intensity = 0;
    if (*intensity <= start) {
        total += dualStereoRsv;
        dualStereoRsv = 0;
    }
    if (dualStereoRsv > 0) *dualStereo = ec.bitLogp(1);
    else *dualStereo = 0;

    int32_t left = total - psum;
    const int32_t percoeff = left / (m.eBands[codedBands] - m.eBands[start]);
    left -= (m.eBands[codedBands] - m.eBands[start]) * percoeff;
    for (int j = start; j < codedBands; j++) bits[j] += (int)percoeff * (m.eBands[j + 1] - m.eBands[j]);
    for (int j = start; j < codedBands; j++) {
        const int tmp = (int)std::min<int32_t>(left, m.eBands[j + 1] - m.eBands[j]);
        bits[j] += tmp;
        left -= tmp;
    }
    int32_t balance = 0;
    int j;
    for (j = start; j < codedBands; j++) {                            // split PVQ bits / fine energy bits
        const int N0 = m.eBands[j + 1] - m.eBands[j];
        const int N = N0 << LM;
        const int32_t bit = (int32_t)bits[j] + balance;
        int32_t excess;
        if (N > 1) {
            excess = std::max<int32_t>(bit - cap[j], 0);
            bits[j] = (int)(bit - excess);
            const int den = C * N + ((C == 2 && N > 2 && !*dualStereo && j < *intensity) ? 1 : 0);
            const int NClogN = den * (m.logN[j] + logM);
            int offset = (NClogN >> 1) - den * kFineOffset;
            if (N == 2) offset += den << kBitRes >> 2;
            if (bits[j] + offset < den * 2 << kBitRes) offset += NClogN >> 2;
            else if (bits[j] + offset < den * 3 << kBitRes) offset += NClogN >> 3;
            ebits[j] = std::max(0, (bits[j] + offset + (den << (kBitRes - 1))) / (den << kBitRes));
            if (C * ebits[j] > (bits[j] >> kBitRes)) ebits[j] = bits[j] >> stereo >> kBitRes;
            ebits[j] = std::min(ebits[j], kMaxFineBits);
            finePriority[j] = ebits[j] * (den << kBitRes) >= bits[j] + offset;
            bits[j] -= C * ebits[j] << kBitRes;
        } else {
            excess = std::max<int32_t>(0, bit - (C << kBitRes));
            bits[j] = (int)(bit - excess);
            ebits[j] = 0;
            finePriority[j] = 1;
        }
        if (excess > 0) {
            const int extraFine = (int)std::min<int32_t>(excess >> (stereo + kBitRes), kMaxFineBits - ebits[j]);
            ebits[j] += extraFine;
            const int extraBits = extraFine * C << kBitRes;
            finePriority[j] = extraBits >= excess - balance;
            excess -= extraBits;
        }
        balance = excess;
    }
    *balanceOut = balance;
    for (; j < end; j++) {                                            // skipped bands: fine energy only
        ebits[j] = bits[j] >> stereo >> kBitRes;
        bits[j] = 0;
        finePriority[j] = ebits[j] < 1;
    }
    return codedBands;
}

int computeAllocation(const CeltMode &m, int start, int end, const int *offsets, const int *cap, int allocTrim,
                      int *intensity, int *dualStereo, int32_t total, int32_t *balance, int *pulses, int *ebits,
                      int *finePriority, int C, int LM, RangeDecoder &ec) {
    total = std::max<int32_t>(total, 0);
    int skipStart = start;
    const int skipRsv = total >= 1 << kBitRes ? 1 << kBitRes : 0;
    total -= skipRsv;
    int intensityRsv = 0, dualStereoRsv = 0;
    if (C == 2) {
        intensityRsv = kLog2Frac[end - start];
        if (intensityRsv > total) {
            intensityRsv = 0;
        } else {
            total -= intensityRsv;
            dualStereoRsv = total >= 1 << kBitRes ? 1 << kBitRes : 0;
            total -= dualStereoRsv;
        }
    }
    int bits1[kBands], bits2[kBands], thresh[kBands], trimOffset[kBands];
    for (int j = start; j < end; j++) {
        const int w = m.eBands[j + 1] - m.eBands[j];
        thresh[j] = std::max(C << kBitRes, (3 * w << LM << kBitRes) >> 4);
        trimOffset[j] = C * w * (allocTrim - 5 - LM) * (end - j - 1) * (1 << (LM + kBitRes)) >> 6;
        if ((w << LM) == 1) trimOffset[j] -= C << kBitRes;
    }
    int lo = 1, hi = kAllocVectors - 1;
    do {
        int done = 0, psum = 0;
        const int mid = (lo + hi) >> 1;
        for (int j = end; j-- > start;) {
            const int N = m.eBands[j + 1] - m.eBands[j];
            int bitsj = C * N * m.alloc[mid * kBands + j] << LM >> 2;
            if (bitsj > 0) bitsj = std::max(0, bitsj + trimOffset[j]);
            bitsj += offsets[j];
            if (bitsj >= thresh[j] || done) {
                done = 1;
                psum += std::min(bitsj, cap[j]);
            } else if (bitsj >= C << kBitRes) {
                psum += C << kBitRes;
            }
        }
        if (psum > total) hi = mid - 1;
        else lo = mid + 1;
    } while (lo <= hi);
    hi = lo--;
    for (int j = start; j < end; j++) {
        const int N = m.eBands[j + 1] - m.eBands[j];
        int bits1j = C * N * m.alloc[lo * kBands + j] << LM >> 2;
        int bits2j = hi >= kAllocVectors ? cap[j] : C * N * m.alloc[hi * kBands + j] << LM >> 2;
        if (bits1j > 0) bits1j = std::max(0, bits1j + trimOffset[j]);
        if (bits2j > 0) bits2j = std::max(0, bits2j + trimOffset[j]);
        if (lo > 0) bits1j += offsets[j];
        bits2j += offsets[j];
        if (offsets[j] > 0) skipStart = j;
        bits2j = std::max(0, bits2j - bits1j);
        bits1[j] = bits1j;
        bits2[j] = bits2j;
    }
    return interpBits2Pulses(m, start, end, skipStart, bits1, bits2, thresh, cap, total, balance, skipRsv, intensity,
                             intensityRsv, dualStereo, dualStereoRsv, pulses, ebits, finePriority, C, LM, ec);
}

// ---- PVQ shape decoding (cwrs.c:469-540, vq.c) ----------------------------------------------------
void decodePulseVector(int n, int k, uint32_t idx, int *y) {             // cwrsi
    // U is symmetric, so every lookup of a step reads row n of the table: contiguous in k
    const uint64_t *T = pvqTable();
    uint64_t i = idx;
    while (n > 2) {
        const uint64_t *row = T + (size_t)n * kPvqTableDim;
        uint64_t p, q;
        int s, k0;
        if (k >= n) {                                                    // many pulses
            p = row[k + 1];
            s = -(i >= p);
            i -= p & (uint64_t)(int64_t)s;
            k0 = k;
            q = row[n];
            if (q > i) {
                k = n;
                do p = row[--k];
                while (p > i);
            } else {
                for (p = row[k]; p > i; p = row[k]) k--;
            }
            i -= p;
            *y++ = (k0 - k + s) ^ s;
        } else {                                                         // many dimensions
            p = row[k];
            q = row[k + 1];
            if (p <= i && i < q) {
                i -= p;
                *y++ = 0;
            } else {
                s = -(i >= q);
                i -= q & (uint64_t)(int64_t)s;
                k0 = k;
                do p = row[--k];
                while (p > i);
                i -= p;
                *y++ = (k0 - k + s) ^ s;
            }
        }
        n--;
    }
    uint64_t p = 2 * (uint64_t)k + 1;                                    // n == 2
    int s = -(i >= p);
    i -= p & (uint64_t)(int64_t)s;
    int k0 = k;
    k = (int)((i + 1) >> 1);
    if (k) i -= 2 * (uint64_t)k - 1;
    *y++ = (k0 - k + s) ^ s;
    s = -(int)i;                                                         // n == 1
    *y = (k + s) ^ s;
}

void expRotation1(float *X, int len, int stride, float c, float s) {       // vq.c:40-63
    float *p = X;
    for (int i = 0; i < len - stride; i++) {
        const float x1 = p[0], x2 = p[stride];
        p[stride] = c * x2 + s * x1;
        *p++ = c * x1 - s * x2;
    }
    p = &X[len - 2 * stride - 1];
    for (int i = len - 2 * stride - 1; i >= 0; i--) {
        const float x1 = p[0], x2 = p[stride];
        p[stride] = c * x2 + s * x1;
        *p-- = c * x1 - s * x2;
    }
}

void expRotation(float *X, int len, int dir, int stride, int K, int spread) {   // vq.c:65-111
    static const int factorOf[3] = {15, 10, 5};
    if (2 * K >= len || spread == kSpreadNone) return;
    const int factor = factorOf[spread - 1];
    // (c, s) depend on (len, K, spread) only, and a stream keeps hitting the same few hundred triples frame
    // after frame: a small per-thread direct-mapped memo replaces the two libm cos() calls (a fifth of the
    // whole entropy stage) by a lookup, with the very same values
    struct Memo { uint32_t key; float c, s; };
    static thread_local Memo memo[1024];
    const uint32_t key = ((uint32_t)len << 12) | ((uint32_t)K << 2) | (uint32_t)spread;   // len <= 176, K <= 128: never 0
    Memo &mm = memo[(key * 2654435761u) >> 22];
    if (mm.key != key) {
        const float gain = (float)(1.0f * len) / (float)(len + factor * K);
        const float theta = .5f * (gain * gain);
        // celt_cos_norm (mathops.h): float argument, C library cos() in double, rounded once
        mm.c = (float)std::cos((double)((.5f * 3.141592653f) * theta));
        mm.s = (float)std::cos((double)((.5f * 3.141592653f) * (1.0f - theta)));
        mm.key = key;
    }
    const float c = mm.c, s = mm.s;
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    len /= stride;
    for (int i = 0; i < stride; i++) {
        if (dir < 0) {
            if (stride2) expRotation1(X + i * len, len, stride2, s, c);
            expRotation1(X + i * len, len, 1, c, s);
        } else {
            expRotation1(X + i * len, len, 1, c, -s);
            if (stride2) expRotation1(X + i * len, len, stride2, s, -c);
        }
    }
}

void renormalise(float *X, int N, float gain) {                            // vq.c:354-382
    float E = 1e-15f;
    for (int i = 0; i < N; i++) E += X[i] * X[i];
    const float g = (1.f / (float)std::sqrt(E)) * gain;
    for (int i = 0; i < N; i++) X[i] = g * X[i];
}

unsigned algUnquant(float *X, int N, int K, int spread, int B, RangeDecoder &dec, float gain) {   // vq.c:327-352
    int iy[176];
    decodePulseVector(N, K, dec.uint((uint32_t)pvqV(N, K)), iy);
    float Ryy = 0;
    for (int i = 0; i < N; i++) Ryy += (float)iy[i] * (float)iy[i];
    const float g = (1.f / (float)std::sqrt(Ryy)) * gain;                   // normalise_residual
    for (int i = 0; i < N; i++) X[i] = g * iy[i];
    expRotation(X, N, -1, B, K, spread);
    if (B <= 1) return 1;                                                   // extract_collapse_mask
    const int N0 = N / B;
    unsigned mask = 0;
    for (int i = 0; i < B; i++)
        for (int j = 0; j < N0; j++) mask |= (unsigned)(iy[i * N0 + j] != 0) << i;
    return mask;
}

// ---- band shapes: splits, folding, stereo (bands.c:541-1518, decoder side) -------------------------
const int kOrdery[] = {1, 0, 3, 0, 2, 1, 7, 0, 4, 3, 6, 1, 5, 2, 15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5};

void deinterleaveHadamard(float *X, int N0, int stride, int hadamard) {
    float tmp[176];
    const int N = N0 * stride;
    if (hadamard) {
        const int *ordery = kOrdery + stride - 2;
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[ordery[i] * N0 + j] = X[j * stride + i];
    } else {
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[i * N0 + j] = X[j * stride + i];
    }
    std::memcpy(X, tmp, sizeof(float) * N);
}

void interleaveHadamard(float *X, int N0, int stride, int hadamard) {
    float tmp[176];
    const int N = N0 * stride;
    if (hadamard) {
        const int *ordery = kOrdery + stride - 2;
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[ordery[i] * N0 + j];
    } else {
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[i * N0 + j];
    }
    std::memcpy(X, tmp, sizeof(float) * N);
}

void haar1(float *X, int N0, int stride) {
    N0 >>= 1;
    for (int i = 0; i < stride; i++)
        for (int j = 0; j < N0; j++) {
            const float t1 = .70710678f * X[stride * 2 * j + i];
            const float t2 = .70710678f * X[stride * (2 * j + 1) + i];
            X[stride * 2 * j + i] = t1 + t2;
            X[stride * (2 * j + 1) + i] = t1 - t2;
        }
}

int bitexactCos(int x) {                                                    // bands.c:68-78
    const int32_t tmp = (4096 + (int32_t)x * x) >> 13;
    int x2 = (int16_t)tmp;
    x2 = (32767 - x2) + fracMul16(x2, (-7651 + fracMul16(x2, (8277 + fracMul16(-626, x2)))));
    return 1 + x2;
}

int bitexactLog2tan(int isin, int icos) {                                   // bands.c:80-92
    const int lc = ilog((uint32_t)icos), ls = ilog((uint32_t)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + fracMul16(isin, fracMul16(isin, -2597) + 7932) -
           fracMul16(icos, fracMul16(icos, -2597) + 7932);
}

int computeQn(int N, int b, int offset, int pulseCap, int stereo) {         // bands.c:614-636
    static const int16_t exp2Table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
    int N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    int qb = std::min(b - pulseCap - (4 << kBitRes), (b + N2 * offset) / N2);
    qb = std::min(8 << kBitRes, qb);
    int qn;
    if (qb < (1 << kBitRes >> 1)) {
        qn = 1;
    } else {
        qn = exp2Table8[qb & 0x7] >> (14 - (qb >> kBitRes));
        qn = (qn + 1) >> 1 << 1;
    }
    return qn;
}

struct BandState {
    const CeltMode *m;
    RangeDecoder *ec;
    int band, intensity, spread, tfChange;
    int32_t remainingBits;
    uint32_t seed;
};

struct Split {
    int inv, imid, iside, delta, itheta, qalloc;
};

void computeTheta(BandState &ctx, Split &sp, int N, int *b, int B, int B0, int LM, int stereo, int *fill) {   // bands.c:661-832
    const CeltMode &m = *ctx.m;
    RangeDecoder &ec = *ctx.ec;
    const int i = ctx.band;
    int itheta = 0, inv = 0;
    const int pulseCap = m.logN[i] + LM * (1 << kBitRes);
    const int offset = (pulseCap >> 1) - (stereo && N == 2 ? kQThetaOffsetTwoPhase : kQThetaOffset);
    int qn = computeQn(N, *b, offset, pulseCap, stereo);
    if (stereo && i >= ctx.intensity) qn = 1;
    const int32_t tell = (int32_t)ec.tellFrac();
    if (qn != 1) {
        if (stereo && N > 2) {                                              // step pdf
            const int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            const int fs = (int)ec.decode(ft);
            int x;
            if (fs < (x0 + 1) * p0) x = fs / p0;
            else x = x0 + 1 + (fs - (x0 + 1) * p0);
            ec.update(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0, x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            itheta = x;
        } else if (B0 > 1 || stereo) {                                      // uniform pdf
            itheta = (int)ec.uint(qn + 1);
        } else {                                                            // triangular pdf
            const int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            const int fm = (int)ec.decode(ft);
            int fs, fl;
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
                itheta = (int)(isqrt32(8 * (uint32_t)fm + 1) - 1) >> 1;
                fs = itheta + 1;
                fl = itheta * (itheta + 1) >> 1;
            } else {
                itheta = (int)(2 * (qn + 1) - isqrt32(8 * (uint32_t)(ft - fm - 1) + 1)) >> 1;
                fs = qn + 1 - itheta;
                fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            ec.update(fl, fl + fs, ft);
        }
        itheta = (int32_t)itheta * 16384 / qn;
    } else if (stereo) {
        if (*b > 2 << kBitRes && ctx.remainingBits > 2 << kBitRes) inv = ec.bitLogp(2);
        else inv = 0;
        itheta = 0;
    }
    const int qalloc = (int)ec.tellFrac() - tell;
    *b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) {
        imid = 32767; iside = 0; *fill &= (1 << B) - 1; delta = -16384;
    } else if (itheta == 16384) {
        imid = 0; iside = 32767; *fill &= ((1 << B) - 1) << B; delta = 16384;
    } else {
        imid = bitexactCos((int16_t)itheta);
        iside = bitexactCos((int16_t)(16384 - itheta));
        delta = fracMul16((N - 1) << 7, bitexactLog2tan(iside, imid));
    }
    sp = {inv, imid, iside, delta, itheta, qalloc};
}

unsigned quantBandN1(BandState &ctx, float *X, float *Y, float *lowbandOut) {     // bands.c:834-872
    float *x = X;
    for (int c = 0; c < 1 + (Y != nullptr); c++) {
        int sign = 0;
        if (ctx.remainingBits >= 1 << kBitRes) {
            sign = (int)ctx.ec->bits(1);
            ctx.remainingBits -= 1 << kBitRes;
        }
        x[0] = sign ? -1.f : 1.f;
        x = Y;
    }
    if (lowbandOut) lowbandOut[0] = X[0];
    return 1;
}

unsigned quantPartition(BandState &ctx, float *X, int N, int b, int B, float *lowband, int LM, float gain, int fill) {   // bands.c:879-1055
    const CeltMode &m = *ctx.m;
    const int i = ctx.band;
    const int B0 = B;
    unsigned cm = 0;
    const uint8_t *cache = m.cacheFor(i, LM);
    if (LM != -1 && b > cache[cache[0]] + 12 && N > 2) {                     // split the band in two
        N >>= 1;
        float *Y = X + N;
        LM -= 1;
        if (B == 1) fill = (fill & 1) | (fill << 1);
        B = (B + 1) >> 1;
        Split sp;
        computeTheta(ctx, sp, N, &b, B, B0, LM, 0, &fill);
        int delta = sp.delta;
        const int itheta = sp.itheta;
        const float mid = (1.f / 32768) * sp.imid, side = (1.f / 32768) * sp.iside;
        if (B0 > 1 && (itheta & 0x3fff)) {
            if (itheta > 8192) delta -= delta >> (4 - LM);
            else delta = std::min(0, delta + (N << kBitRes >> (5 - LM)));
        }
        int mbits = std::max(0, std::min(b, (b - delta) / 2));
        int sbits = b - mbits;
        ctx.remainingBits -= sp.qalloc;
        float *nextLowband2 = lowband ? lowband + N : nullptr;
        int32_t rebalance = ctx.remainingBits;
        if (mbits >= sbits) {
            cm = quantPartition(ctx, X, N, mbits, B, lowband, LM, gain * mid, fill);
            rebalance = mbits - (rebalance - ctx.remainingBits);
            if (rebalance > 3 << kBitRes && itheta != 0) sbits += rebalance - (3 << kBitRes);
            cm |= quantPartition(ctx, Y, N, sbits, B, nextLowband2, LM, gain * side, fill >> B) << (B0 >> 1);
        } else {
            cm = quantPartition(ctx, Y, N, sbits, B, nextLowband2, LM, gain * side, fill >> B) << (B0 >> 1);
            rebalance = sbits - (rebalance - ctx.remainingBits);
            if (rebalance > 3 << kBitRes && itheta != 16384) mbits += rebalance - (3 << kBitRes);
            cm |= quantPartition(ctx, X, N, mbits, B, lowband, LM, gain * mid, fill);
        }
        return cm;
    }
    int q = m.bits2pulses(i, LM, b);                                         // no split
    int currBits = m.pulses2bits(i, LM, q);
    ctx.remainingBits -= currBits;
    while (ctx.remainingBits < 0 && q > 0) {
        ctx.remainingBits += currBits;
        q--;
        currBits = m.pulses2bits(i, LM, q);
        ctx.remainingBits -= currBits;
    }
    if (q != 0) return algUnquant(X, N, CeltMode::pulsesOf(q), ctx.spread, B, *ctx.ec, gain);
    const unsigned cmMask = (unsigned)(1UL << B) - 1;                         // no pulses: fold or noise
    fill &= (int)cmMask;
    if (!fill) {
        for (int j = 0; j < N; j++) X[j] = 0;
        return 0;
    }
    if (lowband == nullptr) {
        for (int j = 0; j < N; j++) {
            ctx.seed = lcg(ctx.seed);
            X[j] = (float)((int32_t)ctx.seed >> 20);
        }
        cm = cmMask;
    } else {
        for (int j = 0; j < N; j++) {
            ctx.seed = lcg(ctx.seed);
            const float tmp = (ctx.seed & 0x8000) ? 1.0f / 256 : -1.0f / 256;
            X[j] = lowband[j] + tmp;
        }
        cm = (unsigned)fill;
    }
    renormalise(X, N, gain);
    return cm;
}

unsigned quantBand(BandState &ctx, float *X, int N, int b, int B, float *lowband, int LM, float *lowbandOut, float gain,
                   float *lowbandScratch, int fill) {                        // bands.c:1060-1191
    const int N0 = N;
    int N_B = N;
    int B0 = B;
    int timeDivide = 0, recombine = 0;
    const int longBlocks = B0 == 1;
    int tfChange = ctx.tfChange;
    N_B /= B;
    if (N == 1) return quantBandN1(ctx, X, nullptr, lowbandOut);
    if (tfChange > 0) recombine = tfChange;
    if (lowbandScratch && lowband && (recombine || ((N_B & 1) == 0 && tfChange < 0) || B0 > 1)) {
        std::memcpy(lowbandScratch, lowband, sizeof(float) * N);
        lowband = lowbandScratch;
    }
    for (int k = 0; k < recombine; k++) {
        static const uint8_t bitInterleave[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
        if (lowband) haar1(lowband, N >> k, 1 << k);
        fill = bitInterleave[fill & 0xF] | bitInterleave[fill >> 4] << 2;
    }
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tfChange < 0) {                                  // more time resolution
        if (lowband) haar1(lowband, N_B, B);
        fill |= fill << B;
        B <<= 1;
        N_B >>= 1;
        timeDivide++;
        tfChange++;
    }
    B0 = B;
    const int N_B0 = N_B;
    if (B0 > 1 && lowband) deinterleaveHadamard(lowband, N_B >> recombine, B0 << recombine, longBlocks);
    unsigned cm = quantPartition(ctx, X, N, b, B, lowband, LM, gain, fill);
    if (B0 > 1) interleaveHadamard(X, N_B >> recombine, B0 << recombine, longBlocks);
    N_B = N_B0;
    B = B0;
    for (int k = 0; k < timeDivide; k++) {
        B >>= 1;
        N_B <<= 1;
        cm |= cm >> B;
        haar1(X, N_B, B);
    }
    for (int k = 0; k < recombine; k++) {
        static const uint8_t bitDeinterleave[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F,
                                                    0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
        cm = bitDeinterleave[cm];
        haar1(X, N0 >> k, 1 << k);
    }
    B <<= recombine;
    if (lowbandOut) {                                                         // scaled copy for later folding
        const float n = (float)std::sqrt((float)N0);
        for (int j = 0; j < N0; j++) lowbandOut[j] = n * X[j];
    }
    cm &= (1u << B) - 1;
    return cm;
}

void stereoMerge(float *X, float *Y, float mid, int N) {                      // bands.c:391-441
    float xp = 0, side = 0;
    for (int j = 0; j < N; j++) {
        xp += Y[j] * X[j];
        side += Y[j] * Y[j];
    }
    xp = mid * xp;
    const float mid2 = mid;
    const float El = mid2 * mid2 + side - 2 * xp;
    const float Er = mid2 * mid2 + side + 2 * xp;
    if (Er < 6e-4f || El < 6e-4f) {
        std::memcpy(Y, X, sizeof(float) * N);
        return;
    }
    const float lgain = 1.f / (float)std::sqrt(El), rgain = 1.f / (float)std::sqrt(Er);
    for (int j = 0; j < N; j++) {
        const float l = mid * X[j], r = Y[j];
        X[j] = lgain * (l - r);
        Y[j] = rgain * (l + r);
    }
}

unsigned quantBandStereo(BandState &ctx, float *X, float *Y, int N, int b, int B, float *lowband, int LM, float *lowbandOut,
                         float *lowbandScratch, int fill) {                   // bands.c:1194-1353
    if (N == 1) return quantBandN1(ctx, X, Y, lowbandOut);
    const int origFill = fill;
    Split sp;
    computeTheta(ctx, sp, N, &b, B, B, LM, 1, &fill);
    const int inv = sp.inv, itheta = sp.itheta, delta = sp.delta, qalloc = sp.qalloc;
    const float mid = (1.f / 32768) * sp.imid, side = (1.f / 32768) * sp.iside;
    unsigned cm = 0;
    if (N == 2) {
        int mbits = b, sbits = 0;
        if (itheta != 0 && itheta != 16384) sbits = 1 << kBitRes;
        mbits -= sbits;
        const int c = itheta > 8192;
        ctx.remainingBits -= qalloc + sbits;
        float *x2 = c ? Y : X, *y2 = c ? X : Y;
        int sign = 0;
        if (sbits) sign = (int)ctx.ec->bits(1);
        sign = 1 - 2 * sign;
        cm = quantBand(ctx, x2, N, mbits, B, lowband, LM, lowbandOut, 1.0f, lowbandScratch, origFill);
        y2[0] = -sign * x2[1];
        y2[1] = sign * x2[0];
        X[0] = mid * X[0];
        X[1] = mid * X[1];
        Y[0] = side * Y[0];
        Y[1] = side * Y[1];
        float tmp = X[0];
        X[0] = tmp - Y[0];
        Y[0] = tmp + Y[0];
        tmp = X[1];
        X[1] = tmp - Y[1];
        Y[1] = tmp + Y[1];
    } else {
        int mbits = std::max(0, std::min(b, (b - delta) / 2));
        int sbits = b - mbits;
        ctx.remainingBits -= qalloc;
        int32_t rebalance = ctx.remainingBits;
        if (mbits >= sbits) {
            cm = quantBand(ctx, X, N, mbits, B, lowband, LM, lowbandOut, 1.0f, lowbandScratch, fill);
            rebalance = mbits - (rebalance - ctx.remainingBits);
            if (rebalance > 3 << kBitRes && itheta != 0) sbits += rebalance - (3 << kBitRes);
            cm |= quantBand(ctx, Y, N, sbits, B, nullptr, LM, nullptr, side, nullptr, fill >> B);
        } else {
            cm = quantBand(ctx, Y, N, sbits, B, nullptr, LM, nullptr, side, nullptr, fill >> B);
            rebalance = sbits - (rebalance - ctx.remainingBits);
            if (rebalance > 3 << kBitRes && itheta != 16384) mbits += rebalance - (3 << kBitRes);
            cm |= quantBand(ctx, X, N, mbits, B, lowband, LM, lowbandOut, 1.0f, lowbandScratch, fill);
        }
    }
    if (N != 2) stereoMerge(X, Y, mid, N);
    if (inv)
        for (int j = 0; j < N; j++) Y[j] = -Y[j];
    return cm;
}

void quantAllBands(const CeltMode &m, int start, int end, float *X_, float *Y_, uint8_t *collapseMasks, const int *pulses,
                   int shortBlocks, int spread, int dualStereo, int intensity, const int *tfRes, int32_t totalBits,
                   int32_t balance, RangeDecoder &ec, int LM, int codedBands, uint32_t *seed) {   // bands.c:1355-1518
    const int16_t *eBands = m.eBands;
    const int M = 1 << LM;
    const int B = shortBlocks ? M : 1;
    const int C = Y_ ? 2 : 1;
    const int normOffset = M * eBands[start];
    std::vector<float> normBuf((size_t)C * (M * eBands[kBands - 1] - normOffset));
    float *norm = normBuf.data();
    float *norm2 = norm + M * eBands[kBands - 1] - normOffset;
    float *lowbandScratch = X_ + M * eBands[kBands - 1];     // the last band doubles as scratch
    int lowbandOffset = 0;
    int updateLowband = 1;
    BandState ctx;
    ctx.m = &m;
    ctx.ec = &ec;
    ctx.intensity = intensity;
    ctx.seed = *seed;
    ctx.spread = spread;
    for (int i = start; i < end; i++) {
        ctx.band = i;
        const int last = (i == end - 1);
        float *X = X_ + M * eBands[i];
        float *Y = Y_ ? Y_ + M * eBands[i] : nullptr;
        const int N = M * eBands[i + 1] - M * eBands[i];
        const int32_t tell = (int32_t)ec.tellFrac();
        if (i != start) balance -= tell;
        const int32_t remainingBits = totalBits - tell - 1;
        ctx.remainingBits = remainingBits;
        int b;
        if (i <= codedBands - 1) {
            const int32_t currBalance = balance / std::min(3, codedBands - i);
            b = std::max(0, std::min(16383, (int)std::min<int32_t>(remainingBits + 1, pulses[i] + currBalance)));
        } else {
            b = 0;
        }
        if (M * eBands[i] - N >= M * eBands[start] && (updateLowband || lowbandOffset == 0)) lowbandOffset = i;
        const int tfChange = tfRes[i];
        ctx.tfChange = tfChange;
        if (i >= kBands) {   // i >= m->effEBands never happens for the 48 kHz mode (effEBands == nbEBands)
            X = norm;
            if (Y_) Y = norm;
            lowbandScratch = nullptr;
        }
        if (i == end - 1) lowbandScratch = nullptr;
        int effectiveLowband = -1;
        unsigned xCm, yCm;
        if (lowbandOffset != 0 && (spread != kSpreadAggressive || B > 1 || tfChange < 0)) {
            effectiveLowband = std::max(0, M * eBands[lowbandOffset] - normOffset - N);
            int foldStart = lowbandOffset;
            while (M * eBands[--foldStart] > effectiveLowband + normOffset) {}
            int foldEnd = lowbandOffset - 1;
            while (M * eBands[++foldEnd] < effectiveLowband + normOffset + N) {}
            xCm = yCm = 0;
            int foldI = foldStart;
            do {
                xCm |= collapseMasks[foldI * C + 0];
                yCm |= collapseMasks[foldI * C + C - 1];
            } while (++foldI < foldEnd);
        } else {
            xCm = yCm = (1u << B) - 1;
        }
        if (dualStereo && i == intensity) {
            dualStereo = 0;
            for (int j = 0; j < M * eBands[i] - normOffset; j++) norm[j] = .5f * (norm[j] + norm2[j]);
        }
        float *lowX = effectiveLowband != -1 ? norm + effectiveLowband : nullptr;
        float *outX = last ? nullptr : norm + M * eBands[i] - normOffset;
        if (dualStereo) {
            float *lowY = effectiveLowband != -1 ? norm2 + effectiveLowband : nullptr;
            float *outY = last ? nullptr : norm2 + M * eBands[i] - normOffset;
            xCm = quantBand(ctx, X, N, b / 2, B, lowX, LM, outX, 1.0f, lowbandScratch, (int)xCm);
            yCm = quantBand(ctx, Y, N, b / 2, B, lowY, LM, outY, 1.0f, lowbandScratch, (int)yCm);
        } else {
            if (Y) xCm = quantBandStereo(ctx, X, Y, N, b, B, lowX, LM, outX, lowbandScratch, (int)(xCm | yCm));
            else xCm = quantBand(ctx, X, N, b, B, lowX, LM, outX, 1.0f, lowbandScratch, (int)(xCm | yCm));
            yCm = xCm;
        }
        collapseMasks[i * C + 0] = (uint8_t)xCm;
        collapseMasks[i * C + C - 1] = (uint8_t)yCm;
        balance += pulses[i] + tell;
        updateLowband = b > (N << kBitRes);
    }
    *seed = ctx.seed;
}

void antiCollapse(const CeltMode &m, float *X_, const uint8_t *collapseMasks, int LM, int C, int size, int start, int end,
                  const float *logE, const float *prev1logE, const float *prev2logE, const int *pulses, uint32_t seed) {   // bands.c:258-351
    for (int i = start; i < end; i++) {
        const int N0 = m.eBands[i + 1] - m.eBands[i];
        const int depth = (1 + pulses[i]) / ((m.eBands[i + 1] - m.eBands[i]) << LM);
        const float thresh = .5f * exp2f_ref(-.125f * depth);
        const float sqrt1 = 1.f / (float)std::sqrt((float)(N0 << LM));
        for (int c = 0; c < C; c++) {
            float prev1 = prev1logE[c * kBands + i], prev2 = prev2logE[c * kBands + i];
            if (C == 1) {
                prev1 = std::max(prev1, prev1logE[kBands + i]);
                prev2 = std::max(prev2, prev2logE[kBands + i]);
            }
            float Ediff = logE[c * kBands + i] - std::min(prev1, prev2);
            Ediff = std::max(0.f, Ediff);
            float r = 2.f * exp2f_ref(-Ediff);
            if (LM == 3) r *= 1.41421356f;
            r = std::min(thresh, r);
            r = r * sqrt1;
            float *X = X_ + c * size + (m.eBands[i] << LM);
            int renorm = 0;
            for (int k = 0; k < 1 << LM; k++) {
                if (!(collapseMasks[i * C + c] & 1 << k)) {
                    for (int j = 0; j < N0; j++) {
                        seed = lcg(seed);
                        X[(j << LM) + k] = (seed & 0x8000 ? r : -r);
                    }
                    renorm = 1;
                }
            }
            if (renorm) renormalise(X, N0 << LM, 1.0f);
        }
    }
}

void denormalise(const CeltMode &m, const float *X, float *freq, const float *bandLogE, int start, int end, int C, int M) {   // bands.c:192-256
    const int N = M * kShortMdct;
    for (int c = 0; c < C; c++) {
        float *f = freq + c * N;
        const float *x = X + c * N + M * m.eBands[start];
        for (int i = 0; i < M * m.eBands[start]; i++) *f++ = 0;
        for (int i = start; i < end; i++) {
            int j = M * m.eBands[i];
            const int bandEnd = M * m.eBands[i + 1];
            const float lg = bandLogE[i + c * kBands] + m.eMeans[i];
            const float g = exp2f_ref(lg);
            do {
                *f++ = *x++ * g;
            } while (++j < bandEnd);
        }
        for (int i = M * m.eBands[end]; i < N; i++) *f++ = 0;
    }
}

}  // namespace

// ---- the frame decoder --------------------------------------------------------------------------
CeltDecoder::CeltDecoder(int channels) : m_(mode48k()), channels_(channels), streamChannels_(channels) { reset(); }

void CeltDecoder::reset() {
    rng_ = 0;
    for (int i = 0; i < 2 * kBands; i++) {
        oldBandE_[i] = 0.f;
        oldLogE_[i] = oldLogE2_[i] = -28.f;      // celt_decoder_clean.c:855-856
        backgroundLogE_[i] = 0.f;
    }
}

int CeltDecoder::decode(const uint8_t *data, int len, int frameSize, float *freqOut, CeltFrame &info) {
    const CeltMode &m = m_;
    const int CC = channels_, C = streamChannels_;
    // A mono decoder may be handed stereo-coded packets (the TOC's stereo flag is per packet): both channels are
    // decoded and then mixed down (celt_decoder_clean.c:648-652), so the work buffer is max(CC, C) channels wide
    // (:396 ALLOC(freq, IMAX(CC,C)*N)) while the caller's holds CC.
    float *freq = C > CC ? wide_ : freqOut;
    int LM;
    for (LM = 0; LM <= kMaxLM; LM++)
        if ((kShortMdct << LM) == frameSize) break;
    if (LM > kMaxLM) return -1;
    if (len < 0 || len > 1275 || !data || !freqOut) return -1;
    const int M = 1 << LM;
    const int N = M * kShortMdct;
    const int start = start_, end = end_;
    const int effEnd = std::min(end, kBands);

    RangeDecoder dec;
    dec.init(data, (uint32_t)len);
    if (C == 1)
        for (int i = 0; i < kBands; i++) oldBandE_[i] = std::max(oldBandE_[i], oldBandE_[kBands + i]);

    int32_t totalBits = len * 8;
    int32_t tell = dec.tell();
    int silence;
    if (tell >= totalBits) silence = 1;
    else if (tell == 1) silence = dec.bitLogp(15);
    else silence = 0;
    if (silence) {
        tell = len * 8;                      // pretend every remaining bit was read
        dec.skipTo(tell);
    }
    float pfGain = 0.f;
    int pfPitch = 0, pfTapset = 0;
    if (start == 0 && tell + 16 <= totalBits) {
        if (dec.bitLogp(1)) {
            const int octave = (int)dec.uint(6);
            pfPitch = (16 << octave) + (int)dec.bits(4 + octave) - 1;
            const int qg = (int)dec.bits(3);
            if (dec.tell() + 2 <= totalBits) pfTapset = dec.icdf(kTapsetIcdf, 2);
            pfGain = .09375f * (qg + 1);
        }
        tell = dec.tell();
    }
    int isTransient = 0;
    if (LM > 0 && tell + 3 <= totalBits) {
        isTransient = dec.bitLogp(3);
        tell = dec.tell();
    }
    const int shortBlocks = isTransient ? M : 0;
    const int intraEner = tell + 3 <= totalBits ? dec.bitLogp(3) : 0;
    unquantCoarse(m, start, end, oldBandE_, intraEner, dec, C, LM);

    int tfRes[kBands];
    tfDecode(start, end, isTransient, tfRes, LM, dec);

    tell = dec.tell();
    int spreadDecision = kSpreadNormal;
    if (tell + 4 <= totalBits) spreadDecision = dec.icdf(kSpreadIcdf, 5);

    int cap[kBands];
    m.initCaps(cap, LM, C);
    int offsets[kBands];
    int dynallocLogp = 6;
    totalBits <<= kBitRes;
    tell = (int32_t)dec.tellFrac();
    for (int i = start; i < end; i++) {
        const int width = C * (m.eBands[i + 1] - m.eBands[i]) << LM;
        const int quanta = std::min(width << kBitRes, std::max(6 << kBitRes, width));
        int loopLogp = dynallocLogp;
        int boost = 0;
        while (tell + (loopLogp << kBitRes) < totalBits && boost < cap[i]) {
            const int flag = dec.bitLogp(loopLogp);
            tell = (int32_t)dec.tellFrac();
            if (!flag) break;
            boost += quanta;
            totalBits -= quanta;
            loopLogp = 1;
        }
        offsets[i] = boost;
        if (boost > 0) dynallocLogp = std::max(2, dynallocLogp - 1);
    }
    int fineQuant[kBands], pulses[kBands], finePriority[kBands];
    const int allocTrim = tell + (6 << kBitRes) <= totalBits ? dec.icdf(kTrimIcdf, 7) : 5;
    int32_t bits = (((int32_t)len * 8) << kBitRes) - (int32_t)dec.tellFrac() - 1;
    const int antiCollapseRsv = isTransient && LM >= 2 && bits >= ((LM + 2) << kBitRes) ? (1 << kBitRes) : 0;
    bits -= antiCollapseRsv;
    int intensity = 0, dualStereo = 0;
    int32_t balance = 0;
    const int codedBands = computeAllocation(m, start, end, offsets, cap, allocTrim, &intensity, &dualStereo, bits, &balance,
                                             pulses, fineQuant, finePriority, C, LM, dec);
    unquantFine(start, end, oldBandE_, fineQuant, dec, C);

    uint8_t collapseMasks[2 * kBands];
    std::memset(collapseMasks, 0, sizeof collapseMasks);
    std::vector<float> Xbuf((size_t)C * N, 0.f);
    float *X = Xbuf.data();
    quantAllBands(m, start, end, X, C == 2 ? X + N : nullptr, collapseMasks, pulses, shortBlocks, spreadDecision, dualStereo,
                  intensity, tfRes, len * (8 << kBitRes) - antiCollapseRsv, balance, dec, LM, codedBands, &rng_);
    int antiCollapseOn = 0;
    if (antiCollapseRsv > 0) antiCollapseOn = (int)dec.bits(1);
    unquantFinalise(start, end, oldBandE_, fineQuant, finePriority, len * 8 - dec.tell(), dec, C);
    if (antiCollapseOn)
        antiCollapse(m, X, collapseMasks, LM, C, N, start, end, oldBandE_, oldLogE_, oldLogE2_, pulses, rng_);

    if (silence) {
        for (int i = 0; i < C * kBands; i++) oldBandE_[i] = -28.f;
        std::memset(freq, 0, sizeof(float) * (size_t)std::max(CC, C) * N);
    } else {
        denormalise(m, X, freq, oldBandE_, start, effEnd, C, M);
    }
    for (int c = 0; c < C; c++) {                                   // celt_decoder_clean.c:628-636
        const int bound = M * m.eBands[effEnd];
        for (int i = bound; i < N; i++) freq[c * N + i] = 0;
    }
    if (CC == 2 && C == 1) std::memcpy(freq + N, freq, sizeof(float) * N);          // :643-647
    if (CC == 1 && C == 2)
        for (int i = 0; i < N; i++) freqOut[i] = .5f * (freq[i] + freq[N + i]);    // :648-652

    if (C == 1) std::memcpy(oldBandE_ + kBands, oldBandE_, sizeof(float) * kBands);  // :685-689
    if (!isTransient) {                                                               // :691-703
        std::memcpy(oldLogE2_, oldLogE_, sizeof oldLogE_);
        std::memcpy(oldLogE_, oldBandE_, sizeof oldBandE_);
        for (int i = 0; i < 2 * kBands; i++) backgroundLogE_[i] = std::min(backgroundLogE_[i] + M * 0.001f, oldBandE_[i]);
    } else {
        for (int i = 0; i < 2 * kBands; i++) oldLogE_[i] = std::min(oldLogE_[i], oldBandE_[i]);
    }
    for (int c = 0; c < 2; c++) {                                                     // :704-718
        for (int i = 0; i < start; i++) {
            oldBandE_[c * kBands + i] = 0;
            oldLogE_[c * kBands + i] = oldLogE2_[c * kBands + i] = -28.f;
        }
        for (int i = end; i < kBands; i++) {
            oldBandE_[c * kBands + i] = 0;
            oldLogE_[c * kBands + i] = oldLogE2_[c * kBands + i] = -28.f;
        }
    }
    rng_ = dec.range();
    info.LM = LM;
    info.channels = CC;
    info.transient = isTransient != 0;
    info.silence = silence != 0;
    info.pfPitch = pfPitch;
    info.pfGain = pfGain;
    info.pfTapset = pfTapset;
    info.rangeFinal = rng_;
    if (dec.tell() > 8 * len) return -3;
    if (dec.error()) return -4;
    return 0;
}

}  // namespace nyq_host
