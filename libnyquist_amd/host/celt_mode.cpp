// celt_mode.cpp -- see celt_mode.hpp.
#include "celt_mode.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace nyq_host {

namespace {

constexpr int kMaxDim = kPvqTableDim;   // widest band at LM 3 is 22*8 = 176 bins; k up to 129 is needed
uint64_t g_U[kMaxDim][kMaxDim];
std::once_flag g_once;
CeltMode g_mode;

void buildU() {
    const uint64_t cap = ~0ull >> 1;
    for (int n = 0; n < kMaxDim; n++)
        for (int k = 0; k < kMaxDim; k++) {
            if (n == 0 || k == 0) {
                g_U[n][k] = (n == 0 && k == 0) ? 1 : 0;
                continue;
            }
            uint64_t a = g_U[n - 1][k], b = g_U[n][k - 1], c = g_U[n - 1][k - 1];
            uint64_t s = a + b;
            if (s > cap) s = cap;
            s += c;
            if (s > cap) s = cap;
            g_U[n][k] = s;
        }
}

int widthOf(const CeltMode &m, int j) { return m.eBands[j + 1] - m.eBands[j]; }
uint64_t rawV(int n, int k) { return g_U[n][k] + g_U[n][k + 1]; }   // valid once buildU() has run

// rate.c:73-245 restated: one shared table of "bits needed for pseudo-pulse count q" per distinct
// band size, and per (LM, C, band) the rate above which extra bits stop being useful.
void buildPulseCache(CeltMode &m) {
    const int LM = kMaxLM;
    m.cacheIndex.assign((LM + 2) * kBands, -1);
    struct Entry { int N, K, off; };
    std::vector<Entry> entries;
    int total = 0;
    for (int i = 0; i <= LM + 1; i++)
        for (int j = 0; j < kBands; j++) {
            const int N = (widthOf(m, j) << i) >> 1;
            int found = -1;
            bool hit = false;
            for (int k = 0; k <= i && !hit; k++)
                for (int n = 0; n < kBands && (k != i || n < j); n++)
                    if (N == ((widthOf(m, n) << k) >> 1)) {
                        found = m.cacheIndex[k * kBands + n];
                        hit = true;
                        break;
                    }
            if (hit) {
                m.cacheIndex[i * kBands + j] = (int16_t)found;
            } else if (N != 0) {
                int K = 0;
                while (K < kMaxPseudo && rawV(N, CeltMode::pulsesOf(K + 1)) <= 0xFFFFFFFFull) K++;
                m.cacheIndex[i * kBands + j] = (int16_t)total;
                entries.push_back({N, K, total});
                total += K + 1;
            }
        }
    m.cacheBits.assign(total, 0);
    for (const Entry &e : entries) {
        uint8_t *p = m.cacheBits.data() + e.off;
        p[0] = (uint8_t)e.K;
        for (int q = 1; q <= e.K; q++)
            p[q] = (uint8_t)(log2Frac((uint32_t)rawV(e.N, CeltMode::pulsesOf(q)), kBitRes) - 1);
    }
    m.cacheCaps.assign((LM + 1) * 2 * kBands, 0);
    uint8_t *cap = m.cacheCaps.data();
    for (int i = 0; i <= LM; i++)
        for (int C = 1; C <= 2; C++)
            for (int j = 0; j < kBands; j++) {
                int N0 = widthOf(m, j);
                int maxBits;
                if ((N0 << i) == 1) {
                    maxBits = (C * (1 + kMaxFineBits)) << kBitRes;       // a sign bit plus fine energy
                } else {
                    int LM0 = 0;
                    if (N0 > 2) { N0 >>= 1; LM0--; }                     // one more split is possible
                    else if (N0 <= 1) { LM0 = std::min(i, 1); N0 <<= LM0; }
                    const uint8_t *pc = m.cacheBits.data() + m.cacheIndex[(LM0 + 1) * kBands + j];
                    maxBits = pc[pc[0]] + 1;                             // fully split band, deepest PVQ
                    int N = N0;
                    for (int k = 0; k < i - LM0; k++) {                  // plus every time split's theta
                        maxBits <<= 1;
                        const int offset = ((m.logN[j] + (LM0 + k) * (1 << kBitRes)) >> 1) - kQThetaOffset;
                        const int32_t num = 459 * (int32_t)((2 * N - 1) * offset + maxBits);
                        const int32_t den = ((int32_t)(2 * N - 1) << 9) - 459;
                        maxBits += std::min((num + (den >> 1)) / den, 57);
                        N <<= 1;
                    }
                    if (C == 2) {                                        // plus the stereo split
                        maxBits <<= 1;
                        const int offset = ((m.logN[j] + (i << kBitRes)) >> 1) - (N == 2 ? kQThetaOffsetTwoPhase : kQThetaOffset);
                        const int ndof = 2 * N - 1 - (N == 2);
                        const int32_t num = (N == 2 ? 512 : 487) * (int32_t)(maxBits + ndof * offset);
                        const int32_t den = ((int32_t)ndof << 9) - (N == 2 ? 512 : 487);
                        maxBits += std::min((num + (den >> 1)) / den, (N == 2 ? 64 : 61));
                    }
                    const int ndof = C * N + ((C == 2 && N > 2) ? 1 : 0); // plus the fine energy bits
                    int offset = ((m.logN[j] + (i << kBitRes)) >> 1) - kFineOffset;
                    if (N == 2) offset += (1 << kBitRes) >> 2;
                    const int32_t num = maxBits + ndof * offset;
                    const int32_t den = (ndof - 1) << kBitRes;
                    maxBits += (C * std::min((num + (den >> 1)) / den, kMaxFineBits)) << kBitRes;
                }
                maxBits = (4 * maxBits / (C * (widthOf(m, j) << i))) - 64;
                *cap++ = (uint8_t)maxBits;
            }
}

void buildMode() {
    buildU();
    CeltMode &m = g_mode;
    // Band edges of the Opus CELT layer at 48 kHz, in 2.5 ms MDCT bins (RFC 6716 table 55; modes.c:41-44)
    static const int16_t edges[kBands + 1] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
    std::memcpy(m.eBands, edges, sizeof edges);
    for (int j = 0; j < kBands; j++) m.logN[j] = (int16_t)log2Frac((uint32_t)widthOf(m, j), kBitRes);
    // Static bit allocation matrix in 1/32 bit per sample (RFC 6716 table 57; modes.c:49-62)
    static const uint8_t alloc[kAllocVectors * kBands] = {
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        90, 80, 75, 69, 63, 56, 49, 40, 34, 29, 20, 18, 10, 0, 0, 0, 0, 0, 0, 0, 0,
        110, 100, 90, 84, 78, 71, 65, 58, 51, 45, 39, 32, 26, 20, 12, 0, 0, 0, 0, 0, 0,
        118, 110, 103, 93, 86, 80, 75, 70, 65, 59, 53, 47, 40, 31, 23, 15, 4, 0, 0, 0, 0,
        126, 119, 112, 104, 95, 89, 83, 78, 72, 66, 60, 54, 47, 39, 32, 25, 17, 12, 1, 0, 0,
        134, 127, 120, 114, 103, 97, 91, 85, 78, 72, 66, 60, 54, 47, 41, 35, 29, 23, 16, 10, 1,
        144, 137, 130, 124, 113, 107, 101, 95, 88, 82, 76, 70, 64, 57, 51, 45, 39, 33, 26, 15, 1,
        152, 145, 138, 132, 123, 117, 111, 105, 98, 92, 86, 80, 74, 67, 61, 55, 49, 43, 36, 20, 1,
        162, 155, 148, 142, 133, 127, 121, 115, 108, 102, 96, 90, 84, 77, 71, 65, 59, 53, 46, 30, 1,
        172, 165, 158, 152, 143, 137, 131, 125, 118, 112, 106, 100, 94, 87, 81, 75, 69, 63, 56, 45, 20,
        200, 200, 200, 200, 200, 200, 200, 200, 198, 193, 188, 183, 178, 173, 168, 163, 158, 153, 148, 129, 104};
    std::memcpy(m.alloc, alloc, sizeof alloc);
    // Mean band energies in log2 units (Q4 constants of the specification, quant_bands.c:43-60)
    static const int8_t meansQ4[25] = {103, 100, 92, 85, 81, 77, 72, 70, 78, 75, 73, 71, 78, 74, 69, 72, 70, 74, 76, 71, 60, 60, 60, 60, 60};
    for (int i = 0; i < 25; i++) m.eMeans[i] = meansQ4[i] / 16.0f;
    buildPulseCache(m);
}

}  // namespace

const CeltMode &mode48k() {
    std::call_once(g_once, buildMode);
    return g_mode;
}

uint64_t pvqU(int n, int k) {
    std::call_once(g_once, buildMode);
    return g_U[n][k];
}

const uint64_t *pvqTable() {
    std::call_once(g_once, buildMode);
    return &g_U[0][0];
}

const uint32_t *pvqTable32() {
    static const std::vector<uint32_t> t = [] {
        const uint64_t *u = pvqTable();
        std::vector<uint32_t> v((size_t)kPvqTableDim * kPvqTableDim);
        for (size_t i = 0; i < v.size(); i++) v[i] = u[i] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)u[i];
        return v;
    }();
    return t.data();
}

uint64_t pvqV(int n, int k) {
    std::call_once(g_once, buildMode);
    return rawV(n, k);
}

int log2Frac(uint32_t val, int frac) {
    int l = ilog(val);
    if (val & (val - 1)) {
        if (l > 16) val = ((val - 1) >> (l - 16)) + 1;   // (val >> l-16) rounded up
        else val <<= 16 - l;
        l = (l - 1) << frac;
        do {
            const int b = (int)(val >> 16);
            l += b << frac;
            val = (val + b) >> b;
            val = (val * val + 0x7FFF) >> 15;
        } while (frac-- > 0);
        return l + (val > 0x8000);
    }
    return (l - 1) << frac;
}

unsigned isqrt32(uint32_t v) {
    unsigned g = 0;
    int bshift = (ilog(v) - 1) >> 1;
    unsigned b = 1u << bshift;
    do {
        const uint32_t t = (((uint32_t)g << 1) + b) << bshift;
        if (t <= v) { g += b; v -= t; }
        b >>= 1;
        bshift--;
    } while (bshift >= 0);
    return g;
}

int CeltMode::bits2pulses(int band, int LM, int bits) const {
    const uint8_t *cache = cacheFor(band, LM);
    int lo = 0, hi = cache[0];
    bits--;
    for (int i = 0; i < kLogMaxPseudo; i++) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)cache[mid] >= bits) hi = mid;
        else lo = mid;
    }
    return (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) ? lo : hi;
}

int CeltMode::pulses2bits(int band, int LM, int pulses) const {
    return pulses == 0 ? 0 : cacheFor(band, LM)[pulses] + 1;
}

void CeltMode::initCaps(int *cap, int LM, int C) const {
    for (int i = 0; i < kBands; i++) {
        const int N = (eBands[i + 1] - eBands[i]) << LM;
        cap[i] = ((cacheCaps[kBands * (2 * LM + C - 1) + i] + 64) * C * N) >> 2;
    }
}

}  // namespace nyq_host
