// tools/scripts/pvq_unrank_bench.cpp -- variants of the PVQ index -> pulse vector step timed on the (n, k, index) triples of a real stream
// (a trace dumped from the decoder: 867980 vectors of sb-reverie.opus); every variant is checked against the row-walk form first.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdint>
#include "celt_mode.hpp"
using namespace nyq_host;
struct T3 { int n, k; unsigned idx; };
static const long D = kPvqTableDim;
// variant A: current decoder's
int32_t unrankA(int n, int k, uint32_t idx, int16_t *y) {
    const uint32_t *T = pvqTable32();
    const uint32_t *here = T + (size_t)k * D;
    int32_t yy = 0;
    while (n > 2) {
        const uint32_t above = here[D + n];
        const uint32_t neg = idx >= above ? ~0u : 0u;
        idx -= above & neg;
        const uint32_t h0 = here[n], h1 = k > 0 ? here[n - D] : 0;
        const bool zero = h0 <= idx, unit = !zero && h1 <= idx;
        if (__builtin_expect(zero || unit, 1)) {
            idx -= zero ? h0 : h1;
            const int v = unit ? 1 : 0;
            *y++ = (int16_t)((v ^ (int)neg) - (int)neg);
            yy += v; k -= v; here -= unit ? D : 0;
        } else {
            int kk = k - 1;
            const uint32_t *p = here + n - D;
            if (k > n && T[(size_t)n * D + n] > idx) { kk = n; p = T + (size_t)n * D + n; }
            do { kk--; p -= D; } while (*p > idx);
            idx -= *p;
            const int v = k - kk;
            *y++ = (int16_t)(neg ? -v : v);
            yy += v * v; k = kk; here = T + (size_t)k * D;
        }
        n--;
    }
    const uint32_t a = 2 * (uint32_t)k + 1;
    const int neg = idx >= a; if (neg) idx -= a;
    const int kk = (int)((idx + 1) >> 1);
    if (kk) idx -= 2 * (uint32_t)kk - 1;
    const int v = k - kk;
    *y++ = (int16_t)(neg ? -v : v); *y = (int16_t)(idx ? -kk : kk);
    return yy + v * v + kk * kk;
}
// variant B: simple row-walk (first new version)
int32_t unrankB(int n, int k, uint32_t idx, int16_t *y) {
    const uint32_t *T = pvqTable32();
    const uint32_t *here = T + (size_t)k * D;
    int32_t yy = 0;
    while (n > 2) {
        uint32_t a = here[D + n];
        const int neg = idx >= a;
        if (neg) idx -= a;
        a = here[n];
        if (a <= idx) { idx -= a; *y++ = 0; }
        else {
            int kk = k; const uint32_t *p = here + n;
            if (kk > n && T[(size_t)n * D + n] > idx) { kk = n; p = T + (size_t)n * D + n; }
            do { kk--; p -= D; } while (*p > idx);
            idx -= *p; const int v = k - kk; *y++ = (int16_t)(neg ? -v : v); yy += v * v; k = kk; here = T + (size_t)k * D;
        }
        n--;
    }
    const uint32_t a = 2 * (uint32_t)k + 1;
    const int neg = idx >= a; if (neg) idx -= a;
    const int kk = (int)((idx + 1) >> 1);
    if (kk) idx -= 2 * (uint32_t)kk - 1;
    const int v = k - kk;
    *y++ = (int16_t)(neg ? -v : v); *y = (int16_t)(idx ? -kk : kk);
    return yy + v * v + kk * kk;
}
// variant C: branch-free lattice walk: one step = one table compare; either the coordinate is finished (n--) or one more
// pulse goes to it (k--).  Signs are resolved when a coordinate starts.
int32_t unrankC(int n, int k, uint32_t idx, int16_t *y) {
    const uint32_t *T = pvqTable32();
    int32_t yy = 0;
    // per coordinate: sign, then walk k down
    const uint32_t *col = T + n;            // col[k*D] = U(n,k)
    long off = (long)k * D;                 // current k row offset
    while (n > 2) {
        const uint32_t above = col[off + D];
        const uint32_t neg = idx >= above ? ~0u : 0u;
        idx -= above & neg;
        int v = 0;
        // walk: while U(n,k) > idx: k--, v++
        uint32_t p = col[off];
        while (p > idx) { off -= D; v++; p = col[off]; }
        idx -= p;
        *y++ = (int16_t)((v ^ (int)neg) - (int)neg);
        yy += v * v;
        n--; col--;
    }
    k = (int)(off / D);
    const uint32_t a = 2 * (uint32_t)k + 1;
    const int neg = idx >= a; if (neg) idx -= a;
    const int kk = (int)((idx + 1) >> 1);
    if (kk) idx -= 2 * (uint32_t)kk - 1;
    const int v = k - kk;
    *y++ = (int16_t)(neg ? -v : v); *y = (int16_t)(idx ? -kk : kk);
    return yy + v * v + kk * kk;
}
// variant D: fully branch-free inner step with a fixed trip count N + K (each step: finish coordinate or add a pulse)
int32_t unrankD(int n, int k, uint32_t idx, int16_t *y) {
    const uint32_t *T = pvqTable32();
    int32_t yy = 0;
    const uint32_t *cell = T + (long)k * D + n;     // U(n,k); cell[D] = U(n,k+1)
    int v = 0;
    uint32_t neg = 0;
    bool fresh = true;
    while (n > 2) {
        if (fresh) {                                // (predictable: alternates with the data only through `take`)
            const uint32_t above = cell[D];
            neg = idx >= above ? ~0u : 0u;
            idx -= above & neg;
        }
        const uint32_t p = *cell;
        const bool take = p <= idx;
        idx -= take ? p : 0;
        *y = (int16_t)((v ^ (int)neg) - (int)neg);
        yy += take ? v * v : 0;
        y += take;
        n -= take;
        cell -= take ? 1 : D;
        k -= take ? 0 : 1;
        v = take ? 0 : v + 1;
        fresh = take;
    }
    const uint32_t a = 2 * (uint32_t)k + 1;
    const int ng = idx >= a; if (ng) idx -= a;
    const int kk = (int)((idx + 1) >> 1);
    if (kk) idx -= 2 * (uint32_t)kk - 1;
    const int vv = k - kk;
    *y++ = (int16_t)(ng ? -vv : vv); *y = (int16_t)(idx ? -kk : kk);
    return yy + vv * vv + kk * kk;
}
// variant E: the lattice walk with both possible next cells loaded ahead (the load leaves the dependent chain)
int32_t unrankE(int n, int k, uint32_t idx, int16_t *y) {
    const uint32_t *T = pvqTable32();
    int32_t yy = 0;
    const uint32_t *cell = T + (long)k * D + n;     // U(n,k); cell[D] = U(n,k+1)
    uint32_t p = cell[0], above = cell[D];
    int v = 0;
    uint32_t neg = 0, fresh = ~0u;
    while (n > 2) {
        // candidates of the next step, whichever way this one goes
        const uint32_t pTake = cell[-1], aboveTake = cell[D - 1], pMore = cell[-D];
        const uint32_t isNeg = (idx >= above ? ~0u : 0u) & fresh;
        neg = (neg & ~fresh) | isNeg;
        const uint32_t base = above & isNeg;        // what the sign costs (0 when positive or not fresh)
        const bool take = (uint64_t)p + base <= idx;
        idx -= take ? p + base : base;
        *y = (int16_t)((v ^ (int)neg) - (int)neg);
        yy += take ? v * v : 0;
        y += take;
        n -= take;
        k -= take ? 0 : 1;
        cell -= take ? 1 : D;
        p = take ? pTake : pMore;
        above = aboveTake;                           // (only read when fresh)
        v = take ? 0 : v + 1;
        fresh = take ? ~0u : 0u;
    }
    const uint32_t a = 2 * (uint32_t)k + 1;
    // a coordinate in progress when n hit 2?  (cannot: the loop leaves only on `take`, or before any step)
    const int ng = idx >= a; if (ng) idx -= a;
    const int kk = (int)((idx + 1) >> 1);
    if (kk) idx -= 2 * (uint32_t)kk - 1;
    const int vv = k - kk;
    *y++ = (int16_t)(ng ? -vv : vv); *y = (int16_t)(idx ? -kk : kk);
    return yy + vv * vv + kk * kk;
}
int main() {
    FILE *f = fopen("/tmp/dec/pvq_trace.bin", "rb");
    std::vector<T3> t(867980);
    size_t n = fread(t.data(), sizeof(T3), t.size(), f); fclose(f); t.resize(n);
    int16_t ya[192], yb[192];
    // correctness vs B
    for (auto &e : t) {
        int32_t r0 = unrankB(e.n, e.k, e.idx, yb);
        int32_t (*fs[4])(int,int,uint32_t,int16_t*) = {unrankA, unrankC, unrankD, unrankE};
        for (auto fn : fs) { memset(ya, 0, sizeof ya); int32_t r = fn(e.n, e.k, e.idx, ya); if (r != r0 || memcmp(ya, yb, e.n * 2)) { printf("MISMATCH n %d k %d\n", e.n, e.k); return 1; } }
    }
    auto bench = [&](const char *nm, int32_t (*fn)(int,int,uint32_t,int16_t*)) {
        double best = 1e9; long acc = 0;
        for (int r = 0; r < 7; r++) {
            auto t0 = std::chrono::steady_clock::now();
            for (auto &e : t) acc += fn(e.n, e.k, e.idx, ya);
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        printf("%s: %.2f ns per vector, %.2f us per frame (%ld)\n", nm, best / t.size() * 1e9, best / 11184 * 1e6, acc);
    };
    for (int r = 0; r < 2; r++) { bench("B row-walk", unrankB); bench("C column walk", unrankC); bench("D branch-free", unrankD); bench("E speculative", unrankE); }
}
