// tools/kbench.hip -- A/B harness: times kernel configurations of nyq_kernels.hpp against each
// other in ONE process, interleaved round-robin (cdna_hip_programming.md section 5.4 rule 24),
// on the bench workload (2^20 nfft-480 rows, U(-1,1)).  Tuning tool, not part of the product.
// Build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -Ilibnyquist_amd/csrc -o tools/kbench tools/kbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "nyq_kernels.hpp"
using namespace nyq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Variant {
    std::string name;
    std::function<void()> run;
    std::vector<float> ms;
};

static float *d_in, *d_fin, *d_tail, *d_trig, *d_win;
static long rows = 1 << 20;
static int cus = 256;

template <typename Cfg>
static void add(std::vector<Variant> &v, const char *name, int blocks_per_cu, bool with_tail = true) {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, imdct_rows_kernel<32, Cfg>, kWave * Cfg::WPB, 0));
    int bpc = blocks_per_cu > 0 ? blocks_per_cu : occ;
    char nm[160];
    snprintf(nm, sizeof nm, "%-28s wpb%d blk/CU %d (occ %d)", name, Cfg::WPB, bpc, occ);
    unsigned grid = (unsigned)(cus * bpc);
    float *tl = with_tail ? d_tail : nullptr;
    v.push_back({nm, [grid, tl] {
        hipLaunchKernelGGL((imdct_rows_kernel<32, Cfg>), dim3(grid), dim3(kWave * Cfg::WPB), 0, 0, d_in, nullptr, d_fin, tl, rows, d_trig, d_win);
    }, {}});
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 12;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); cus = prop.multiProcessorCount;
    size_t n = (size_t)rows * 960;
    CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_fin, n * 4)); CK(hipMalloc(&d_tail, (size_t)rows * 60 * 4));
    CK(hipMalloc(&d_trig, 481 * 4)); CK(hipMalloc(&d_win, 120 * 4));
    {
        std::vector<float> h(n);
        unsigned s = 480;
        for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
        CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> t(481), w(120);
        for (int i = 0; i <= 480; i++) t[i] = (float)cos(2 * 3.141592653f * i / 1920);
        for (int i = 0; i < 120; i++) { double x = sin(.5 * M_PI * (i + .5) / 120); w[i] = (float)sin(.5 * M_PI * x * x); }
        CK(hipMemcpy(d_trig, t.data(), 481 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_win, w.data(), 120 * 4, hipMemcpyHostToDevice));
    }
    std::vector<Variant> v;
#include "kbench_variants.inc"
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto &x : v) { x.run(); }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; r++)
        for (auto &x : v) {
            CK(hipEventRecord(a)); x.run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); x.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("%-52s %9s %9s %9s %8s\n", "variant", "median ms", "min ms", "alg GB/s", "frac 8T");
    for (auto &x : v) {
        std::sort(x.ms.begin(), x.ms.end());
        float med = x.ms[x.ms.size() / 2], mn = x.ms[0];
        double gbs = 7680.0 * rows / (med * 1e-3) / 1e9;
        printf("%-52s %9.4f %9.4f %9.1f %8.3f\n", x.name.c_str(), med, mn, gbs, gbs / 8000.0);
    }
    return 0;
}
