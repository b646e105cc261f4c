// tools/kbench.hip -- A/B harness: times kernel configurations of nyq_kernels.hpp against each
// other in ONE process, interleaved round-robin (cdna_hip_programming.md section 5.4 rule 24),
// on the bench workload (2^20 nfft-480 rows, U(-1,1)).  Tuning tool, not part of the product.
// Build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -Ilibnyquist_amd/csrc -o tools/kbench tools/kbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "nyq_kernels.hpp"
using namespace nyq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// ---- experiment: LDS-DMA stage-in ---------------------------------------------------------
// The raw rows of the NEXT group land in the wave's LDS by global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPRs
// while in flight), issued right after the current group's rows have been pre-rotated out of the landing buffer; the
// pre-rotation reads LDS instead of freshly loaded registers.  One landing buffer of 4 x 3840 B per wave next to the
// 15.9 KB slice: 31 KB of LDS per wave, five waves per CU.
__device__ __forceinline__ void dma_row_to_lds(const float *grow, float *lrow, int lane) {
    // four full-wave pieces of 1 KiB over the 3840 B of a row: the last one starts at float4 176 and overlaps the third
    // (a piece with inactive lanes did not land where expected: measured, so every piece keeps all 64 lanes)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int v0 = q < 3 ? q * 64 : 240 - 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(grow + 4 * (v0 + lane)),
                                         (__attribute__((address_space(3))) void *)(lrow + 4 * v0), 16, 0, 0);
    }
}

template <int WAIT_STORES>
__global__ __launch_bounds__(64) void imdct_rows_dma_kernel(const float *__restrict__ in, float *__restrict__ fin,
                                                            float *__restrict__ tail, long nrows, const float *__restrict__ trig,
                                                            const float *__restrict__ window) {
    using Gm = Geo<32>;
    __shared__ __attribute__((aligned(16))) float raw[Gm::G * Gm::NIN];
    __shared__ __attribute__((aligned(16))) float smem[2 * Gm::LDS_CPX];
    const int lane = threadIdx.x;
    cpx *lds = reinterpret_cast<cpx *>(smem);
    LaneConst<32> K;
    lane_init<32>(K, lane, trig, window);
    const long ngroups = (nrows + Gm::G - 1) / Gm::G, nwaves = gridDim.x;
    long gi = blockIdx.x;
    auto issue = [&](long g0) {
        for (int g = 0; g < Gm::G; g++)
            if (g0 * Gm::G + g < nrows) dma_row_to_lds(in + (g0 * Gm::G + g) * (long)Gm::NIN, raw + g * Gm::NIN, lane);
    };
    if (gi < ngroups) issue(gi);
    for (; gi < ngroups; gi += nwaves) {
        // the landing buffer is complete when every older vector-memory operation is: the DMA of this group was issued
        // BEFORE the previous group's stores, so (WAIT_STORES == 0) all but the youngest 28 operations suffice for whole
        // groups; the last, possibly ragged, group and WAIT_STORES == 1 wait for everything
        if (WAIT_STORES == 0 && (gi + 1) * Gm::G <= nrows && gi != (long)blockIdx.x) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NYQ_WAVE_SYNC();
        IndepRows<32> rows{in, nullptr, fin, tail, gi * Gm::G, nrows};
        IndepRows<32> lrows{raw, nullptr, nullptr, nullptr, 0, nrows - gi * Gm::G < Gm::G ? nrows - gi * Gm::G : (long)Gm::G};
        StageRegs<32> R;
        stage_in_load<32, 0>(R, lane, lrows);                   // LDS -> registers (ds_read_b128)
        NYQ_WAVE_SYNC();
        stage_in_store<32>(R, K, lane, lds);
        NYQ_WAVE_SYNC();
        if (gi + nwaves < ngroups) issue(gi + nwaves);          // next group's rows on their way during the transform
        fft_passes<32>(lane, lds);
        HeadRegs<32> H;
        stage_out<32, 0>(K, lane, lds, nullptr, rows, H);
    }
}

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

template <int WAIT_STORES>
static void add_dma(std::vector<Variant> &v, const char *name, int blocks_per_cu) {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, imdct_rows_dma_kernel<WAIT_STORES>, kWave, 0));
    const int bpc = blocks_per_cu > 0 ? std::min(blocks_per_cu, occ) : occ;
    char nm[160];
    snprintf(nm, sizeof nm, "%-28s wpb1 blk/CU %d (occ %d)", name, bpc, occ);
    const unsigned grid = (unsigned)(cus * bpc);
    v.push_back({nm, [grid] {
        hipLaunchKernelGGL((imdct_rows_dma_kernel<WAIT_STORES>), dim3(grid), dim3(kWave), 0, 0, d_in, d_fin, d_tail, rows, d_trig, d_win);
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
    // every variant must produce the first variant's output bit for bit (same lane program, same order of operations)
    {
        const size_t take = 4096 * 960, off = (size_t)(rows - 4096) * 960;
        std::vector<float> want(2 * take), got(2 * take), wt(4096 * 60), gt(4096 * 60);
        for (size_t k = 0; k < v.size(); k++) {
            CK(hipMemset(d_fin, 0xff, n * 4));
            CK(hipMemset(d_tail, 0xff, (size_t)rows * 60 * 4));
            v[k].run();
            CK(hipDeviceSynchronize());
            std::vector<float> &dst = k == 0 ? want : got;
            CK(hipMemcpy(dst.data(), d_fin, take * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(dst.data() + take, d_fin + off, take * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy((k == 0 ? wt : gt).data(), d_tail + (size_t)(rows - 4096) * 60, 4096 * 60 * 4, hipMemcpyDeviceToHost));
            if (k > 0 && v[k].name.find("no-tail") == std::string::npos) {
                size_t bad = 0;
                size_t shown = 0;
                for (size_t i = 0; i < 2 * take; i++)
                    if (memcmp(&want[i], &got[i], 4) != 0) {
                        bad++;
                        if (shown++ < 0) printf("   fin row %zu col %zu: want %.9g got %.9g\n", i / 960, i % 960, want[i], got[i]);
                    }
                shown = 0;
                for (size_t i = 0; i < wt.size(); i++)
                    if (memcmp(&wt[i], &gt[i], 4) != 0) {
                        bad++;
                        if (shown++ < 0) printf("   tail row %zu col %zu: want %.9g got %.9g\n", i / 60, i % 60, wt[i], gt[i]);
                    }
                double worst = 0;
                for (size_t i = 0; i < 2 * take; i++) worst = std::max(worst, std::fabs((double)want[i] - got[i]) / (std::fabs((double)want[i]) + 1.0));
                printf("check %-52s %s (%zu of %zu floats differ in the last bits, worst |d|/(|x|+1) = %.2g)\n", v[k].name.c_str(),
                       bad == 0 ? "identical" : worst < 1e-5 ? "same to rounding" : "MISMATCH", bad, 2 * take + wt.size(), worst);
            }
        }
    }
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
