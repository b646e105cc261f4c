// tools/chainbench.hip -- A/B harness of the frames -> PCM chain (post-filter stage forms; fused forms), ONE process,
// variants interleaved round-robin (cdna_hip_programming.md section 5.4 rule 24).  Tuning tool, not part of the product.
//
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -Ilibnyquist_amd/csrc -o tools/chainbench tools/chainbench.hip
//   hipcc ... -DNYQ_PIPE_STAMPS -o tools/chainbench_stamps tools/chainbench.hip     (per-role cycle accounting, diagnostic)
//   ./tools/chainbench [rounds] [nstreams] [nframes] [case]     case: mix | short | long | all | off
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "nyq_kernels.hpp"
#include "nyq_post_kernels.hpp"
#include "nyq_post_pipe.hpp"
using namespace nyq;
#ifdef CHB_MONO
#define CHB_PAIR false
#else
#define CHB_PAIR true
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Variant {
    std::string name;
    std::function<void()> run;
    std::vector<float> ms;
};

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 10;
    const long ns = argc > 2 ? atol(argv[2]) : 1024, nf = argc > 3 ? atol(argv[3]) : 256;
    const char *cs = argc > 4 ? argv[4] : "mix";
#ifdef CHB_MONO   // mono streams through the generic instance (-DCHB_MONO: pass twice the stream count for the same bytes)
    const int ch = 1, N = 960;
#else
    const int ch = 2, N = 960;
#endif
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t nx = (size_t)ns * ch * nf * N;
    float *d_pcm, *d_out, *d_win, *d_gain;
    int *d_pitch, *d_tap;
    CK(hipMalloc(&d_pcm, nx * 4)); CK(hipMalloc(&d_out, nx * 4)); CK(hipMalloc(&d_win, 120 * 4));
    CK(hipMalloc(&d_gain, ns * nf * 4)); CK(hipMalloc(&d_pitch, ns * nf * 4)); CK(hipMalloc(&d_tap, ns * nf * 4));
    {
        std::vector<float> h(nx);
        unsigned s = 4;
        for (size_t i = 0; i < nx; i++) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 300.f; }
        CK(hipMemcpy(d_pcm, h.data(), nx * 4, hipMemcpyHostToDevice));
        std::vector<float> w(120);
        for (int i = 0; i < 120; i++) { double x = sin(.5 * M_PI * (i + .5) / 120); w[i] = (float)sin(.5 * M_PI * x * x); }
        CK(hipMemcpy(d_win, w.data(), 120 * 4, hipMemcpyHostToDevice));
        std::mt19937 g(7);
        std::vector<int> pp(ns * nf), pt(ns * nf);
        std::vector<float> pg(ns * nf);
        int lo = 15, hi = 80;
        double on = 0.7;
        if (!strcmp(cs, "short")) { hi = 60; on = 1.0; }
        if (!strcmp(cs, "long")) { lo = 300; hi = 1000; on = 1.0; }
        if (!strcmp(cs, "all")) { hi = 1000; on = 8.0 / 9.0; }
        if (!strcmp(cs, "off")) on = 0.0;
        for (long i = 0; i < ns * nf; i++) {
            pp[i] = lo + (int)(g() % (unsigned)(hi - lo));
            pt[i] = (int)(g() % 3u);
            pg[i] = ((g() % 1000000u) / 1e6 < on) ? (1 + (int)(g() % 8u)) * 0.09375f : 0.f;
        }
        if (!strncmp(cs, "file:", 5)) {
            // parameters from a file: ns*nf int32 periods, ns*nf float gains, ns*nf int32 tapsets (tools/chain_time.py's
            // 'real' case dumped with numpy: python tools/dump_real_params.py)
            FILE *fp = fopen(cs + 5, "rb");
            if (!fp || fread(pp.data(), 4, ns * nf, fp) != (size_t)(ns * nf) || fread(pg.data(), 4, ns * nf, fp) != (size_t)(ns * nf) ||
                fread(pt.data(), 4, ns * nf, fp) != (size_t)(ns * nf)) { printf("cannot read %s\n", cs + 5); return 1; }
            fclose(fp);
        }
        CK(hipMemcpy(d_pitch, pp.data(), ns * nf * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tap, pt.data(), ns * nf * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_gain, pg.data(), ns * nf * 4, hipMemcpyHostToDevice));
    }
    PostArgs A{};
    A.pcm = d_pcm; A.pf_pitch = d_pitch; A.pf_gain = d_gain; A.pf_tapset = d_tap; A.pf_state = nullptr; A.pf_state_out = nullptr;
    A.hist = nullptr; A.deemph = nullptr; A.out = d_out; A.nstreams = ns; A.nframes = nf; A.channels = ch;

    std::vector<Variant> v;
    {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, celt_post_kernel<3, 4, 1>, kWave * 4, 0));
        const unsigned grid = (unsigned)std::min<long>((ns * ch + 3) / 4, (long)occ * cus);
        char nm[128]; snprintf(nm, sizeof nm, "r1 wave per channel (occ %d blk/CU)", occ);
        v.push_back({nm, [=] { hipLaunchKernelGGL((celt_post_kernel<3, 4, 1>), dim3(grid), dim3(kWave * 4), 0, 0, A, d_win); }, {}});
    }
#ifndef CHB_MONO
    {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, celt_post_kernel<3, 2, 2>, kWave * 2, 0));
        const unsigned grid = (unsigned)std::min<long>((ns + 1) / 2, (long)occ * cus);
        char nm[128]; snprintf(nm, sizeof nm, "r1 wave per stereo pair (occ %d blk/CU)", occ);
        v.push_back({nm, [=] { hipLaunchKernelGGL((celt_post_kernel<3, 2, 2>), dim3(grid), dim3(kWave * 2), 0, 0, A, d_win); }, {}});
    }
#else
    v.push_back({"(no stereo-pair form for mono)", [=] {}, {}});
#endif
    {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, celt_post_pipe_kernel<3, CHB_PAIR>, kWave * kPipeWaves, 0));
        const unsigned grid = (unsigned)std::min<long>((ns * ch + 1) / 2, (long)occ * cus);
        char nm[128]; snprintf(nm, sizeof nm, "r2 workgroup pipeline (occ %d blk/CU, grid %u)", occ, grid);
        v.push_back({nm, [=] { hipLaunchKernelGGL((celt_post_pipe_kernel<3, CHB_PAIR>), dim3(grid), dim3(kWave * kPipeWaves), 0, 0, A, d_win); }, {}});
    }
    // experiment: the pipeline kernel behind a large memset node (a fill kernel of the runtime) on unrelated memory
    {
        static float *dummy = nullptr;
        const size_t dbytes = (size_t)128 << 20;
        CK(hipMalloc(&dummy, dbytes));
        auto base = v.back().run;
        v.push_back({"r2 workgroup pipeline behind a 128 MB memset node", [=] { CK(hipMemsetAsync(dummy, 0, dbytes, 0)); base(); }, {}});
        v.push_back({"the 128 MB memset node alone", [=] { CK(hipMemsetAsync(dummy, 0, dbytes, 0)); }, {}});
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto &x : v) x.run();
    CK(hipDeviceSynchronize());
#ifdef NYQ_PIPE_STAMPS
    {
        unsigned long long z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_stamps), z, sizeof z));
        v[2].run();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_pipe_stamps), sizeof z));
        const double nwg = (double)((ns * ch + 1) / 2), per = nwg * nf;   // per workgroup-frame (comb: wave 0 only)
        printf("cycles per frame (s_memtime ticks, mean over workgroups), case %s\n", cs);
        printf("  comb wave : set-up %.0f  first-120 %.0f  cross-fade %.0f  constant part %.0f  barrier wait %.0f\n", z[2] / per, z[3] / per,
               z[4] / per, z[0] / per, z[1] / per);
        printf("  I/O  wave : keep-copy %.0f  de-emphasis %.0f  pick-up + land + store %.0f  fetch %.0f  tail %.0f  barrier wait %.0f\n",
               z[8] / per, z[9] / per, z[10] / per, z[11] / per, z[12] / per, z[13] / per);
        // placement trace: where and when every workgroup of the pipeline kernel ran, alone and behind the memset node
        auto placement = [&](const char *what, std::function<void()> run) {
            static unsigned long long wg[4096][4];
            CK(hipDeviceSynchronize());
            run();
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_pipe_wg), sizeof wg));
            const int n = (int)std::min<long>(4096, (ns * ch + 1) / 2);
            std::vector<int> percu(8 * 64, 0);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < n; i++) t0 = std::min(t0, wg[i][2]), t1 = std::max(t1, wg[i][3]);
            std::vector<double> st, du;
            for (int i = 0; i < n; i++) {
                const unsigned hw = (unsigned)wg[i][0], xcc = (unsigned)wg[i][1] & 15;
                const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 3;
                percu[xcc * 64 + se * 16 + sh * 8 * 0 + cu + sh * 0] += 1;   // (sh is always 0 on this part)
                st.push_back((wg[i][2] - t0) * 0.01);
                du.push_back((wg[i][3] - wg[i][2]) * 0.01);
            }
            int hist[16] = {0}, used = 0;
            for (int v : percu) { if (v) used++; hist[std::min(v, 15)]++; }
            std::sort(st.begin(), st.end());
            std::sort(du.begin(), du.end());
            printf("placement, %s: %d workgroups on %d CUs; workgroups per CU:", what, n, used);
            for (int k = 1; k < 16; k++) if (hist[k]) printf(" %d x%d", k, hist[k]);
            printf("\n   start after first [us]: p50 %.1f p90 %.1f p99 %.1f max %.1f; lifetime [us]: min %.1f p50 %.1f max %.1f; span %.1f us\n",
                   st[n / 2], st[n * 9 / 10], st[n * 99 / 100], st[n - 1], du[0], du[n / 2], du[n - 1], (t1 - t0) * 0.01);
        };
        placement("alone", v[2].run);
        placement("behind a 128 MB memset node", v[3].run);
        placement("alone again", v[2].run);
    }
#endif
    for (int r = 0; r < rounds; r++)
        for (auto &x : v) {
            CK(hipEventRecord(a)); x.run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); x.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("case %s: %ld streams x %ld frames x %d ch\n%-56s %9s %9s %9s %8s\n", cs, ns, nf, ch, "variant", "median ms", "min ms", "alg GB/s", "frac 8T");
    for (auto &x : v) {
        std::sort(x.ms.begin(), x.ms.end());
        const float med = x.ms[x.ms.size() / 2], mn = x.ms[0];
        const double gbs = 7680.0 * ns * nf * ch / (med * 1e-3) / 1e9;
        printf("%-56s %9.4f %9.4f %9.1f %8.3f\n", x.name.c_str(), med, mn, gbs, gbs / 8000.0);
    }
    return 0;
}
